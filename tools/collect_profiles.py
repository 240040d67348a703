#!/usr/bin/env python3
"""Collect the rocprofv3 evidence behind the bench lines of a round, on the GPU box:

  python3 tools/collect_profiles.py gpurun_out/prof_r2      (then copy the summaries it names into profiles/rN/)

For every workload below: one `rocprofv3 --kernel-trace --stats` run of bench.py (kernel durations) and three
`--kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE and one SQ group, each in its own pass: TCC slots do not hold both
sizes, MI355X_MICROARCH.md "rocprofv3 PMC slots"; counters are never combined with tracing other than --kernel-trace).
Writes <out>/kernel_stats_<workload>.csv and <out>/pmc_index.json, whose records bench.py matches by
(kernel, batch, workload) and by the hash of the kernel sources they were taken on.

This driver never touches the GPU itself: every measured program is a child `rocprofv3 ... -- python3 bench.py ...`.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB.  Which correction of MI355X_MICROARCH.md's HBM section a record's
FETCH_SIZE got is WRITTEN INTO THE RECORD (`fetch_size_rule`), per kernel (FETCH_RULES below): "as read" for kernels whose
loads are 8-byte-per-lane record loads (the raw counter matched the known input byte count to 1 %,
profiles/r1/hbm_traffic_pmc_bench_static_b4096.json), "x2" for kernels that stream 16 bytes per lane (the guide's gfx950
correction).  WRITE_SIZE is taken as read everywhere (it matched the algorithmic output of the dynamics kernel exactly).
A rocprofv3 pass that exits non-zero or yields no counter rows marks the record `incomplete` and this script exits non-zero.
"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = [
    # name, kernel substring, batch, bench.py arguments
    # (+warm: bench.py's default method since round 6 -- the caller's placed + warm-started loop on a trajectory; +placed: the
    # same steps with every QP started cold, round 5's method; no suffix: the plain entry, the headline of rounds 1-4)
    ("static-survey+warm", "balance_coop_kernel", 4096, ["--gait", "static", "--errors", "survey", "--method", "warm"]),
    ("static-calm+warm", "balance_coop_kernel", 4096, ["--gait", "static", "--errors", "calm", "--method", "warm"]),
    ("trot+warm", "balance_coop_kernel", 4096, ["--gait", "trot", "--method", "warm"]),
    ("trot+warm", "balance_coop_kernel", 8192, ["--gait", "trot", "--batch", "8192", "--method", "warm"]),
    ("trot+warm", "balance_coop_kernel", 65536, ["--gait", "trot", "--batch", "65536", "--method", "warm"]),
    ("static-survey+placed", "balance_coop_kernel", 4096, ["--gait", "static", "--errors", "survey", "--method", "placed"]),
    ("trot+placed", "balance_coop_kernel", 8192, ["--gait", "trot", "--batch", "8192", "--method", "placed"]),
    ("trot+placed", "balance_coop_kernel", 65536, ["--gait", "trot", "--batch", "65536", "--method", "placed"]),
    ("static-survey", "balance_coop_kernel", 4096, ["--gait", "static", "--errors", "survey", "--method", "plain"]),
    ("trot", "balance_coop_kernel", 4096, ["--gait", "trot", "--method", "plain"]),
    ("trot", "balance_coop_kernel", 65536, ["--gait", "trot", "--batch", "65536", "--method", "plain"]),
    ("pose_sqp", "pose_sqp_coop_kernel", 4096, ["--workload", "pose_sqp"]),
    ("wholebody-trot", "wholebody_solve_kernel", 4096, ["--workload", "wholebody", "--gait", "trot"]),
    ("wholebody_dynamics", "wholebody_dynamics_leg_kernel", 1048576, ["--workload", "wholebody_dynamics", "--batch", "1048576"]),
    ("wholebody_dynamics", "wholebody_dynamics_kernel", 4096, ["--workload", "wholebody_dynamics", "--batch", "4096"]),
]
# FETCH_SIZE correction per kernel (see the docstring): every kernel here reads its inputs with 8-byte-per-lane loads
# (24-/32-byte records and the 12 joint angles coalesced over 12 lanes; the whole-body kernels read q, qd the same way)
FETCH_RULES = {"balance_coop_kernel": "as read", "pose_sqp_coop_kernel": "as read", "wholebody_solve_kernel": "as read",
               "wholebody_dynamics_leg_kernel": "as read", "wholebody_dynamics_kernel": "as read"}
SQ_GROUP = "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"


def run(cmd, log):
    with open(log, "w") as f:
        return subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))


def kernel_rows(db):
    con = sqlite3.connect(db)
    return con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                       "group by name order by 6 desc").fetchall()


def counters(db, kernel):
    con = sqlite3.connect(db)
    acc = {}
    for cname, _, v in con.execute("select counter_name, dispatch_id, sum(value) from counters_collection where kernel_name like ? "
                                   "group by counter_name, dispatch_id order by dispatch_id", ("%" + kernel + "%",)):
        acc.setdefault(cname, []).append(v)
    return {c: sum(v[5:]) / len(v[5:]) if len(v) > 10 else sum(v) / len(v) for c, v in acc.items()}


def derive(rec):
    """Derived figures of a record, from its own counters and kernel average: the fraction of the chip's VALU issue slots the
    launch used (a wave64 VALU instruction holds its SIMD16 for 4 cycles; 1024 SIMDs at 2.4 GHz) and the mean lifetime of a
    wavefront (SQ_WAVE_CYCLES counts in units of 4 cycles)."""
    c = rec.get("counters", {})
    if "SQ_INSTS_VALU" in c and rec.get("kernel_avg_us"):
        rec["valu_issue_frac"] = c["SQ_INSTS_VALU"] * 4.0 / (1024 * rec["kernel_avg_us"] * 1e-6 * 2.4e9)
    if c.get("SQ_WAVES") and "SQ_WAVE_CYCLES" in c:
        rec["mean_wave_lifetime_us"] = c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"] * 4.0 / 2.4e3


def main():
    if sys.argv[1] == "--derive":  # add the derived figures to an index collected before they existed
        path = sys.argv[2]
        idx = json.load(open(path))
        for rec in idx["records"]:
            derive(rec)
        json.dump(idx, open(path, "w"), indent=1, sort_keys=True)
        return
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    import bench
    records = []
    bench_py = os.path.join(ROOT, "bench.py")
    for name, kernel, batch, args in WORKLOADS:
        tag = "%s_b%d" % (name.replace("-", "_").replace("+", "_"), batch)
        rec = dict(kernel=kernel, batch=batch, workload=name, source_hash=bench.source_hash(), files=[], failed_passes=[])
        # kernel durations (the K steps as one hipGraph, as the bench line is measured)
        d = os.path.join(out, "raw", tag + "_stats")
        if run(["rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "s", "--", "python3", bench_py, *args, "--steps", "50",
                "--warmup", "10", "--no-cpu-baseline", "--no-also", "--replays", "5"], os.path.join(out, tag + "_stats.log")) != 0:
            rec["failed_passes"].append("kernel-trace --stats")
        dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        if dbs:
            rows = kernel_rows(dbs[0])
            csv = os.path.join(out, "kernel_stats_%s.csv" % tag)
            with open(csv, "w") as f:
                f.write("kernel,calls,avg_us,min_us,max_us,total_us\n")
                for nm, n, avg, mn, mx, tot in rows:
                    short = nm.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
                    f.write("%s,%d,%.2f,%.2f,%.2f,%.2f\n" % (short, n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e3))
                    if kernel in nm:
                        rec["kernel_avg_us"], rec["kernel_calls"] = avg / 1e3, n
            rec["files"].append(os.path.basename(csv))
        # counters: eager launches, one counter group per pass
        for group in ("FETCH_SIZE", "WRITE_SIZE", SQ_GROUP):
            d = os.path.join(out, "raw", tag + "_pmc_" + group.split()[0])
            rc = run(["rocprofv3", "--kernel-trace", "--pmc", *group.split(), "-d", d, "-o", "p", "--", "python3", bench_py, *args,
                      "--no-graph", "--steps", "25", "--warmup", "5", "--no-cpu-baseline", "--no-also", "--replays", "1"],
                     os.path.join(out, tag + "_pmc_" + group.split()[0] + ".log"))
            got = 0
            for db in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
                for c, v in counters(db, kernel).items():
                    rec.setdefault("counters", {})[c] = v
                    got += 1
            if rc != 0 or got == 0:
                rec["failed_passes"].append("pmc " + group.split()[0])
        c = rec.get("counters", {})
        if "FETCH_SIZE" in c:
            rule = FETCH_RULES[kernel]
            rec["fetch_size_rule"] = "FETCH_SIZE %s (KB -> bytes), WRITE_SIZE as read" % rule
            rec["fetch_bytes"] = c["FETCH_SIZE"] * 1024.0 * (2.0 if rule == "x2" else 1.0)
        if "WRITE_SIZE" in c:
            rec["write_bytes"] = c["WRITE_SIZE"] * 1024.0
        if "SQ_INSTS_VALU" in c:
            rec["valu_insts"] = c["SQ_INSTS_VALU"]
        derive(rec)
        rec["incomplete"] = bool(rec["failed_passes"]) or not all(k in rec for k in ("fetch_bytes", "write_bytes", "valu_insts", "kernel_avg_us"))
        records.append(rec)
        print(json.dumps(rec))
    json.dump({"records": records, "collected_with": "tools/collect_profiles.py (rocprofv3 --kernel-trace [--stats | --pmc <one group>])"},
              open(os.path.join(out, "pmc_index.json"), "w"), indent=1, sort_keys=True)
    bad = [(r["workload"], r["batch"], r["failed_passes"]) for r in records if r["incomplete"]]
    if bad:
        sys.exit("incomplete records: %s" % bad)


if __name__ == "__main__":
    main()
