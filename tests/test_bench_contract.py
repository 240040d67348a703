"""The one-line JSON contract of bench.py: every key the driver reads, with the types and relations it relies on.
CPU: the committed line of the last measured run (profiles/r6/bench_static_b4096.json).  GPU: short live runs."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

TOP = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
       "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}
ROOFLINE = {"bound": str, "achieved": float, "peak": float, "unit": str, "frac": float}
CPU = {"value": float, "unit": str, "cores": int, "kind": str, "sample": str}


def check(line, want_cpu=True):
    d = json.loads(line)
    for k, t in TOP.items():
        assert isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None            # BASELINE.md holds no published number for this metric
    assert d["metric"].startswith("control-step QP solves/sec") and d["unit"] == "solves/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * d["config"]["robots_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k, t in ROOFLINE.items():
        assert isinstance(r[k], t), k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["peak"] == 8000.0
    if want_cpu:
        c = d["cpu_baseline"]
        for k, t in CPU.items():
            assert isinstance(c[k], t), k
        assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    return d


ALSO = {"static-calm", "trot", "trot_b8192", "trot_b65536", "pose_sqp_b4096", "full_tick_b4096", "wholebody_trot_b4096"}
ALSO_KEYS = ("value", "ms_per_step", "kernel_ms", "roofline_frac", "all_status_ok")
TRAJECTORY_KEYS = ("states_per_replay", "ticks", "regions", "dt", "switched_per_tick", "switched_at_wrap", "placement_give_ups", "working_set_unchanged",
                   "working_set_unchanged_at_wrap", "warm_rejected")


def check_round6(d, steps):
    """Round 6: every timed region runs on a trajectory (K consecutive control ticks, contact switches included), the method is the
    caller's placed + warm-started loop, the same steps with every QP started cold ride along (`cold_start`: the placed loop,
    round 5's method; `unplaced`: the plain entry, rounds 1-4's), and `also` carries every other BASELINE config plus the whole
    tick and the whole-body step."""
    c = d["config"]
    assert c["method"] == "warm" and "working set" in c["method_note"] and "placement" in c["method_note"]
    assert c["states_per_replay"] == steps and "consecutive control ticks" in c["workload"]
    t = c["trajectory"]
    for k in TRAJECTORY_KEYS:
        assert k in t, k
    assert t["states_per_replay"] == steps and t["dt"] == 0.0025 and t["switched_per_tick"] == 0.0      # a static stance switches nothing
    assert t["ticks"] == max(steps, steps * (200 // steps)) and t["regions"] * steps == t["ticks"]   # regions continue the trajectory
    assert 0.9 < t["working_set_unchanged"] <= 1.0 and t["warm_rejected"] == 0 and t["placement_give_ups"] == 0
    for key, m in (("cold_start", "placed"), ("unplaced", "plain")):
        u = d[key]
        assert u["method"] == m and u["all_status_ok"] is True and u["batch"] == 4096 and u["ms_per_step"] > 0, key
        assert u["trajectory"]["states_per_replay"] == steps
    assert d["ms_per_step"] < d["cold_start"]["ms_per_step"] < d["unplaced"]["ms_per_step"] * 1.02
    assert set(d["also"]) == ALSO
    for name, a in d["also"].items():
        assert "error" not in a, (name, a)
        for k in ALSO_KEYS:
            assert k in a, (name, k)
        assert a["all_status_ok"] is True and a["value"] > 0 and a["kernel_ms"] > 0 and 0 < a["roofline_frac"] < 1, name
        if name != "pose_sqp_b4096":
            assert a["cold_start"]["ms_per_step"] > 0 and a["cold_start"]["all_status_ok"] is True, name
    for name in ("static-calm", "trot", "trot_b8192", "trot_b65536"):
        a = d["also"][name]
        assert a["method"] == "warm" and a["unplaced"]["all_status_ok"] is True and "pmc_source" in a, name
        assert a["trajectory"]["warm_rejected"] == 0
        if name != "static-calm":   # a trot steps through its contact switches: about 1.1 % of the robots per tick
            assert 0.007 < a["trajectory"]["switched_per_tick"] < 0.016, name
    assert d["also"]["static-calm"]["tracking_error"] == [0.004, 0.005, 0.01]
    assert d["also"]["trot_b8192"]["batch"] == 8192 and d["also"]["trot_b65536"]["batch"] == 65536
    assert d["also"]["trot_b65536"]["trajectory"]["ticks"] == (64 if steps >= 64 else steps * (64 // steps))
    ft, wb = d["also"]["full_tick_b4096"], d["also"]["wholebody_trot_b4096"]
    assert ft["unit"] == "ticks/s" and ft["message_bytes"] > 3000 and 0.007 < ft["switched_per_tick"] < 0.016
    assert wb["unit"] == "whole-body control steps/s" and wb["warm_rejected"] == 0
    assert d["records"]["all_status_ok"] is True and d["records"]["ms_per_step"] < 1.05 * d["ms_per_step"] and "qlamd_state_record" in d["records"]["note"]
    assert d["also"]["trot_b65536"]["records"]["ms_per_step"] > 0
    sp = d["scale_point"]
    assert sp["robots_per_gpu"] == 8192 and sp["gait"] == "trot" and sp["n_gpus"] == d["n_gpus"] and sp["method"] == "warm"
    assert sp["value"] > 0 and sp["without_gather"] > 0 and "efficiency" in sp["definition"]
    assert abs(sp["value"] - d["also"]["trot_b8192"]["value"]) < 1e-6 * sp["value"]
    assert sp["cold_start"]["ms_per_step"] > 0 and sp["unplaced"]["ms_per_step"] > 0


def test_committed_bench_line_follows_the_contract():
    path = os.path.join(ROOT, "profiles", "r6", "bench_static_b4096.json")
    d = check(open(path).read().strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["config"]["robots_per_gpu"] == 4096 and d["config"]["all_status_ok"] is True
    # round 2: the input is disclosed, the timed region is a median of samples, PMC numbers carry their source
    assert len(d["config"]["samples_ms"]) >= 11 and "traffic_source" in d["roofline"]
    # round 3: the driver-timed workload is SURVEY 8(d)'s literal one, the other presets ride along, the CPU baseline is a
    # sustained rate of pinned threads, every PMC record says which FETCH_SIZE rule it got
    assert d["config"]["gait"] == "static" and d["config"]["tracking_error"] == [0.02, 0.05, 0.1]
    c = d["cpu_baseline"]
    assert c["cpu_model"] and c["value"] >= 0.9 * max(c["thread_sweep"].values()) and c["thread_sweep_seconds_each"] >= 1.5
    assert d["roofline"]["traffic"] and "FETCH_SIZE" in d["roofline"]["traffic_rule"]
    assert d["valu_issue"]["frac"] >= 0.25
    check_round6(d, d["steps"])


@pytest.mark.gpu
def test_live_bench_line_follows_the_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--cpu-seconds", "4",
                          "--no-also"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line"
    d = check(lines[0])
    assert d["steps"] == 20 and d["warmup"] == 3 and d["config"]["all_status_ok"] is True
    # the CPU baseline is a sustained rate: close to the best count of its own sweep (every count measured >= 1.5 s; the
    # committed 12-second line is held to 10 %, this short run on a shared host to 25 %)
    c = d["cpu_baseline"]
    assert c["value"] >= 0.75 * max(c["thread_sweep"].values()) and c["thread_sweep_seconds_each"] >= 1.5
    assert "OMP_PLACES=cores" in c["threads"] and "OMP_PROC_BIND=close" in c["threads"]


def _bench(*argv, rc=0):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == rc, (out.returncode, out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line (RCCL's shutdown banner included): %r" % lines
    return lines[0]


@pytest.mark.gpu
def test_live_default_line_is_the_contract_workload_and_carries_the_other_presets():
    """`python bench.py` with no workload flags: BASELINE configs[1] on SURVEY.md 8(d)'s literal tracking errors, and an
    `also` object with static-calm and trot measured in the same process."""
    d = check(_bench("--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--replays", "3"), want_cpu=False)
    assert d["config"]["gait"] == "static" and d["config"]["tracking_error"] == [0.02, 0.05, 0.1]
    check_round6(d, 20)
    assert d["also"]["static-calm"]["kernel_ms"] < d["roofline"]["kernel_ms"]     # the calm preset is the lighter one
    # the plain entry as the method: the placed loop rides along instead; the placed loop: the plain entry does
    d = check(_bench("--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--replays", "3", "--method", "plain", "--no-also"), want_cpu=False)
    assert d["config"]["method"] == "plain" and "cold_start" not in d
    d = check(_bench("--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--replays", "3", "--method", "placed", "--ticks", "7"), want_cpu=False)
    assert d["config"]["method"] == "placed" and d["unplaced"]["method"] == "plain" and d["unplaced"]["all_status_ok"] is True
    assert d["config"]["trajectory"]["ticks"] == 7 and d["config"]["states_per_replay"] == 20


@pytest.mark.gpu
def test_collective_path_with_one_rank():
    """The code path the N > 1 runs take -- RCCL process group, the K steps with their all-gathers captured into one
    hipGraph, gathers on a second captured stream -- exercised with a single rank, then without the gather, then with
    eager launches (what the ranks fall back to when a capture fails)."""
    common = ("--force-collective", "--gait", "trot", "--batch", "8192", "--steps", "20", "--warmup", "3",
              "--no-cpu-baseline", "--replays", "3")
    d = check(_bench(*common, "--overlap-gather"), want_cpu=False)
    c = d["config"]
    assert c["rccl_ranks"] == 1 and d["n_gpus"] == 1 and c["robots_per_gpu"] == 8192 and c["all_status_ok"] is True
    assert c["launch"] == "hipGraph of K steps" and c["gather_stream"] == "second captured stream"
    assert c["result_collection"] == "rccl all_gather of torques" and c["gather_layout_ok"] is True
    assert "also" not in d and "cpu_baseline" not in d
    sp = d["scale_point"]     # every line at every N carries it; here it IS the line's workload
    assert sp["robots_per_gpu"] == 8192 and abs(sp["value"] - d["value"]) < 1e-6 * d["value"] and sp["method"] == d["config"]["method"]
    with_gather = d["value"]
    d = check(_bench(*common, "--no-gather"), want_cpu=False)
    assert d["config"]["result_collection"] == "none (--no-gather)" and d["config"]["rccl_ranks"] == 1
    assert d["config"]["launch"] == "hipGraph of K steps" and d["config"]["gather_stream"] is None
    assert d["value"] > 0.8 * with_gather
    d = check(_bench(*common, "--overlap-gather", "--no-graph"), want_cpu=False)
    assert d["config"]["launch"] == "eager" and d["config"]["gather_layout_ok"] is True and d["config"]["all_status_ok"] is True


@pytest.mark.gpu
def test_collection_cadence_and_peer_copies_with_one_rank():
    """--gather-every K (one all-gather per K steps), --collect peer (copies into the ranks' buffers instead of a
    collective) and the `alternatives` object a several-rank line carries, each through the captured two-stream form."""
    common = ("--force-collective", "--overlap-gather", "--gait", "trot", "--batch", "8192", "--steps", "20", "--warmup", "3",
              "--no-cpu-baseline", "--replays", "3")
    d = check(_bench(*common, "--gather-every", "4"), want_cpu=False)
    c = d["config"]
    assert c["gather_every"] == 4 and c["collect"] == "rccl" and c["gather_layout_ok"] is True and c["all_status_ok"] is True
    assert c["launch"] == "hipGraph of K steps" and "alternatives" not in d
    d = check(_bench(*common, "--collect", "peer"), want_cpu=False)
    c = d["config"]
    assert c["collect"] == "peer" and c["gather_every"] == 1 and c["gather_layout_ok"] is True and c["all_status_ok"] is True
    assert c["result_collection"].startswith("copies of the torque shard") and c["launch"] == "hipGraph of K steps"
    d = check(_bench(*common, "--alternatives"), want_cpu=False)  # RCCL every step, the others measured beside it on request
    assert d["config"]["collect"] == "rccl" and d["config"]["gather_every"] == 1
    alt = d["alternatives"]
    assert set(alt) == {"rccl_every_8", "peer_every_1", "peer_every_8"}
    for name, a in alt.items():
        assert "error" not in a, (name, a)
        assert a["value"] > 0 and a["layout_ok"] is True, (name, a)
    # alternatives that do not come back in time cost the line their object only: the watchdog prints the line and ends the rank
    # -- with a non-zero exit code: a launcher must not read success from a rank that did not finish what it was asked to do
    d = check(_bench(*common, "--alternatives", "--alternatives-timeout", "0.05", rc=3), want_cpu=False)
    assert d["config"]["gather_layout_ok"] is True and d["value"] > 0
    assert "not finished" in d["alternatives"]["error"]


def test_profile_collection_names_exist_in_the_sources():
    """tools/collect_profiles.py keys its PMC records by kernel name and bench.py looks them up by the same strings: every
    kernel named there must be a __global__ function of the library, and the committed index must carry one record per
    workload of the list."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("collect_profiles", os.path.join(ROOT, "tools", "collect_profiles.py"))
    cp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cp)
    csrc = os.path.join(ROOT, "quadruped_locomotion_amd", "csrc")
    text = "".join(open(os.path.join(csrc, f)).read() for f in os.listdir(csrc) if f.endswith(".hip"))
    kernels = set(re.findall(r"__global__[^;{]*?\bvoid\s+(\w+)\s*\(", text))
    for name, kernel, batch, args in cp.WORKLOADS:
        assert kernel in kernels, kernel
    idx = json.load(open(os.path.join(ROOT, "profiles", "r6", "pmc_index.json")))
    have = {(r["kernel"], r["batch"], r["workload"]) for r in idx["records"]}
    assert have == {(k, b, n) for n, k, b, _ in cp.WORKLOADS}
    assert all("fetch_bytes" in r and "write_bytes" in r and "valu_insts" in r for r in idx["records"])
    assert all(r["incomplete"] is False and r["fetch_size_rule"].startswith("FETCH_SIZE") for r in idx["records"])
