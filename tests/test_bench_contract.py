"""The one-line JSON contract of bench.py: every key the driver reads, with the types and relations it relies on.
CPU: the committed line of the last measured run (profiles/r5/bench_static_b4096.json).  GPU: short live runs."""
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


ALSO = {"static-calm", "trot", "static-survey-prev-tick-hints", "static-survey-warm", "trot_b8192", "trot_b65536", "pose_sqp_b4096"}
ALSO_KEYS = ("value", "ms_per_step", "kernel_ms", "roofline_frac", "valu_issue_frac", "all_status_ok")


def check_round5(d):
    """What round 5 added to the default line: the placed loop as the method, the same steps through the plain entry beside
    it, every BASELINE config in `also`, and the scale point a weak-scaling curve is drawn from."""
    assert d["config"]["method"] == "placed" and "placement" in d["config"]["method_note"]
    u = d["unplaced"]
    assert u["method"] == "plain" and u["all_status_ok"] is True and u["batch"] == 4096 and u["ms_per_step"] > 0
    assert set(d["also"]) == ALSO
    for name, a in d["also"].items():
        for k in ALSO_KEYS:
            assert k in a, (name, k)
        assert a["all_status_ok"] is True and a["value"] > 0 and a["kernel_ms"] > 0 and 0 < a["roofline_frac"] < 1, name
    assert d["also"]["static-calm"]["tracking_error"] == [0.004, 0.005, 0.01] and d["also"]["static-calm"]["method"] == "plain"
    assert d["also"]["trot_b8192"]["batch"] == 8192 and d["also"]["trot_b65536"]["batch"] == 65536
    assert d["also"]["trot_b8192"]["unplaced"]["ms_per_step"] > 0 and d["also"]["trot_b65536"]["unplaced"]["all_status_ok"] is True
    assert "note" in d["also"]["static-survey-prev-tick-hints"] and d["also"]["static-survey-prev-tick-hints"]["method"] == "placed"
    w = d["also"]["static-survey-warm"]
    assert w["method"] == "warm" and "working set" in w["note"] and w["ms_per_step"] < d["unplaced"]["ms_per_step"]
    assert d["also"]["trot_b8192"]["warm"]["all_status_ok"] is True and d["also"]["trot_b65536"]["warm"]["ms_per_step"] > 0
    assert d["scale_point"]["warm"]["value"] > 0
    assert d["warm_started"] == w   # the line's own workload by the three ways a caller can step it: value, unplaced, warm_started
    sp = d["scale_point"]
    assert sp["robots_per_gpu"] == 8192 and sp["gait"] == "trot" and sp["n_gpus"] == d["n_gpus"]
    assert sp["value"] > 0 and sp["without_gather"] > 0 and "efficiency" in sp["definition"]
    assert abs(sp["value"] - d["also"]["trot_b8192"]["value"]) < 1e-6 * sp["value"]


def test_committed_bench_line_follows_the_contract():
    path = os.path.join(ROOT, "profiles", "r5", "bench_static_b4096.json")
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
    # round 5
    check_round5(d)
    assert d["ms_per_step"] < d["unplaced"]["ms_per_step"]   # the placement pays on the headline preset


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


def _bench(*argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "bench.py must print exactly one line (RCCL's shutdown banner included): %r" % lines
    return lines[0]


@pytest.mark.gpu
def test_live_default_line_is_the_contract_workload_and_carries_the_other_presets():
    """`python bench.py` with no workload flags: BASELINE configs[1] on SURVEY.md 8(d)'s literal tracking errors, and an
    `also` object with static-calm and trot measured in the same process."""
    d = check(_bench("--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--replays", "3"), want_cpu=False)
    assert d["config"]["gait"] == "static" and d["config"]["tracking_error"] == [0.02, 0.05, 0.1]
    check_round5(d)
    for name, a in d["also"].items():
        assert "pmc_source" in a, name
    assert d["also"]["static-calm"]["kernel_ms"] < d["roofline"]["kernel_ms"]     # the calm preset is the lighter one
    # the plain entry as the method: the placed loop rides along instead
    d = check(_bench("--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--replays", "3", "--method", "plain"), want_cpu=False)
    assert d["config"]["method"] == "plain" and d["placed"]["method"] == "placed" and d["placed"]["all_status_ok"] is True


@pytest.mark.gpu
def test_collective_path_with_one_rank():
    """The code path the N > 1 runs take -- RCCL process group, the K steps with their all-gathers captured into one
    hipGraph, gathers on a second captured stream -- exercised with a single rank, then without the gather, then with
    eager launches (what the ranks fall back to when a capture fails)."""
    common = ("--force-collective", "--gait", "trot", "--batch", "8192", "--steps", "20", "--warmup", "3",
              "--no-cpu-baseline", "--replays", "3", "--no-alternatives")
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
    d = check(_bench(*common, "--gather-every", "4", "--no-alternatives"), want_cpu=False)
    c = d["config"]
    assert c["gather_every"] == 4 and c["collect"] == "rccl" and c["gather_layout_ok"] is True and c["all_status_ok"] is True
    assert c["launch"] == "hipGraph of K steps" and "alternatives" not in d
    d = check(_bench(*common, "--collect", "peer", "--no-alternatives"), want_cpu=False)
    c = d["config"]
    assert c["collect"] == "peer" and c["gather_every"] == 1 and c["gather_layout_ok"] is True and c["all_status_ok"] is True
    assert c["result_collection"].startswith("copies of the torque shard") and c["launch"] == "hipGraph of K steps"
    d = check(_bench(*common), want_cpu=False)  # the default: RCCL every step, the others measured beside it
    assert d["config"]["collect"] == "rccl" and d["config"]["gather_every"] == 1
    alt = d["alternatives"]
    assert set(alt) == {"rccl_every_8", "peer_every_1", "peer_every_8"}
    for name, a in alt.items():
        assert "error" not in a, (name, a)
        assert a["value"] > 0 and a["layout_ok"] is True, (name, a)
    # alternatives that do not come back in time cost the line their object only (the watchdog prints the line and ends the rank)
    d = check(_bench(*common, "--alternatives-timeout", "0.05"), want_cpu=False)
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
    idx = json.load(open(os.path.join(ROOT, "profiles", "r5", "pmc_index.json")))
    have = {(r["kernel"], r["batch"], r["workload"]) for r in idx["records"]}
    assert have == {(k, b, n) for n, k, b, _ in cp.WORKLOADS}
    assert all("fetch_bytes" in r and "write_bytes" in r and "valu_insts" in r for r in idx["records"])
    assert all(r["incomplete"] is False and r["fetch_size_rule"].startswith("FETCH_SIZE") for r in idx["records"])
