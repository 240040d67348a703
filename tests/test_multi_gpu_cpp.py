"""The C++ host of the sharded solve (tests/cpp/multi_gpu_demo.cpp: qlamd.h + HIP runtime + rccl.h, no torch): it builds
in the CPU-only container, its sharding arithmetic is checked there, and on the GPU one rank runs the whole pipeline --
context on the device, solves and RCCL all-gathers on two streams -- and must reproduce the Python path bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, has_gpu

from quadruped_locomotion_amd import synth
from quadruped_locomotion_amd.capi import FIELD_OF_KEY

BIN = os.path.join(ROOT, "tests", "cpp", "multi_gpu_demo")
HIPCC = "/opt/rocm/bin/hipcc"
needs_hipcc = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")


@pytest.fixture(scope="module")
def gpu():
    import torch
    from quadruped_locomotion_amd import capi
    assert torch.cuda.is_available(), "these tests need the MI355X"
    capi.lib()  # raises if the HIP extension is missing: no silent fallback
    ctx = capi.Context(device=0)
    yield capi, ctx, torch
    ctx.close()


def build_demo():
    from quadruped_locomotion_amd import build
    build.build()
    pkg = os.path.join(ROOT, "quadruped_locomotion_amd")
    subprocess.check_call([HIPCC, "-std=c++17", "-O1", "-Wall", "-x", "c++", "-D__HIP_PLATFORM_AMD__",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(pkg, "host"), "-I/opt/rocm/include", "-o", BIN,
                           os.path.join(ROOT, "tests", "cpp", "multi_gpu_demo.cpp"), "-L" + pkg, "-lqlamd",
                           "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + pkg], stderr=subprocess.DEVNULL)


def run(*args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.update(env_extra or {})
    return subprocess.run([BIN, *args], capture_output=True, text=True, env=env, timeout=timeout)


def write_states(path, state):
    """The global batch in the order of qlamd_state_batch, support flags last; a list of states: a trajectory, tick after tick
    (the demo's --ticks)."""
    with open(path, "wb") as f:
        for st in (state if isinstance(state, (list, tuple)) else [state]):
            for key, _field, k in FIELD_OF_KEY:
                f.write(np.ascontiguousarray(st[key], dtype=np.float64).reshape(-1, k).tobytes())
            f.write(np.ascontiguousarray(st["stance"], dtype=np.uint8).tobytes())


@needs_hipcc
def test_demo_builds_and_shards_partition_the_batch():
    build_demo()
    p = run("--selftest-sharding")
    assert p.returncode == 0
    seen = 0
    for line in p.stdout.splitlines():
        w = line.split()
        robots, ranks, slot = int(w[1]), int(w[2]), int(w[4])
        shards = [tuple(int(v) for v in s.split("+")) for s in w[6:]]
        assert len(shards) == ranks
        nxt = 0
        for first, count in shards:  # contiguous, in rank order, nothing lost, nothing twice
            assert first == nxt and 0 < count <= slot
            nxt = first + count
        assert nxt == robots and slot == max(c for _, c in shards)
        seen += 1
    assert seen == 16
    if not has_gpu():
        p = run("--states", "/nonexistent", "--robots", "8")
        assert p.returncode == 3  # no device: the demo, like the library, has no CPU path


@needs_hipcc
def test_id_file_rendezvous_between_two_processes(tmp_path):
    """What ranks > 0 do before ncclCommInitRank: wait for the file rank 0 publishes (written aside and renamed, so that a
    reader sees nothing or all of it).  Two processes go through exchange_id() of host/qlamd/sharded.hpp -- the waiting rank
    started first -- with a 128-byte pattern in place of ncclGetUniqueId's id; no GPU, no RCCL call."""
    import time
    build_demo()
    idf = str(tmp_path / "nccl.id")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    waiter = subprocess.Popen([BIN, "--selftest-rendezvous", "--rank", "1", "--ranks", "2", "--id-file", idf],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    time.sleep(0.5)
    assert waiter.poll() is None          # still waiting: nothing has been published
    p0 = run("--selftest-rendezvous", "--rank", "0", "--ranks", "2", "--id-file", idf)
    out1, err1 = waiter.communicate(timeout=60)
    assert p0.returncode == 0 and waiter.returncode == 0, p0.stderr + err1
    c0, c1 = p0.stdout.split()[-1], out1.split()[-1]
    assert c0 == c1 and p0.stdout.startswith("rank 0 of 2") and out1.startswith("rank 1 of 2")
    want = 0
    for k in range(128):
        want = (want * 131 + (37 * k + 11) % 256) % (1 << 64)
    assert int(c0) == want
    assert not os.path.exists(idf + ".tmp")
    # nobody publishes: the waiting rank gives up with an error instead of hanging (timeout shortened through the file's absence)
    p = run("--selftest-rendezvous", "--rank", "0", "--ranks", "2", "--id-file", str(tmp_path / "no_such_dir" / "id"))
    assert p.returncode == 4


@pytest.mark.gpu
@pytest.mark.parametrize("gather_every", [1, 4])
def test_one_rank_pipeline_equals_the_python_path(gpu, tmp_path, gather_every):
    capi, ctx, torch = gpu
    build_demo()
    B = 4096 + 3  # a ragged last wavefront
    state = synth.make_states(B, "trot")
    states, out = str(tmp_path / "states.bin"), str(tmp_path / "tau.bin")
    write_states(states, state)
    got = None
    for extra in ((), ("--plain",)):  # the placed loop of the header and the plain entry: the same efforts
        p = run("--states", states, "--robots", str(B), "--ranks", "1", "--rank", "0", "--steps", "10",
                "--gather-every", str(gather_every), "--out", out, *extra)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "0 robots with status != ok" in p.stdout and ("plain" if extra else "placed") in p.stdout
        now = np.fromfile(out, dtype=np.float64).reshape(B, 12)
        assert got is None or np.array_equal(got, now)
        got = now
    d = capi.to_device(state)
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ctx.balance_solve_device(d, tau, None, status, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    assert np.array_equal(got, tau.cpu().numpy())  # bit for bit: same library, same kernel, gathered through RCCL
    # the placed loop with the working sets carried from step to step: the same minimisers to the solver's accuracy
    p = run("--states", states, "--robots", str(B), "--ranks", "1", "--rank", "0", "--steps", "10",
            "--gather-every", str(gather_every), "--out", out, "--warm")
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 robots with status != ok" in p.stdout and "placed+warm" in p.stdout
    assert np.abs(np.fromfile(out, dtype=np.float64).reshape(B, 12) - got).max() < 1e-7
    # a trajectory (--ticks): step k solves tick k % T, so the loop's hints come from earlier ticks; the last step's efforts
    # are those of the last tick, whatever the method
    T = 10
    traj = synth.trajectory(B, "trot", T)
    write_states(states, traj)
    dl = capi.to_device(traj[-1])
    ctx.balance_solve_device(dl, tau, None, status, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    want = tau.cpu().numpy()
    # (--graph: the K steps captured into one hipGraph by ShardedBalanceSolver::capture_steps -- solves and in-place all-gathers on
    # two captured streams -- and replayed: the same efforts)
    for extra, tol in ((("--plain",), 0.0), ((), 0.0), (("--warm",), 1e-7), (("--graph",), 0.0), (("--graph", "--warm"), 1e-7)):
        p = run("--states", states, "--robots", str(B), "--ticks", str(T), "--ranks", "1", "--rank", "0", "--steps", str(T),
                "--gather-every", str(gather_every), "--out", out, *extra)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "0 robots with status != ok" in p.stdout
        assert np.abs(np.fromfile(out, dtype=np.float64).reshape(B, 12) - want).max() <= tol, extra
