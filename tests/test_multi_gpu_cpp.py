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
                           "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-o", BIN,
                           os.path.join(ROOT, "tests", "cpp", "multi_gpu_demo.cpp"), "-L" + pkg, "-lqlamd",
                           "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + pkg], stderr=subprocess.DEVNULL)


def run(*args, env_extra=None, timeout=300):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.update(env_extra or {})
    return subprocess.run([BIN, *args], capture_output=True, text=True, env=env, timeout=timeout)


def write_states(path, state):
    """The global batch in the order of qlamd_state_batch, support flags last."""
    with open(path, "wb") as f:
        for key, _field, k in FIELD_OF_KEY:
            f.write(np.ascontiguousarray(state[key], dtype=np.float64).reshape(-1, k).tobytes())
        f.write(np.ascontiguousarray(state["stance"], dtype=np.uint8).tobytes())


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


@pytest.mark.gpu
@pytest.mark.parametrize("gather_every", [1, 4])
def test_one_rank_pipeline_equals_the_python_path(gpu, tmp_path, gather_every):
    capi, ctx, torch = gpu
    build_demo()
    B = 4096 + 3  # a ragged last wavefront
    state = synth.make_states(B, "trot")
    states, out = str(tmp_path / "states.bin"), str(tmp_path / "tau.bin")
    write_states(states, state)
    p = run("--states", states, "--robots", str(B), "--ranks", "1", "--rank", "0", "--steps", "10",
            "--gather-every", str(gather_every), "--out", out)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "0 robots with status != ok" in p.stdout
    got = np.fromfile(out, dtype=np.float64).reshape(B, 12)
    d = capi.to_device(state)
    tau = torch.zeros(B, 12, dtype=torch.float64, device="cuda:0")
    status = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    ctx.balance_solve_device(d, tau, None, status, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert (status.cpu().numpy() == 0).all()
    assert np.array_equal(got, tau.cpu().numpy())  # bit for bit: same library, same kernel, gathered through RCCL
