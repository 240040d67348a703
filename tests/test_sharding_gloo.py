"""Multi-rank path on CPU (gloo, world size 2): contiguous robot shards per rank, no data-path
collective, one all-gather of the torque shards for result collection -- the same calls bench.py
makes over RCCL.  The per-rank solve is stood in by the oracle (no GPU here); what is under test is
the sharding arithmetic, the shard-stable synthetic generator and the gather layout."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    from quadruped_locomotion_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    state = synth.make_states(B, "trot", offset=rank * B)       # bench.py: rank r owns [r*B, (r+1)*B)
    tau, _, status = O.balance_batch(state)
    assert (status == 0).all()
    shard = torch.from_numpy(tau)
    gathered = torch.zeros(world * B, 12, dtype=torch.float64)
    work = dist.all_gather_into_tensor(gathered, shard, async_op=True)  # as in bench.py
    work.wait()
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == float(world)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_equal_one_global_batch(oracle, tmp_path):
    from quadruped_locomotion_amd import synth
    world, B = 2, 96
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    gathered = np.load(tmp_path / "gathered.npy")
    whole = synth.make_states(world * B, "trot")
    tau, _, _ = oracle.balance_batch(whole)
    assert np.array_equal(gathered, tau)  # rank order == robot order, bitwise


def _grouped_worker(rank, world, port, B, G, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py --gather-every G: the torques of G steps sit in one [G][B][12] buffer and travel with one all-gather into
    # [world][G][B][12]; the stand-in value of (rank, step, robot, joint) makes the layout readable
    tau = torch.zeros(G, B, 12, dtype=torch.float64)
    for g in range(G):
        tau[g] = rank * 1e6 + g * 1e3 + torch.arange(B * 12, dtype=torch.float64).reshape(B, 12) * 1e-3
    gathered = torch.zeros(world, G, B, 12, dtype=torch.float64)
    dist.all_gather_into_tensor(gathered.view(world * G * B, 12), tau.view(G * B, 12))
    assert torch.equal(gathered[rank], tau)
    if rank == 0:
        np.save(os.path.join(out_dir, "grouped.npy"), gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_grouped_gather_layout(tmp_path):
    world, B, G = 2, 5, 3
    mp.spawn(_grouped_worker, args=(world, _free_port(), B, G, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "grouped.npy")
    for r in range(world):
        for g in range(G):
            want = r * 1e6 + g * 1e3 + np.arange(B * 12).reshape(B, 12) * 1e-3
            assert np.array_equal(got[r, g], want)


def _run_bench(*argv, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=600, cwd=ROOT, env=e)


def test_bench_launches_its_own_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` without torch.distributed.run around it: the parent starts the two ranks as a child
    process; here over gloo with the launcher self-test standing in for the solve (no GPU in this container).  The
    N > 1 defaults are BASELINE configs[3]: 8192 trot robots per GPU."""
    import json
    out = _run_bench("--gpus", "2", "--selftest-launcher")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["world_size"] == 2 and d["ranks_seen"] == 2 and d["gather_layout_ok"] and d["max_over_ranks_ok"]
    assert d["robots_per_rank"] == 8192 and d["gait"] == "trot"


def test_bench_reports_a_failed_rank_with_a_nonzero_exit():
    """Without a GPU the ranks refuse to run (no CPU fallback); the parent must hand that failure on."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    out = _run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64")
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
