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
