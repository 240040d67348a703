"""bench.py --collect peer across processes: two ranks on ONE GPU (gloo for the control plane), each maps the other's
hipMalloc'd buffer through an IPC handle and copies its shard into it -- the code path the several-GPU runs take, which a
single RCCL rank cannot exercise (with one rank no handle is exchanged)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = "cuda:0"
    torch.cuda.set_device(0)
    G, B = 2, 24

    class CpuFlags:  # the consensus flags go through gloo: CPU tensors
        int32 = torch.int32

        @staticmethod
        def tensor(v, dtype, device):
            return torch.tensor(v, dtype=dtype)

    pb = bench.PeerBuffers(G * B * 12, rank, world, dev, dist, CpuFlags)
    shard = [torch.full((G, B, 12), 100.0 * rank + buf, dtype=torch.float64, device=dev) +
             torch.arange(G * B * 12, dtype=torch.float64, device=dev).reshape(G, B, 12) * 1e-3 for buf in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    for buf in range(2):
        pb.scatter(buf, shard[buf].data_ptr(), stream)
    torch.cuda.synchronize()
    pb.agree_no_failure()
    dist.barrier()  # every rank's copies have landed
    ok = True
    for buf in range(2):
        for r in range(world):  # slot r of MY buffer holds rank r's shard
            want = (100.0 * r + buf) + torch.arange(G * B * 12, dtype=torch.float64).reshape(G, B, 12) * 1e-3
            ok = ok and pb.slot_equals(buf, r, want)
    np.save(os.path.join(out_dir, "ok%d.npy" % rank), np.array([ok]))
    dist.barrier()
    pb.close()
    dist.destroy_process_group()


def test_two_ranks_exchange_shards_through_ipc_mapped_buffers(tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(tmp_path / ("ok%d.npy" % r))[0])
