"""The peer-write assembly of the walk matrix (pecanpy_amd/sharding.py: PeerRowWriter) on ONE GPU: two processes, gloo for
the control messages, CUDA IPC for the matrix -- rank 1 writes its rows straight into rank 0's allocation.  (RCCL refuses
two ranks on one device, so the multi-GPU form proper cannot run on the one-GPU box; what runs here is everything except
the xGMI hop: the IPC mapping, the row selection, the chunked posts, the completion protocol.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist

    from pecanpy_amd.engine import shard_bounds, tapered_bounds
    from pecanpy_amd.sharding import PeerRowWriter, isolated_row_filler

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    n_rows, L = 50000, 20
    rs = np.random.RandomState(3)
    starts = rs.randint(0, 1000, n_rows).astype(np.uint32)
    known = rs.random_sample(n_rows) < 0.4                      # rows nobody sends (isolated starts)
    want = rs.randint(1, 2**31 - 1, (n_rows, L + 2)).astype(np.int32)
    want[known] = 0
    want[known, 0] = starts[known].view(np.int32)
    want[known, L + 1] = 1
    bounds = shard_bounds(n_rows, world, 0.5)
    pw = PeerRowWriter(n_rows, L + 2, bounds, torch.int32, dev, dst=0, known=known, fill_known=isolated_row_filler(starts, L, dev))
    lo, hi = bounds[rank]
    mine = torch.from_numpy(want[lo:hi]).to(dev)
    for a, b in tapered_bounds(hi - lo, 3):
        if rank == 0:
            pw.own_rows()[a:b] = mine[a:b]
            pw.expect([])
        else:
            pw.post(lo + a, lo + b, mine[a:b])
    full = pw.finish()
    if rank == 0:
        ret["ok"] = bool(np.array_equal(full.cpu().numpy(), want))
    dist.barrier()
    del pw, full
    dist.destroy_process_group()


def test_peer_row_writer_two_processes_one_gpu():
    import torch.multiprocessing as mp

    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["ok"]
