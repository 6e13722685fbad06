"""CPU test of the N>1 path with torch.distributed (gloo, world_size 2): the source is split by the
library's own shard arithmetic (symmicp_shard_range, the partition symmicp_set_source applies), every
rank reduces its share (CPU oracle standing in for the GPU pass, which needs a device), the 40-double
records are summed with one all-reduce -- the exchange step the GPU path does with RCCL -- and every
rank runs the identical host solve (symmicp_solve).  Checks: shards tile the cloud exactly once, the
all-reduced record equals the single-process record, all ranks get the same 4x4, and that 4x4 equals
the single-process one."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, GOLDEN


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    for p in (ROOT, os.path.join(ROOT, "icp-symm_amd", "py")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import symmicp
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    src, _ = O.pcd_read(os.path.join(GOLDEN, "cat.pcd"))
    tgt, _ = O.pcd_read(os.path.join(GOLDEN, "cat_out.pcd"))
    g = np.load(os.path.join(GOLDEN, "cat_golden.npz"))
    sn, tn = g["src_n"], g["tgt_n"]
    n = src.shape[0]
    b, c = symmicp.shard_range(n, world, rank)
    # this rank's share, nearest-neighbour pairs against the replicated target
    idx, _ = O.nn_brute(src[b:b + c], tgt)
    pivot = tgt.astype(np.float64).mean(0).astype(np.float32)
    S = O.reduce40(src[b:b + c], sn[b:b + c], tgt, tn, idx=idx, pivot=pivot)
    t = torch.from_numpy(S.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)           # the one exchange step of the path
    st, pb, qb, a, tt, rc, X = symmicp.solve(symmicp.MODE_PAPER, t.numpy(), pivot)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), begin=b, count=c, sums=t.numpy(), X=X, status=st)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reduce_allreduce_solve_gloo(world, tmp_path, oracle, cat):
    import torch.multiprocessing as mp
    import symmicp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    outs = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    # shards tile [0, n) exactly once, in rank order
    n = cat["src"].shape[0]
    assert int(outs[0]["begin"]) == 0
    for r in range(world - 1):
        assert int(outs[r]["begin"]) + int(outs[r]["count"]) == int(outs[r + 1]["begin"])
    assert int(outs[-1]["begin"]) + int(outs[-1]["count"]) == n
    # single-process record and solve
    idx, _ = oracle.nn_brute(cat["src"], cat["tgt"])
    pivot = cat["tgt"].astype(np.float64).mean(0).astype(np.float32)
    S = oracle.reduce40(cat["src"], cat["src_n"], cat["tgt"], cat["tgt_n"], idx=idx, pivot=pivot)
    st, pb, qb, a, t, rc, X = symmicp.solve(symmicp.MODE_PAPER, S, pivot)
    for o in outs:
        assert int(o["status"]) == 0
        assert np.abs(o["sums"] - S).max() <= 1e-12 * np.abs(S).max()
        assert np.array_equal(o["X"], outs[0]["X"])               # every rank solves identically
        assert np.abs(o["X"] - X).max() < 1e-6


def test_shard_range_edge_cases():
    import symmicp
    for n, w in ((0, 1), (1, 4), (7, 8), (1_000_000, 8), (8_000_001, 3)):
        tot = 0
        prev_end = 0
        for r in range(w):
            b, c = symmicp.shard_range(n, w, r)
            assert b == prev_end
            prev_end = b + c
            tot += c
        assert tot == n
    with pytest.raises(symmicp.SymmIcpError):
        symmicp.shard_range(10, 2, 2)
