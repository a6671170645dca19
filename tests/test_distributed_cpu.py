"""CPU, world_size 2, gloo: the boundary-voxel exchange of the multi-GPU schedule (N > 1 path).
After the exchange each rank's map restricted to its FBV equals the un-sharded union of all maps
restricted to that FBV (lower rank wins duplicates); voxels outside the FBV are untouched."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def rank_map(rank, dim=12, c=5):
    rng = np.random.default_rng(100 + rank)
    lo = np.array([0, 0, 0]) if rank == 0 else np.array([8, 0, 0])
    pts = rng.integers(-4, 24, size=(400, 3))
    pts = np.unique(pts, axis=0)
    feat = rng.standard_normal((len(pts), c)).astype(np.float32) + 10 * rank
    return pts.astype(np.int32), feat, lo


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eprecon_amd import distributed as D
    pts, feat, lo = rank_map(rank)
    c, f = D.exchange_boundary_voxels(torch.from_numpy(pts), torch.from_numpy(feat), lo.tolist(), 12)
    parts = D.all_gather_variable(torch.arange(rank + 2, dtype=torch.float32).reshape(-1, 1))
    q.put((rank, c.numpy(), f.numpy(), [p.numpy() for p in parts]))
    dist.barrier()
    dist.destroy_process_group()


def test_boundary_exchange_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, c, f, parts = q.get(timeout=120)
        res[r] = (c, f, parts)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    maps = [rank_map(r) for r in range(2)]
    for r in range(2):
        c, f, parts = res[r]
        assert [len(p) for p in parts] == [2, 3]                     # variable-size all-gather
        own_c, own_f, lo = maps[r]
        other_c, other_f, _ = maps[1 - r]
        assert np.array_equal(c[: len(own_c)], own_c) and np.array_equal(f[: len(own_f)], own_f)
        inside = ((other_c >= lo) & (other_c < lo + 12)).all(1)
        own_keys = {tuple(x) for x in own_c}
        expect = [(tuple(x), y) for x, y, m in zip(other_c, other_f, inside) if m and tuple(x) not in own_keys]
        got = {tuple(x): y for x, y in zip(c[len(own_c):], f[len(own_c):])}
        assert len(got) == len(expect) == len(c) - len(own_c) and len(expect) > 5
        for k, v in expect:
            assert np.array_equal(got[k], v)
