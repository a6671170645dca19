"""Multi-GPU sharding of the fragment stream (BASELINE.json config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Fragment windows are independent units of work and shard one-per-rank with no collective
in any per-voxel kernel.  Fragments of the SAME scene are coupled only through GRUFusion's global
map, so there is exactly one exchange step per scale before GRU fusion: every rank contributes the
map voxels it owns that fall inside another rank's fragment bounding volume ("boundary voxels":
int32[n,3] scene-grid coordinates + f32[n,C] features, C = 176 / 88 / 48), by a variable-size
all-gather (counts first, then one padded all-gather per tensor — a few hundred KB to a few MB per
rank: latency-bound, one collective per tensor rather than a ring of small sends).
Received voxels that lie inside the local FBV and are not yet in the local map are appended to it.

The reference has no such step (it fuses fragments strictly sequentially on one GPU,
models/gru_fusion.py:275); parity for this schedule is defined against the un-sharded union, see
tests/test_distributed_cpu.py.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_gather_variable(t, group=None):
    """all-gather of tensors whose first dimension differs per rank -> list (one tensor per rank)"""
    world = dist.get_world_size(group)
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return [b[:c] for b, c in zip(bufs, counts)]


def _pack(c):
    """exact int64 key of int32[n,3] scene-grid coordinates (|c| < 2^20)"""
    c = c.to(torch.int64) + (1 << 20)
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def _inside(c, lo, dim):
    return ((c >= lo) & (c < lo + dim)).all(dim=1)


def exchange_boundary_voxels(map_c, map_f, fbv_lo, dim, group=None):
    """map_c int32[M,3], map_f f32[M,C]: this rank's global map at one scale; fbv_lo int[3]: the local
    fragment's origin in scene-grid units; dim: FBV edge length in voxels of this scale.
    Returns the local map extended by the other ranks' voxels that fall inside the local FBV
    (lower rank wins when several ranks own the same voxel).  Collective: call on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = map_f.device
    lo = torch.as_tensor(fbv_lo, dtype=torch.int32, device=dev).reshape(1, 3)
    boxes = [torch.zeros_like(lo) for _ in range(world)]
    dist.all_gather(boxes, lo, group=group)
    wanted = torch.zeros(map_c.shape[0], dtype=torch.bool, device=dev)
    for r in range(world):
        if r != rank:
            wanted |= _inside(map_c, boxes[r], dim)
    got_c = all_gather_variable(map_c[wanted].contiguous(), group)
    got_f = all_gather_variable(map_f[wanted].contiguous(), group)
    have = _pack(map_c)
    new_c, new_f = [map_c], [map_f]
    for r in range(world):
        if r == rank or got_c[r].shape[0] == 0:
            continue
        c, f = got_c[r], got_f[r]
        keep = _inside(c, lo, dim)
        if keep.any():
            key = _pack(c)
            keep &= ~torch.isin(key, have)
            new_c.append(c[keep])
            new_f.append(f[keep])
            have = torch.cat([have, key[keep]])
    return torch.cat(new_c), torch.cat(new_f)
