"""Multi-GPU sharding of the fragment stream (BASELINE.json config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Fragment windows are independent units of work and shard one-per-rank with no collective
in any per-voxel kernel.  Fragments of the SAME scene are coupled only through GRUFusion's global
map, so there is exactly one exchange step per fragment, before GRU fusion, for all three scales
at once (`BoundaryExchange.exchange`):

  1. all-gather of the ranks' fragment bounding volumes (3 scales x 3 ints);
  2. every rank selects the map voxels IT FUSED ITSELF (never ones it merely received) that fall
     inside another rank's volume, at every scale; all-gather of the per-scale counts;
  3. ONE padded all-gather of a packed payload: per voxel (x, y, z, stamp, features), the three
     scales back to back, integers bit-cast into the float32 buffer.
  A received voxel inside the local volume replaces the local copy when its stamp — the global
  index of the fragment that fused it — is NEWER (or the voxel is absent); it is marked "received"
  so that it is not re-broadcast.  After the local fusion the union voxels are stamped with the
  local fragment index and become "locally fused".

Three collectives per fragment (the payload is a few hundred KB to a few MB per rank: latency-bound,
so one direct all-gather over the fully connected xGMI links rather than a ring of small sends), one
host read (the counts).  The reference has no such step (it fuses fragments strictly sequentially on
one GPU, models/gru_fusion.py:275); parity for this schedule is defined against a single-process
simulation of the same schedule, tests/test_distributed_cpu.py.
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


def pack_key(c):
    """exact int64 key of int32[n,3] scene-grid coordinates (|c| < 2^20)"""
    c = c.to(torch.int64) + (1 << 20)
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


def _inside(c, lo, dim):
    return ((c >= lo) & (c < lo + dim)).all(dim=1)


class _Stamps:
    """Per scale: a dense int32 volume over the scene grid holding, for every voxel, the index of the fragment that
    produced its current features and whether this rank fused it itself: 0 = unknown, +(stamp + 1) = fused here,
    -(stamp + 1) = received.  (A sorted key table cost three sorts of the whole key set per fragment; 288 GB of HBM make the
    dense form the cheap one: `extent`^3 x 4 bytes, 512 MB for a 20 m scene at 4 cm, allocated on first use.)
    Coordinates outside [-extent/2, extent/2) set `bad`, which the exchange reports at its one host read."""

    def __init__(self, device, extent=None):
        self.device = device
        self.extent = int(extent or (512 if device.type == "cuda" else 128))
        self.grid = None
        self.bad = torch.zeros((), dtype=torch.bool, device=device)

    def _index(self, coords):
        e, h = self.extent, self.extent // 2
        c = coords.to(torch.int64) + h
        ok = ((c >= 0) & (c < e)).all(dim=1)
        idx = (c[:, 0] * e + c[:, 1]) * e + c[:, 2]
        return torch.where(ok, idx, torch.zeros_like(idx)), ok

    def put(self, coords, stamp, local):
        """coords int32[n,3] (distinct); stamp int32[n] or int; existing entries are overwritten"""
        n = coords.shape[0]
        if n == 0:
            return
        if self.grid is None:
            self.grid = torch.zeros(self.extent ** 3, dtype=torch.int32, device=self.device)
        idx, ok = self._index(coords)
        self.bad |= ~ok.all()
        val = (stamp.to(torch.int32) if torch.is_tensor(stamp) else torch.full((n,), int(stamp), dtype=torch.int32, device=self.device)) + 1
        val = val if local else -val
        self.grid[idx] = torch.where(ok, val, self.grid[idx])

    def get(self, coords):
        """-> (stamp int32[n] (-1 when unknown), local bool[n])"""
        n = coords.shape[0]
        if n == 0 or self.grid is None:
            return (torch.full((n,), -1, dtype=torch.int32, device=self.device), torch.zeros(n, dtype=torch.bool, device=self.device))
        idx, ok = self._index(coords)
        v = torch.where(ok, self.grid[idx], torch.zeros_like(self.grid[idx]))
        return v.abs() - 1, v > 0

    def local_count(self):
        return 0 if self.grid is None else int((self.grid > 0).sum())


class BoundaryExchange:
    """State and protocol of the boundary-voxel exchange for one GRUFusion (all scales)."""

    def __init__(self, n_scales, device, group=None, extent=None):
        """extent: edge of the finest scale's stamp volume in voxels (coarser scales halve it), default 512 on a GPU"""
        self.n_scales, self.device, self.group = n_scales, device, group
        self.extent = int(extent or (512 if device.type == "cuda" else 128))
        self.stamps = [self.new_stamps(s) for s in range(n_scales)]
        self.collectives = 0        # issued so far (tests / bench reporting)

    def new_stamps(self, scale):
        return _Stamps(self.device, max(self.extent >> (self.n_scales - 1 - scale), 32))

    def reset(self):
        self.stamps = [self.new_stamps(s) for s in range(self.n_scales)]

    def mark_fused(self, scale, coords, fragment_index):
        """the voxels `coords` (int32[n,3], scene grid of `scale`) were just fused locally by fragment `fragment_index`"""
        self.stamps[scale].put(coords, fragment_index, True)

    def exchange(self, maps, boxes_lo, dims):
        """maps: per scale (C int32[M,3], F f32[M,C_s]); boxes_lo: per scale int[3] (local FBV origin, scene grid);
        dims: per scale FBV edge length.  Returns the per-scale maps extended / refreshed by the other ranks'
        locally-fused voxels inside the local FBV.  Collective: every rank calls it once per fragment."""
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        dev, ns = self.device, self.n_scales
        lo = torch.as_tensor(boxes_lo, dtype=torch.int32, device=dev).reshape(ns, 3)
        boxes = [torch.zeros_like(lo) for _ in range(world)]
        dist.all_gather(boxes, lo, group=self.group)                                   # collective 1
        send, counts = [], torch.zeros(ns + 1, dtype=torch.int64, device=dev)   # last slot: a stamp volume overflowed
        for s, (c, f) in enumerate(maps):
            stamp, local = self.stamps[s].get(c)
            wanted = torch.zeros(c.shape[0], dtype=torch.bool, device=dev)
            for r in range(world):
                if r != rank:
                    wanted |= _inside(c, boxes[r][s], dims[s])
            wanted &= local
            cs, fs, ss = c[wanted], f[wanted], stamp[wanted]
            block = torch.cat([cs.contiguous().view(torch.float32), ss.contiguous().view(torch.float32).unsqueeze(1),
                               fs.float()], dim=1)                                     # [n, 4 + C_s], ints bit-cast
            send.append(block.reshape(-1))
            counts[s] = cs.shape[0]
            counts[ns] += self.stamps[s].bad.to(torch.int64)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=self.group)                          # collective 2
        all_counts = torch.stack(all_counts).tolist()                                  # the one host read
        if any(row[ns] for row in all_counts):   # every rank sees it and raises: no one is left waiting in a collective
            raise RuntimeError("boundary exchange: a map voxel lies outside the stamp volume (scene larger than "
                               f"{self.extent} finest voxels per axis); construct BoundaryExchange with a larger extent")
        widths = [4 + f.shape[1] for _, f in maps]
        sizes = [sum(all_counts[r][s] * widths[s] for s in range(ns)) for r in range(world)]
        cap = max(max(sizes), 1)
        payload = torch.zeros(cap, dtype=torch.float32, device=dev)
        mine = torch.cat(send) if send else payload[:0]
        payload[: mine.numel()] = mine
        bufs = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(bufs, payload, group=self.group)                               # collective 3
        self.collectives += 3
        out = []
        for s, (c, f) in enumerate(maps):
            width = widths[s]
            got_c, got_f, got_s = [], [], []
            for r in range(world):
                n = all_counts[r][s]
                if r == rank or n == 0:
                    continue
                off = sum(all_counts[r][t] * widths[t] for t in range(s))
                block = bufs[r][off: off + n * width].reshape(n, width)
                rc = block[:, :3].contiguous().view(torch.int32)
                keep = _inside(rc, lo[s], dims[s])
                got_c.append(rc[keep])
                got_s.append(block[:, 3].contiguous().view(torch.int32)[keep])
                got_f.append(block[:, 4:][keep])
            if not got_c or sum(x.shape[0] for x in got_c) == 0:
                out.append((c, f))
                continue
            rc, rs, rf = torch.cat(got_c), torch.cat(got_s), torch.cat(got_f)
            # several ranks may send the same voxel: keep the newest per key
            key = pack_key(rc)
            order = torch.sort(rs.to(torch.int64), descending=True, stable=True)[1]
            order = order[torch.sort(key[order], stable=True)[1]]
            key_o = key[order]
            first = torch.ones_like(key_o, dtype=torch.bool)
            first[1:] = key_o[1:] != key_o[:-1]
            sel = order[first]
            rc, rs, rf, key = rc[sel], rs[sel], rf[sel], key[sel]
            # against the local copy: newer wins, absent voxels are appended
            my_key = pack_key(c)
            my_stamp, _ = self.stamps[s].get(c)
            if my_key.numel():
                srt, perm = torch.sort(my_key)
                pos = torch.searchsorted(srt, key).clamp(max=srt.numel() - 1)
                present = srt[pos] == key
                row = perm[pos]
            else:
                present = torch.zeros_like(key, dtype=torch.bool)
                row = torch.zeros_like(key)
            newer = present & (rs > my_stamp[row])
            f = f.clone()
            f[row[newer]] = rf[newer].to(f.dtype)
            add = ~present
            new_c, new_f = torch.cat([c, rc[add]]), torch.cat([f, rf[add].to(f.dtype)])
            take = newer | add
            self.stamps[s].put(rc[take], rs[take], False)
            out.append((new_c, new_f))
        return out


def exchange_boundary_voxels(map_c, map_f, fbv_lo, dim, group=None):
    """One-scale, stateless form kept for callers that hold a plain (C, F) map: every voxel counts as locally
    fused with stamp 0, so voxels already present locally are left alone (first owner wins) and missing ones
    inside the local FBV are appended."""
    ex = BoundaryExchange(1, map_f.device, group)
    ex.stamps[0].put(map_c, 0, True)
    (c, f), = ex.exchange([(map_c, map_f)], [list(fbv_lo)], [dim])
    return c, f
