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
    """Reference (torch) form of the per-voxel origin stamps, used by the CPU / gloo tests and the stateless wrapper; on the
    GPU the stamps are a column of the map handle (csrc/global_map.hip) and no volume exists.
    Per scale: a dense int32 volume holding, for every voxel, the index of the fragment that produced its current features
    and whether this rank fused it itself: 0 = unknown, +(stamp + 1) = fused here, -(stamp + 1) = received.
    The volume is anchored at the first voxels it sees (lower corner minus a margin) and GROWS when a later voxel falls
    outside (re-allocation + copy; one host read per put, acceptable on this path): no scene-size limit, no wasted half
    range for the mostly non-negative map coordinates."""

    MARGIN = 16

    def __init__(self, device, extent=None):
        self.device = device
        self.extent = int(extent or 64)         # edge of the first allocation; doubles on overflow
        self.origin = None                      # int64[3] lower corner (host list)
        self.shape = None
        self.grid = None

    def _fit(self, coords):
        lo = coords.min(dim=0).values.tolist()
        hi = coords.max(dim=0).values.tolist()
        if self.grid is None:
            self.origin = [int(v) - self.MARGIN for v in lo]
            self.shape = [max(self.extent, int(h) - int(l) + 1 + 2 * self.MARGIN) for l, h in zip(lo, hi)]
            self.grid = torch.zeros(self.shape, dtype=torch.int32, device=self.device)
            return
        new_lo = [min(o, int(v) - self.MARGIN) if int(v) < o else o for o, v in zip(self.origin, lo)]
        new_hi = [max(o + n, int(v) + 1 + self.MARGIN) if int(v) >= o + n else o + n for o, n, v in zip(self.origin, self.shape, hi)]
        if new_lo == self.origin and all(h == o + n for h, o, n in zip(new_hi, self.origin, self.shape)):
            return
        # grow geometrically so that a scene scanned along one axis re-allocates O(log) times
        shape = [max(h - l, 2 * n if (h - l) > n else n) for l, h, n in zip(new_lo, new_hi, self.shape)]
        grid = torch.zeros(shape, dtype=torch.int32, device=self.device)
        off = [o - l for o, l in zip(self.origin, new_lo)]
        grid[off[0]:off[0] + self.shape[0], off[1]:off[1] + self.shape[1], off[2]:off[2] + self.shape[2]] = self.grid
        self.origin, self.shape, self.grid = new_lo, shape, grid

    def _index(self, coords):
        c = coords.to(torch.int64) - torch.tensor(self.origin, dtype=torch.int64, device=self.device)
        dims = torch.tensor(self.shape, dtype=torch.int64, device=self.device)
        ok = ((c >= 0) & (c < dims)).all(dim=1)
        idx = (c[:, 0] * self.shape[1] + c[:, 1]) * self.shape[2] + c[:, 2]
        return torch.where(ok, idx, torch.zeros_like(idx)), ok

    def put(self, coords, stamp, local):
        """coords int32[n,3] (distinct); stamp int32[n] or int; existing entries are overwritten"""
        n = coords.shape[0]
        if n == 0:
            return
        self._fit(coords)
        idx, _ = self._index(coords)
        val = (stamp.to(torch.int32) if torch.is_tensor(stamp) else torch.full((n,), int(stamp), dtype=torch.int32, device=self.device)) + 1
        self.grid.view(-1)[idx] = val if local else -val

    def get(self, coords):
        """-> (stamp int32[n] (-1 when unknown), local bool[n])"""
        n = coords.shape[0]
        if n == 0 or self.grid is None:
            return (torch.full((n,), -1, dtype=torch.int32, device=self.device), torch.zeros(n, dtype=torch.bool, device=self.device))
        idx, ok = self._index(coords)
        v = torch.where(ok, self.grid.view(-1)[idx], torch.zeros(n, dtype=torch.int32, device=self.device))
        return v.abs() - 1, v > 0

    def local_count(self):
        return 0 if self.grid is None else int((self.grid > 0).sum())


class BoundaryExchange:
    """State and protocol of the boundary-voxel exchange for one GRUFusion (all scales)."""

    def __init__(self, n_scales, device, group=None, extent=None):
        """extent: edge of the first allocation of the finest scale's reference stamp volume (it grows on demand)"""
        self.n_scales, self.device, self.group = n_scales, device, group
        self.extent = int(extent or 128)
        self.stamps = [self.new_stamps(s) for s in range(n_scales)]
        self.collectives = 0        # issued so far (tests / bench reporting)
        self.rows_sent = 0          # payload rows this rank contributed so far
        self.bytes_sent = 0         # ... and their size in the packed payload (4 + C floats per row)

    def new_stamps(self, scale):
        return _Stamps(self.device, max(self.extent >> (self.n_scales - 1 - scale), 32))

    def exchange_handles(self, gmaps, boxes_lo, dims):
        """The exchange on the map HANDLES (GPU): selection, packing and merge are kernels of libeprecon_hip.so
        (eprecon_map_select_boundary_async / pack / merge); the stamps are a column of the map, no full-map export, no
        boolean indexing / sort / searchsorted.  Same protocol and collectives as `exchange`; when no rank has anything to
        send (every rank sees the same counts) the payload all-gather is skipped.  gmaps: per scale a GlobalMap."""
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        dev, ns = self.device, self.n_scales
        lo = torch.as_tensor(boxes_lo, dtype=torch.int32, device=dev).reshape(ns, 3)
        boxes = [torch.zeros_like(lo) for _ in range(world)]
        dist.all_gather(boxes, lo, group=self.group)                                   # collective 1
        all_boxes = torch.stack(boxes)                                                 # [world, ns, 3]
        counts = torch.zeros(ns, dtype=torch.int32, device=dev)
        for s, g in enumerate(gmaps):
            g.select_boundary(all_boxes[:, s].contiguous(), rank, dims[s], counts[s:s + 1])
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=self.group)                          # collective 2
        all_counts = torch.stack(all_counts).tolist()                                  # the one host read
        self.collectives += 2
        widths = [4 + g.channels for g in gmaps]
        sizes = [sum(all_counts[r][s] * widths[s] for s in range(ns)) for r in range(world)]
        if max(sizes) == 0:
            return 0
        payload = torch.empty(max(sizes), dtype=torch.float32, device=dev)
        off = 0
        for s, g in enumerate(gmaps):
            n = all_counts[rank][s]
            g.pack_boundary(payload[off:], n)
            off += n * widths[s]
        self.rows_sent += sum(all_counts[rank])
        self.bytes_sent += 4 * sizes[rank]
        bufs = [torch.empty_like(payload) for _ in range(world)]
        dist.all_gather(bufs, payload, group=self.group)                               # collective 3
        self.collectives += 1
        added = 0
        lo_host = [list(map(int, b)) for b in boxes_lo]
        for s, g in enumerate(gmaps):
            for r in range(world):                                                     # rank order: deterministic appends
                n = all_counts[r][s]
                if r == rank or n == 0:
                    continue
                off = sum(all_counts[r][t] * widths[t] for t in range(s))
                added += g.merge_boundary(bufs[r][off: off + n * widths[s]], n, lo_host[s], dims[s])
        return added

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
        send, counts = [], torch.zeros(ns, dtype=torch.int64, device=dev)
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
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=self.group)                          # collective 2
        all_counts = torch.stack(all_counts).tolist()                                  # the one host read
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
    ex = BoundaryExchange(1, map_f.device, group, extent=32)   # (anchored at the map's own lower corner, grows on demand)
    ex.stamps[0].put(map_c, 0, True)
    (c, f), = ex.exchange([(map_c, map_f)], [list(fbv_lo)], [dim])
    return c, f
