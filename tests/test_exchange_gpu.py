"""The boundary exchange's device side on the map handle (csrc/global_map.hip: select / pack / merge, stamps as a map column)
against a plain numpy restatement of the protocol of eprecon_amd/distributed.py — the schedule that emulates the sequential
map updates of models/gru_fusion.py:195-215,275 across ranks.  The merge is fed the payload of a SYNTHETIC second (and third)
rank at cfg4 map sizes: present / absent voxels, newer / older stamps, the same voxel from two senders, rows outside the box."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

DEV = "cuda"


def make_map(rng, n, channels, extent):
    from eprecon_amd.global_map import GlobalMap
    cells = rng.choice(extent ** 3, size=n, replace=False)
    c = np.stack([cells // (extent * extent), (cells // extent) % extent, cells % extent], 1).astype(np.int32) - 7
    f = rng.standard_normal((n, channels)).astype(np.float32)
    stamp = rng.integers(0, 9, n).astype(np.int32) * rng.choice([-1, 0, 1], n, p=[0.3, 0.1, 0.6]).astype(np.int32)
    g = GlobalMap(channels, torch.device(DEV))
    g.set(torch.from_numpy(c), torch.from_numpy(f))
    g.set_stamps(torch.from_numpy(stamp))
    return g, c, f, stamp


def inside(c, lo, dim):
    return np.all((c >= lo) & (c < lo + dim), axis=1)


@pytest.mark.parametrize("n,channels,dim", [(60000, 48, 96), (9000, 176, 24), (1, 8, 4)])
def test_select_and_pack_match_the_definition(n, channels, dim):
    rng = np.random.default_rng(n)
    g, c, f, stamp = make_map(rng, n, channels, max(dim + 30, 16))
    boxes = np.array([[0, 0, 0], [dim // 2, 3, -5], [-4, dim // 3, dim // 2]], np.int32)   # box 0 is this rank's own
    count = torch.zeros(1, dtype=torch.int32, device=DEV)
    g.select_boundary(torch.from_numpy(boxes).to(DEV), 0, dim, count)
    want = (stamp > 0) & (inside(c, boxes[1], dim) | inside(c, boxes[2], dim))
    k = int(count.item())
    assert k == int(want.sum())
    payload = torch.full((max(k, 1) * (4 + channels),), float("nan"), device=DEV)
    g.pack_boundary(payload, k)
    rows = payload[: k * (4 + channels)].reshape(k, 4 + channels).cpu()
    assert np.array_equal(rows[:, :3].contiguous().view(torch.int32).numpy(), c[want])          # map order
    assert np.array_equal(rows[:, 3].contiguous().view(torch.int32).numpy(), stamp[want] - 1)   # fragment index
    assert np.array_equal(rows[:, 4:].numpy(), f[want])
    # nothing to select: the own box only
    g.select_boundary(torch.from_numpy(boxes[:1]).to(DEV), 0, dim, count)
    assert int(count.item()) == 0


def reference_merge(c, f, stamp, blocks, lo, dim):
    """numpy restatement: per cell the newest received copy; overwrite when newer than the local row, append when absent
    (in payload order, blocks in rank order)"""
    key = {tuple(v): i for i, v in enumerate(c)}
    c, f, stamp = list(map(tuple, c)), [r.copy() for r in f], list(stamp)
    for rc, rs, rf in blocks:                      # one merge call per sender
        ok = inside(rc, lo, dim)
        best = {}
        for i in np.nonzero(ok)[0]:
            k = tuple(rc[i])
            if k not in best or rs[i] > rs[best[k]]:
                best[k] = i
        for i in sorted(best.values()):            # payload order
            k = tuple(rc[i])
            if k in key:
                j = key[k]
                if rs[i] + 1 > abs(stamp[j]):
                    f[j], stamp[j] = rf[i].copy(), -(rs[i] + 1)
            else:
                key[k] = len(c)
                c.append(k); f.append(rf[i].copy()); stamp.append(-(rs[i] + 1))
    return np.array(c, np.int32), np.stack(f), np.array(stamp, np.int32)


@pytest.mark.parametrize("n,channels,dim", [(80000, 48, 96), (30000, 88, 48), (6000, 176, 24)])
def test_merge_of_a_synthetic_second_ranks_payload(n, channels, dim):
    rng = np.random.default_rng(channels)
    g, c, f, stamp = make_map(rng, n, channels, dim + 20)
    lo = np.array([3, -2, 5], np.int32)
    blocks = []
    for sender in range(2):
        m = n // 3
        pick = rng.choice(n, m // 2, replace=False)                                     # voxels the local map holds
        fresh = (rng.integers(-10, dim + 12, (m, 3))).astype(np.int32)                  # mostly new ones, some outside the box
        rc = np.concatenate([c[pick], fresh])
        rc = rc[np.unique(rc, axis=0, return_index=True)[1]]                            # a sender sends a voxel once
        rng.shuffle(rc)
        rs = rng.integers(0, 12, len(rc)).astype(np.int32)
        rf = rng.standard_normal((len(rc), channels)).astype(np.float32)
        blocks.append((rc, rs, rf))
    # sender 1 repeats some of sender 0's voxels with other stamps
    dup = blocks[0][0][: len(blocks[0][0]) // 4]
    blocks[1] = (np.concatenate([blocks[1][0], dup]), np.concatenate([blocks[1][1], rng.integers(0, 12, len(dup)).astype(np.int32)]),
                 np.concatenate([blocks[1][2], rng.standard_normal((len(dup), channels)).astype(np.float32)]))
    keep = np.unique(blocks[1][0], axis=0, return_index=True)[1]
    blocks[1] = tuple(a[np.sort(keep)] for a in blocks[1])
    ref_c, ref_f, ref_s = reference_merge(c, f, stamp, blocks, lo, dim)
    added = 0
    for rc, rs, rf in blocks:
        pay = np.concatenate([rc.view(np.float32), rs.view(np.float32)[:, None], rf], 1)
        added += g.merge_boundary(torch.from_numpy(np.ascontiguousarray(pay)).to(DEV), len(rc), lo, dim)
    assert g.size == len(ref_c) and added == len(ref_c) - n
    got_c, got_f = g.export()
    assert np.array_equal(got_c.cpu().numpy(), ref_c)            # appended rows in payload order
    assert np.array_equal(got_f.cpu().numpy(), ref_f)
    assert np.array_equal(g.stamps().cpu().numpy(), ref_s)


def test_update_stamps_the_rows_it_appends_and_compaction_keeps_the_others():
    """crop_union + update (the GRU bookkeeping) with a fragment index set: union rows become 'fused here', rows outside the
    FBV keep their stamps, and a second exchange would select exactly the new rows inside the other rank's box"""
    from eprecon_amd.global_map import GlobalMap
    rng = np.random.default_rng(4)
    g, c, f, stamp = make_map(rng, 20000, 24, 60)
    dim, rel = 24, [5, 6, 7]
    cur = np.unique(rng.integers(0, dim, (3000, 3)), axis=0).astype(np.int32)
    cur_c = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.zeros((len(cur), 1), np.int32), cur], 1))).to(DEV)
    cur_f = torch.randn((len(cur), 24), device=DEV)
    updated, _, _, _ = g.crop_union(cur_c, cur_f, dim, 1, rel)
    g.set_fragment(41)
    g.update(updated, torch.randn((updated.shape[0], 24), device=DEV))
    out_c, _ = g.export()
    s = g.stamps().cpu().numpy()
    outside = ~inside(c, np.array(rel), dim)
    assert np.array_equal(out_c.cpu().numpy()[: outside.sum()], c[outside]) and np.array_equal(s[: outside.sum()], stamp[outside])
    assert np.all(s[outside.sum():] == 42) and len(s) == outside.sum() + updated.shape[0]


@pytest.mark.parametrize("world", [2, 4])
def test_virtual_ranks_follow_the_simulated_schedule(monkeypatch, world):
    """BoundaryExchange.exchange_handles at WORLD SIZE 2 and 4 on one GPU (4: uneven per-rank voxel counts and a rank that never
    overlaps anybody — it sends and receives nothing but takes every collective, and the all-counts-zero skip of the payload
    all-gather is taken by all ranks or none): threads stand in for the ranks (a fake
    `torch.distributed` shuttles their all-gathers through a mailbox), each with its own three map handles, streaming four
    overlapping fragments per rank through exchange -> crop_union -> fuse -> update.  Every rank's maps (coordinates AND
    features) must equal the single-process simulation of the "independent windows + exchange" schedule that the gloo test
    checks the torch form against (tests/test_distributed_cpu.py: newest fusion wins, received voxels are not re-broadcast)."""
    import threading
    from eprecon_amd import distributed as D
    from eprecon_amd.global_map import GlobalMap
    from eprecon_amd.gru_fusion import gather_rows
    from test_distributed_cpu import CH, DIMS, STEPS, fragment, simulate_schedule

    dev = torch.device("cuda", 0)

    class FakeDist:
        def __init__(self):
            self.barrier = threading.Barrier(world)
            self.box = [None] * world
            self.local = threading.local()

        def get_world_size(self, group=None):
            return world

        def get_rank(self, group=None):
            return self.local.rank

        def all_gather(self, outs, t, group=None):
            self.box[self.local.rank] = t.clone()
            torch.cuda.synchronize()
            self.barrier.wait()
            for o, src in zip(outs, self.box):
                o.copy_(src)
            torch.cuda.synchronize()
            self.barrier.wait()

    fake = FakeDist()
    monkeypatch.setattr(D, "dist", fake)
    results, errors = [None] * world, []

    def rank_main(rank):
        try:
            fake.local.rank = rank
            torch.cuda.set_device(0)
            ex = D.BoundaryExchange(3, dev)
            gmaps = [GlobalMap(CH[s], dev) for s in range(3)]
            for step in range(STEPS):
                frs = [fragment(rank, step, s, world) for s in range(3)]
                ex.exchange_handles(gmaps, [fr[0].tolist() for fr in frs], DIMS)
                for s in range(3):
                    lo, cc, cf = frs[s]
                    g = gmaps[s]
                    cur_c = torch.from_numpy(np.ascontiguousarray(np.concatenate([np.zeros((len(cc), 1), np.int32), cc - lo.astype(np.int32)], 1))).to(dev)
                    cur_f = torch.from_numpy(cf).to(dev)
                    updated, src_cur, src_glob, _ = g.crop_union(cur_c, cur_f, DIMS[s], 1, lo.tolist())
                    old = g.gather(src_glob, 0, CH[s], torch.empty((updated.shape[0], CH[s]), device=dev))
                    cur = gather_rows(cur_f, src_cur, CH[s])
                    g.set_fragment(step * world + rank)
                    g.update(updated, (0.5 * old + cur).contiguous())          # toy_fuse of the gloo test, on the handle
            out = []
            for g in gmaps:
                c, f = g.export()
                out.append({tuple(k): row for k, row in zip(c.cpu().numpy().tolist(), f.cpu().numpy())})
            results[rank] = (out, ex.collectives, ex.rows_sent)
        except Exception as exc:  # noqa: BLE001
            errors.append((rank, repr(exc)))
            try:
                fake.barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    want, rows_sent = simulate_schedule(world)
    assert sum(results[r][2] for r in range(world)) > 0          # (world 2: only the later fragment's rank has rows to send)
    assert [results[r][2] for r in range(world)] == rows_sent    # every rank sent exactly the rows the schedule says
    if world == 4:
        assert rows_sent[3] == 0 and len(set(rows_sent)) == 4    # a silent rank, three uneven senders
    assert len({results[r][1] for r in range(world)}) == 1       # the same number of collectives on every rank
    for r in range(world):
        maps, collectives, sent = results[r]
        assert 2 * STEPS < collectives <= 3 * STEPS                # the payload all-gather ran whenever somebody had rows
        for s in range(3):
            assert set(maps[s]) == set(want[r][s]), (r, s, len(maps[s]), len(want[r][s]))
            worst = max(float(np.abs(maps[s][k] - want[r][s][k]).max()) for k in maps[s])
            assert worst < 1e-5, (r, s, worst)
