// The body of one SPVCNN pass (models/modules.py:148-175 with the blocks of :15-72 and the point <-> voxel transfers of
// ops/torchsparse_utils.py:40-105) issued from ONE library call: 27 convolutions, their train-mode BatchNorms, 3 scatter-means
// and 3 trilinear devoxelisations — ~115 launches that eprecon_amd/modules.py otherwise issues one by one from Python at ~15 us
// of host time each (here ~3 us).  Nothing new is computed: the call fills the same descriptors with the same rules
// (eprecon_amd/sparse.py: conv_stats / _resolve_map, batchnorm_apply_partials, bn_affine) and calls the same entry points in the
// same order, so the results are bit-identical to the Python-issued pass (tests/test_spvcnn_gpu.py, tests/test_switches_gpu.py).
// What it buys (profiles/r05/cfg4_switches_ab_all.txt): the host runs ~1 ms per pass further ahead of the GPU; a
// fragment that is bound by its kernels (12.8 of ~13.4 ms busy) does not get shorter by it — interleaved A/Bs put the call between
// -0.01 and +0.10 ms per fragment against the Python-issued pass.
// Intermediates live in a caller-provided arena whose size eprecon_spvcnn_forward_workspace_bytes computes by walking the same
// sequence without launching; the arena is managed first-fit with explicit release (see Pass::alloc).
#include <vector>

#include "common.hpp"

namespace {
using namespace ep;

constexpr int kDirectMaxCout = 64;        // eprecon_amd/sparse.py DIRECT_MAX_COUT
constexpr int64_t kK1DirectMinRows = 20000;  // eprecon_amd/sparse.py K1_DIRECT_MIN_ROWS

// conv / BatchNorm slots of the descriptor (eprecon_spvcnn_forward_desc.conv[] / .bn[]): the order of the launches
enum Slot {
    kStem = 0,
    kS1Down, kS1R1a, kS1R1b, kS1R1d, kS1R2a, kS1R2b,
    kS2Down, kS2R1a, kS2R1b, kS2R1d, kS2R2a, kS2R2b,
    kMlp0,
    kU1Up, kU1R1a, kU1R1b, kU1R1d, kU1R2a, kU1R2b,
    kU2Up, kU2R1a, kU2R1b, kU2R1d, kU2R2a, kU2R2b,
    kMlp1,
    kSlots
};
static_assert(kSlots == EPRECON_SPVCNN_CONVS, "slot table out of step with include/eprecon_hip.h");

struct Rows {            // a row-major feature matrix (or a column slice of one)
    float *p;
    int64_t n;
    int c, ld;
};

struct Pass {
    const eprecon_spvcnn_forward_desc *d;
    void *stream;
    bool dry;            // size the arena only: no launch
    char *base;
    size_t used, cap;
    int rc;

    // First-fit allocator over the arena with explicit release: a buffer is handed back as soon as its last consumer is queued
    // (every launch of the pass goes to ONE stream, so a later launch that re-uses the bytes is ordered behind it).  Without the
    // re-use every intermediate of the pass had its own cold address range — ~4x the footprint of the Python-issued pass, whose
    // tensors the caching allocator recycles — and the call measured 0.1 ms per fragment SLOWER than the launches it replaces
    // (profiles/r05/cfg4_switches_ab.txt, first collection).  The dry run replays the same sequence: `peak` is the arena size.
    struct Block { size_t off, bytes; };
    std::vector<Block> free_list, live;
    size_t peak = 0;
    static constexpr uintptr_t kFakeBase = 0x100000;

    void *alloc(size_t bytes)
    {
        bytes = align_up(bytes > 0 ? bytes : 1, 256);
        size_t off = (size_t)-1;
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].bytes >= bytes) {
                off = free_list[i].off;
                if (free_list[i].bytes == bytes) free_list.erase(free_list.begin() + i);
                else { free_list[i].off += bytes; free_list[i].bytes -= bytes; }
                break;
            }
        if (off == (size_t)-1) {
            off = used;
            used += bytes;
            if (used > peak) peak = used;
        }
        live.push_back(Block{off, bytes});
        if (dry) return reinterpret_cast<void *>(kFakeBase + off);     // (never dereferenced)
        if (off + bytes > cap) {
            if (rc == EPRECON_OK) rc = EPRECON_ERR_WORKSPACE;
            return nullptr;
        }
        return base + off;
    }
    void release(const void *p)
    {
        if (!p) return;
        const size_t off = dry ? (size_t)(reinterpret_cast<uintptr_t>(p) - kFakeBase) : (size_t)(static_cast<const char *>(p) - base);
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].off == off) {
                Block b = live[i];
                live.erase(live.begin() + i);
                // insert sorted by offset and merge with the neighbours
                size_t k = 0;
                while (k < free_list.size() && free_list[k].off < b.off) ++k;
                free_list.insert(free_list.begin() + k, b);
                if (k + 1 < free_list.size() && free_list[k].off + free_list[k].bytes == free_list[k + 1].off) {
                    free_list[k].bytes += free_list[k + 1].bytes;
                    free_list.erase(free_list.begin() + k + 1);
                }
                if (k > 0 && free_list[k - 1].off + free_list[k - 1].bytes == free_list[k].off) {
                    free_list[k - 1].bytes += free_list[k].bytes;
                    free_list.erase(free_list.begin() + k);
                }
                if (!free_list.empty() && free_list.back().off + free_list.back().bytes == used) {   // the top of the arena: give it back
                    used = free_list.back().off;
                    free_list.pop_back();
                }
                return;
            }
    }
    void release(const Rows &r) { release(r.p); }
    Rows rows(int64_t n, int c, int ld = 0)
    {
        if (ld == 0) ld = (c + 3) & ~3;      // row pitch rounded up to 4 floats (16-byte gathers), as _segment_mean does
        return Rows{static_cast<float *>(alloc((size_t)(n > 0 ? n : 1) * ld * sizeof(float))), n, c, ld};
    }
    static Rows slice(const Rows &r, int c0, int c) { return Rows{r.p ? r.p + c0 : nullptr, r.n, c, r.ld}; }
    void step(int r)
    {
        if (rc == EPRECON_OK && r != EPRECON_OK && r != EPRECON_EMPTY) rc = r;
    }
    bool ok() const { return rc == EPRECON_OK; }

    // eprecon_amd.sparse.conv_stats: the convolution of slot s with the BatchNorm summaries of its output -> partial rows
    float *conv(int s, const Rows &x, const int32_t *nbr, int64_t n_out, const Rows &out, const float *in_scale,
                const float *in_shift, int64_t *n_partial)
    {
        const eprecon_spvcnn_conv &w = d->conv[s];
        eprecon_conv_desc c = {};
        c.x = x.p; c.n_in = x.n; c.ld_x = x.ld;
        c.nbr = nbr; c.kvol = w.kvol; c.n_out = n_out;
        c.weight = w.weight; c.cin = w.cin; c.cout = w.cout;
        c.out = out.p; c.ld_out = out.ld;
        c.in_scale = in_scale; c.in_shift = in_shift; c.in_relu = in_scale ? 1 : 0;
        // sparse._resolve_map: which operand-order packings ride along (the library picks the kernel)
        if (!nbr && w.kvol == 1 && w.cout <= kDirectMaxCout && x.n >= kK1DirectMinRows) c.packed_weight16 = w.packed_weight16;
        if (nbr && w.kvol == 27 && w.cout <= kDirectMaxCout) c.packed_weight16 = w.packed_weight16;
        if (nbr && w.kvol == 27) c.packed_weight = w.packed_weight;
        if (x.c != w.cin || out.c != w.cout) {
            step(EPRECON_ERR_ARG);
            return nullptr;
        }
        const size_t need = eprecon_conv_desc_workspace_bytes(&c);      // sparse._attach_workspace
        if (need) {
            c.workspace = alloc(need);
            c.workspace_bytes = need;
        }
        int64_t rows_p = eprecon_conv_desc_partial_rows(&c);
        if (rows_p < 1) rows_p = 1;
        float *partial = static_cast<float *>(alloc((size_t)rows_p * 3 * w.cout * sizeof(float)));
        c.bn_partial = partial;
        *n_partial = rows_p;
        if (!dry && ok() && n_out > 0) step(eprecon_conv_desc_async(&c, stream));
        if (need) release(c.workspace);
        return partial;
    }
    // sparse.bn_affine: summaries -> (scale, shift)
    float *affine(int s, const float *partial, int64_t rows_p)
    {
        const eprecon_spvcnn_conv &w = d->conv[s];
        float *aff = static_cast<float *>(alloc((size_t)2 * w.cout * sizeof(float)));
        if (!dry && ok())
            step(eprecon_batchnorm_finalize_affine_async(partial, rows_p, w.cout, d->bn[s].gamma, d->bn[s].beta, d->bn[s].eps, aff,
                                                         aff + w.cout, stream));
        return aff;
    }
    // sparse.batchnorm_apply_partials: second half of the BatchNorm [+ residual (with its own pending BatchNorm)] + ReLU
    void apply(int s, const Rows &x, const float *partial, int64_t rows_p, const Rows *res, const float *res_aff, const Rows &out)
    {
        const int c = d->conv[s].cout;
        const size_t wsb = eprecon_batchnorm_apply_workspace_bytes(c);
        void *ws = alloc(wsb);
        struct Done { Pass *p; void *ws; ~Done() { p->release(ws); } } done{this, ws};
        if (dry || !ok()) return;
        if (res_aff)
            step(eprecon_batchnorm_apply_partials_res_async(x.p, x.n, c, x.ld, partial, rows_p, d->bn[s].gamma, d->bn[s].beta,
                                                            d->bn[s].eps, res->p, res->ld, res_aff, res_aff + c, 1, out.p, out.ld, ws,
                                                            wsb, stream));
        else
            step(eprecon_batchnorm_apply_partials_async(x.p, x.n, c, x.ld, partial, rows_p, d->bn[s].gamma, d->bn[s].beta,
                                                        d->bn[s].eps, res ? res->p : nullptr, res ? res->ld : 0, 1, out.p, out.ld,
                                                        nullptr, nullptr, ws, wsb, stream));
    }
    // BasicConvolutionBlock / BasicDeconvolutionBlock / the stem / a point MLP: conv -> BatchNorm -> ReLU (in place)
    Rows basic(int s, const Rows &x, const int32_t *nbr, int64_t n_out, const Rows *out_slot)
    {
        const Rows y = out_slot ? *out_slot : rows(n_out, d->conv[s].cout, d->conv[s].cout);
        int64_t rp = 0;
        const float *p = conv(s, x, nbr, n_out, y, nullptr, nullptr, &rp);
        apply(s, y, p, rp, nullptr, nullptr, y);
        release(p);
        return y;
    }
    // ResidualBlock: ReLU( [conv-BN-ReLU-conv-BN](x) + [x | conv1x1-BN](x) ); a = first conv, a + 1 = second, a + 2 = the 1x1 skip
    Rows residual(int a, bool has_skip, const Rows &x, const int32_t *nbr, const Rows *out_slot)
    {
        const int64_t n = x.n;
        int64_t rp1 = 0, rp2 = 0, rps = 0;
        // (the block's output first: what is allocated behind it are temporaries, handed back as the block goes)
        const Rows y2 = rows(n, d->conv[a + 1].cout, d->conv[a + 1].cout);
        const Rows out = out_slot ? *out_slot : y2;
        const Rows y = rows(n, d->conv[a].cout, d->conv[a].cout);
        const float *p1 = conv(a, x, nbr, n, y, nullptr, nullptr, &rp1);
        const float *aff = affine(a, p1, rp1);
        release(p1);
        const float *p2 = conv(a + 1, y, nbr, n, y2, aff, aff + d->conv[a].cout, &rp2);
        release(y);
        release(aff);
        if (!has_skip) {
            apply(a + 1, y2, p2, rp2, &x, nullptr, out);
        } else {
            const Rows skip = rows(n, d->conv[a + 2].cout, d->conv[a + 2].cout);
            const float *ps = conv(a + 2, x, nullptr, n, skip, nullptr, nullptr, &rps);
            const float *saff = affine(a + 2, ps, rps);
            release(ps);
            apply(a + 1, y2, p2, rp2, &skip, saff, out);
            release(skip);
            release(saff);
        }
        release(p2);
        if (out_slot) release(y2);
        return out;
    }
    void seg_mean(const Rows &feat, const int32_t *offsets, const int32_t *order, const Rows &out)
    {
        if (!dry && ok())
            step(eprecon_segment_mean_async(feat.p, feat.ld, offsets, order, out.n, feat.c, out.p, out.ld, stream));
    }
    void devox(const Rows &vox, const int32_t *idx8, const float *w8, const Rows &out, int accumulate)
    {
        if (!dry && ok())
            step(eprecon_devoxelize_async(vox.p, vox.ld, idx8, w8, out.n, vox.c, out.p, out.ld, accumulate, stream));
    }

    int run()
    {
        const eprecon_spvcnn_forward_desc &D = *d;
        const int *cs = D.cs;
        const int64_t n = D.n, n1 = D.n1, n2 = D.n2, n4 = D.n4;
        const Rows zf{const_cast<float *>(D.feat), n, D.cin, D.ld_feat};
        // concat buffers: the skip branches are written in place (torchsparse.cat for free)
        const Rows cat0 = rows(n1, cs[4] + cs[0], cs[4] + cs[0]), cat1 = rows(n2, cs[3] + cs[1], cs[3] + cs[1]);
        // x0 = initial_voxelize(z): scatter-mean of the point features into the stride-1 voxels
        const Rows x0 = rows(n1, D.cin);
        seg_mean(zf, D.offsets1, D.order1, x0);
        const Rows f0_slot = slice(cat0, cs[4], cs[0]);
        const Rows f0 = basic(kStem, x0, D.k1, n1, &f0_slot);
        release(x0);
        const Rows z0 = rows(n, cs[0], cs[0]);
        devox(f0, D.idx8_1, D.weight8_1, z0, 0);
        const Rows x1 = rows(n1, cs[0]);
        seg_mean(z0, D.offsets1, D.order1, x1);
        Rows f = basic(kS1Down, x1, D.down12, n2, nullptr);
        release(x1);
        Rows g = residual(kS1R1a, true, f, D.k2, nullptr);
        release(f);
        const Rows f1_slot = slice(cat1, cs[3], cs[1]);
        const Rows f1 = residual(kS1R2a, false, g, D.k2, &f1_slot);
        release(g);
        f = basic(kS2Down, f1, D.down24, n4, nullptr);
        g = residual(kS2R1a, true, f, D.k4, nullptr);
        release(f);
        const Rows f2 = residual(kS2R2a, false, g, D.k4, nullptr);
        release(g);
        // z1 = voxel_to_point(x2, z0) + MLP0(z0.F)
        const Rows z1 = rows(n, cs[2], cs[2]);
        basic(kMlp0, z0, nullptr, n, &z1);
        release(z0);
        devox(f2, D.idx8_4, D.weight8_4, z1, 1);
        release(f2);
        const Rows y3 = rows(n4, cs[2]);
        seg_mean(z1, D.offsets4, D.order4, y3);
        const Rows u1_slot = slice(cat1, 0, cs[3]);
        basic(kU1Up, y3, D.up42, n2, &u1_slot);
        release(y3);
        f = residual(kU1R1a, true, cat1, D.k2, nullptr);
        release(cat1);
        g = residual(kU1R2a, false, f, D.k2, nullptr);
        release(f);
        const Rows u2_slot = slice(cat0, 0, cs[4]);
        basic(kU2Up, g, D.up21, n1, &u2_slot);
        release(g);
        f = residual(kU2R1a, true, cat0, D.k1, nullptr);
        release(cat0);
        g = residual(kU2R2a, false, f, D.k1, nullptr);
        release(f);
        // z3 = voxel_to_point(y4, z1) + MLP1(z1.F) -> the caller's output rows
        const Rows out{D.out, n, cs[4], D.ld_out};
        basic(kMlp1, z1, nullptr, n, &out);
        release(z1);
        devox(g, D.idx8_1, D.weight8_1, out, 1);
        release(g);
        return rc;
    }
};

int check_desc(const eprecon_spvcnn_forward_desc *d)
{
    if (!d || d->n <= 0 || d->n1 <= 0 || d->n2 <= 0 || d->n4 <= 0 || d->cin <= 0 || d->ld_feat < d->cin) return EPRECON_ERR_ARG;
    for (int i = 0; i < 5; ++i)
        if (d->cs[i] <= 0) return EPRECON_ERR_ARG;
    static const int kvol[kSlots] = {27, 8, 27, 27, 1, 27, 27, 8, 27, 27, 1, 27, 27, 1, 8, 27, 27, 1, 27, 27, 8, 27, 27, 1, 27, 27, 1};
    const int *c = d->cs;
    const int cin[kSlots] = {d->cin, c[0], c[0], c[1], c[0], c[1], c[1], c[1], c[1], c[2], c[1], c[2], c[2], c[0],
                             c[2], c[3] + c[1], c[3], c[3] + c[1], c[3], c[3], c[3], c[4] + c[0], c[4], c[4] + c[0], c[4], c[4], c[2]};
    const int cout[kSlots] = {c[0], c[0], c[1], c[1], c[1], c[1], c[1], c[1], c[2], c[2], c[2], c[2], c[2], c[2],
                              c[3], c[3], c[3], c[3], c[3], c[3], c[4], c[4], c[4], c[4], c[4], c[4], c[4]};
    for (int s = 0; s < kSlots; ++s) {
        const eprecon_spvcnn_conv &w = d->conv[s];
        if (!w.weight || w.kvol != kvol[s] || w.cin != cin[s] || w.cout != cout[s]) return EPRECON_ERR_ARG;
        if (w.kvol == 27 && !w.packed_weight) return EPRECON_ERR_ARG;
        if ((w.kvol == 27 || w.kvol == 1) && w.cout <= kDirectMaxCout && !w.packed_weight16) return EPRECON_ERR_ARG;
    }
    return EPRECON_OK;
}

}  // namespace

extern "C" {

size_t eprecon_spvcnn_forward_workspace_bytes(const eprecon_spvcnn_forward_desc *d)
{
    if (check_desc(d) != EPRECON_OK) return 0;
    Pass p{};
    p.d = d; p.dry = true; p.rc = EPRECON_OK;
    p.run();
    return align_up(p.peak, 256) + 256;
}

int eprecon_spvcnn_forward_async(const eprecon_spvcnn_forward_desc *d, void *stream)
{
    int rc = check_desc(d);
    if (rc != EPRECON_OK) return rc;
    if (!d->feat || !d->out || d->ld_out < d->cs[4] || !d->workspace || !d->offsets1 || !d->order1 || !d->offsets4 || !d->order4 ||
        !d->k1 || !d->k2 || !d->k4 || !d->down12 || !d->up21 || !d->down24 || !d->up42 || !d->idx8_1 || !d->weight8_1 || !d->idx8_4 ||
        !d->weight8_4 || (reinterpret_cast<uintptr_t>(d->workspace) & 255))
        return EPRECON_ERR_ARG;
    Pass p{};
    p.d = d; p.stream = stream; p.base = static_cast<char *>(d->workspace); p.cap = d->workspace_bytes; p.rc = EPRECON_OK;
    return p.run();
}

}  // extern "C"
