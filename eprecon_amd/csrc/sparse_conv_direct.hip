// Direct gather form of the 3x3x3 sparse convolution: eligibility rule and dispatch.  The kernels and their launchers live in
// sparse_conv_direct_impl.hpp and are instantiated per column-tile count by sparse_conv_direct_ct{1..4}.hip.
#include "sparse_conv_direct_impl.hpp"

namespace epconv {
int launch_direct16_ct1(const ConvParams &p, hipStream_t st);
int launch_direct16_ct2(const ConvParams &p, hipStream_t st);
int launch_direct16_ct3(const ConvParams &p, hipStream_t st);
int launch_direct16_ct4(const ConvParams &p, hipStream_t st);

// EPRECON_CONV_DIRECT=0: the LDS-resident kernels for every long list (read per launch: tests flip it)
bool direct16_ok(const ConvParams &p)
{
    const char *e = getenv("EPRECON_CONV_DIRECT");
    if (e && e[0] == '0') return false;
    // 3x3x3 kernel maps, the 3x3 pixel maps of the dense 2D stack, and point-wise layers on long lists (K = 1, identity map: a
    // streaming [N, C_in] x [C_in, C_out] product) — the caller packs the weights only where it wants this kernel
    const bool pointwise = p.K == 1 && !p.nbr;
    if (!pointwise && ((p.K != 27 && p.K != 9) || !p.nbr)) return false;
    if (!p.wq16 || (reinterpret_cast<uintptr_t>(p.wq16) & 15) != 0) return false;
    if (p.Cout > 64 || p.accumulate) return false;
    if (p.ld_x % 4 != 0 || (reinterpret_cast<uintptr_t>(p.x) & 15) != 0 || (p.Cin % 4 != 0 && p.ld_x < ((p.Cin + 3) & ~3))) return false;
    if (p.x_bytes <= 0 || p.x_bytes >= 0x7fffffffll || (int64_t)p.ld_x * 4 >= (1 << 24) || p.x_bytes / ((int64_t)p.ld_x * 4) >= (1 << 24))
        return false;
    return true;
}

// rows of one BatchNorm summary block of the launch launch_direct16 would make (eprecon_conv_desc_partial_rows)
int direct16_partial_block_rows(const ConvParams &p) { return persist_ok(p) ? kJobRows : kDirectRows; }

int launch_direct16(const ConvParams &p, hipStream_t st)
{
    switch ((p.Cout + 15) / 16) {
        case 1: return launch_direct16_ct1(p, st);
        case 2: return launch_direct16_ct2(p, st);
        case 3: return launch_direct16_ct3(p, st);
        default: return launch_direct16_ct4(p, st);
    }
}

}  // namespace epconv
