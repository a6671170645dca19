// The direct gather kernels with 1 column tile of 16 (C_out <= 16): see sparse_conv_direct_impl.hpp
#include "sparse_conv_direct_impl.hpp"

namespace epconv {
int launch_direct16_ct1(const ConvParams &p, hipStream_t st) { return launch_ct<1>(p, st); }
}  // namespace epconv
