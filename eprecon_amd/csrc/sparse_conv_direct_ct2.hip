// The direct gather kernels with 2 column tiles of 16 (C_out <= 32): see sparse_conv_direct_impl.hpp
#include "sparse_conv_direct_impl.hpp"

namespace epconv {
int launch_direct16_ct2(const ConvParams &p, hipStream_t st) { return launch_ct<2>(p, st); }
}  // namespace epconv
