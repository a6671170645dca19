// The direct gather kernels with 4 column tiles of 16 (C_out <= 64): see sparse_conv_direct_impl.hpp
#include "sparse_conv_direct_impl.hpp"

namespace epconv {
int launch_direct16_ct4(const ConvParams &p, hipStream_t st) { return launch_ct<4>(p, st); }
}  // namespace epconv
