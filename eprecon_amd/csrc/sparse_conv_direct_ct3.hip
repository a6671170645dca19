// The direct gather kernels with 3 column tiles of 16 (C_out <= 48): see sparse_conv_direct_impl.hpp
#include "sparse_conv_direct_impl.hpp"

namespace epconv {
int launch_direct16_ct3(const ConvParams &p, hipStream_t st) { return launch_ct<3>(p, st); }
}  // namespace epconv
