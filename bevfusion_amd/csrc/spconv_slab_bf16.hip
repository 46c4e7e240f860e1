// bf16 instantiation of the slab submanifold convolution (kernels: spconv_slab.h).
#include "spconv_slab_impl.h"

namespace bevamd {
namespace slab {
int launch_bf16(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream) {
  return launch_impl<tile::T_BF16>(sa, cin, nt, variant, stream);
}
}  // namespace slab
}  // namespace bevamd
