// bf16 instantiation of the tiled sparse convolution (kernels: spconv_tile.h).
#include "spconv_tile_impl.h"

namespace bevamd {
namespace tile {
int launch_bf16(const Args& a, int cinp, int nt, int variant, hipStream_t stream) {
  return launch_impl<T_BF16>(a, cinp, nt, variant, stream);
}
int image_bf16(const void* w, int K, int cin, int cout, int transpose_io, void* img, hipStream_t stream) {
  return image_impl<T_BF16>(w, K, cin, cout, transpose_io, img, stream);
}
}  // namespace tile
}  // namespace bevamd
