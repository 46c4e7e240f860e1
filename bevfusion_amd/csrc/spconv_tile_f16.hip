// fp16 instantiation of the tiled sparse convolution (kernels: spconv_tile.h) + its C-ABI entry points.
#include "spconv_tile_impl.h"

namespace bevamd {
namespace tile {
int launch_f16(const Args& a, int cinp, int nt, int variant, hipStream_t stream) {
  return launch_impl<T_F16>(a, cinp, nt, variant, stream);
}
int image_f16(const void* w, int K, int cin, int cout, int transpose_io, void* img, hipStream_t stream) {
  return image_impl<T_F16>(w, K, cin, cout, transpose_io, img, stream);
}
}  // namespace tile
}  // namespace bevamd

namespace bevamd {
// dst [n, pitch] 16-bit = src [n, c] fp32 rounded, zero padded to the pitch the tiled kernels read: one launch instead of
// zeros + strided copy-cast (two torch kernels in front of the encoder's first convolution)
template <int DT>
__global__ __launch_bounds__(256) void sp_pad_cast_rows_kernel(const float* __restrict__ src, int n, int c, int pitch,
                                                               typename tile::Num<DT>::T* __restrict__ dst) {
  const size_t total = (size_t)n * pitch;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t row = i / pitch;
    const int col = (int)(i - row * pitch);
    dst[i] = tile::Num<DT>::from_f32(col < c ? src[row * c + col] : 0.f);
  }
}

// ---- filter images of MANY layers in one launch (round 6: the training path prepares the forward and the input-gradient image of
// every convolution of a step from the fp32 master weights — 21 casts + 37 image launches + 16 flips before) -------------------
// flags: 1 = transpose_io (input-gradient pass), 2 = mirror the kernel offsets (k -> K - 1 - k: the input gradient of a symmetric
// SubM rulebook walks the SAME table with the mirrored transposed filter), 4 = the source is fp32 (else the image's 16-bit type)
struct ImgDesc {
  const void* w;
  void* img;
  int K, cin, cout, cinp, nt, nchunks, flags;
};
constexpr int IMG_BATCH = 48;
struct ImgBatch {
  ImgDesc d[IMG_BATCH];
};

template <int DT>
__global__ __launch_bounds__(256) void sp_filter_images_batch_kernel(ImgBatch b) {
  typedef typename tile::Num<DT>::T T;
  const ImgDesc& d = b.d[blockIdx.y];
  const size_t total = (size_t)d.nchunks * d.nt * 64 * 8;
  const bool tr = d.flags & 1, mirror = d.flags & 2, f32 = d.flags & 4;
  const int rows = tr ? d.cin : d.cout, cols = tr ? d.cout : d.cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const size_t t = i >> 9;
    const int nt = (int)(t % d.nt);
    const long long j = (long long)(t / d.nt);
    const int c = lane & 15, g = lane >> 4;
    const long long flat = j * 32 + g * 8 + e;
    const int k = (int)(flat / d.cinp), ci = (int)(flat % d.cinp);
    const int co = nt * 16 + c;
    T v = tile::Num<DT>::from_f32(0.f);
    if (k < d.K && ci < cols && co < rows) {
      const int wi = tr ? co : ci, wo = tr ? ci : co;
      const size_t at = ((size_t)(mirror ? d.K - 1 - k : k) * d.cin + wi) * d.cout + wo;
      v = f32 ? tile::Num<DT>::from_f32(((const float*)d.w)[at]) : ((const T*)d.w)[at];
    }
    ((T*)d.img)[i] = v;
  }
}
}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* Filter images (bevamd_spconv_make_filter_image's layout) of n <= 48 convolutions in ONE launch.  Per layer: filters[i]
 * [K, cin, cout] (fp32 when flags[i] & 4, else `dtype`), image[i] of bevamd_spconv_filter_image_elems(K, cin, cout, flags & 1)
 * elements of `dtype` (1 fp16 | 2 bf16); flags[i] & 1 = transpose_io, & 2 = kernel offsets mirrored (k -> K - 1 - k).
 * No reference counterpart: the reference hands torch::mm the filter as it is (spconv_ops.h:322-334). */
int bevamd_spconv_make_filter_images(int n, const void* const* filters, void* const* images, const int* kernel_volume, const int* cin,
                                     const int* cout, const int* flags, int dtype, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype == tile::T_F16 || dtype == tile::T_BF16, "spconv_make_filter_images: dtype %d is not 16-bit", dtype);
  BEVAMD_REQUIRE(n >= 0 && n <= IMG_BATCH, "spconv_make_filter_images: %d layers (at most %d a call)", n, IMG_BATCH);
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(filters && images && kernel_volume && cin && cout && flags, "spconv_make_filter_images: null array");
  ImgBatch b;
  size_t most = 0;
  for (int i = 0; i < n; ++i) {
    const int tr = flags[i] & 1;
    const int rows = tr ? cin[i] : cout[i], cols = tr ? cout[i] : cin[i];
    BEVAMD_REQUIRE(kernel_volume[i] > 0 && cin[i] > 0 && cout[i] > 0 && filters[i] && images[i], "spconv_make_filter_images: layer %d: bad sizes / null buffer", i);
    const int cinp = tile::pad_cin(cols), nt = tile::pad_nt(rows);
    BEVAMD_REQUIRE(cinp && nt, "spconv_make_filter_images: layer %d: channels %d -> %d exceed 128", i, cols, rows);
    b.d[i].w = filters[i]; b.d[i].img = images[i];
    b.d[i].K = kernel_volume[i]; b.d[i].cin = cin[i]; b.d[i].cout = cout[i];
    b.d[i].cinp = cinp; b.d[i].nt = nt; b.d[i].nchunks = tile::image_chunks(kernel_volume[i], cinp); b.d[i].flags = flags[i];
    const size_t total = tile::image_elems(kernel_volume[i], cinp, nt);
    most = total > most ? total : most;
  }
  const unsigned gx = (unsigned)((most + 2047) / 2048 < 64 ? (most + 2047) / 2048 : 64);
  if (dtype == tile::T_F16) sp_filter_images_batch_kernel<tile::T_F16><<<dim3(gx ? gx : 1, n), dim3(256), 0, stream>>>(b);
  else sp_filter_images_batch_kernel<tile::T_BF16><<<dim3(gx ? gx : 1, n), dim3(256), 0, stream>>>(b);
  BEVAMD_LAUNCH_CHECK("sp_filter_images_batch");
  return BEVAMD_OK;
}

/* 1 if (dtype, cin -> cout) is served by the tiled kernels (16-bit features, channels <= 128) */
int bevamd_spconv_tiled_supported(int dtype, int cin, int cout) {
  return (dtype == tile::T_F16 || dtype == tile::T_BF16) && cin > 0 && cout > 0 && tile::pad_cin(cin) && tile::pad_nt(cout);
}

/* ELEMENTS of the filter image of a K-offset cin -> cout convolution (transpose_io: input-gradient pass) */
size_t bevamd_spconv_filter_image_elems(int kernel_volume, int cin, int cout, int transpose_io) {
  const int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  const int cinp = tile::pad_cin(cols), nt = tile::pad_nt(rows);
  if (!cinp || !nt || kernel_volume <= 0) return 0;
  return tile::image_elems(kernel_volume, cinp, nt);
}

int bevamd_spconv_make_filter_image(const void* filters, int dtype, int kernel_volume, int cin, int cout,
                                    int transpose_io, void* image, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype == tile::T_F16 || dtype == tile::T_BF16, "spconv_make_filter_image: dtype %d is not 16-bit", dtype);
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0, "spconv_make_filter_image: bad sizes");
  BEVAMD_REQUIRE(filters && image, "spconv_make_filter_image: null buffer");
  return dtype == tile::T_F16 ? tile::image_f16(filters, kernel_volume, cin, cout, transpose_io, image, stream)
                              : tile::image_bf16(filters, kernel_volume, cin, cout, transpose_io, image, stream);
}

int bevamd_spconv_conv_forward_tiled(const void* features, int dtype, int feat_stride, int num_in, const void* image,
                                     const int* nbr, int nbr_stride, int num_out, const int* num_out_dev,
                                     int kernel_volume, int cin, int cout, void* out, int out_stride, const void* bias,
                                     const float* bn_scale, const float* bn_shift, const void* residual,
                                     int residual_stride, int relu, int variant, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype == tile::T_F16 || dtype == tile::T_BF16, "spconv_conv_forward_tiled: dtype %d is not 16-bit", dtype);
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0 && num_out >= 0 && num_in >= 0, "spconv_conv_forward_tiled: bad sizes");
  if (num_out == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && image && nbr && out, "spconv_conv_forward_tiled: null buffer");
  BEVAMD_REQUIRE(nbr_stride >= num_out, "spconv_conv_forward_tiled: nbr_stride %d < num_out %d", nbr_stride, num_out);
  const int cinp = tile::pad_cin(cin), nt = tile::pad_nt(cout);
  BEVAMD_REQUIRE(cinp && nt, "spconv_conv_forward_tiled: channels %d -> %d exceed 128", cin, cout);
  // rows are read with 16-byte buffer loads of cin_pad channels: the row pitch must cover them
  BEVAMD_REQUIRE(feat_stride >= cinp && feat_stride % 8 == 0 && ((uintptr_t)features & 15) == 0,
                 "spconv_conv_forward_tiled: feature pitch %d must be a multiple of 8 and >= %d (zero-padded), 16-byte aligned",
                 feat_stride, cinp);
  BEVAMD_REQUIRE((unsigned long long)num_in * feat_stride * 2ull < 0x80000000ull,
                 "spconv_conv_forward_tiled: feature matrix must be < 2 GiB");
  BEVAMD_REQUIRE(out_stride >= cout && (!residual || residual_stride >= cout), "spconv_conv_forward_tiled: bad output pitch");
  BEVAMD_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "spconv_conv_forward_tiled: scale and shift go together");
  tile::Args a;
  a.hdr = nullptr; a.slots = nullptr; a.slab_rows = 0;
  a.feat = features; a.wimg = image; a.nbr = nbr; a.m_dev = num_out_dev; a.out = out;
  a.bias = bias; a.scale = bn_scale; a.shift = bn_shift; a.residual = residual;
  a.feat_stride = feat_stride; a.n_in = num_in; a.nbr_stride = nbr_stride; a.m_cap = num_out; a.K = kernel_volume;
  a.cout = cout; a.out_stride = out_stride; a.res_stride = residual_stride; a.relu = relu;
  // 16-byte row-wise epilogue: 8-channel groups must be whole and 16-byte aligned everywhere they are touched
  a.row_epilogue = cout % 8 == 0 && out_stride % 8 == 0 && ((uintptr_t)out & 15) == 0 &&
                   (!residual || (residual_stride % 8 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                   (!bias || ((uintptr_t)bias & 15) == 0) && (!bn_scale || (((uintptr_t)bn_scale | (uintptr_t)bn_shift) & 15) == 0);
  return dtype == tile::T_F16 ? tile::launch_f16(a, cinp, nt, variant, stream) : tile::launch_bf16(a, cinp, nt, variant, stream);
}

/* bevamd_spconv_conv_forward_tiled for a 3x3x3 convolution whose rulebook is given as slab metadata (hdr / slots of
 * bevamd_spconv_slab_build*, block_rows = 128 | 256) instead of the int32 neighbour table: the gather kernels decode
 * `first row of the plane + 16-bit slot` while they load a tile's table into LDS.  Same kernels, same results; half the rulebook
 * bytes, and no table has to be cleared and scattered first (the strided 32->64 / 64->128 layers of the SparseEncoder, whose input
 * ranges are too long for the staged-rows kernels). */
int bevamd_spconv_conv_forward_tiled_slots(const void* features, int dtype, int feat_stride, int num_in, const void* image,
                                           const void* hdr, const void* slots, int block_rows, int num_out,
                                           const int* num_out_dev, int cin, int cout, void* out, int out_stride,
                                           const void* bias, const float* bn_scale, const float* bn_shift, const void* residual,
                                           int residual_stride, int relu, int variant, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype == tile::T_F16 || dtype == tile::T_BF16, "spconv_conv_forward_tiled_slots: dtype %d is not 16-bit", dtype);
  BEVAMD_REQUIRE(cin > 0 && cout > 0 && num_out >= 0 && num_in >= 0, "spconv_conv_forward_tiled_slots: bad sizes");
  BEVAMD_REQUIRE(block_rows == 128 || block_rows == 256, "spconv_conv_forward_tiled_slots: block_rows %d (128 | 256)", block_rows);
  if (num_out == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && image && hdr && slots && out, "spconv_conv_forward_tiled_slots: null buffer");
  const int cinp = tile::pad_cin(cin), nt = tile::pad_nt(cout);
  BEVAMD_REQUIRE(cinp && nt, "spconv_conv_forward_tiled_slots: channels %d -> %d exceed 128", cin, cout);
  BEVAMD_REQUIRE(feat_stride >= cinp && feat_stride % 8 == 0 && ((uintptr_t)features & 15) == 0,
                 "spconv_conv_forward_tiled_slots: feature pitch %d must be a multiple of 8 and >= %d (zero-padded), 16-byte aligned",
                 feat_stride, cinp);
  BEVAMD_REQUIRE((unsigned long long)num_in * feat_stride * 2ull < 0x80000000ull,
                 "spconv_conv_forward_tiled_slots: feature matrix must be < 2 GiB");
  BEVAMD_REQUIRE(out_stride >= cout && (!residual || residual_stride >= cout), "spconv_conv_forward_tiled_slots: bad output pitch");
  BEVAMD_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "spconv_conv_forward_tiled_slots: scale and shift go together");
  tile::Args a;
  a.hdr = (const int2*)hdr; a.slots = (const uint16_t*)slots; a.slab_rows = block_rows;
  a.feat = features; a.wimg = image; a.nbr = nullptr; a.m_dev = num_out_dev; a.out = out;
  a.bias = bias; a.scale = bn_scale; a.shift = bn_shift; a.residual = residual;
  a.feat_stride = feat_stride; a.n_in = num_in; a.nbr_stride = 0; a.m_cap = num_out; a.K = 27;
  a.cout = cout; a.out_stride = out_stride; a.res_stride = residual_stride; a.relu = relu;
  a.row_epilogue = cout % 8 == 0 && out_stride % 8 == 0 && ((uintptr_t)out & 15) == 0 &&
                   (!residual || (residual_stride % 8 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                   (!bias || ((uintptr_t)bias & 15) == 0) && (!bn_scale || (((uintptr_t)bn_scale | (uintptr_t)bn_shift) & 15) == 0);
  return dtype == tile::T_F16 ? tile::launch_f16(a, cinp, nt, variant, stream) : tile::launch_bf16(a, cinp, nt, variant, stream);
}

/* dst [n, pitch] (fp16 / bf16) = src [n, c] fp32 rounded to nearest, columns c .. pitch-1 zero: the encoder's input rows in the
 * padded pitch the 16-bit convolution kernels read (SparseEncoder.forward's @auto_fp16 cast, sparse_encoder.py:99). */
int bevamd_spconv_pad_cast_rows(const float* src, int n, int c, int pitch, int dtype, void* dst, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype == tile::T_F16 || dtype == tile::T_BF16, "spconv_pad_cast_rows: dtype %d is not 16-bit", dtype);
  BEVAMD_REQUIRE(n >= 0 && c > 0 && pitch >= c, "spconv_pad_cast_rows: bad sizes");
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(src && dst, "spconv_pad_cast_rows: null buffer");
  const size_t total = (size_t)n * pitch;
  const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == tile::T_F16)
    sp_pad_cast_rows_kernel<tile::T_F16><<<dim3(blocks), dim3(256), 0, stream>>>(src, n, c, pitch, (_Float16*)dst);
  else
    sp_pad_cast_rows_kernel<tile::T_BF16><<<dim3(blocks), dim3(256), 0, stream>>>(src, n, c, pitch, (uint16_t*)dst);
  BEVAMD_LAUNCH_CHECK("sp_pad_cast_rows");
  return BEVAMD_OK;
}

}  // extern "C"
