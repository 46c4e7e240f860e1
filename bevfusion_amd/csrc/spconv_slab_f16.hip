// fp16 instantiation of the slab (staged-rows) submanifold convolution (kernels: spconv_slab.h) + its C-ABI entry points.
#include "spconv_slab_impl.h"

namespace bevamd {
namespace slab {
int launch_f16(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream) {
  return launch_impl<tile::T_F16>(sa, cin, nt, variant, stream);
}
}  // namespace slab
}  // namespace bevamd

using namespace bevamd;

static unsigned long long* g_slab_prof = nullptr;   // profiling builds: device buffer the next slab launches add their cycle sums to

extern "C" {

/* -DBEVAMD_PROFILING builds only: device buffer [5] uint64 (issue, multiply, dma-wait, barrier cycle sums over all waves; wave
 * count) that every following bevamd_spconv_conv_forward_slab launch accumulates into; NULL switches it off. */
void bevamd_spconv_slab_set_profile_buffer(void* buf) { g_slab_prof = (unsigned long long*)buf; }
/* the ablation mask this library was COMPILED with (-DBEVAMD_SLAB_ABL=mask; 0 in every shipped build) */
int bevamd_spconv_slab_ablation_mask(void) { return slab::SLAB_ABL; }

/* Rows per block of slab variant `variant` (0 = default) for a cin-channel 3x3x3 convolution, 0 if none is built (cin 32, 64 or
 * 128 with cout == cin; cin <= 16 = the narrow-row kernels, 256 rows or 3000128 -> 128).  The block size fixes the metadata
 * layout of bevamd_spconv_slab_build*. */
int bevamd_spconv_slab_block_rows(int cin, int variant) {
  return slab::block_rows_of(cin, variant);
}

/* The variant codes built for `cin` (for sweeps): writes up to max_n codes, returns how many exist. */
int bevamd_spconv_slab_variants(int cin, int* codes, int max_n) {
  if (cin >= 1 && cin <= 16) {
    if (max_n > 0) codes[0] = slab::SMALL_BASE + 256;
    if (max_n > 1) codes[1] = slab::SMALL_BASE + 128;
    if (cin <= 8) return 2;
    // 16 -> 32 only: both output tiles in one wave
    if (max_n > 2) codes[2] = slab::SMALL_BASE + slab::SMALL_WHOLE + 256;
    if (max_n > 3) codes[3] = slab::SMALL_BASE + slab::SMALL_WHOLE + 128;
    return 4;
  }
  int n = 0, nr = 0;
  const slab::Shape* s = slab::shapes_of(cin, &n);
  const slab::ShapeR* r = slab::shapes_r_of(cin, &nr);
  int k = 0;
  auto put = [&](int code) { if (k < max_n) codes[k] = code; ++k; };
  for (int i = 0; s && i < n; ++i) put(slab::variant_code(s[i]));
  for (int i = 0; r && i < nr; ++i) put(slab::variant_code(r[i]));
  for (int i = 0; r && i < nr; ++i)   // persistent twins: plain shapes only (no kernel flags, no 64-row blocks)
    if (slab::has_persistent_twin(r[i])) put(slab::variant_code(r[i]) + (slab::PERSIST_BASE - slab::REGW_BASE));
#define BEVAMD_ROW(CAP) if (cin == 32) put(slab::FSTAT_BASE + CAP);
  BEVAMD_SLABF_SHAPES_32(BEVAMD_ROW)
#undef BEVAMD_ROW
#define BEVAMD_ROW(CAP) if (cin == 32) put(slab::FSTAT2_BASE + CAP);
  BEVAMD_SLABF2_SHAPES_32(BEVAMD_ROW)
#undef BEVAMD_ROW
  return k;
}

/* 1 if a voxel set on a [batch, X, Y, Z] grid whose rows are in ascending linear index can use the slab kernels: the input
 * range of a kernel plane is at most block_rows + (Y + 2) * Z + 2 rows, which must fit the 16-bit slots. */
int bevamd_spconv_slab_grid_ok(const int* shape, int block_rows) {
  block_rows = slab::rows_of_code(block_rows);
  if (!shape || block_rows <= 0) return 0;
  const long long bound = (long long)block_rows + ((long long)shape[1] + 2) * shape[2] + 2;
  return bound < 0xFFFE;
}

size_t bevamd_spconv_slab_hdr_bytes(int m_cap, int block_rows) {
  block_rows = slab::rows_of_code(block_rows);
  if (m_cap <= 0 || block_rows <= 0) return 0;
  return (size_t)((m_cap + block_rows - 1) / block_rows) * slab::PLANES * sizeof(int2);
}
size_t bevamd_spconv_slab_slot_bytes(int m_cap, int block_rows) {
  block_rows = slab::rows_of_code(block_rows);
  if (m_cap <= 0 || block_rows <= 0) return 0;
  return (size_t)((m_cap + block_rows - 1) / block_rows) * 27 * block_rows * sizeof(uint16_t);
}

/* Block metadata of a 3x3x3 SubM neighbour table nbr [27, nbr_stride] over m rows (m = *m_dev clamped to m_cap, or m_cap):
 * per block of block_rows (128 | 256) rows and kernel plane kx: hdr = (first input row, row count) of the range its nine
 * (ky, kz) taps read, slots = the table as 16-bit offsets into that range (0xFFFF = no neighbour).  Built once per voxel
 * set, shared by every SubM convolution over it.  status (optional int32, device): bit 0 is set if a range exceeded the
 * 16-bit slots (rows not in linear-index order on a grid bevamd_spconv_slab_grid_ok rejects). */
int bevamd_spconv_slab_build(const int* nbr, int nbr_stride, int m_cap, const int* m_dev, int block_rows, void* hdr,
                             void* slots, int* status, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int fmt = slab::fmt_of_code(block_rows);   // upper half: slot format (spconv_slab_meta.h)
  block_rows = slab::rows_of_code(block_rows);
  BEVAMD_REQUIRE(block_rows == 64 || block_rows == 128 || block_rows == 256, "spconv_slab_build: block_rows %d (64 | 128 | 256)", block_rows);
  BEVAMD_REQUIRE(fmt == 0 || fmt == slab::FMT_BAKED128 || (slab::fmt_is_wg(fmt) && block_rows == 128), "spconv_slab_build: slot format %d (the filter-gradient formats need 128-row blocks)", fmt);
  BEVAMD_REQUIRE(m_cap >= 0 && nbr_stride >= m_cap, "spconv_slab_build: bad sizes");
  if (m_cap == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(nbr && hdr && slots, "spconv_slab_build: null buffer");
  const unsigned nblk = (unsigned)((m_cap + block_rows - 1) / block_rows);
  if (block_rows == 64)
    slab::slab_build_kernel<64><<<dim3(nblk), dim3(64), 0, stream>>>(nbr, nbr_stride, m_cap, m_dev, (int2*)hdr, (uint16_t*)slots, status, fmt);
  else if (block_rows == 128)
    slab::slab_build_kernel<128><<<dim3(nblk), dim3(128), 0, stream>>>(nbr, nbr_stride, m_cap, m_dev, (int2*)hdr, (uint16_t*)slots, status, fmt);
  else
    slab::slab_build_kernel<256><<<dim3(nblk), dim3(256), 0, stream>>>(nbr, nbr_stride, m_cap, m_dev, (int2*)hdr, (uint16_t*)slots, status, fmt);
  BEVAMD_LAUNCH_CHECK("spconv_slab_build");
  return BEVAMD_OK;
}

/* Replaces sparse_conv_ext.indice_conv_half (spconv/src/all.cc:30-33 -> spconv_ops.h:260-361) for a 3x3x3 convolution whose
 * input rows are in ascending linear index: submanifold with cin == cout in {32, 64, 128}, or the narrow layers — cin <= 16
 * (feature pitch = the padded width 8 | 16), cout in {16, 32}, submanifold or strided (only the metadata differs:
 * bevamd_spconv_slab_build_from_sorted) — with the same fused epilogue as bevamd_spconv_conv_forward_tiled and the same
 * results (bit-identical for cin <= 64).  hdr / slots: bevamd_spconv_slab_build* with
 * block_rows = bevamd_spconv_slab_block_rows(cin, variant). */
int bevamd_spconv_conv_forward_slab(const void* features, int dtype, int feat_stride, int num_in, const void* image,
                                    const void* hdr, const void* slots, int block_rows, int num_out,
                                    const int* num_out_dev, int cin, int cout, void* out, int out_stride, const void* bias,
                                    const float* bn_scale, const float* bn_shift, const void* residual,
                                    int residual_stride, int relu, int variant, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype == tile::T_F16 || dtype == tile::T_BF16, "spconv_conv_forward_slab: dtype %d is not 16-bit", dtype);
  const bool narrow = cin >= 1 && cin <= 16 && (cout == 16 || cout == 32) && (cout == 16 || cin > 8);
  BEVAMD_REQUIRE(narrow || (cin == cout && (cin == 32 || cin == 64 || cin == 128)), "spconv_conv_forward_slab: %d -> %d channels", cin, cout);
  const int cinp = narrow ? (cin <= 8 ? 8 : 16) : cin;
  BEVAMD_REQUIRE(num_out >= 0 && num_in >= 0, "spconv_conv_forward_slab: bad sizes");
  if (num_out == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && image && hdr && slots && out, "spconv_conv_forward_slab: null buffer");
  BEVAMD_REQUIRE(block_rows == bevamd_spconv_slab_block_rows(cin, variant),
                 "spconv_conv_forward_slab: metadata built for %d-row blocks, variant %d wants %d", block_rows, variant,
                 bevamd_spconv_slab_block_rows(cin, variant));
  BEVAMD_REQUIRE(feat_stride >= cinp && feat_stride % 8 == 0 && ((uintptr_t)features & 15) == 0,
                 "spconv_conv_forward_slab: feature pitch %d must be a multiple of 8 and >= %d, 16-byte aligned", feat_stride, cinp);
  BEVAMD_REQUIRE((unsigned long long)num_in * (unsigned long long)feat_stride * 2ull < 0x100000000ull,
                 "spconv_conv_forward_slab: the feature tensor must be smaller than 4 GiB (buffer descriptor)");
  BEVAMD_REQUIRE(!residual || (unsigned long long)num_out * (unsigned long long)residual_stride * 2ull < 0x100000000ull,
                 "spconv_conv_forward_slab: the residual tensor must be smaller than 4 GiB (buffer descriptor)");
  BEVAMD_REQUIRE(((uintptr_t)image & 15) == 0 && ((uintptr_t)slots & 15) == 0, "spconv_conv_forward_slab: image / slots must be 16-byte aligned");
  BEVAMD_REQUIRE(out_stride >= cout && (!residual || residual_stride >= cout), "spconv_conv_forward_slab: bad output pitch");
  BEVAMD_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "spconv_conv_forward_slab: scale and shift go together");
  slab::SlabArgs sa;
  tile::Args& a = sa.a;
  a.hdr = nullptr; a.slots = nullptr; a.slab_rows = 0;   // (the slab kernels read sa.hdr / sa.slots)
  a.feat = features; a.wimg = image; a.nbr = nullptr; a.m_dev = num_out_dev; a.out = out;
  a.bias = bias; a.scale = bn_scale; a.shift = bn_shift; a.residual = residual;
  a.feat_stride = feat_stride; a.n_in = num_in; a.nbr_stride = 0; a.m_cap = num_out; a.K = 27;
  a.cout = cout; a.out_stride = out_stride; a.res_stride = residual_stride; a.relu = relu;
  a.row_epilogue = cout % 8 == 0 && out_stride % 8 == 0 && ((uintptr_t)out & 15) == 0 &&
                   (!residual || (residual_stride % 8 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                   (!bias || ((uintptr_t)bias & 15) == 0) && (!bn_scale || (((uintptr_t)bn_scale | (uintptr_t)bn_shift) & 15) == 0);
  sa.hdr = (const int2*)hdr;
  sa.slots = (const uint16_t*)slots;
  sa.wimg_bytes = (unsigned)(tile::image_elems(27, cinp, cout / 16) * 2);
  {
    const size_t sb = bevamd_spconv_slab_slot_bytes(num_out, block_rows);
    BEVAMD_REQUIRE(sb < 0x100000000ull, "spconv_conv_forward_slab: slot table of 4 GiB or more");
    sa.slot_bytes = (unsigned)sb;
  }
  sa.prof = g_slab_prof;
  return dtype == tile::T_F16 ? slab::launch_f16(sa, cinp, cout / 16, variant, stream)
                              : slab::launch_bf16(sa, cinp, cout / 16, variant, stream);
}

}  // extern "C"
