// Per-dtype instantiation of the slab (staged-rows) submanifold convolution (included by spconv_slab_{f16,bf16}.hip).
#pragma once
#include "spconv_slab.h"
#include "spconv_slab_regw.h"
#include "spconv_slab_persist.h"
#include "spconv_slab_small.h"
#include "spconv_slab_fstat.h"
#include "spconv_slab_fstat2.h"

namespace bevamd {
namespace slab {

// variant = KC*10000 + MT*1000 + (NW/4)*100 + SPS*10 + WR
//   KC: channels staged per row (32 | 64)   MT: 16-row tiles per wave   NW: waves per workgroup
//   SPS: kernel taps per barrier (1 | 3 | 9)   WR: filter ring slots (2 | 3)
struct Shape { int kc, mt, nw, sps, wr, cap; };

constexpr int MAX_DEVICES = 64;
// Per kernel instantiation AND per device: the opt-in to > 64 KiB of dynamic LDS is a per-device function attribute, and so is
// the occupancy a persistent grid is sized by (a process may drive several GPUs).  Plain ints written with the same value by
// every thread that gets here first: a benign race.
struct PerDevice {
  int raised[MAX_DEVICES];
  int wg_per_xcd[MAX_DEVICES];
};
static inline int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return dev >= 0 && dev < MAX_DEVICES ? dev : 0;
}
template <typename K>
static inline void raise_lds_limit(PerDevice& pd, K kern, int bytes) {
  if (bytes <= 65536) return;
  const int dev = current_device();
  if (pd.raised[dev]) return;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)hipGetLastError();
  pd.raised[dev] = 1;
}
// tuning knob (percent of the resident workgroups a persistent grid is sized to; default 100; > 100 = oversubscribed: workgroups
// that start late take over from finished ones instead of becoming stragglers): BEVAMD_SLAB_GRID_PCT
static inline long long persistent_cap(int wg_per_xcd) {
  static int pct = -1;
  if (pct < 0) { const char* e = getenv("BEVAMD_SLAB_GRID_PCT"); pct = e ? atoi(e) : 100; if (pct < 10 || pct > 1600) pct = 100; }
  const long long c = (long long)wg_per_xcd * pct / 100;
  return c < 1 ? 1 : c;
}
constexpr int SMALL_BASE = 3000000;   // variant codes of the narrow-row kernels: SMALL_BASE + rows per block
constexpr int SMALL_WHOLE = 100000;   // + SMALL_WHOLE: 16 -> 32 with BOTH output tiles in one wave (each operand fragment read from
                                      // LDS once instead of once per tile: the kernels are LDS-bound, EXPERIMENTS C.10)
static inline int small_block_rows(int cinp, int variant) {
  if (cinp != 8 && cinp != 16) return 0;
  if (variant == 0 || variant == SMALL_BASE + 256) return 256;
  if (variant == SMALL_BASE + 128) return 128;
  if (cinp == 16 && variant == SMALL_BASE + SMALL_WHOLE + 256) return 256;
  if (cinp == 16 && variant == SMALL_BASE + SMALL_WHOLE + 128) return 128;
  return 0;
}
// resident workgroups per XCD of a persistent kernel on the current device (0 on error)
template <typename K>
static inline int resident_per_xcd(PerDevice& pd, K kern, int threads, int bytes) {
  const int dev = current_device();
  if (pd.wg_per_xcd[dev] == 0) {
    raise_lds_limit(pd, kern, bytes);
    int cus = 0, occ = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)kern, threads, bytes) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    if (occ < 1) occ = 1;
    pd.wg_per_xcd[dev] = (cus + 7) / 8 * occ;
  }
  return pd.wg_per_xcd[dev];
}

template <int DT, int KC, int CIN, int NT, int MT, int NW, int SPS, int WR, int CAP>
static int run(const SlabArgs& sa, hipStream_t stream) {
  typedef Plan<KC, CIN, NT, MT, NW, SPS, WR, CAP> P;
  static_assert(P::BYTES <= 160 * 1024, "LDS plan exceeds the CU");
  auto kern = &spconv_slab_kernel<DT, KC, CIN, NT, MT, NW, SPS, WR, CAP>;
  static PerDevice pd = {};
  raise_lds_limit(pd, kern, P::BYTES);
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  const long long blocks = (nblk + 7) / 8 * 8;
  kern<<<dim3((unsigned)blocks), dim3(NW * 64), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slab");
  return BEVAMD_OK;
}

// the built configurations per input width (first entry = what variant 0 means); X(...) expands once per configuration
// MI355X, 8 flagship frames per launch, fp16 (profiles/r02_slab_sweep_b8.txt; gather kernel in brackets):
//   32->32 2.08 M rows: 322113 218 us [292]   64->64 788 k rows: 642113 206 us / 642232 204 us [246]
//   128->128 192 k rows: 642213 167 us / 322232 166 us [193].  64 rows per wave (MT = 4) was measured too: 274-389 us, one
//   workgroup per CU is too little latency hiding.  The encoder's defaults are the register-filter / persistent shapes further
//   down (spconv/fused.py: _SLAB_DEFAULT); these LDS-filter kernels stay built as the reference point of the sweeps.
// (round 3: the lists are pruned to what the sweeps still compare against — every dropped shape is in profiles/r02_slab_sweep_*)
#define BEVAMD_SLAB_SHAPES_32(X) X(32, 2, 4, 1, 3, 192)
#define BEVAMD_SLAB_SHAPES_64(X) X(64, 2, 4, 1, 3, 184) X(32, 2, 4, 1, 3, 192)
#define BEVAMD_SLAB_SHAPES_128(X) X(64, 2, 8, 1, 3, 320) X(32, 2, 8, 3, 2, 384)

static inline const Shape* shapes_of(int cin, int* n) {
#define BEVAMD_ROW(KC, MT, NW, SPS, WR, CAP) {KC, MT, NW, SPS, WR, CAP},
  static const Shape s32[] = {BEVAMD_SLAB_SHAPES_32(BEVAMD_ROW)};
  static const Shape s64[] = {BEVAMD_SLAB_SHAPES_64(BEVAMD_ROW)};
  static const Shape s128[] = {BEVAMD_SLAB_SHAPES_128(BEVAMD_ROW)};
#undef BEVAMD_ROW
  switch (cin) {
    case 32: *n = (int)(sizeof(s32) / sizeof(Shape)); return s32;
    case 64: *n = (int)(sizeof(s64) / sizeof(Shape)); return s64;
    case 128: *n = (int)(sizeof(s128) / sizeof(Shape)); return s128;
    default: *n = 0; return nullptr;
  }
}
static inline int variant_code(const Shape& s) { return s.kc * 10000 + s.mt * 1000 + (s.nw / 4) * 100 + s.sps * 10 + s.wr; }
static inline const Shape* find_shape(int cin, int variant) {
  int n = 0;
  const Shape* s = shapes_of(cin, &n);
  if (!s) return nullptr;
  if (variant == 0) return s;
  for (int i = 0; i < n; ++i)
    if (variant_code(s[i]) == variant) return s + i;
  return nullptr;
}

// ---- register-filter kernels (spconv_slab_regw.h): variant = 1000000 + KC*10000 + MT*1000 + RW*100 + CW*10 + ID ----------
struct ShapeR { int kc, mt, rw, cw, cap, id; };

template <int DT, int KC, int CIN, int NT, int MT, int RW, int CW, int CAP, int FLAGS = 0>
static int run_r(const SlabArgs& sa, hipStream_t stream) {
  typedef PlanR<KC, CIN, NT, MT, RW, CW, CAP> P;
  static_assert(P::BYTES <= 160 * 1024, "LDS plan exceeds the CU");
  auto kern = &spconv_slabr_kernel<DT, KC, CIN, NT, MT, RW, CW, CAP, FLAGS>;
  static PerDevice pd = {};
  raise_lds_limit(pd, kern, P::BYTES);
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  const long long blocks = (nblk + 7) / 8 * 8;
  kern<<<dim3((unsigned)blocks), dim3(P::NW * 64), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slabr");
  return BEVAMD_OK;
}

// persistent flavour of the same shapes (spconv_slab_persist.h): variant = 2000000 + the same fields
template <int DT, int KC, int CIN, int NT, int MT, int RW, int CW, int CAP>
static int run_p(const SlabArgs& sa, hipStream_t stream) {
  typedef PlanP<KC, CIN, NT, MT, RW, CW, CAP> P;
  static_assert(P::BYTES <= 160 * 1024, "LDS plan exceeds the CU");
  auto kern = &spconv_slabp_kernel<DT, KC, CIN, NT, MT, RW, CW, CAP>;
  static PerDevice pd = {};
  const int wg_per_xcd = resident_per_xcd(pd, kern, P::NW * 64, P::BYTES);
  if (wg_per_xcd <= 0) { set_error("spconv slab: occupancy query failed"); return BEVAMD_ERR_HIP; }
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  long long gx = (nblk + 7) / 8;
  if (gx > persistent_cap(wg_per_xcd)) gx = persistent_cap(wg_per_xcd);
  kern<<<dim3((unsigned)(gx * 8)), dim3(P::NW * 64), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slabp");
  return BEVAMD_OK;
}

#define BEVAMD_SLABR_SHAPES_32(X) X(32, 4, 4, 1, 384, 0) X(32, 2, 4, 1, 192, 0)
// IDs >= 4 carry kernel flags (spconv_slab_regw.h: 8 = baked slot metadata, block_rows code rows | FMT_BAKED128 << 16); (64, 2, 2, 2): 64-row blocks (one frame's
// level 4 is 188 blocks of 128 rows on 256 CUs), reading the baked metadata the filter-stationary kernels share
#ifdef BEVAMD_SLABR_EXTRA_SHAPES   // experiment builds: 64 x 64 wave tiles in 2-wave blocks (1644218)
#define BEVAMD_SLABR_SHAPES_64(X) X(64, 4, 2, 2, 184, 0) X(64, 4, 4, 2, 320, 0) X(64, 4, 2, 2, 168, 2) X(64, 4, 2, 2, 168, 8) X(64, 4, 2, 1, 168, 8) X(64, 2, 4, 1, 168, 8) X(32, 4, 4, 1, 320, 0)
#else
#define BEVAMD_SLABR_SHAPES_64(X) X(64, 4, 2, 2, 184, 0) X(64, 4, 4, 2, 320, 0) X(64, 4, 2, 2, 168, 2) X(64, 4, 2, 2, 168, 8)
#endif
#define BEVAMD_SLABR_SHAPES_128(X) X(64, 4, 2, 2, 184, 0) X(64, 4, 4, 2, 320, 0) X(64, 4, 2, 2, 184, 8) X(64, 2, 2, 2, 120, 0)

static inline const ShapeR* shapes_r_of(int cin, int* n) {
#define BEVAMD_ROW(KC, MT, RW, CW, CAP, ID) {KC, MT, RW, CW, CAP, ID},
  static const ShapeR s32[] = {BEVAMD_SLABR_SHAPES_32(BEVAMD_ROW)};
  static const ShapeR s64[] = {BEVAMD_SLABR_SHAPES_64(BEVAMD_ROW)};
  static const ShapeR s128[] = {BEVAMD_SLABR_SHAPES_128(BEVAMD_ROW)};
#undef BEVAMD_ROW
  switch (cin) {
    case 32: *n = (int)(sizeof(s32) / sizeof(ShapeR)); return s32;
    case 64: *n = (int)(sizeof(s64) / sizeof(ShapeR)); return s64;
    case 128: *n = (int)(sizeof(s128) / sizeof(ShapeR)); return s128;
    default: *n = 0; return nullptr;
  }
}
constexpr int REGW_BASE = 1000000, PERSIST_BASE = 2000000;
static inline int variant_code(const ShapeR& s) { return REGW_BASE + s.kc * 10000 + s.mt * 1000 + s.rw * 100 + s.cw * 10 + s.id; }
static inline bool has_persistent_twin(const ShapeR& s) { return (s.id & ~3) == 0 && s.rw * 16 * s.mt != BAKED_ROWS; }
static inline const ShapeR* find_shape_r(int cin, int variant) {
  const bool persistent = variant >= PERSIST_BASE;
  if (persistent) variant -= PERSIST_BASE - REGW_BASE;   // the persistent kernels are built for the same (plain) shapes
  int n = 0;
  const ShapeR* s = shapes_r_of(cin, &n);
  for (int i = 0; s && i < n; ++i)
    if (variant_code(s[i]) == variant) return (persistent && !has_persistent_twin(s[i])) ? nullptr : s + i;
  return nullptr;
}
// ---- filter-stationary kernels (spconv_slab_fstat.h), 32 -> 32: variant = 4000000 + CAP; 64-row blocks, BAKED slots --
constexpr int FSTAT_BASE = 4000000;
#ifdef BEVAMD_PROFILING   // the 96-row buffers measured slower (one wave: 3 % ; wave pairs: 148 against 143 us): sweeps only
#define BEVAMD_SLABF_SHAPES_32(X) X(112) X(96)
#else
#define BEVAMD_SLABF_SHAPES_32(X) X(112)
#endif
static inline bool fstat_built(int cin, int variant) {
#define BEVAMD_ROW(CAP) if (cin == 32 && variant == FSTAT_BASE + CAP) return true;
  BEVAMD_SLABF_SHAPES_32(BEVAMD_ROW)
#undef BEVAMD_ROW
  return false;
}
template <int DT, int CAP>
static int run_f(const SlabArgs& sa, hipStream_t stream) {
  typedef PlanF<CAP> P;
  static_assert(P::BYTES <= 40 * 1024, "four waves per CU: 40 KiB of LDS each");
  auto kern = &spconv_slabf_kernel<DT, CAP>;
  static PerDevice pd = {};
  const int wg_per_xcd = resident_per_xcd(pd, kern, 64, P::BYTES);
  if (wg_per_xcd <= 0) { set_error("spconv slab: occupancy query failed"); return BEVAMD_ERR_HIP; }
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  long long gx = (nblk + 7) / 8;
  if (gx > persistent_cap(wg_per_xcd)) gx = persistent_cap(wg_per_xcd);
  kern<<<dim3((unsigned)(gx * 8)), dim3(64), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slabf");
  return BEVAMD_OK;
}
// ---- filter-stationary wave pairs (spconv_slab_fstat2.h), 32 -> 32: variant = 4100000 + CAP; same metadata as 4000000 + CAP --
constexpr int FSTAT2_BASE = 4100000;
#ifdef BEVAMD_PROFILING
#define BEVAMD_SLABF2_SHAPES_32(X) X(96) X(128)
#else
#define BEVAMD_SLABF2_SHAPES_32(X) X(128)
#endif
static inline bool fstat2_built(int cin, int variant) {
#define BEVAMD_ROW(CAP) if (cin == 32 && variant == FSTAT2_BASE + CAP) return true;
  BEVAMD_SLABF2_SHAPES_32(BEVAMD_ROW)
#undef BEVAMD_ROW
  return false;
}
template <int DT, int CAP>
static int run_f2(const SlabArgs& sa, hipStream_t stream) {
  typedef PlanF2<CAP> P;
  static_assert(P::BYTES <= 53 * 1024, "three wave pairs per CU at least");
  auto kern = &spconv_slabf2_kernel<DT, CAP>;
  static PerDevice pd = {};
  const int wg_per_xcd = resident_per_xcd(pd, kern, 128, P::BYTES);
  if (wg_per_xcd <= 0) { set_error("spconv slab: occupancy query failed"); return BEVAMD_ERR_HIP; }
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  long long gx = (nblk + 7) / 8;
  if (gx > persistent_cap(wg_per_xcd)) gx = persistent_cap(wg_per_xcd);
  kern<<<dim3((unsigned)(gx * 8)), dim3(128), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slabf2");
  return BEVAMD_OK;
}
template <int DT>
int launch_f_impl(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream) {
  if (cin == 32 && nt == 2 && !sa.a.row_epilogue) {
    set_error("spconv slab: the filter-stationary kernels need 16-byte aligned rows (pitches multiples of 8)");
    return BEVAMD_ERR_UNSUPPORTED;
  }
#define BEVAMD_ROW(CAP) if (cin == 32 && nt == 2 && variant == FSTAT_BASE + CAP) return run_f<DT, CAP>(sa, stream);
  BEVAMD_SLABF_SHAPES_32(BEVAMD_ROW)
#undef BEVAMD_ROW
#define BEVAMD_ROW(CAP) if (cin == 32 && nt == 2 && variant == FSTAT2_BASE + CAP) return run_f2<DT, CAP>(sa, stream);
  BEVAMD_SLABF2_SHAPES_32(BEVAMD_ROW)
#undef BEVAMD_ROW
  set_error("spconv slab: no filter-stationary kernel for cin=%d, cout tiles=%d, variant=%d", cin, nt, variant);
  return BEVAMD_ERR_UNSUPPORTED;
}

// rows per block of any variant code (0 = none built)
static inline int block_rows_of(int cin, int variant) {
  if (cin <= 16) return cin > 0 ? small_block_rows(cin <= 8 ? 8 : 16, variant) : 0;
  if (variant >= FSTAT_BASE) return (fstat_built(cin, variant) || fstat2_built(cin, variant)) ? BAKED_ROWS : 0;
  if (variant >= REGW_BASE) {
    const ShapeR* r = find_shape_r(cin, variant);
    if (!r) return 0;
    return r->rw * 16 * r->mt + ((r->id & 8) ? (FMT_BAKED128 << FMT_SHIFT) : 0);   // flag 8: baked 128-byte-row slot format
  }
  const Shape* s = find_shape(cin, variant);
  return s ? s->nw * 16 * s->mt : 0;
}

// ---- narrow-row kernels (spconv_slab_small.h): cin padded to 8 | 16, cout 16 | 32; variant = 3000000 + block rows ----------
template <int DT, int CIN, int NT, int MT, int NW, int CW, int CAP>
static int run_s(const SlabArgs& sa, hipStream_t stream) {
  typedef PlanS<CIN, NT, MT, NW, CW, CAP> P;
  static_assert(P::BYTES <= 160 * 1024, "LDS plan exceeds the CU");
  auto kern = &spconv_slabs_kernel<DT, CIN, NT, MT, NW, CW, CAP>;
  static PerDevice pd = {};
  const int wg_per_xcd = resident_per_xcd(pd, kern, NW * 64, P::BYTES);
  if (wg_per_xcd <= 0) { set_error("spconv slab: occupancy query failed"); return BEVAMD_ERR_HIP; }
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  long long gx = (nblk + 7) / 8;
  if (gx > persistent_cap(wg_per_xcd)) gx = persistent_cap(wg_per_xcd);
  kern<<<dim3((unsigned)(gx * 8)), dim3(NW * 64), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slabs");
  return BEVAMD_OK;
}
template <int DT>
int launch_s_impl(const SlabArgs& sa, int cinp, int nt, int variant, hipStream_t stream) {
  const int bm = small_block_rows(cinp, variant);
  if (variant >= SMALL_BASE + SMALL_WHOLE && cinp == 16 && nt == 2) {
    if (bm == 256) return run_s<DT, 16, 2, 4, 4, 1, 384>(sa, stream);
    if (bm == 128) return run_s<DT, 16, 2, 2, 4, 1, 256>(sa, stream);
  }
  if (bm == 256) {
    if (cinp == 8 && nt == 1) return run_s<DT, 8, 1, 4, 4, 1, 384>(sa, stream);
    if (cinp == 16 && nt == 1) return run_s<DT, 16, 1, 4, 4, 1, 384>(sa, stream);
    if (cinp == 16 && nt == 2) return run_s<DT, 16, 2, 4, 8, 2, 384>(sa, stream);
  } else if (bm == 128) {
    if (cinp == 8 && nt == 1) return run_s<DT, 8, 1, 2, 4, 1, 256>(sa, stream);
    if (cinp == 16 && nt == 1) return run_s<DT, 16, 1, 2, 4, 1, 256>(sa, stream);
    if (cinp == 16 && nt == 2) return run_s<DT, 16, 2, 2, 8, 2, 256>(sa, stream);
  }
  set_error("spconv slab: no narrow-row kernel for padded cin=%d, cout tiles=%d, variant=%d", cinp, nt, variant);
  return BEVAMD_ERR_UNSUPPORTED;
}

template <int DT>
int launch_r_impl(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream) {
  const ShapeR* s = find_shape_r(cin, variant);
  if (!s || nt != cin / 16) {
    set_error("spconv slab: no register-filter kernel for cin=%d, cout tiles=%d, variant=%d", cin, nt, variant);
    return BEVAMD_ERR_UNSUPPORTED;
  }
#define BEVAMD_CODE(KC, MT, RW, CW, ID) (KC * 10000 + MT * 1000 + RW * 100 + CW * 10 + ID)
#define BEVAMD_CASE32(KC, MT, RW, CW, CAP, ID)                                                                              \
  if (cin == 32 && variant == REGW_BASE + BEVAMD_CODE(KC, MT, RW, CW, ID)) return run_r<DT, KC, 32, 2, MT, RW, CW, CAP, ((ID) & ~3)>(sa, stream); \
  if constexpr (((ID) & ~3) == 0 && RW * 16 * MT != BAKED_ROWS) if (cin == 32 && variant == PERSIST_BASE + BEVAMD_CODE(KC, MT, RW, CW, ID)) return run_p<DT, KC, 32, 2, MT, RW, CW, CAP>(sa, stream);
#define BEVAMD_CASE64(KC, MT, RW, CW, CAP, ID)                                                                              \
  if (cin == 64 && variant == REGW_BASE + BEVAMD_CODE(KC, MT, RW, CW, ID)) return run_r<DT, KC, 64, 4, MT, RW, CW, CAP, ((ID) & ~3)>(sa, stream); \
  if constexpr (((ID) & ~3) == 0 && RW * 16 * MT != BAKED_ROWS) if (cin == 64 && variant == PERSIST_BASE + BEVAMD_CODE(KC, MT, RW, CW, ID)) return run_p<DT, KC, 64, 4, MT, RW, CW, CAP>(sa, stream);
#define BEVAMD_CASE128(KC, MT, RW, CW, CAP, ID)                                                                               \
  if (cin == 128 && variant == REGW_BASE + BEVAMD_CODE(KC, MT, RW, CW, ID)) return run_r<DT, KC, 128, 8, MT, RW, CW, CAP, ((ID) & ~3)>(sa, stream); \
  if constexpr (((ID) & ~3) == 0 && RW * 16 * MT != BAKED_ROWS) if (cin == 128 && variant == PERSIST_BASE + BEVAMD_CODE(KC, MT, RW, CW, ID)) return run_p<DT, KC, 128, 8, MT, RW, CW, CAP>(sa, stream);
  BEVAMD_SLABR_SHAPES_32(BEVAMD_CASE32)
  BEVAMD_SLABR_SHAPES_64(BEVAMD_CASE64)
  BEVAMD_SLABR_SHAPES_128(BEVAMD_CASE128)
#undef BEVAMD_CASE32
#undef BEVAMD_CASE64
#undef BEVAMD_CASE128
#undef BEVAMD_CODE
  set_error("spconv slab: variant %d is listed but not built for cin=%d", variant, cin);
  return BEVAMD_ERR_UNSUPPORTED;
}

template <int DT>
int launch_impl(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream) {
  if (cin <= 16) return launch_s_impl<DT>(sa, cin, nt, variant, stream);
  if (variant >= FSTAT_BASE) return launch_f_impl<DT>(sa, cin, nt, variant, stream);
  if (variant >= REGW_BASE) return launch_r_impl<DT>(sa, cin, nt, variant, stream);
  const Shape* s = find_shape(cin, variant);
  if (!s || nt != cin / 16) {
    set_error("spconv slab: no kernel for cin=%d, cout tiles=%d, variant=%d", cin, nt, variant);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  const int code = variant_code(*s);
#define BEVAMD_CASE32(KC, MT, NW, SPS, WR, CAP) \
  if (cin == 32 && code == KC * 10000 + MT * 1000 + (NW / 4) * 100 + SPS * 10 + WR) return run<DT, KC, 32, 2, MT, NW, SPS, WR, CAP>(sa, stream);
#define BEVAMD_CASE64(KC, MT, NW, SPS, WR, CAP) \
  if (cin == 64 && code == KC * 10000 + MT * 1000 + (NW / 4) * 100 + SPS * 10 + WR) return run<DT, KC, 64, 4, MT, NW, SPS, WR, CAP>(sa, stream);
#define BEVAMD_CASE128(KC, MT, NW, SPS, WR, CAP) \
  if (cin == 128 && code == KC * 10000 + MT * 1000 + (NW / 4) * 100 + SPS * 10 + WR) return run<DT, KC, 128, 8, MT, NW, SPS, WR, CAP>(sa, stream);
  BEVAMD_SLAB_SHAPES_32(BEVAMD_CASE32)
  BEVAMD_SLAB_SHAPES_64(BEVAMD_CASE64)
  BEVAMD_SLAB_SHAPES_128(BEVAMD_CASE128)
#undef BEVAMD_CASE32
#undef BEVAMD_CASE64
#undef BEVAMD_CASE128
  set_error("spconv slab: variant %d is listed but not built for cin=%d", code, cin);
  return BEVAMD_ERR_UNSUPPORTED;
}

}  // namespace slab
}  // namespace bevamd
