// Per-dtype instantiation of the tiled sparse-convolution kernels (included by spconv_tile_{f16,bf16}.hip).
#pragma once
#include "spconv_tile.h"

namespace bevamd {
namespace tile {

template <int DT, int CINP, int NT, int MT, int NW, int SPS>
static int run_resident(const Args& a, hipStream_t stream) {
  constexpr int CPO = SPS * Chunks<CINP>::CPB;
  const size_t lds = (size_t)StepShape<CINP, CPO>::nsteps(a.K) * CPO * NT * 1024 + (size_t)(NW * a.K * 16 * MT + 3) / 4 * 16 + (size_t)NW * EpiScratch<NT>::U4 * 16;
  if (lds > 65536) {
    set_error("spconv tiled: resident variant needs %zu B of LDS (> 64 KiB)", lds);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  const long long ntiles = ((long long)a.m_cap + 16 * MT - 1) / (16 * MT);
  long long per_cu = 163840 / (long long)(lds + 512);          // workgroups per CU by LDS
  const long long by_threads = 2048 / (NW * 64);               // and by wave slots
  if (per_cu > by_threads) per_cu = by_threads;
  if (per_cu < 1) per_cu = 1;
  long long blocks = (ntiles + NW - 1) / NW;
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;
  blocks = (blocks + 7) / 8 * 8;
  spconv_resident_kernel<DT, CINP, NT, MT, NW, CPO><<<dim3((unsigned)blocks), dim3(NW * 64), lds, stream>>>(a);
  BEVAMD_LAUNCH_CHECK("spconv_resident");
  return BEVAMD_OK;
}

template <int DT, int CINP, int NT, int MT, int NW, int SPS>
static int run_stream(const Args& a, hipStream_t stream) {
  constexpr int CPO = SPS * Chunks<CINP>::CPB;
  const size_t lds = (size_t)2 * CPO * NT * 1024 + (size_t)(NW * a.K * 16 * MT + 3) / 4 * 16 + (size_t)NW * EpiScratch<NT>::U4 * 16;  // filter ring + neighbour tables
  if (lds > 160 * 1024) {
    set_error("spconv tiled: stream variant needs %zu B of LDS (> 160 KiB)", lds);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  if (lds > 65536) {  // gfx950 has 160 KiB per CU; above 64 KiB the launch needs the opt-in
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute((const void*)&spconv_stream_kernel<DT, CINP, NT, MT, NW, CPO>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError();
      raised = true;
    }
  }
  constexpr int BM = NW * 16 * MT;
  const long long nblk = ((long long)a.m_cap + BM - 1) / BM;
  const long long blocks = (nblk + 7) / 8 * 8;
  spconv_stream_kernel<DT, CINP, NT, MT, NW, CPO><<<dim3((unsigned)blocks), dim3(NW * 64), lds, stream>>>(a);
  BEVAMD_LAUNCH_CHECK("spconv_stream");
  return BEVAMD_OK;
}

// which (MT, NW, SPS) combinations are built for a shape: bounded by registers (gathered sets 2*MT*CPO*4 VGPRs,
// staged filter CPO*NT/NW*4, accumulators MT*NT*4) and by 64 KiB of LDS for the stream ring (2*CPO*NT KiB)
template <int CINP, int NT, int MT, int NW, int SPS>
constexpr bool stream_built() {
  constexpr int CPO = SPS * Chunks<CINP>::CPB;
  return 2 * CPO * NT <= 64 && 2 * MT * CPO * 4 + (CPO * NT + NW - 1) / NW * 4 + MT * NT * 4 <= 200;
}
template <int CINP, int NT, int MT, int SPS>
constexpr bool resident_built() {
  constexpr int CPO = SPS * Chunks<CINP>::CPB;
  return CINP <= 32 && NT <= 2 && 2 * MT * CPO * 4 + MT * NT * 4 <= 160;
}

// profiling builds only (-DBEVAMD_PROFILING, `python -m bevfusion_amd.build --profiling`): variant 9000 + mask launches the
// shipped stream configuration of the 64->64 / 128->128 layers with parts of the kernel compiled out (see WaveTile's ABL;
// results are wrong by construction) — tools/sweep_spconv.py --ablate.  The shipped library does not contain them.
#ifdef BEVAMD_PROFILING
template <int DT, int CINP, int NT, int MT, int NW, int SPS, int ABL>
static int run_ablation(const Args& a, hipStream_t stream) {
  constexpr int CPO = SPS * Chunks<CINP>::CPB;
  const size_t lds = (size_t)2 * CPO * NT * 1024 + (size_t)(NW * a.K * 16 * MT + 3) / 4 * 16 + (size_t)NW * EpiScratch<NT>::U4 * 16;
  if (lds > 65536)
    (void)hipFuncSetAttribute((const void*)&spconv_stream_kernel<DT, CINP, NT, MT, NW, CPO, ABL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  constexpr int BM = NW * 16 * MT;
  const long long blocks = ((((long long)a.m_cap + BM - 1) / BM) + 7) / 8 * 8;
  spconv_stream_kernel<DT, CINP, NT, MT, NW, CPO, ABL><<<dim3((unsigned)blocks), dim3(NW * 64), lds, stream>>>(a);
  BEVAMD_LAUNCH_CHECK("spconv_stream(ablation)");
  return BEVAMD_OK;
}
#endif

template <int DT, int CINP, int NT>
static int run_shape(const Args& a, int variant, hipStream_t stream) {
  if (variant >= 9000) {
#ifdef BEVAMD_PROFILING
    if constexpr (DT == T_F16 && ((CINP == 64 && NT == 4) || (CINP == 128 && NT == 8))) {
      constexpr int MT = CINP == 64 ? 2 : 1, NW = CINP == 64 ? 4 : 8;
      switch (variant - 9000) {
#define BEVAMD_ABL(M) case M: return run_ablation<DT, CINP, NT, MT, NW, 1, M>(a, stream)
        BEVAMD_ABL(0); BEVAMD_ABL(1); BEVAMD_ABL(2); BEVAMD_ABL(4); BEVAMD_ABL(8); BEVAMD_ABL(16); BEVAMD_ABL(32);
        BEVAMD_ABL(3); BEVAMD_ABL(6); BEVAMD_ABL(7); BEVAMD_ABL(24); BEVAMD_ABL(31); BEVAMD_ABL(63); BEVAMD_ABL(62);
        BEVAMD_ABL(64); BEVAMD_ABL(127); BEVAMD_ABL(255); BEVAMD_ABL(191);
#undef BEVAMD_ABL
        default: break;
      }
    }
#endif
    set_error("spconv tiled: ablation variant %d needs a -DBEVAMD_PROFILING build of the library (and the 64->64 / 128->128 fp16 shapes)", variant);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  if (variant == 0) {
    // measured on MI355X over the SparseEncoder layer shapes (tools/sweep_spconv.py, profiles/r01_spconv_sweep.txt)
    const size_t img = (size_t)StepShape<CINP, Chunks<CINP>::CPB>::nsteps(a.K) * Chunks<CINP>::CPB * NT * 1024;  // unpadded image
    if (NT >= 8) variant = 2121;
    else if (CINP >= 64) variant = 2211;
    else if (CINP == 32) variant = 2213;   // 3 offsets per barrier: 37.7 vs 38.8 us on 32->32, 22.1 vs 23.2 on 32->64
    else if (resident_built<CINP, NT, 2, 1>() &&
             img + (size_t)8 * a.K * 32 * 4 + (size_t)8 * EpiScratch<NT>::U4 * 16 <= 65536) variant = 1221;   // 8 waves
    else if (resident_built<CINP, NT, 2, 1>() &&
             img + (size_t)4 * a.K * 32 * 4 + (size_t)4 * EpiScratch<NT>::U4 * 16 <= 65536) variant = 1211;   // 4 waves
    else variant = 2212;
  }
#define BEVAMD_RES(MT, NW, SPS)                                                                       \
  case 1000 + MT * 100 + (NW / 4) * 10 + SPS:                                                          \
    if constexpr (resident_built<CINP, NT, MT, SPS>()) return run_resident<DT, CINP, NT, MT, NW, SPS>(a, stream); \
    break
#define BEVAMD_STR(MT, NW, SPS)                                                                       \
  case 2000 + MT * 100 + (NW / 4) * 10 + SPS:                                                          \
    if constexpr (stream_built<CINP, NT, MT, NW, SPS>()) return run_stream<DT, CINP, NT, MT, NW, SPS>(a, stream); \
    break
  switch (variant) {
    BEVAMD_RES(2, 8, 1); BEVAMD_RES(2, 8, 2); BEVAMD_RES(2, 8, 3);
    BEVAMD_RES(4, 8, 1); BEVAMD_RES(4, 8, 2);
    BEVAMD_RES(2, 4, 1);
    BEVAMD_STR(1, 4, 1); BEVAMD_STR(1, 4, 2); BEVAMD_STR(1, 4, 3);
    BEVAMD_STR(1, 8, 1); BEVAMD_STR(1, 8, 2); BEVAMD_STR(1, 8, 3);
    BEVAMD_STR(2, 4, 1); BEVAMD_STR(2, 4, 2); BEVAMD_STR(2, 4, 3);
    BEVAMD_STR(2, 8, 1); BEVAMD_STR(2, 8, 2);
    default: break;
  }
#undef BEVAMD_RES
#undef BEVAMD_STR
  set_error("spconv tiled: variant %d is not built for cin_pad=%d, cout tiles=%d", variant, CINP, NT);
  return BEVAMD_ERR_UNSUPPORTED;
}

template <int DT>
int launch_impl(const Args& a, int cinp, int nt, int variant, hipStream_t stream) {
#define BEVAMD_SHAPE(C, N) if (cinp == C && nt == N) return run_shape<DT, C, N>(a, variant, stream)
  BEVAMD_SHAPE(8, 1);
  BEVAMD_SHAPE(8, 2);
  BEVAMD_SHAPE(8, 4);
  BEVAMD_SHAPE(8, 8);
  BEVAMD_SHAPE(16, 1);
  BEVAMD_SHAPE(16, 2);
  BEVAMD_SHAPE(16, 4);
  BEVAMD_SHAPE(16, 8);
  BEVAMD_SHAPE(32, 1);
  BEVAMD_SHAPE(32, 2);
  BEVAMD_SHAPE(32, 4);
  BEVAMD_SHAPE(32, 8);
  BEVAMD_SHAPE(64, 1);
  BEVAMD_SHAPE(64, 2);
  BEVAMD_SHAPE(64, 4);
  BEVAMD_SHAPE(64, 8);
  BEVAMD_SHAPE(128, 1);
  BEVAMD_SHAPE(128, 2);
  BEVAMD_SHAPE(128, 4);
  BEVAMD_SHAPE(128, 8);
#undef BEVAMD_SHAPE
  set_error("spconv tiled: no kernel for cin_pad=%d, cout tiles=%d", cinp, nt);
  return BEVAMD_ERR_UNSUPPORTED;
}

template <int DT>
int image_impl(const void* w, int K, int cin, int cout, int transpose_io, void* img, hipStream_t stream) {
  typedef typename Num<DT>::T T;
  const int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  const int cinp = pad_cin(cols), nt = pad_nt(rows);
  if (!cinp || !nt) {
    set_error("spconv filter image: channels %d -> %d exceed 128", cols, rows);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  const size_t total = image_elems(K, cinp, nt);
  const int nchunks = image_chunks(K, cinp);
  dim3 grid((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), block(256);
#define BEVAMD_IMG(C) case C: spconv_filter_image_kernel<DT, C><<<grid, block, 0, stream>>>((const T*)w, K, cin, cout, nt, nchunks, transpose_io, (T*)img); break
  switch (cinp) {
    BEVAMD_IMG(8);
    BEVAMD_IMG(16);
    BEVAMD_IMG(32);
    BEVAMD_IMG(64);
    default: BEVAMD_IMG(128);
  }
#undef BEVAMD_IMG
  BEVAMD_LAUNCH_CHECK("spconv_filter_image");
  return BEVAMD_OK;
}

}  // namespace tile
}  // namespace bevamd
