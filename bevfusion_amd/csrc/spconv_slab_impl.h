// Per-dtype instantiation of the slab (staged-rows) submanifold convolution (included by spconv_slab_{f16,bf16}.hip).
#pragma once
#include "spconv_slab.h"

namespace bevamd {
namespace slab {

// variant = KC*1000 + MT*100 + (NW/4)*10 + SPS   (KC: channels staged per row, 32 | 64; SPS: kernel taps per barrier, 1 | 3)
struct Shape { int kc, mt, nw, sps, cap; };

template <int DT, int KC, int CIN, int NT, int MT, int NW, int SPS, int CAP>
static int run(const SlabArgs& sa, hipStream_t stream) {
  typedef Plan<KC, CIN, NT, MT, NW, SPS, CAP> P;
  static_assert((CAP + 1) * P::RB < 65536, "row offsets are packed into 16 bits");
  static_assert(P::BYTES <= 160 * 1024, "LDS plan exceeds the CU");
  auto kern = &spconv_slab_kernel<DT, KC, CIN, NT, MT, NW, SPS, CAP>;
  if (P::BYTES > 65536) {
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError();
      raised = true;
    }
  }
  const long long nblk = ((long long)sa.a.m_cap + P::BM - 1) / P::BM;
  const long long blocks = (nblk + 7) / 8 * 8;
  kern<<<dim3((unsigned)blocks), dim3(NW * 64), P::BYTES, stream>>>(sa);
  BEVAMD_LAUNCH_CHECK("spconv_slab");
  return BEVAMD_OK;
}

// the built configurations per input width (first entry = default)
static inline const Shape* shapes_of(int cin, int* n) {
  static const Shape s32[] = {{32, 2, 4, 3, 192}, {32, 4, 4, 3, 384}, {32, 2, 4, 1, 192}};
  static const Shape s64[] = {{64, 2, 4, 1, 192}, {64, 2, 8, 1, 320}, {32, 2, 4, 1, 192}, {32, 2, 4, 3, 192}};
  static const Shape s128[] = {{64, 2, 4, 1, 144}, {32, 2, 4, 1, 192}, {64, 1, 8, 1, 160}};
  switch (cin) {
    case 32: *n = 3; return s32;
    case 64: *n = 4; return s64;
    case 128: *n = 3; return s128;
    default: *n = 0; return nullptr;
  }
}
static inline int variant_code(const Shape& s) { return s.kc * 1000 + s.mt * 100 + (s.nw / 4) * 10 + s.sps; }
static inline const Shape* find_shape(int cin, int variant) {
  int n = 0;
  const Shape* s = shapes_of(cin, &n);
  if (!s) return nullptr;
  if (variant == 0) return s;
  for (int i = 0; i < n; ++i)
    if (variant_code(s[i]) == variant) return s + i;
  return nullptr;
}

template <int DT>
int launch_impl(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream) {
  const Shape* s = find_shape(cin, variant);
  if (!s || nt != cin / 16) {
    set_error("spconv slab: no kernel for cin=%d, cout tiles=%d, variant=%d", cin, nt, variant);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  const int code = variant_code(*s);
#define BEVAMD_SLAB(CIN, KC, MT, NW, SPS, CAP) \
  if (cin == CIN && code == KC * 1000 + MT * 100 + (NW / 4) * 10 + SPS) return run<DT, KC, CIN, CIN / 16, MT, NW, SPS, CAP>(sa, stream)
  BEVAMD_SLAB(32, 32, 2, 4, 3, 192);
  BEVAMD_SLAB(32, 32, 4, 4, 3, 384);
  BEVAMD_SLAB(32, 32, 2, 4, 1, 192);
  BEVAMD_SLAB(64, 64, 2, 4, 1, 192);
  BEVAMD_SLAB(64, 64, 2, 8, 1, 320);
  BEVAMD_SLAB(64, 32, 2, 4, 1, 192);
  BEVAMD_SLAB(64, 32, 2, 4, 3, 192);
  BEVAMD_SLAB(128, 64, 2, 4, 1, 144);
  BEVAMD_SLAB(128, 32, 2, 4, 1, 192);
  BEVAMD_SLAB(128, 64, 1, 8, 1, 160);
#undef BEVAMD_SLAB
  set_error("spconv slab: variant %d is listed but not built for cin=%d", code, cin);
  return BEVAMD_ERR_UNSUPPORTED;
}

}  // namespace slab
}  // namespace bevamd
