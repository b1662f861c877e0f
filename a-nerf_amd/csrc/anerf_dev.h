// anerf_dev.h -- shared device/host definitions for libanerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf.h"

namespace anerf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int TILE = 128;                       // samples per workgroup (4 waves x 32)
constexpr int FRAG_FLOATS = 256;                // one MFMA weight fragment x 4 k-steps: 64 lanes x float4
constexpr int FRAG_BYTES = 1024;
constexpr int STAGE_FRAGS = 32;                 // fragments per LDS stage
constexpr int STAGE_FLOATS = STAGE_FRAGS * FRAG_FLOATS;
constexpr int STAGE_BYTES = STAGE_FRAGS * FRAG_BYTES;   // 32 KiB
constexpr int MAX_TILE_RAYS = 18;               // rays a 128-sample tile can touch when N_samples >= 8
constexpr int MIN_SAMPLES = 8;
constexpr int MAX_SAMPLES = 512;

// aux image (natural order): biases of pts_linears.0..7, feature, views; alpha / rgb head weights and biases
constexpr int AUX_B0 = 0;          // 8 x 256
constexpr int AUX_BF = 2048;       // 256
constexpr int AUX_BV = 2304;       // 128
constexpr int AUX_WA = 2432;       // 256
constexpr int AUX_BA = 2688;       // 1 (+3 pad)
constexpr int AUX_WC = 2692;       // 3 x 128
constexpr int AUX_BC = 3076;       // 3 (+1 pad)
constexpr int AUX_FLOATS = 3080;

int set_error(int code, const char* msg);
int check_launch(const char* what);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device, and a host that
// drives several GPUs from one process (one thread per GPU, as nn.DataParallel does) reaches every launcher on each of them.
// `done` is the launcher's own static bit mask (bit = device ordinal).
inline void ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long* done) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done, __ATOMIC_RELAXED) & bit) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
}

// a schedule vector comes with its length (the pack kernels take the column of an element modulo the layer's input width).
// sched_scale() below hard-codes the trunk width 256 -- the only width config_ok() admits -- so the lengths must be those of
// the supported layouts: dim_x = 24 (1 + 2*7) + 72 = 432, u_width = 72 (1 + 2*multires_views) [+ 16 frame-code columns]
// in {72, 648, 664}.  A direct C caller with another length gets ANERF_E_SHAPE, not silently mis-scaled weight columns; the
// entry points that know the configuration compare with it exactly (sched_matches).
inline bool sched_ok(const AnerfNetParams& p) {
  const bool x_ok = p.sched_x ? p.sched_dim_x == 432 : true;
  const bool u_ok = p.sched_u ? (p.sched_dim_u == 72 || p.sched_dim_u == 648 || p.sched_dim_u == 664) : true;
  return x_ok && u_ok;
}

#if defined(__HIPCC__)
// sin and cos of x for |x| < ~1e4: Cody-Waite reduction by pi/2 (3 constants, exact products for |k| < 2^16)
// + cephes-style minimax polynomials on [-pi/4, pi/4]; max abs error ~1e-7 (covers 2^6 * distance, 2^3 * unit dir).
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
#ifdef ANERF_EXP_CHEAPSINCOS   // ablation build only: wrong values, ~2 VALU instead of ~25
  s = x * 0.5f; c = x + 1.0f; return;
#endif
  const float k = rintf(x * 0.636619772367581343f);
  float r = fmaf(k, -1.5703125f, x);
  r = fmaf(k, -4.837512969970703125e-4f, r);
  r = fmaf(k, -7.54978995489188216e-8f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  const float sr = fmaf(ps * r2, r, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  const float cr = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
  const int q = (int)k;
  const float ss = (q & 1) ? cr : sr;
  const float cc = (q & 1) ? sr : cr;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

// ABI revision 5, frequency schedule folded into the weight images: factor of element `off` of parameter tensor `id` (0..11 weights,
// 12..23 biases; torch Linear [out][in] row-major).  Only pts_linears.0 (id 0, in = dim_x), the input columns of the skip layer
// pts_linears.5 (id 5, in = dim_x + 256, encoded input first: nerf.py:139-141) and the columns of views_linears.0 behind its 256
// feature columns (id 10, in = 256 + u_width) consume the encoding.
__device__ __forceinline__ float sched_scale(const AnerfNetParams& P, int id, int off) {
  if (P.sched_x) {
    if (id == 0) return P.sched_x[off % P.sched_dim_x];
    if (id == 5) {
      const int c = off % (P.sched_dim_x + 256);
      return c < P.sched_dim_x ? P.sched_x[c] : 1.f;
    }
  }
  if (P.sched_u && id == 10) {
    const int c = off % (256 + P.sched_dim_u) - 256;
    return c >= 0 ? P.sched_u[c] : 1.f;
  }
  return 1.f;
}

// sample index -> ray index.  P < 2^32 is checked by the launchers (mlp_dispatch): one 32-bit division (~25 instructions)
// instead of the 64-bit software division (~150, three to five of them at the head of every tile).
__device__ __forceinline__ long long div_samples(long long p, int S) { return (long long)((unsigned)p / (unsigned)S); }

// sin and cos for |x| <= 1 (components of a unit vector): no reduction, Taylor to x^9 / x^10 (truncation 2.5e-8 / 2e-9).
__device__ __forceinline__ void sincos_unit_f32(float x, float& s, float& c) {
  const float x2 = x * x;
  float ps = fmaf(x2, 2.7557319224e-6f, -1.9841269841e-4f);
  ps = fmaf(ps, x2, 8.3333333333e-3f);
  ps = fmaf(ps, x2, -1.6666666667e-1f);
  s = fmaf(ps * x2, x, x);
  float pc = fmaf(x2, -2.7557319224e-7f, 2.4801587302e-5f);
  pc = fmaf(pc, x2, -1.3888888889e-3f);
  pc = fmaf(pc, x2, 4.1666666667e-2f);
  pc = fmaf(pc, x2, -0.5f);
  c = fmaf(pc, x2, 1.0f);
}

// 1 / x from v_rcp_f32 (1 ulp) + one Newton step: 3 VALU; the IEEE-correct `1.0f / x` hipcc emits is a 10-instruction
// v_div_scale / v_div_fmas / v_div_fixup sequence, issued into the MFMA stream ~60 times per tile
__device__ __forceinline__ float rcp_nr(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// cutoff gate w = 1 - sigmoid(tau * (dist - cutoff))  (core/cutoff_embedder.py:149-155), evaluated as
// 1 / (1 + exp(a)): same value without the cancellation of "1 - sigmoid".
__device__ __forceinline__ float cutoff_gate(float tau, float dist, float cutoff) {
  const float a = fminf(tau * (dist - cutoff), 80.f);   // exp stays finite (the Newton step of rcp_nr would make inf * 0)
  return rcp_nr(1.0f + __expf(a));
}
#endif

}  // namespace anerf
