// anerf_mlp_bwd.hip -- backward kernels of the fused MLP (gfx950): k_mlp_bwd (backward-data on the W^T image) and
// k_mlp_bwd_in (input gradients for pose optimisation / frame codes).  Same register-resident transposed fp32-MFMA
// scheme as the forward (see anerf_mlp.hip / DESIGN.md 4.2).
#include "anerf_mlp_common.h"
#include "anerf_fwd_common.h"

namespace anerf {

// backward-data kernel: d(raw) -> d(pre-activation) of every layer, same register-resident transposed scheme
// with the W^T weight image (anerf_layout which=1).  Per layer l:  dh_{l-1} = W_l^T dz_l ;  dz_{l-1} = dh_{l-1}
// * [h_{l-1} > 0].  Writes dz0..dz7 [8][Ppad][256], dF [Ppad][256], dZv [Ppad][128] for the weight-gradient GEMMs
// (anerf_gemm.hip).  Autograd of NeRF.forward (core/networks/nerf.py:94-148) w.r.t. activations.
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
  const float* packed_t;   // W^T image
  const float* aux;        // natural-order head weights (forward aux)
  const float* draw;       // [P][4]
  const float* save_h;     // [8][Ppad][256]
  const float* save_g;     // [Ppad][128]
  float* dz;               // [8][Ppad][256]
  float* df;               // [Ppad][256]
  float* dzv;              // [Ppad][128]
  long long P, Ppad;
  int nstages;
};

// ---- helpers on the accumulator layout (register r of block nb, lane half h = feature 32nb + (r&3) + 8(r>>2) + 4h)
template <int NB>
__device__ __forceinline__ void load_mask(f32x4 (&mk)[32], const float* __restrict__ row_h) {
#ifdef ANERF_EXP_BWD_NOMASK   // ablation build only: masks are not loaded (results are wrong)
#pragma unroll
  for (int i = 0; i < 4 * NB; ++i) mk[i] = f32x4{1.f, 1.f, 1.f, 1.f};
  return;
#endif
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) mk[4 * nb + q] = *reinterpret_cast<const f32x4*>(row_h + 32 * nb + 8 * q);
}
// acc <- acc * [saved activation > 0], in place (the result is read directly as the next layer's MFMA B operands)
template <int NB>
__device__ __forceinline__ void mask_pass(f32x16 (&acc)[NB], const f32x4 (&mk)[32]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[nb][4 * q + 0] = mk[4 * nb + q].x > 0.f ? acc[nb][4 * q + 0] : 0.f;
      acc[nb][4 * q + 1] = mk[4 * nb + q].y > 0.f ? acc[nb][4 * q + 1] : 0.f;
      acc[nb][4 * q + 2] = mk[4 * nb + q].z > 0.f ? acc[nb][4 * q + 2] : 0.f;
      acc[nb][4 * q + 3] = mk[4 * nb + q].w > 0.f ? acc[nb][4 * q + 3] : 0.f;
    }
}
template <int NB>
__device__ __forceinline__ void store_acc(float* __restrict__ row_h, const f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
      *reinterpret_cast<f32x4*>(row_h + 32 * nb + 8 * q) = o;
    }
}
template <int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
}

// One trunk layer of the backward:  out = (W^T prev) * [saved h > 0], stored as dz; `prev` stays intact (it is the
// B operand).  The ReLU-mask row of the layer's output is fetched when the layer starts, so the 32 loads have the MFMA
// stream of the layer to land; the dz stores are issued after the in-place mask pass and drain under the next layer's
// first stage.  KG0 = 32 (any non-zero multiple of the k-groups per stage): the weight stream is one continuous
// segment, so the first fragments of every layer come from the cross-barrier register prefetch.
// Known cost: all 256 CUs run their tiles in lock step, so these layer-sized bursts (32 MB chip-wide each) outlast
// the one stage (3.4 us) they have before the next s_waitcnt vmcnt(0): ablations say no stores -8 %, no mask loads
// -8 %.  Slicing the traffic per stage (block j at stage j) was tried twice; both variants pushed the kernel past
// 512 registers (2 x 128 accumulators + 128 mask values + prefetch), spilled pointers, and the scratch reloads --
// in-order VMEM ops behind the weight-stream loads -- made it slower (2.6 vs 2.4 ms).  The clean fix is to apply the
// mask one block ahead of its use in the NEXT layer (32 live mask registers instead of 128); not done yet.
__device__ __forceinline__ void bwd_layer(Pipe3F& pipe, f32x16 (&out)[8], const f32x16 (&prev)[8], f32x4 (&mk)[32],
                                          const float* __restrict__ mask_row_h, float* __restrict__ dz_row_h, bool valid,
                                          bool last) {
  load_mask<8>(mk, mask_row_h);
  zero_acc<8>(out);
  hidden_part<8, 32>(pipe, out, prev, false, last);
  mask_pass<8>(out, mk);
  if (valid) store_acc<8>(dz_row_h, out);
}

__global__ __launch_bounds__(256) void k_mlp_bwd(const BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe3F pipe;
  pipe.init(A.packed_t, smem, wave, lane, A.nstages);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
#ifdef ANERF_EXP_BWD_NOSTORE   // ablation build only: results are not written
  const bool valid = p < A.P && A.nstages < 0;   // never true at run time, opaque to the compiler
#else
  const bool valid = p < A.P;
#endif
  const long long pc = p < A.P ? p : A.P - 1;
  // head rows (w_c, w_alpha) -> LDS copy of the aux image, read back as float4 per 8-feature group
  float* aux_l = reinterpret_cast<float*>(smem + LDS_AUX_OFF);
  for (int i = tid; i < AUX_FLOATS / 4; i += 256)
    reinterpret_cast<f32x4*>(aux_l)[i] = reinterpret_cast<const f32x4*>(A.aux)[i];
  const float* aux_h = aux_l + 4 * h;
  const f32x4 dr = *reinterpret_cast<const f32x4*>(A.draw + pc * 4);
  f32x4 mk[32];
  load_mask<4>(mk, A.save_g + pc * 128 + 4 * h);
  pipe.begin();   // barrier: aux visible, weight stages 0/1 landed
  pipe.prime();

  f32x16 accA[8], accB[8];   // ping-pong: a layer's output set is the next layer's B-operand set
  f32x16 accv[4];
  // ---- rgb head (VALU): dg = Wc^T dc ; dzv = dg * [g > 0]
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = 32 * nb + 8 * q;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(aux_h + AUX_WC + o);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(aux_h + AUX_WC + 128 + o);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(aux_h + AUX_WC + 256 + o);
      accv[nb][4 * q + 0] = w0.x * dr.x + w1.x * dr.y + w2.x * dr.z;
      accv[nb][4 * q + 1] = w0.y * dr.x + w1.y * dr.y + w2.y * dr.z;
      accv[nb][4 * q + 2] = w0.z * dr.x + w1.z * dr.y + w2.z * dr.z;
      accv[nb][4 * q + 3] = w0.w * dr.x + w1.w * dr.y + w2.w * dr.z;
    }
  mask_pass<4>(accv, mk);
  if (valid) store_acc<4>(A.dzv + p * 128 + 4 * h, accv);
  // ---- view layer, feature columns: df = Wv[:, :256]^T dzv      (16 k-groups over the 128 view units)
  load_mask<8>(mk, A.save_h + (7 * A.Ppad + pc) * 256 + 4 * h);     // h7 mask, needed after the feature layer
  zero_acc<8>(accA);
#pragma unroll
  for (int kg = 0; kg < 16; ++kg)
    kgroup<8>(pipe, accA, kg, kg == 0, false, accv[kg >> 2][4 * (kg & 3) + 0], accv[kg >> 2][4 * (kg & 3) + 1],
              accv[kg >> 2][4 * (kg & 3) + 2], accv[kg >> 2][4 * (kg & 3) + 3]);
  if (valid) store_acc<8>(A.df + p * 256 + 4 * h, accA);
  // ---- feature layer + density head: dh7 = Wf^T df + w_alpha * dsigma ; dz7 = dh7 * [h7 > 0]
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wa = *reinterpret_cast<const f32x4*>(aux_h + AUX_WA + 32 * nb + 8 * q);
      accB[nb][4 * q + 0] = wa.x * dr.w;
      accB[nb][4 * q + 1] = wa.y * dr.w;
      accB[nb][4 * q + 2] = wa.z * dr.w;
      accB[nb][4 * q + 3] = wa.w * dr.w;
    }
  hidden_part<8, 16>(pipe, accB, accA, false, false);
  mask_pass<8>(accB, mk);
  if (valid) store_acc<8>(A.dz + (7 * A.Ppad + p) * 256 + 4 * h, accB);
  // ---- trunk: dz_{l-1} = (W_l^T dz_l) * [h_{l-1} > 0],  l = 7..1   (W_5: hidden columns only)
  const float* hrow = A.save_h + pc * 256 + 4 * h;
  float* zrow = A.dz + p * 256 + 4 * h;
  const long long plane = A.Ppad * 256;
#pragma unroll 1
  for (int L = 7; L >= 3; L -= 2) {
    bwd_layer(pipe, accA, accB, mk, hrow + (L - 1) * plane, zrow + (L - 1) * plane, valid, false);
    bwd_layer(pipe, accB, accA, mk, hrow + (L - 2) * plane, zrow + (L - 2) * plane, valid, false);
  }
  bwd_layer(pipe, accA, accB, mk, hrow, zrow, valid, true);
}

// ------------------------------------------------------------------------------------------------
// input-gradient kernel (pose optimisation / frame codes only): gradients w.r.t. the ENCODED inputs, in stream
// column order, on the which=2 weight image:
//     dX'[p][432] = W0'^T dz0 + W5x'^T dz5          dU'[p][UW] = Wvu'^T dzv
// Output columns are produced 256 at a time (8 blocks); lane (m,h) receives exactly the columns whose forward
// B operands it generated (k-group 4*nb+q, half h), which is what k_encode_bwd consumes.
// ------------------------------------------------------------------------------------------------
struct BwdInArgs {
  const float* packed_i;
  const float* dz;     // [8][Ppad][256]
  const float* dzv;    // [Ppad][128]
  float* dx;           // [Ppad][432]
  float* du;           // [Ppad][UW]
  long long P, Ppad;
  int nstages, uw;
};

__device__ __forceinline__ void store_cols(float* __restrict__ row, const f32x16 (&acc)[8], int c0, int w, int h) {
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 32 * nb + 8 * q + 4 * h;
      if (c < w) {
        f32x4 o = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
        *reinterpret_cast<f32x4*>(row + c) = o;
      }
    }
}

__global__ __launch_bounds__(256) void k_mlp_bwd_in(const BwdInArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe pipe;
  pipe.init(A.packed_i, smem, wave, lane, A.nstages);
  pipe.issue(0);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  float d[128];
  f32x16 acc[8];
#pragma unroll 1
  for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    load_row<8>(d, A.dz + pc * 256, h);                          // dz0
    hidden_part<8, 0>(pipe, acc, d);
    load_row<8>(d, A.dz + (5 * A.Ppad + pc) * 256, h);           // dz5
    hidden_part<8, 0>(pipe, acc, d);
    if (valid) store_cols(A.dx + p * 432, acc, 256 * gi, 432, h);
  }
  const int ngu = (A.uw + 255) / 256;
#pragma unroll
  for (int i = 64; i < 128; ++i) d[i] = 0.f;
  load_row<4>(d, A.dzv + pc * 128, h);
#pragma unroll 1
  for (int gi = 0; gi < ngu; ++gi) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) kgroup<8>(pipe, acc, kg, d[4 * kg], d[4 * kg + 1], d[4 * kg + 2], d[4 * kg + 3]);
    if (valid) store_cols(A.du + p * A.uw, acc, 256 * gi, A.uw, h);
  }
}

int mlp_bwd_in_entry(const float* packed_i, const float* dz, const float* dzv, float* dx, float* du, long long P,
                     long long Ppad, int nstages, int uw, hipStream_t st) {
  BwdInArgs b;
  b.packed_i = packed_i; b.dz = dz; b.dzv = dzv; b.dx = dx; b.du = du; b.P = P; b.Ppad = Ppad; b.nstages = nstages; b.uw = uw;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = 2 * STAGE_BYTES;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_in), (int)lds, &lds_set);
  hipLaunchKernelGGL(k_mlp_bwd_in, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd_in");
}

int mlp_bwd_entry(const float* packed_t, const float* aux, const float* draw, const AnerfSaved* sv, float* dz, float* df,
                  float* dzv, long long P, int nstages, hipStream_t st) {
  BwdArgs b;
  b.packed_t = packed_t; b.aux = aux; b.draw = draw; b.save_h = sv->h; b.save_g = sv->g;
  b.dz = dz; b.df = df; b.dzv = dzv; b.P = P; b.Ppad = sv->p_pad; b.nstages = nstages;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = LDS_BONES_OFF;   // 3-slot weight ring + the aux copy (no bone staging in the backward)
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd), (int)lds, &lds_set);
  hipLaunchKernelGGL(k_mlp_bwd, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd");
}

}  // namespace anerf
