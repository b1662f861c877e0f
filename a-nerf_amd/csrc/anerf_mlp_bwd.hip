// anerf_mlp_bwd.hip -- backward kernels of the fused MLP (gfx950): k_mlp_bwd (backward-data on the W^T image) and
// k_mlp_bwd_in (input gradients for pose optimisation / frame codes).  Same register-resident transposed fp32-MFMA
// scheme as the forward (see anerf_mlp.hip / DESIGN.md 4.2).
#include "anerf_mlp_common.h"

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

// act[i] <- act[i] * (saved[i] > 0)
__global__ __launch_bounds__(256) void k_mlp_bwd(const BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe pipe;
  pipe.init(A.packed_t, smem, wave, lane, A.nstages);
  pipe.issue(0);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  const f32x4 dr = *reinterpret_cast<const f32x4*>(A.draw + pc * 4);

  float d[128];
  f32x16 acc[8];
  // ---- rgb head: dg = Wc^T dc ; dzv = dg * [g > 0]
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = 32 * nb + 8 * q + 4 * h;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(A.aux + AUX_WC + o);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(A.aux + AUX_WC + 128 + o);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(A.aux + AUX_WC + 256 + o);
      d[nb * 16 + 4 * q + 0] = w0.x * dr.x + w1.x * dr.y + w2.x * dr.z;
      d[nb * 16 + 4 * q + 1] = w0.y * dr.x + w1.y * dr.y + w2.y * dr.z;
      d[nb * 16 + 4 * q + 2] = w0.z * dr.x + w1.z * dr.y + w2.z * dr.z;
      d[nb * 16 + 4 * q + 3] = w0.w * dr.x + w1.w * dr.y + w2.w * dr.z;
    }
#pragma unroll
  for (int i = 64; i < 128; ++i) d[i] = 0.f;
  relu_mask<4>(d, A.save_g + pc * 128, h);
  if (valid) store_row<4>(A.dzv + p * 128, d, h);
  // ---- view layer, feature columns: df = Wv[:, :256]^T dzv      (16 k-groups of the 128 view units)
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
#pragma unroll
  for (int kg = 0; kg < 16; ++kg) kgroup<8>(pipe, acc, kg, d[4 * kg], d[4 * kg + 1], d[4 * kg + 2], d[4 * kg + 3]);
  to_hidden<8, false>(d, acc);
  if (valid) store_row<8>(A.df + p * 256, d, h);
  // ---- feature layer + density head: dh7 = Wf^T df + w_alpha * dsigma
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wa = *reinterpret_cast<const f32x4*>(A.aux + AUX_WA + 32 * nb + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = wa.x * dr.w;
      acc[nb][4 * q + 1] = wa.y * dr.w;
      acc[nb][4 * q + 2] = wa.z * dr.w;
      acc[nb][4 * q + 3] = wa.w * dr.w;
    }
  hidden_part<8, 0>(pipe, acc, d);
  to_hidden<8, false>(d, acc);
  relu_mask<8>(d, A.save_h + (7 * A.Ppad + pc) * 256, h);
  if (valid) store_row<8>(A.dz + (7 * A.Ppad + p) * 256, d, h);
  // ---- trunk: dz_{l-1} = (W_l^T dz_l) * [h_{l-1} > 0],  l = 7..1   (W_5: hidden columns only)
#pragma unroll 1
  for (int L = 7; L >= 1; --L) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    hidden_part<8, 0>(pipe, acc, d);
    to_hidden<8, false>(d, acc);
    relu_mask<8>(d, A.save_h + ((long long)(L - 1) * A.Ppad + pc) * 256, h);
    if (valid) store_row<8>(A.dz + ((long long)(L - 1) * A.Ppad + p) * 256, d, h);
  }
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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd_in), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
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
  const size_t lds = 2 * STAGE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_mlp_bwd, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd");
}

}  // namespace anerf
