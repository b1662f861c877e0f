// anerf_mlp.hip -- fused "encode + 11-layer MLP" forward kernel for gfx950 (MI355X, CDNA4).
//
// Replaces, per sample point, the reference op sequence
//   transform_batch_pts / transform_batch_rays      core/encoders.py:8-37
//   RelDistEncoder / VecNormEncoder                 core/encoders.py:110-122,181-193
//   CutoffEmbedder._embed (x2) + identity embedder  core/cutoff_embedder.py:111-174
//   run_network cat + NeRF.forward                  core/raycasters.py:557-577, core/networks/nerf.py:94-148
// The 1080-wide encoding and the 256-wide activations never exist in HBM.
//
// Execution model (see DESIGN.md for the derivation)
//   * workgroup = 4 waves = one 128-sample tile; every wave owns 32 samples for the whole network.
//   * each layer is computed TRANSPOSED on the fp32 matrix cores: D^T[n][m] = sum_k W[n][k] * act[m][k] with
//     v_mfma_f32_32x32x2_f32, A = weight fragment (from LDS), B = activation (a register), D = 32 features x
//     32 samples.  Lane l holds sample m = l&31; lanes 0-31 / 32-63 hold the two k-halves.  The accumulator layout
//     of layer i (feature n = 32nb + (r&3) + 8(r>>2) + 4(l>>5) in register r of block nb) IS the B-operand layout
//     of layer i+1, so activations stay in registers across all layers: no LDS round trip, no barrier for them.
//   * weights are pre-packed (anerf_pack.hip) into a linear stream of 1 KiB MFMA fragments in consumption order
//     and streamed HBM/L2 -> LDS with global_load_lds_dwordx4 through a 2 x 32 KiB ring, one barrier per stage
//     (= 8192 MFMA cycles per wave), shared by the 4 waves.
//   * the encoding is produced in registers just in time as B operands: lane half h owns joints
//     {j : ((j>>2)&1) == h}; bone matrices of the tile's rays are staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf_dev.h"

namespace anerf {

// ------------------------------------------------------------------------------------------------
// weight-stream pipe: global -> LDS ring (2 stages x 32 KiB), all 4 waves cooperate
// ------------------------------------------------------------------------------------------------
struct Pipe {
  const char* gsrc;   // per-lane source of this wave's first fragment of stage 0
  char* smem;
  unsigned wave_dst;  // wave-uniform LDS byte offset of this wave's 8 fragments inside a stage
  unsigned lane16;    // lane * 16
  unsigned cur;       // LDS byte offset (lane-relative) of the stage being consumed
  int stage;          // index of the next stage to consume
  int nstages;

  __device__ __forceinline__ void init(const float* packed, char* smem_, int wave, int lane, int nstages_) {
    gsrc = reinterpret_cast<const char*>(packed) + wave * (8 * FRAG_BYTES) + lane * 16;
    smem = smem_;
    wave_dst = wave * (8 * FRAG_BYTES);
    lane16 = lane * 16;
    cur = 0;
    stage = 0;
    nstages = nstages_;
  }
  __device__ __forceinline__ void issue(int s) {
    const char* g = gsrc + (size_t)s * STAGE_BYTES;
    char* l = smem + (s & 1) * STAGE_BYTES + wave_dst;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + i * FRAG_BYTES), (lds_ptr_t)(l + i * FRAG_BYTES), 16, 0, 0);
  }
  // Called before the first k-group of every stage.
  __device__ __forceinline__ void next_stage() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of stage `stage` has landed
    __syncthreads();                                   // ... everybody's has; slot (stage+1)&1 is no longer read
    if (stage + 1 < nstages) issue(stage + 1);
    cur = lane16 + (stage & 1) * STAGE_BYTES;
    ++stage;
  }
};

// One k-group (8 contraction indices: 4 from each lane half) against NB 32-row feature blocks.
// kg = k-group index relative to the segment start (compile-time after unrolling).
template <int NB>
__device__ __forceinline__ void kgroup(Pipe& pipe, f32x16 (&acc)[NB], int kg, float b0, float b1, float b2, float b3) {
  constexpr int KPS = STAGE_FRAGS / NB;  // k-groups per stage
  if (kg % KPS == 0) pipe.next_stage();
  const unsigned off = pipe.cur + (kg % KPS) * NB * FRAG_BYTES;
  f32x4 a[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) a[nb] = *reinterpret_cast<const f32x4*>(pipe.smem + off + nb * FRAG_BYTES);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].x, b0, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].y, b1, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].z, b2, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].w, b3, acc[nb], 0, 0, 0);
}

// acc[nb][r] <- bias[n(nb,r,h)]; natural-order bias vector, float4 per (nb,q).
template <int NB>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NB], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * nb + 8 * q + 4 * h);
      acc[nb][4 * q + 0] = b.x;
      acc[nb][4 * q + 1] = b.y;
      acc[nb][4 * q + 2] = b.z;
      acc[nb][4 * q + 3] = b.w;
    }
}

// 32 k-groups whose B operands are the 256 hidden activations held in registers.
template <int NB, int KG0>
__device__ __forceinline__ void hidden_part(Pipe& pipe, f32x16 (&acc)[NB], const float (&hin)[128]) {
#pragma unroll
  for (int kg = 0; kg < 32; ++kg)
    kgroup<NB>(pipe, acc, KG0 + kg, hin[4 * kg + 0], hin[4 * kg + 1], hin[4 * kg + 2], hin[4 * kg + 3]);
}

template <int NB, bool RELU>
__device__ __forceinline__ void to_hidden(float (&hin)[128], const f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) hin[nb * 16 + r] = RELU ? fmaxf(acc[nb][r], 0.f) : acc[nb][r];
}

// dot of the lane's 16*NB held activations with a natural-order weight row, reduced over both lane halves
template <int NB>
__device__ __forceinline__ float head_dot(const float (&act)[128], const float* __restrict__ wrow, int h) {
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + 32 * nb + 8 * q + 4 * h);
      s = fmaf(act[nb * 16 + 4 * q + 0], w.x, s);
      s = fmaf(act[nb * 16 + 4 * q + 1], w.y, s);
      s = fmaf(act[nb * 16 + 4 * q + 2], w.z, s);
      s = fmaf(act[nb * 16 + 4 * q + 3], w.w, s);
    }
  return s + __shfl_xor(s, 32);
}

// ------------------------------------------------------------------------------------------------
// the x-part (432 = 360 distance-PE + 72 bone-direction channels) as 54 k-groups.
// Encoded just in time (PRE = false) or read from a pre-encoded row (PRE = true, NeRF.forward seam).
// ------------------------------------------------------------------------------------------------
// STORE: also write the 4 operands of every k-group to xsave[8*kg + 4h .. +3] (stream column order X').
template <int LV, bool PRE, bool STORE>
__device__ __forceinline__ void x_part(Pipe& pipe, f32x16 (&acc)[8], const float (&v)[12], const float (&wv)[12],
                                       const float (&rh)[36], const float* __restrict__ xrow, int h,
                                       float* __restrict__ xsave) {
  auto KG = [&](int kg, float b0, float b1, float b2, float b3) __attribute__((always_inline)) {
    if constexpr (STORE) {
      if (xsave) {
        f32x4 o = {b0, b1, b2, b3};
        *reinterpret_cast<f32x4*>(xsave + 8 * kg + 4 * h) = o;
      }
    }
    kgroup<8>(pipe, acc, kg, b0, b1, b2, b3);
  };
  if constexpr (PRE) {
#pragma unroll
    for (int kg = 0; kg < 3 * (1 + 2 * LV); ++kg) {
      const float* c = xrow + 8 * kg + 4 * h;
      KG(kg, c[0], c[1], c[2], c[3]);
    }
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      float b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int a = (4 * g + t) / 3, c = (4 * g + t) % 3;
        b[t] = xrow[24 * (1 + 2 * LV) + 3 * (8 * (a >> 2) + (a & 3)) + c + 12 * h];
      }
      KG(3 * (1 + 2 * LV) + g, b[0], b[1], b[2], b[3]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 3; ++g)
      KG(g, v[4 * g] * wv[4 * g], v[4 * g + 1] * wv[4 * g + 1], v[4 * g + 2] * wv[4 * g + 2],
                v[4 * g + 3] * wv[4 * g + 3]);
#pragma unroll
    for (int f = 0; f < LV; ++f) {
      float sv[12], cv[12];
#pragma unroll
      for (int a = 0; a < 12; ++a) {
        float s, c;
        sincos_f32(v[a] * (float)(1 << f), s, c);
        sv[a] = s * wv[a];
        cv[a] = c * wv[a];
      }
#pragma unroll
      for (int g = 0; g < 3; ++g)
        KG(3 + 6 * f + g, sv[4 * g], sv[4 * g + 1], sv[4 * g + 2], sv[4 * g + 3]);
#pragma unroll
      for (int g = 0; g < 3; ++g)
        KG(6 + 6 * f + g, cv[4 * g], cv[4 * g + 1], cv[4 * g + 2], cv[4 * g + 3]);
    }
#pragma unroll
    for (int g = 0; g < 9; ++g)
      KG(3 * (1 + 2 * LV) + g, rh[4 * g], rh[4 * g + 1], rh[4 * g + 2], rh[4 * g + 3]);
  }
}

struct MlpArgs {
  const float* packed;
  const float* aux;
  const float* rays;
  const float* z;
  const float* skts;
  const float* cam;
  const float* codes;
  const float* cut_v;
  const float* cut_d;
  const float* x;  // PRE
  float* raw;
  // TRAIN: saved activations, row-major planes with Ppad rows (rows >= P are never written)
  float* save_h;   // [8][Ppad][256]  h0..h7 (post-ReLU)
  float* save_f;   // [Ppad][256]     feature (no activation)
  float* save_g;   // [Ppad][128]     view-layer output (post-ReLU)
  float* save_x;   // [Ppad][432]     x in stream column order
  float* save_u;   // [Ppad][UW]      view inputs (D, code) in stream column order
  long long P;
  long long Ppad;
  long long skt_stride;
  int S, N, ray_stride, n_codes, x_width, nstages;
  float tau_v, tau_d;
};

// row-major store of the lane's 16*NB activations: features 32nb+8q+4h .. +3 of row `row`
template <int NB>
__device__ __forceinline__ void store_row(float* __restrict__ row, const float (&a)[128], int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {a[nb * 16 + 4 * q], a[nb * 16 + 4 * q + 1], a[nb * 16 + 4 * q + 2], a[nb * 16 + 4 * q + 3]};
      *reinterpret_cast<f32x4*>(row + 32 * nb + 8 * q + 4 * h) = o;
    }
}

template <int LV, int LD, int CODE, bool PRE, bool TRAIN>
__global__ __launch_bounds__(256) void k_mlp_fwd(const MlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;

  Pipe pipe;
  pipe.init(A.packed, smem, wave, lane, A.nstages);
  pipe.issue(0);

  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  const bool save = TRAIN && valid;

  float v[12], wv[12], rh[36];
  float dray[3] = {0.f, 0.f, 0.f};
  int lr = 0;
  long long ray = 0;
  const float* xrow = nullptr;
  const f32x4* bones4 = reinterpret_cast<const f32x4*>(smem + 2 * STAGE_BYTES);

  if constexpr (PRE) {
    xrow = A.x + pc * A.x_width;
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] = wv[a] = 0.f;
#pragma unroll
    for (int a = 0; a < 36; ++a) rh[a] = 0.f;
  } else {
    // ---- stage the bone matrices (rows 0..2 of each 4x4 world->bone matrix) of this tile's rays in LDS
    const long long tile_p0 = (long long)blockIdx.x * TILE;
    const long long ray0 = tile_p0 / A.S;
    long long ray1 = (tile_p0 + TILE - 1) / A.S;
    if (ray1 > A.N - 1) ray1 = A.N - 1;
    ray = pc / A.S;
    const int n_stage_rays = A.skt_stride == 0 ? 1 : (int)(ray1 - ray0 + 1);
    lr = A.skt_stride == 0 ? 0 : (int)(ray - ray0);
    f32x4* bw = reinterpret_cast<f32x4*>(smem + 2 * STAGE_BYTES);
    for (int i = tid; i < n_stage_rays * 72; i += 256) {
      const int ri = i / 72, rem = i - ri * 72, j = rem / 3, row = rem - 3 * j;
      bw[i] = *reinterpret_cast<const f32x4*>(A.skts + (ray0 + ri) * A.skt_stride + j * 16 + row * 4);
    }
    __syncthreads();
    const float* rp = A.rays + ray * A.ray_stride;
    const float z = A.z[pc];
    dray[0] = rp[3];
    dray[1] = rp[4];
    dray[2] = rp[5];
    const float x0 = fmaf(dray[0], z, rp[0]), x1 = fmaf(dray[1], z, rp[1]), x2 = fmaf(dray[2], z, rp[2]);
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const int j = 8 * (a >> 2) + 4 * h + (a & 3);
      const f32x4 r0 = bones4[(lr * 24 + j) * 3 + 0], r1 = bones4[(lr * 24 + j) * 3 + 1],
                  r2 = bones4[(lr * 24 + j) * 3 + 2];
      const float y0 = r0.x * x0 + r0.y * x1 + r0.z * x2 + r0.w;
      const float y1 = r1.x * x0 + r1.y * x1 + r1.z * x2 + r1.w;
      const float y2 = r2.x * x0 + r2.y * x1 + r2.z * x2 + r2.w;
      const float n = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
      const float inv = 1.0f / fmaxf(n, 1e-12f);
      v[a] = n;
      rh[3 * a + 0] = y0 * inv;
      rh[3 * a + 1] = y1 * inv;
      rh[3 * a + 2] = y2 * inv;
      wv[a] = cutoff_gate(A.tau_v, n, A.cut_v[j]);
    }
  }

  constexpr int DIMX = 24 * (1 + 2 * LV) + 72;
  constexpr int DIMD = 72 * (1 + 2 * LD);
  constexpr int UW = DIMD + CODE;
  float hin[128];
  f32x16 acc[8];

  // ---- layer 0: x(432) -> 256
  init_bias<8>(acc, A.aux + AUX_B0, h);
  x_part<LV, PRE, TRAIN>(pipe, acc, v, wv, rh, xrow, h, save ? A.save_x + p * DIMX : nullptr);
  to_hidden<8, true>(hin, acc);
  if (save) store_row<8>(A.save_h + p * 256, hin, h);
  // ---- layers 1..4
#pragma unroll 1
  for (int L = 1; L <= 4; ++L) {
    init_bias<8>(acc, A.aux + AUX_B0 + 256 * L, h);
    hidden_part<8, 0>(pipe, acc, hin);
    to_hidden<8, true>(hin, acc);
    if (save) store_row<8>(A.save_h + ((long long)L * A.Ppad + p) * 256, hin, h);
  }
  // ---- layer 5: [x(432); h4(256)] -> 256   (skip connection: x is re-encoded, never stored)
  init_bias<8>(acc, A.aux + AUX_B0 + 256 * 5, h);
  x_part<LV, PRE, false>(pipe, acc, v, wv, rh, xrow, h, nullptr);
  hidden_part<8, 3 * (1 + 2 * LV) + 9>(pipe, acc, hin);
  to_hidden<8, true>(hin, acc);
  if (save) store_row<8>(A.save_h + (5 * A.Ppad + p) * 256, hin, h);
  // ---- layers 6, 7
#pragma unroll 1
  for (int L = 6; L <= 7; ++L) {
    init_bias<8>(acc, A.aux + AUX_B0 + 256 * L, h);
    hidden_part<8, 0>(pipe, acc, hin);
    to_hidden<8, true>(hin, acc);
    if (save) store_row<8>(A.save_h + ((long long)L * A.Ppad + p) * 256, hin, h);
  }
  // ---- density head (VALU): sigma_raw = w_alpha . h7 + b_alpha
  const float sigma_raw = head_dot<8>(hin, A.aux + AUX_WA, h) + A.aux[AUX_BA];
  // ---- feature layer (no activation)
  init_bias<8>(acc, A.aux + AUX_BF, h);
  hidden_part<8, 0>(pipe, acc, hin);
  to_hidden<8, false>(hin, acc);
  if (save) store_row<8>(A.save_f + p * 256, hin, h);
  // ---- view layer: [feature(256); D(72*(1+2LD)); code(CODE)] -> 128, ReLU
  f32x16 accv[4];
  init_bias<4>(accv, A.aux + AUX_BV, h);
  hidden_part<4, 0>(pipe, accv, hin);
  float* usave = save ? A.save_u + p * UW : nullptr;
  auto KGV = [&](int kgu, float b0, float b1, float b2, float b3) __attribute__((always_inline)) {
    if constexpr (TRAIN) {
      if (usave) {
        f32x4 o = {b0, b1, b2, b3};
        *reinterpret_cast<f32x4*>(usave + 8 * kgu + 4 * h) = o;
      }
    }
    kgroup<4>(pipe, accv, 32 + kgu, b0, b1, b2, b3);
  };
  if constexpr (PRE) {
#pragma unroll
    for (int b = 0; b < 1 + 2 * LD; ++b)
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        float bb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int a = (4 * g + t) / 3, c = (4 * g + t) % 3;
          bb[t] = xrow[DIMX + 72 * b + 3 * (8 * (a >> 2) + (a & 3)) + c + 12 * h];
        }
        KGV(9 * b + g, bb[0], bb[1], bb[2], bb[3]);
      }
  } else {
    // per-ray unit direction in each owned bone frame, gated per sample by the distance gate (tau_d, cut_d)
    float e[36], wd[12];
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const int j = 8 * (a >> 2) + 4 * h + (a & 3);
      const f32x4 r0 = bones4[(lr * 24 + j) * 3 + 0], r1 = bones4[(lr * 24 + j) * 3 + 1],
                  r2 = bones4[(lr * 24 + j) * 3 + 2];
      const float y0 = r0.x * dray[0] + r0.y * dray[1] + r0.z * dray[2];
      const float y1 = r1.x * dray[0] + r1.y * dray[1] + r1.z * dray[2];
      const float y2 = r2.x * dray[0] + r2.y * dray[1] + r2.z * dray[2];
      const float inv = 1.0f / fmaxf(sqrtf(y0 * y0 + y1 * y1 + y2 * y2), 1e-12f);
      e[3 * a + 0] = y0 * inv;
      e[3 * a + 1] = y1 * inv;
      e[3 * a + 2] = y2 * inv;
      wd[a] = cutoff_gate(A.tau_d, v[a], A.cut_d[j]);
    }
#pragma unroll
    for (int g = 0; g < 9; ++g)
      KGV(g, e[4 * g] * wd[(4 * g) / 3], e[4 * g + 1] * wd[(4 * g + 1) / 3], e[4 * g + 2] * wd[(4 * g + 2) / 3],
          e[4 * g + 3] * wd[(4 * g + 3) / 3]);
#pragma unroll
    for (int f = 0; f < LD; ++f) {
      float se[36], ce[36];
#pragma unroll
      for (int i = 0; i < 36; ++i) {
        float s, c;
        sincos_f32(e[i] * (float)(1 << f), s, c);
        se[i] = s * wd[i / 3];
        ce[i] = c * wd[i / 3];
      }
#pragma unroll
      for (int g = 0; g < 9; ++g) KGV(9 * (1 + 2 * f) + g, se[4 * g], se[4 * g + 1], se[4 * g + 2], se[4 * g + 3]);
#pragma unroll
      for (int g = 0; g < 9; ++g) KGV(9 * (2 + 2 * f) + g, ce[4 * g], ce[4 * g + 1], ce[4 * g + 2], ce[4 * g + 3]);
    }
  }
  if constexpr (CODE > 0) {
    float fidx;
    if constexpr (PRE) fidx = xrow[DIMX + DIMD];
    else fidx = A.cam[ray];
    int ci = (int)fidx;
    ci = ci < 0 ? 0 : (ci >= A.n_codes ? A.n_codes - 1 : ci);
    const float* crow = A.codes + (long long)ci * CODE;
#pragma unroll
    for (int g = 0; g < CODE / 8; ++g) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(crow + 8 * g + 4 * h);
      KGV(9 * (1 + 2 * LD) + g, c4.x, c4.y, c4.z, c4.w);
    }
  }
  float gact[128];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) gact[nb * 16 + r] = fmaxf(accv[nb][r], 0.f);
  if (save) store_row<4>(A.save_g + p * 128, gact, h);
  // ---- rgb head (VALU)
  const float c0 = head_dot<4>(gact, A.aux + AUX_WC + 0, h) + A.aux[AUX_BC + 0];
  const float c1 = head_dot<4>(gact, A.aux + AUX_WC + 128, h) + A.aux[AUX_BC + 1];
  const float c2 = head_dot<4>(gact, A.aux + AUX_WC + 256, h) + A.aux[AUX_BC + 2];
  if (valid && h == 0) {
    f32x4 o = {c0, c1, c2, sigma_raw};
    *reinterpret_cast<f32x4*>(A.raw + p * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------------
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
template <int NB>
__device__ __forceinline__ void relu_mask(float (&d)[128], const float* __restrict__ row, int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 s = *reinterpret_cast<const f32x4*>(row + 32 * nb + 8 * q + 4 * h);
      d[nb * 16 + 4 * q + 0] = s.x > 0.f ? d[nb * 16 + 4 * q + 0] : 0.f;
      d[nb * 16 + 4 * q + 1] = s.y > 0.f ? d[nb * 16 + 4 * q + 1] : 0.f;
      d[nb * 16 + 4 * q + 2] = s.z > 0.f ? d[nb * 16 + 4 * q + 2] : 0.f;
      d[nb * 16 + 4 * q + 3] = s.w > 0.f ? d[nb * 16 + 4 * q + 3] : 0.f;
    }
}

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

template <int NB>
__device__ __forceinline__ void load_row(float (&a)[128], const float* __restrict__ row, int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * nb + 8 * q + 4 * h);
      a[nb * 16 + 4 * q + 0] = v.x;
      a[nb * 16 + 4 * q + 1] = v.y;
      a[nb * 16 + 4 * q + 2] = v.z;
      a[nb * 16 + 4 * q + 3] = v.w;
    }
}

// store the 8 accumulator blocks as columns [c0, c0+256) of a row of width `w` (columns >= w dropped)
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

template <int LV, int LD, int CODE, bool PRE, bool TRAIN>
static int launch(const MlpArgs& a, hipStream_t st) {
  const long long nblk = (a.P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = 2 * STAGE_BYTES + (PRE ? 0 : MAX_TILE_RAYS * 72 * 16);
  auto kern = k_mlp_fwd<LV, LD, CODE, PRE, TRAIN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
  return check_launch("k_mlp_fwd");
}

int mlp_dispatch(const AnerfConfig* cfg, const MlpArgs& a, bool pre, bool train, hipStream_t st) {
  const int lv = cfg->multires, ld = cfg->multires_views, cd = cfg->framecode_ch;
  if (lv != 7) return set_error(ANERF_E_CONFIG, "multires must be 7");
  if (pre && train) return set_error(ANERF_E_CONFIG, "training forward needs the fused (not pre-encoded) path");
#define ANERF_CASE(LD_, CD_)                                               \
  if (ld == LD_ && cd == CD_) {                                            \
    if (pre) return launch<7, LD_, CD_, true, false>(a, st);               \
    return train ? launch<7, LD_, CD_, false, true>(a, st) : launch<7, LD_, CD_, false, false>(a, st); \
  }
  ANERF_CASE(4, 0)
  ANERF_CASE(4, 16)
  ANERF_CASE(0, 0)
#undef ANERF_CASE
  return set_error(ANERF_E_CONFIG, "unsupported (multires_views, framecode_ch); built: (4,0) (4,16) (0,0)");
}

int mlp_raw_entry(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays, int ray_stride,
                  const float* z, const float* skts, long long skt_stride, const float* cam, const float* codes,
                  int n_codes, float tau_v, float tau_d, const float* cut_v, const float* cut_d, const float* x,
                  int x_width, long long P, int N, int S, int nstages, float* raw, bool pre, const AnerfSaved* sv,
                  hipStream_t st) {
  MlpArgs a;
  a.packed = packed; a.aux = aux; a.rays = rays; a.z = z; a.skts = skts; a.cam = cam; a.codes = codes;
  a.cut_v = cut_v; a.cut_d = cut_d; a.x = x; a.raw = raw; a.P = P; a.skt_stride = skt_stride;
  a.S = S; a.N = N; a.ray_stride = ray_stride; a.n_codes = n_codes; a.x_width = x_width; a.nstages = nstages;
  a.tau_v = tau_v; a.tau_d = tau_d;
  a.save_h = a.save_f = a.save_g = a.save_x = a.save_u = nullptr;
  a.Ppad = P;
  if (sv) {
    a.save_h = sv->h; a.save_f = sv->f; a.save_g = sv->g; a.save_x = sv->x; a.save_u = sv->u; a.Ppad = sv->p_pad;
  }
  return mlp_dispatch(cfg, a, pre, sv != nullptr, st);
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
