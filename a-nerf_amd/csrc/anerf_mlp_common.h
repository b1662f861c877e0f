// anerf_mlp_common.h -- primitives shared by the backward kernels: 2-slot weight-stream pipe, k-group MFMA step,
// register <-> row-major helpers.  (The forward kernel in anerf_mlp.hip carries its own 3-slot / lazy-activation set.)
#pragma once
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
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0 * (NB / 4) + (nb >> 2)][nb & 3], b0, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1 * (NB / 4) + (nb >> 2)][nb & 3], b1, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * (NB / 4) + (nb >> 2)][nb & 3], b2, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3 * (NB / 4) + (nb >> 2)][nb & 3], b3, acc[nb], 0, 0, 0);
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
__device__ __forceinline__ void store_row(float* __restrict__ row, const float (&a)[128], int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {a[nb * 16 + 4 * q], a[nb * 16 + 4 * q + 1], a[nb * 16 + 4 * q + 2], a[nb * 16 + 4 * q + 3]};
      *reinterpret_cast<f32x4*>(row + 32 * nb + 8 * q + 4 * h) = o;
    }
}

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
}  // namespace anerf
