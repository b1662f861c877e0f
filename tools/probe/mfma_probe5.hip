// mfma_probe5.hip -- feasibility of a bf16x3 (hi/lo split) path: v_mfma_f32_32x32x16_bf16 stream, 1 wave/SIMD,
// A fragments (hi, lo) from LDS, B operands split from fp32 registers with VALU.   (timing experiment)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_hi(float a, float b) {       // truncating split (exact residual)
  return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
}
__device__ __forceinline__ float hi_of(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }

// MODE 0: MFMA only (operands fixed)   1: + A hi/lo from LDS   2: + B hi/lo split of 8 fp32 values per k-step (VALU)
// 3: as 2 plus NV extra independent VALU per k-step
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 7);
  __syncthreads();
  f32x16 acc[8], src[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[nb][r] = 0.f; src[nb][r] = 1e-3f * (lane + r + nb); }
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1e-3f * (lane + i);
  bf16x8 bh = {1, 2, 3, 4, 5, 6, 7, 8}, bl = {8, 7, 6, 5, 4, 3, 2, 1};
  bf16x8 ah0 = bh, al0 = bl;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {          // 16 k-steps of 16 = one 256-wide layer
      if (MODE >= 2) {
        const int nbk = ks >> 1, r0 = (ks & 1) * 8;
        u32x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = src[nbk][r0 + 2 * j], b = src[nbk][r0 + 2 * j + 1];
          h[j] = pack_hi(a, b);
          l[j] = pack_hi(a - hi_of(a), b - hi_of(b));
        }
        bh = __builtin_bit_cast(bf16x8, h);
        bl = __builtin_bit_cast(bf16x8, l);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) x[j & 7] = fmaf(x[j & 7], 1.0001f, x[(j + 3) & 7]);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) {
        bf16x8 ah = ah0, al = al0;
        if (MODE >= 1) {
          ah = *(const bf16x8*)(smem + ((ks & 1) * 16 + 2 * nb) * 1024 + lane * 16);
          al = *(const bf16x8*)(smem + ((ks & 1) * 16 + 2 * nb + 1) * 1024 + lane * 16);
        }
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[nb], 0, 0, 0);
      }
    }
    if (MODE >= 2) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) { f32x16 t = src[nb]; src[nb] = acc[nb]; acc[nb] = t; }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[nb][r] + src[nb][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV>
static void run(float* out, int iters, int blocks, const char* name) {
  hipFuncSetAttribute((const void*)k_probe<MODE, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_probe<MODE, NV><<<blocks, 256, 100 * 1024>>>(out, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_probe<MODE, NV><<<blocks, 256, 100 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // "fp32-equivalent" FLOPs: one 256x256 layer for 32 samples per wave per iteration = 2*256*256*32
  const double flop = (double)blocks * 4 * iters * 2.0 * 256 * 256 * 32;
  printf("%-52s %8.3f ms  %8.2f eff-TFLOP/s (x3 = %7.1f bf16 TF)\n", name, ms, flop / ms / 1e9, 3 * flop / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 400, blocks = 256 * 4;
  run<0, 0>(out, iters, blocks, "B0 bf16x3 MFMA only");
  run<1, 0>(out, iters, blocks, "B1 + A hi/lo from LDS (16 ds_read_b128 / k-step)");
  run<2, 0>(out, iters, blocks, "B2 + B hi/lo split from fp32 regs (VALU)");
  run<3, 16>(out, iters, blocks, "B3 + 16 extra VALU / k-step");
  run<3, 48>(out, iters, blocks, "B4 + 48 extra VALU / k-step");
  return 0;
}
