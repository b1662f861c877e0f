// mfma_probe.hip -- microbenchmark of the k-group instruction pattern used by k_mlp_fwd (timing experiments only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)smem)[i] = 1e-3f * (i & 7);
  __syncthreads();
  f32x16 acc[8];
  f32x16 prev[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[nb][r] = 0.f; prev[nb][r] = 1e-3f * (lane + r + nb); }
  float b0 = lane * 1e-3f, b1 = b0 + 1.f, b2 = b0 + 2.f, b3 = b0 + 3.f;
  float n0 = b0, n1 = b1, n2 = b2, n3 = b3;
  for (int it = 0; it < iters; ++it) {
    if (V == 5) { n0 = fmaxf(prev[0][0], 0.f); n1 = fmaxf(prev[0][1], 0.f); n2 = fmaxf(prev[0][2], 0.f); n3 = fmaxf(prev[0][3], 0.f); }
#pragma unroll
    for (int kg = 0; kg < 32; ++kg) {
      f32x4 a[8];
      if (V == 0) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) a[nb] = f32x4{b0, b1, b2, b3};
      } else {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) a[nb] = *(const f32x4*)(smem + ((kg & 3) * 8 + nb) * 1024 + lane * 16);
      }
      float v0 = b0, v1 = b1, v2 = b2, v3 = b3;
      if (V == 6) {
        v0 = prev[kg >> 2][4 * (kg & 3) + 0]; v1 = prev[kg >> 2][4 * (kg & 3) + 1];
        v2 = prev[kg >> 2][4 * (kg & 3) + 2]; v3 = prev[kg >> 2][4 * (kg & 3) + 3];
      }
      if (V == 5) { v0 = n0; v1 = n1; v2 = n2; v3 = n3; }
      if (V == 2 || V == 4) {
        v0 = fmaxf(prev[kg >> 2][4 * (kg & 3) + 0], 0.f); v1 = fmaxf(prev[kg >> 2][4 * (kg & 3) + 1], 0.f);
        v2 = fmaxf(prev[kg >> 2][4 * (kg & 3) + 2], 0.f); v3 = fmaxf(prev[kg >> 2][4 * (kg & 3) + 3], 0.f);
      }
      if (V == 3 || V == 4) {   // nb-major: 4 back-to-back MFMAs per accumulator
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].x, v0, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].y, v1, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].z, v2, acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].w, v3, acc[nb], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].x, v0, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].y, v1, acc[nb], 0, 0, 0);
        if (V == 5 && kg < 31) {   // operands of the NEXT k-group, formed in the middle of this one, in fresh registers
          const int k2 = kg + 1;
          n0 = fmaxf(prev[k2 >> 2][4 * (k2 & 3) + 0], 0.f); n1 = fmaxf(prev[k2 >> 2][4 * (k2 & 3) + 1], 0.f);
          n2 = fmaxf(prev[k2 >> 2][4 * (k2 & 3) + 2], 0.f); n3 = fmaxf(prev[k2 >> 2][4 * (k2 & 3) + 3], 0.f);
        }
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].z, v2, acc[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].w, v3, acc[nb], 0, 0, 0);
      }
    }
    if (V == 2 || V == 4 || V == 5 || V == 6) {   // ping-pong: outputs become next inputs
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) { f32x16 t = prev[nb]; prev[nb] = acc[nb]; acc[nb] = t; }
    }
    if (V == 6) {   // in-place ReLU epilogue on the new inputs
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[nb][r] = fmaxf(prev[nb][r], 0.f);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[nb][r] + prev[nb][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
static float run(float* out, int iters, int blocks) {
  hipFuncSetAttribute((const void*)k_probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_probe<V><<<blocks, 256, 100 * 1024>>>(out, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_probe<V><<<blocks, 256, 100 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 200, blocks = 256 * 4;
  const double flop = (double)blocks * 4 * iters * 1024.0 * 4096.0;   // waves * iters * MFMAs * FLOP per MFMA
  const char* names[] = {"V0 regs only, t-major", "V1 +8 ds_read_b128/kgroup", "V2 V1 + lazy relu + ping-pong",
                         "V3 V1 nb-major", "V4 V2 nb-major", "V5 lazy relu 1 kgroup ahead", "V6 B from acc + in-place relu epilogue"};
  float ms[7] = {run<0>(out, iters, blocks), run<1>(out, iters, blocks), run<2>(out, iters, blocks), run<3>(out, iters, blocks),
                 run<4>(out, iters, blocks), run<5>(out, iters, blocks), run<6>(out, iters, blocks)};
  for (int i = 0; i < 7; ++i) printf("%-34s %8.3f ms  %7.2f TFLOP/s\n", names[i], ms[i], flop / ms[i] / 1e9);
  return 0;
}
