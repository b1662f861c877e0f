// mfma_probe4.hip -- 2 waves/SIMD (8-wave workgroup, <=256 regs) with v_mfma_f32_16x16x4_f32: does VALU hide now?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per k-group (16 input features): 16 ds_read_b128 (one per 16-row output block) + 64 MFMA 16x16x4 + NV VALU
template <int NV>
__global__ __launch_bounds__(512) void k_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += 512) ((float*)smem)[i] = 1e-4f * ((i & 7) - 3);
  __syncthreads();
  f32x4 acc[16];
#pragma unroll
  for (int nb = 0; nb < 16; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1e-3f * (lane + i);
  float b0 = lane * 1e-3f, b1 = b0 + 1.f, b2 = b0 + 2.f, b3 = b0 + 3.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) {
      f32x4 a[16];
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) a[nb] = *(const f32x4*)(smem + ((kg & 1) * 16 + nb) * 1024 + lane * 16);
#pragma unroll
      for (int j = 0; j < NV; ++j) x[j & 7] = fmaf(x[j & 7], 1.0001f, x[(j + 3) & 7]);
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb].x, b0, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb].y, b1, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb].z, b2, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb].w, b3, acc[nb], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < 16; ++nb) s += acc[nb].x + acc[nb].y + acc[nb].z + acc[nb].w;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV>
static void run(float* out, int iters, int blocks, const char* name) {
  hipFuncSetAttribute((const void*)k_probe<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_probe<NV><<<blocks, 512, 100 * 1024>>>(out, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_probe<NV><<<blocks, 512, 100 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flop = (double)blocks * 8 * iters * 1024.0 * 2048.0;   // waves * iters * (16 kg * 64 MFMA) * 2*16*16*4
  printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 512 * 4);
  const int iters = 200, blocks = 256 * 4;
  run<0>(out, iters, blocks, "Q0 2 waves/SIMD 16x16x4,   0 VALU / k-group");
  run<64>(out, iters, blocks, "Q1 2 waves/SIMD 16x16x4,  64 VALU / k-group");
  run<128>(out, iters, blocks, "Q2 2 waves/SIMD 16x16x4, 128 VALU / k-group");
  run<256>(out, iters, blocks, "Q3 2 waves/SIMD 16x16x4, 256 VALU / k-group");
  return 0;
}
