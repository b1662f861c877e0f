// mfma_probe3.hip -- what does a VALU instruction cost inside a 1-wave/SIMD fp32-MFMA stream?  (timing experiment)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// per k-group: 8 ds_read_b128 + 32 MFMA + NV VALU.  KIND 0: independent fma chain on 8 VGPRs (never feeds the MFMAs)
// KIND 1: the VALU results become the B operands of the NEXT k-group (VGPR sources)
template <int NV, int KIND>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)smem)[i] = 1e-4f * ((i & 7) - 3);
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1e-3f * (lane + i);
  float b0 = lane * 1e-3f, b1 = b0 + 1.f, b2 = b0 + 2.f, b3 = b0 + 3.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kg = 0; kg < 32; ++kg) {
      f32x4 a[8];
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) a[nb] = *(const f32x4*)(smem + ((kg & 3) * 8 + nb) * 1024 + lane * 16);
      const float v0 = b0, v1 = b1, v2 = b2, v3 = b3;
#pragma unroll
      for (int j = 0; j < NV; ++j) x[j & 7] = fmaf(x[j & 7], 1.0001f, x[(j + 3) & 7]);
      if (KIND == 1) { b0 = x[0]; b1 = x[1]; b2 = x[2]; b3 = x[3]; }
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].x, v0, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].y, v1, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].z, v2, acc[nb], 0, 0, 0);
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].w, v3, acc[nb], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[nb][r];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND>
static void run(float* out, int iters, int blocks, const char* name) {
  hipFuncSetAttribute((const void*)k_probe<NV, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k_probe<NV, KIND><<<blocks, 256, 100 * 1024>>>(out, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_probe<NV, KIND><<<blocks, 256, 100 * 1024>>>(out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flop = (double)blocks * 4 * iters * 1024.0 * 4096.0;
  printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int iters = 200, blocks = 256 * 4;
  run<0, 0>(out, iters, blocks, "P0  0 VALU / k-group");
  run<16, 0>(out, iters, blocks, "P1  16 independent VALU / k-group");
  run<64, 0>(out, iters, blocks, "P2  64 independent VALU / k-group");
  run<128, 0>(out, iters, blocks, "P3 128 independent VALU / k-group");
  run<256, 0>(out, iters, blocks, "P4 256 independent VALU / k-group");
  run<16, 1>(out, iters, blocks, "P5  16 VALU feeding next k-group's B");
  run<64, 1>(out, iters, blocks, "P6  64 VALU feeding next k-group's B");
  run<128, 1>(out, iters, blocks, "P7 128 VALU feeding next k-group's B");
  return 0;
}
