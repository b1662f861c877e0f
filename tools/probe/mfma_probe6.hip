// mfma_probe6.hip -- the inner loop of k_gemm_tn in isolation: per row pair 2 ds_read_b128 (a4, b4) and NA x NB
// independent v_mfma_f32_32x32x2_f32 on NA*NB accumulator blocks (timing experiment only).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NA, int NB, int V, int RANDOM>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    ((float*)smem)[i] = RANDOM ? ((int)(h & 0xFFFF) - 32768) * (1.0f / 65536.0f) : 1e-3f * (i & 7);
  }
  __syncthreads();
  f32x16 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const char* abase = smem + (wave >> 1) * 8192 + (lane >> 5) * 512 + (lane & 31) * 16;
  const char* bbase = smem + 16384 + (wave & 1) * 8192 + (lane >> 5) * 512 + (lane & 31) * 16;
  f32x4 asum = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 a4 = *(const f32x4*)abase, b4 = *(const f32x4*)bbase;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      f32x4 an = a4, bn = b4;
      if (s + 1 < 8) { an = *(const f32x4*)(abase + (s + 1) * 1024); bn = *(const f32x4*)(bbase + (s + 1) * 1024); }
      if (V == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[a], b4[b], acc[a][b], 0, 0, 0);
      if (V == 1) { __builtin_amdgcn_sched_barrier(0); asum += a4; }
      a4 = an; b4 = bn;
    }
  }
  float s = asum[0] + asum[1] + asum[2] + asum[3];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int NB, int V, int RANDOM>
float run(float* out, int iters, int blocks) {
  hipFuncSetAttribute((const void*)k_probe<NA, NB, V, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<NA, NB, V, RANDOM>), dim3(blocks), dim3(256), 65536, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<NA, NB, V, RANDOM>), dim3(blocks), dim3(256), 65536, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  const int blocks = 256 * 4;
  auto rep = [&](const char* name, float ms, int nacc, int iters) {
    const double flop = (double)blocks * 4 * iters * 8.0 * nacc * 4096.0;
    printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", name, ms, flop / ms / 1e9);
  };
  rep("G0 4x4 acc (256 AGPR), plain", run<4, 4, 0, 0>(out, 200, blocks), 16, 200);
  rep("G1 4x4 acc, sched_barrier + column sums", run<4, 4, 1, 0>(out, 200, blocks), 16, 200);
  rep("G2 2x4 acc (128 AGPR)", run<2, 4, 0, 0>(out, 400, blocks), 8, 400);
  rep("G3 2x2 acc (64 AGPR)", run<2, 2, 0, 0>(out, 800, blocks), 4, 800);
  rep("G4 4x2 acc (128 AGPR)", run<4, 2, 0, 0>(out, 400, blocks), 8, 400);
  {
    const int b1 = 255, it1 = 91;
    float ms = run<4, 4, 1, 1>(out, it1, b1);
    printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", "S0 255 blocks x 91 iters (one round)", ms, (double)b1 * 4 * it1 * 8.0 * 16 * 4096.0 / ms / 1e9);
    ms = run<4, 4, 1, 1>(out, 232, 756);
    printf("%-44s %8.3f ms  %7.2f TFLOP/s\n", "S1 756 blocks x 232 iters (three rounds)", ms, 756.0 * 4 * 232 * 8.0 * 16 * 4096.0 / ms / 1e9);
  }
  rep("R0 4x4 acc, random operands", run<4, 4, 0, 1>(out, 200, blocks), 16, 200);
  rep("R1 4x4 acc, sched_barrier + sums, random", run<4, 4, 1, 1>(out, 200, blocks), 16, 200);
  rep("R2 2x4 acc, random operands", run<2, 4, 0, 1>(out, 400, blocks), 8, 400);
  rep("R3 4x4 acc, random, 2000 iters (long run)", run<4, 4, 0, 1>(out, 2000, blocks), 16, 2000);
  rep("G5 4x4 acc, plain, 2000 iters (long run)", run<4, 4, 0, 0>(out, 2000, blocks), 16, 2000);
  return 0;
}
