// mfma_probe2.hip -- how should layer i's accumulators become layer i+1's MFMA B operands?  (timing experiment)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one 256->256 layer: 32 k-groups x (8 ds_read_b128 + 32 MFMA); B operands from `in` according to MODE
// MODE 0: raw accumulator values (no ReLU at all: upper bound)     MODE 1: lazy fmaxf right before use
// MODE 2: raw values, caller applies an in-place ReLU pass after the layer
template <int MODE>
__device__ __forceinline__ void layer(f32x16 (&out)[8], const f32x16 (&in)[8], const char* smem, int lane) {
#pragma unroll
  for (int kg = 0; kg < 32; ++kg) {
    f32x4 a[8];
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) a[nb] = *(const f32x4*)(smem + ((kg & 3) * 8 + nb) * 1024 + lane * 16);
    float v0 = in[kg >> 2][4 * (kg & 3) + 0], v1 = in[kg >> 2][4 * (kg & 3) + 1];
    float v2 = in[kg >> 2][4 * (kg & 3) + 2], v3 = in[kg >> 2][4 * (kg & 3) + 3];
    if (MODE == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) out[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].x, v0, out[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) out[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].y, v1, out[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) out[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].z, v2, out[nb], 0, 0, 0);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) out[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].w, v3, out[nb], 0, 0, 0);
  }
}
__device__ __forceinline__ void relu_inplace(f32x16 (&x)[8]) {
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[nb][r] = fmaxf(x[nb][r], 0.f);
}
__device__ __forceinline__ void init(f32x16 (&x)[8], float v) {
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) x[nb][r] = v;
}

// V: 0 = no relu, 1 = lazy, 2 = in-place epilogue, 3 = v1 style (relu-copy into a float[128] then B from it)
template <int V>
__global__ __launch_bounds__(256) void k_probe(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += 256) ((float*)smem)[i] = 1e-4f * ((i & 7) - 3);
  __syncthreads();
  f32x16 A[8], B[8];
  init(A, 1e-3f * lane);
  if (V == 3) {
    float hin[128];
#pragma unroll
    for (int i = 0; i < 128; ++i) hin[i] = 1e-3f * (lane + i);
    for (int it = 0; it < iters; ++it) {
      init(A, 0.25f);
#pragma unroll
      for (int kg = 0; kg < 32; ++kg) {
        f32x4 a[8];
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) a[nb] = *(const f32x4*)(smem + ((kg & 3) * 8 + nb) * 1024 + lane * 16);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) A[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].x, hin[4 * kg], A[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) A[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].y, hin[4 * kg + 1], A[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) A[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].z, hin[4 * kg + 2], A[nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) A[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb].w, hin[4 * kg + 3], A[nb], 0, 0, 0);
      }
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) hin[nb * 16 + r] = fmaxf(A[nb][r], 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 128; ++i) s += hin[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    return;
  }
  for (int it = 0; it < iters; it += 2) {
    init(B, 0.25f);
    layer<V == 1 ? 1 : 0>(B, A, smem, lane);
    if (V == 2) relu_inplace(B);
    init(A, 0.25f);
    layer<V == 1 ? 1 : 0>(A, B, smem, lane);
    if (V == 2) relu_inplace(A);
  }
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += A[nb][r];
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
  const double flop = (double)blocks * 4 * iters * 1024.0 * 4096.0;
  const char* names[] = {"W0 ping-pong, no relu (bound)", "W1 ping-pong, lazy relu", "W2 ping-pong, in-place relu pass",
                         "W3 v1: relu-copy acc->hin[128]"};
  float ms[4] = {run<0>(out, iters, blocks), run<1>(out, iters, blocks), run<2>(out, iters, blocks), run<3>(out, iters, blocks)};
  for (int i = 0; i < 4; ++i) printf("%-36s %8.3f ms  %7.2f TFLOP/s\n", names[i], ms[i], flop / ms[i] / 1e9);
  return 0;
}
