// anerf_gemm.hip -- weight gradients of the MLP as a GROUPED fp32-MFMA "TN" GEMM over the sample axis:
//     dW[m][n] = sum_p A[p][m] * B[p][n]      A = d(pre-activation) rows, B = layer-input rows (both row-major,
//                                             saved by k_mlp_fwd<TRAIN> / k_mlp_bwd), p = sample index.
// One launch covers every layer of a network (13 problems): blockIdx -> (problem, 128x128 output tile, p-chunk).
// Each workgroup reduces its p-chunk into registers (4 waves x 64x64, v_mfma_f32_32x32x2_f32, operands staged
// through a double-buffered LDS tile pair filled with global_load_lds_dwordx4) and writes a partial tile;
// k_reduce_dw sums the chunks in a fixed order (deterministic) and scatters into the torch-layout gradient
// tensors (undoing the stream column order of X'/U').  Bias gradients (column sums of A) ride along.
// Autograd of the 12 nn.Linear layers of NeRF (core/networks/nerf.py:57-88) w.r.t. weights and biases.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf_dev.h"
#include "anerf_gemm.h"

namespace anerf {

constexpr int GT = 128;          // output tile edge
constexpr int GP = 32;           // p rows per LDS tile
constexpr int TILE_BYTES = GP * GT * 4;   // 16 KiB

__device__ __forceinline__ void stage_tile(const float* __restrict__ src, int ld, int col0, int ncols, long long row0,
                                           char* lds_tile, int wave, int lane) {
  // 32 rows x 128 cols; one wave-instruction = 2 rows (64 lanes x 16 B).  Columns past the matrix edge are
  // clamped to the last valid float4 (their products land in output entries that are never written).
  int c = col0 + (lane & 31) * 4;
  const int cmax = ncols - 4;
  c = c > cmax ? cmax : c;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = wave * 8 + i * 2 + (lane >> 5);
    const float* g = src + (row0 + r) * ld + c;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)(lds_tile + (wave * 8 + i * 2) * (GT * 4)), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256) void k_gemm_tn(const GemmBatch G, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_id = blockIdx.x % G.total_tiles;
  const int chunk = blockIdx.x / G.total_tiles;
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < G.nprob; ++i)
    if (tile_id >= G.p[i].tile_base) pi = i;
  const GemmProb& pr = G.p[pi];
  const int lt = tile_id - pr.tile_base;
  const int tm = lt / pr.tiles_n, tn = lt - tm * pr.tiles_n;
  const long long r0 = (long long)chunk * G.rows_per_chunk;
  long long r1 = r0 + G.rows_per_chunk;
  if (r1 > G.p_pad) r1 = G.p_pad;
  const int ntile = (int)((r1 - r0) / GP);
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float asum0 = 0.f, asum1 = 0.f;

  // LDS: [buffer 0: A tile | B tile][buffer 1: A tile | B tile]
  if (ntile > 0) {
    stage_tile(pr.A, pr.lda, tm * GT, pr.lda_cols, r0, smem, wave, lane);
    stage_tile(pr.B, pr.ldb, tn * GT, pr.ldb_cols, r0, smem + TILE_BYTES, wave, lane);
  }
  for (int t = 0; t < ntile; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntile) {
      char* nxt = smem + ((t + 1) & 1) * (2 * TILE_BYTES);
      stage_tile(pr.A, pr.lda, tm * GT, pr.lda_cols, r0 + (long long)(t + 1) * GP, nxt, wave, lane);
      stage_tile(pr.B, pr.ldb, tn * GT, pr.ldb_cols, r0 + (long long)(t + 1) * GP, nxt + TILE_BYTES, wave, lane);
    }
    const char* cur = smem + (t & 1) * (2 * TILE_BYTES);
    const float* a_s = reinterpret_cast<const float*>(cur) + wm * 64 + i;
    const float* b_s = reinterpret_cast<const float*>(cur + TILE_BYTES) + wn * 64 + i;
#pragma unroll
    for (int s = 0; s < GP / 2; ++s) {
      const float a0 = a_s[(2 * s + kk) * GT], a1 = a_s[(2 * s + kk) * GT + 32];
      const float b0 = b_s[(2 * s + kk) * GT], b1 = b_s[(2 * s + kk) * GT + 32];
      asum0 += a0;
      asum1 += a1;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // ---- partial tile -> workspace [chunk][M][N]
  float* part = ws + pr.part_off + (long long)chunk * pr.M * pr.N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = tn * GT + wn * 64 + b * 32 + i;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mrow = tm * GT + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (mrow < pr.M && n < pr.N) part[(long long)mrow * pr.N + n] = acc[a][b][r];
      }
    }
  if (pr.bias_off >= 0 && tn == 0 && wn == 0) {
    asum0 += __shfl_xor(asum0, 32);
    asum1 += __shfl_xor(asum1, 32);
    float* bp = ws + pr.bias_off + (long long)chunk * pr.M;
    const int m0 = tm * GT + wm * 64 + i;
    if (kk == 0) {
      if (m0 < pr.M) bp[m0] = asum0;
      if (m0 + 32 < pr.M) bp[m0 + 32] = asum1;
    }
  }
}

// Sum the chunk partials in index order and scatter into the gradient tensors.
__global__ void k_reduce_dw(const GemmBatch G, const float* __restrict__ ws) {
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (gid >= G.total_out) return;
  int pi = 0;
#pragma unroll 1
  for (int i = 1; i < G.nprob; ++i)
    if (gid >= G.p[i].out_base) pi = i;
  const GemmProb& pr = G.p[pi];
  const long long e = gid - pr.out_base;
  const long long mn = (long long)pr.M * pr.N;
  if (e < mn) {
    const int mrow = (int)(e / pr.N), n = (int)(e - (long long)mrow * pr.N);
    const float* src = ws + pr.part_off + e;
    float s = 0.f;
    for (int c = 0; c < G.chunks; ++c) s += src[(long long)c * mn];
    if (mrow >= pr.m_first && mrow < pr.m_first + pr.m_count) {
      const int col = pr.colmap ? pr.colmap[n] : n;
      pr.dst[(long long)(mrow - pr.m_first) * pr.dst_ld + pr.dst_col0 + col] = s;
    }
  } else {
    const int mrow = (int)(e - mn);
    const float* src = ws + pr.bias_off + mrow;
    float s = 0.f;
    for (int c = 0; c < G.chunks; ++c) s += src[(long long)c * pr.M];
    if (mrow >= pr.bm_first && mrow < pr.bm_first + pr.bm_count) pr.bias_dst[mrow - pr.bm_first] = s;
  }
}

int launch_weight_grads(GemmBatch& G, float* ws, hipStream_t st) {
  const size_t lds = 4 * TILE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_tn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_gemm_tn, dim3((unsigned)(G.total_tiles * G.chunks)), dim3(256), lds, st, G, ws);
  int rc = check_launch("k_gemm_tn");
  if (rc) return rc;
  hipLaunchKernelGGL(k_reduce_dw, dim3((unsigned)((G.total_out + 255) / 256)), dim3(256), 0, st, G, (const float*)ws);
  return check_launch("k_reduce_dw");
}

}  // namespace anerf
