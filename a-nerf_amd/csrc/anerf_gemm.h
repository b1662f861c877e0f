// anerf_gemm.h -- descriptors of the grouped weight-gradient GEMM (passed by value as kernel arguments, < 4 KiB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace anerf {

// ---- reduction / scatter side: one entry per GEMM problem  dW[m][n] = sum_p A[p][m] * B[p][n]
struct GemmProb {
  float* dst;            // gradient tensor rows m_first.. -> dst[(m - m_first) * dst_ld + dst_col0 + colmap[n]]
  float* bias_dst;       // bias gradient (column sums of A) or nullptr
  const int* colmap;     // output-column permutation (stream order -> torch order) or nullptr
  long long part_off;    // float offset of the [chunks][M][N] partials in the workspace
  long long bias_off;    // float offset of the [chunks][M] bias partials, -1 if none
  long long out_base;    // prefix of (M*N + (bias ? M : 0)) over the problems, for k_reduce_dw
  int M, N, chunks;
  int dst_ld, dst_col0, m_first, m_count, bm_first, bm_count;
};

struct GemmBatch {
  GemmProb p[16];
  int nprob;
  int accumulate;        // 0: gradients written; 1: added to what dst holds (p.grad accumulated in place)
  long long total_out;
};

// ---- compute side: a block = 4 waves, each owning one 128x128 (or, "skinny", 4x128) output tile of some problem,
// sharing up to 5 LDS operand tiles (16 sample rows x 128 columns each) per stage.
struct GemmMat {         // a row-major [p_pad][ld] operand matrix
  const float* ptr;
  int ld, ncols;
};
struct GemmTile {        // one LDS operand tile = columns [col0, col0+128) of a matrix (clamped at ncols)
  int mat, col0;
};
struct GemmWave {
  int a_tile, b_tile;    // LDS tile indices of the operands; a_tile < 0: idle wave
  int part_off;          // float offset of the problem's partial [chunk][M][N] in the workspace
  int bias_off;          // float offset of the bias partial [chunk][M], -1: none
  int M, N, m0, n0;      // problem dims (partial row stride N) and this tile's origin
};
struct GemmBlock {
  int ntiles, skinny;
  GemmTile t[5];
  GemmWave w[4];
};
struct GemmPlan {
  GemmMat mat[24];
  GemmBlock blk[16];     // heavy blocks first, then the skinny ones
  int nheavy, nskinny;
  int rows_h, chunks_h;  // sample rows per heavy block (multiple of 16) and number of row chunks
  int rows_s, chunks_s;  // same for skinny blocks (currently identical to the heavy chunking)
  long long p_pad;
};

constexpr int GEMM_ROWS = 16;   // sample rows per LDS stage

// rows-per-block search shared by anerf_train_layout (workspace size) and anerf_weight_grads
void gemm_plan_rows(long long p_pad, int nheavy, int nskinny, int* rows_h, int* chunks_h, int* rows_s, int* chunks_s);

int launch_weight_grads(const GemmPlan& P, const GemmBatch& G, float* ws, bool b3, hipStream_t st);

}  // namespace anerf
