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
  const float* colscale; // per torch column (relative to dst_col0) factor of the gradient, or nullptr (frequency schedule, ABI 5)
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

// ---- compute side: a block = 4 waves, each owning one 128x128 output tile of some problem, sharing up to 5 LDS operand
// tiles (16 sample rows x 128 columns each) per stage.  The two head problems (alpha: 1 x 256, rgb: 3 x 128; A = the
// [p][4] head-gradient rows) have no block of their own: they ride as VALU FMAs in two of the heavy blocks (GemmHead).
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
  int ntiles, pad_;
  GemmTile t[5];
  GemmWave w[4];
};
// A head problem dW[m][n] = sum_p draw[p][comp0 + m] * Hm[p][n] hosted by block `blk`: every stage the block also stages the
// 16 draw rows (256 B behind the five operand tiles) and its four waves share the 16 rows -- alpha: wave (r, c) of the
// feature layer's 2 x 2 block takes the rows of row pairs s = r (mod 2) against ITS OWN B tile (h7 columns 128c..), i.e. 2
// row slots; rgb: wave k takes s = k (mod 4) against LDS tile `tile` (the g rows, loaded as the block's fifth tile), 4 row
// slots.  Partials: [chunk * slots + slot][M][N] at part_off, column sums of draw [chunk * slots + slot][M] at bias_off.
struct GemmHead {
  int blk, tile, part_off, bias_off;
};
struct GemmPlan {
  GemmMat mat[24];
  GemmBlock blk[16];
  GemmHead head[2];      // [0] alpha (M = 1, N = 256, draw column 3), [1] rgb (M = 3, N = 128, draw columns 0..2)
  int draw_mat;          // index of the [p_pad][4] head-gradient matrix
  int nheavy;
  int rows_h, chunks_h;  // sample rows per block (multiple of 16) and number of row chunks
  long long p_pad;
};

static_assert(sizeof(GemmPlan) + 8 <= 4096 && sizeof(GemmBatch) + 8 <= 4096, "passed by value: HIP kernel arguments are limited to 4 KiB");

constexpr int GEMM_ROWS = 16;   // sample rows per LDS stage

// rows-per-block search shared by anerf_train_layout (workspace size) and anerf_weight_grads
void gemm_plan_rows(long long p_pad, int nheavy, int* rows_h, int* chunks_h);

// reduce = false: only the GEMM is enqueued; the caller reduces later (launch_reduce_dw2: both passes of a step in ONE launch)
int launch_weight_grads(const GemmPlan& P, const GemmBatch& G, float* ws, bool b3, hipStream_t st, bool reduce = true);
int launch_reduce_dw2(const GemmBatch& G1, const float* ws1, const GemmBatch& G2, const float* ws2, hipStream_t st);
static_assert(2 * sizeof(GemmBatch) + 32 <= 4096, "two batches by value in one kernel-argument block");

}  // namespace anerf
