// anerf_gemm.h -- problem descriptors of the grouped weight-gradient GEMM (passed by value as kernel arguments).
#pragma once
#include <stdint.h>

namespace anerf {

struct GemmProb {
  const float* A;        // [p_pad][lda]   d(pre-activation) rows
  const float* B;        // [p_pad][ldb]   layer-input rows
  float* dst;            // gradient tensor rows m_first.. -> dst[(m - m_first) * dst_ld + dst_col0 + colmap[n]]
  float* bias_dst;       // bias gradient (column sums of A) or nullptr
  const int* colmap;     // output-column permutation (stream order -> torch order) or nullptr
  long long part_off;    // float offset of the [chunks][M][N] partial tiles in the workspace
  long long bias_off;    // float offset of the [chunks][M] bias partials, -1 if none
  long long out_base;    // prefix of (M*N + (bias ? M : 0)) over the problems, for k_reduce_dw
  int lda, ldb, lda_cols, ldb_cols, M, N, tiles_m, tiles_n, tile_base;
  int dst_ld, dst_col0, m_first, m_count, bm_first, bm_count;
};

struct GemmBatch {
  GemmProb p[16];
  int nprob, total_tiles, chunks;
  long long rows_per_chunk, p_pad, total_out;
};

}  // namespace anerf
