// anerf_gemm.hip -- weight gradients of the MLP as a GROUPED fp32-MFMA "TN" GEMM over the sample axis:
//     dW[m][n] = sum_p A[p][m] * B[p][n]      A = d(pre-activation) rows, B = layer-input rows (both row-major,
//                                             saved by k_mlp_fwd<TRAIN> / k_mlp_bwd), p = sample index.
// Autograd of the 12 nn.Linear layers of NeRF (core/networks/nerf.py:57-88) w.r.t. weights and biases.
//
// One launch covers every layer of a network.  Work unit = one WAVE computing a 128x128 output tile over a chunk
// of sample rows with v_mfma_f32_32x32x2_f32 (16 accumulator blocks = all 256 AGPRs).  Operand rows are staged
// [16 rows][128 cols] in LDS exactly as they lie in memory (global_load_lds_dwordx4, 2 rows per wave-instruction);
// lane (i, kk) then reads row 2s+kk, columns 4i..4i+3 with ONE ds_read_b128: the four floats are its operands for
// the four 32-row blocks of the tile (block b <-> columns 4i+b).  That column interleave costs nothing -- block b,
// row i' of the accumulator is simply output row 4i'+b, and on the way out four blocks' registers form a float4 of
// four consecutive output columns -- and it makes the inner loop 2 LDS reads + 16 MFMAs (1024 MFMA cycles) with
// NO address arithmetic.  (The fp32 MFMA does not overlap VALU on this chip, tools/probe: every VALU instruction
// in the loop is paid for in matrix time; the previous 64x64-per-wave version spent 2.3 VALU per MFMA.)
// Where its time goes (round-2 ablations, tools/microbench_gemm.py at 245 760 rows, 3.6-3.8 ms): one round of 255 blocks x
// 904 stages x 8192 MFMA cycles = 3.1 ms at 2.4 GHz (3.3 ms at the ~2.25 GHz the chip holds under this load); without
// operand loads, barriers and column sums 3.41 ms, i.e. the loop itself is at ~97 %; tile padding (432- and 648-wide
// operands in 128-column tiles) executes 6 % more MFMAs than the algorithmic count.  A 3-slot ring with counted
// `s_waitcnt vmcnt(2 x tiles)` (two stage times per load instead of one) measured the same as this double buffer
// (3.63 vs 3.60 ms): the kernel does not wait for HBM.  Issuing the LDS-DMA through inline asm (so that hipcc emits the
// counted `lgkmcnt(2)` the two-pairs-ahead operand reads were written for, instead of `lgkmcnt(0)`) also measured the same
// (3.87 vs 3.89 ms on one box): the loop does not wait for LDS either.
// A block = 4 such waves sharing operand tiles: 2x2 (two A tiles x two B tiles: a whole 256x256 layer per block,
// each activation row read once) or 1x4 (views layer, M = 128).  The two head problems (alpha 1 x 256, rgb 3 x 128; A = the
// [p][4] rows of draw) have no block of their own (rounds 1-2: a "skinny" block of 4-MFMA waves that held a CU per row chunk
// -- 17 of 256 -- for a load-bound trickle): the feature layer's block already stages the h7 rows, so its waves add
// draw[p][3] * h7[p][:] with 16 v_fma per stage, and one more 2x2 block stages the g rows as its fifth tile for the rgb rows
// (24 v_fma per wave and stage); 14 jobs x 18 row chunks = 252 CUs instead of 15 x 17 (head_stage / GemmHead).
// Blocks write partial tiles; k_reduce_dw sums the row chunks in a fixed order (deterministic) and scatters into
// the torch-layout gradient tensors (undoing the stream column order of X'/U').  Bias gradients = column sums of
// A, accumulated by the waves that own a problem's first column tile.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "anerf_dev.h"
#include "anerf_gemm.h"
#include "anerf_split.h"

namespace anerf {

constexpr int GT = 128;                         // tile edge (columns of an operand tile, rows/cols of an output tile)
constexpr int GTILE_BYTES = GEMM_ROWS * GT * 4; // 8 KiB
constexpr int GMINI_OFF = 5 * GTILE_BYTES;      // the stage's 16 draw rows ([16][4] floats) behind the five operand tiles
constexpr int GSTAGE_BYTES = GMINI_OFF + 256;   // 40 KiB + 256 B
constexpr int GLDS_BYTES = 2 * GSTAGE_BYTES;    // double buffer (fp32 kernel)
constexpr int GLDS3_BYTES = 3 * GSTAGE_BYTES;   // 3-slot ring (split-bf16 kernel)

// Chunking: every block covers the same rows_h sample rows; one block per CU at a time (256 AGPRs).
// Pick rows_h (multiple of 16) minimising  rounds(blocks over 256 CUs) x (rows + epilogue).
void gemm_plan_rows(long long p_pad, int nheavy, int* rows_h, int* chunks_h) {
  const int NCU = 256;
  long long best_cost = -1;
  int best_r = 0;
  const long long rmin = p_pad < 256 ? p_pad : 256;
  for (long long r = rmin / 16 * 16; r <= p_pad; r += 16) {
    if (r <= 0) continue;
    const long long ch = (p_pad + r - 1) / r;
    if (ch > 96) continue;
    const long long blocks = nheavy * ch;
    const long long rounds = (blocks + NCU - 1) / NCU;
    const long long cost = rounds * (r + 96);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_r = (int)r; }
  }
  if (best_r == 0) best_r = (int)((p_pad + 95) / 96 + 15) / 16 * 16;
#ifdef ANERF_EXP_GEMM_ROWS   // tuning build only: ANERF_GEMM_ROWS=<rows per block>
  if (const char* e = getenv("ANERF_GEMM_ROWS")) best_r = atoi(e) / 16 * 16;
#endif
  *rows_h = best_r;
  *chunks_h = (int)((p_pad + best_r - 1) / best_r);
}

// ---- loader: every operand tile of a stage is 8 two-row wave-instructions; wave w issues rows {2w, 2w+1} and
// {2w+8, 2w+9} of each tile.  Only 5 per-lane offsets (VGPRs) live across the MFMA loop: a spilled VGPR here would put
// a scratch reload -- an in-order VMEM op -- behind the freshly issued operand loads and expose their full latency at
// every stage.  Tile descriptors are copied into scalars up front: read through the plan reference they would be
// re-fetched with dependent s_load round trips at every stage (the asm "memory" clobber of the stage barrier forbids
// caching them).
struct GemmLoader {
  const char* t_ptr[5];
  int t_ld[5];
  unsigned t_lo[5];
  int ntiles, wave;
  char* smem;
  const char* d_ptr;     // the block's draw rows (head hosts); requested by wave 0 only
  unsigned d_lo;
  int d_on;

  __device__ __forceinline__ void init(const GemmPlan& G, const GemmBlock& B, char* smem_, long long r0, int wave_, int lane,
                                       bool head) {
    const int i = lane & 31, kk = lane >> 5;
    smem = smem_;
    wave = wave_;
    ntiles = B.ntiles;
    d_ptr = reinterpret_cast<const char*>(G.mat[G.draw_mat].ptr + r0 * 4);
    d_lo = (unsigned)lane * 4u;
    d_on = __builtin_amdgcn_readfirstlane((head && wave_ == 0) ? 1 : 0);
#pragma unroll
    for (int tile = 0; tile < 5; ++tile) {
      const int tt = tile < B.ntiles ? tile : 0;
      const GemmMat& M = G.mat[B.t[tt].mat];
      int c = B.t[tt].col0 + i * 4;
      const int cmax = M.ncols - 4;
      c = c > cmax ? cmax : c;   // clamp: columns past the edge repeat the last float4 (their outputs are never stored)
      t_ld[tile] = M.ld;
      t_ptr[tile] = reinterpret_cast<const char*>(M.ptr + (r0 + 2 * wave) * M.ld);    // wave-uniform
      t_lo[tile] = (unsigned)(kk * M.ld + c) * 4u;                                      // per lane
    }
  }
  // one tile's two pieces, predicated on `tile < ntiles` by a branch INSIDE the asm block (nothing for hipcc to restructure
  // around the MFMA loop): for the spread issue below
  __device__ __forceinline__ void issue_tile(int stage, int slot, int tile) const {
#ifdef ANERF_EXP_GEMM_NOLOAD
    (void)stage; (void)slot; (void)tile; return;
#endif
    const char* rowp = t_ptr[tile] + (long long)stage * (GEMM_ROWS * 4) * t_ld[tile];
    char* l = smem + slot * GSTAGE_BYTES + tile * GTILE_BYTES + wave * (2 * GT * 4);
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)l);
    asm volatile("s_cmp_lt_i32 %5, %6\n\ts_cbranch_scc0 1f\n\t"
                 "s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n1:"
                 :: "s"(lds0), "s"(lds0 + 8 * (GT * 4)), "v"(t_lo[tile]), "s"(rowp), "s"(rowp + (long long)t_ld[tile] * 32),
                    "s"(tile), "s"(ntiles) : "memory", "m0", "scc");
  }
  // the stage's 16 draw rows = 256 contiguous bytes = one dword per lane of ONE wave-instruction (predicated like issue_tile)
  __device__ __forceinline__ void issue_mini(int stage, int slot) const {
#ifdef ANERF_EXP_GEMM_NOLOAD
    (void)stage; (void)slot; return;
#endif
    const char* rowp = d_ptr + (long long)stage * (GEMM_ROWS * 16);
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + slot * GSTAGE_BYTES + GMINI_OFF));
    asm volatile("s_cmp_lg_i32 %3, 0\n\ts_cbranch_scc0 1f\n\t"
                 "s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n1:"
                 :: "s"(lds0), "v"(d_lo), "s"(rowp), "s"(d_on) : "memory", "m0", "scc");
  }
  __device__ __forceinline__ void issue(int stage, int slot) const {
#ifdef ANERF_EXP_GEMM_NOLOAD   // ablation build only (tools/ablate.sh): results are wrong
    (void)stage; (void)slot; return;
#endif
    issue_mini(stage, slot);
#pragma unroll
    for (int tile = 0; tile < 5; ++tile)
      if (tile < ntiles) {
        const char* rowp = t_ptr[tile] + (long long)stage * (GEMM_ROWS * 4) * t_ld[tile];
        char* l = smem + slot * GSTAGE_BYTES + tile * GTILE_BYTES + wave * (2 * GT * 4);
#ifdef ANERF_EXP_GEMM_BUILTIN_DMA   // round-1 form: 64-bit per-lane addresses (a v_lshl_add_u64 per piece inside the MFMA stream)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(rowp + t_lo[tile]), (lds_ptr_t)l, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(rowp + (long long)t_ld[tile] * 32 + t_lo[tile]), (lds_ptr_t)(l + 8 * (GT * 4)), 16, 0, 0);
#else
        // scalar row base + 32-bit lane offset: no VALU per piece; issued through asm as in the MLP kernels' weight pipe
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)l);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds0), "v"(t_lo[tile]), "s"(rowp) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                     :: "s"(lds0 + 8 * (GT * 4)), "v"(t_lo[tile]), "s"(rowp + (long long)t_ld[tile] * 32) : "memory", "m0");
#endif
      }
  }
};

// partial tile -> workspace [chunk][M][N]: accumulator block (a, c), register r, lane (i, kk) holds output row
// 4*((r&3) + 8*(r>>2) + 4*kk) + a, column 4*i + c; then the bias partials
__device__ __forceinline__ void gemm_store(const GemmWave& W, float* __restrict__ ws, int chunk, const f32x16 (&acc)[4][4],
                                           f32x4 asum, bool do_bias, int i, int kk) {
  float* part = ws + W.part_off + (long long)chunk * W.M * W.N;
  const int n = W.n0 + 4 * i;
  if (n < W.N) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowi = (r & 3) + 8 * (r >> 2) + 4 * kk;
        const int m = W.m0 + 4 * rowi + a;
        if (m < W.M) {
          const f32x4 o = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
          *reinterpret_cast<f32x4*>(part + (long long)m * W.N + n) = o;
        }
      }
  }
  if (do_bias) {
    float* bp = ws + W.bias_off + (long long)chunk * W.M;
#pragma unroll
    for (int a = 0; a < 4; ++a) asum[a] += __shfl_xor(asum[a], 32);
    if (kk == 0) {
      const int m = W.m0 + 4 * i;
      if (m < W.M) *reinterpret_cast<f32x4*>(bp + m) = asum;   // M is a multiple of 4 for every heavy problem
    }
  }
}

// ---- the head problems, hosted by two heavy blocks (GemmHead).  HEAD = 1 (alpha): wave (r, c) = (wave & 1, wave >> 1) of the
// feature layer's block multiplies draw[p][3] into the h7 columns of its own B tile for the row pairs s = r, r+2, r+4, r+6;
// HEAD = 2 (rgb): wave k multiplies draw[p][0..2] into the g tile's columns for s = k, k+4.  Lane (i, kk) owns row 2s+kk,
// columns 4i..4i+3, like the MFMA operands; the values come from LDS again (re-using the B operands' registers would need a
// per-wave select on every value).  Plain fp32 FMAs in sample order; the two (four) waves' sums and the chunks are added in
// fixed order by k_reduce_dw: deterministic like everything else here.
template <int HEAD>
struct HeadWork {
  static constexpr int NM = HEAD == 2 ? 3 : 1;       // output rows
  static constexpr int NQ = HEAD == 2 ? 2 : 4;       // row pairs of a stage this wave takes
  f32x4 acc[NM];
  float sum[NM];
  f32x4 hv[NQ];
  f32x4 dv[NQ];
  unsigned h_off, d_off;

  __device__ __forceinline__ void init(const GemmPlan& G, const GemmWave& W, int wave, int i, int kk) {
#pragma unroll
    for (int m = 0; m < NM; ++m) { acc[m] = f32x4{0.f, 0.f, 0.f, 0.f}; sum[m] = 0.f; }
    const int s0 = HEAD == 2 ? wave : (wave & 1);
    const int tile = HEAD == 2 ? G.head[1].tile : W.b_tile;
    h_off = tile * GTILE_BYTES + (2 * s0 + kk) * (GT * 4) + i * 16;
    d_off = GMINI_OFF + (2 * s0 + kk) * 16;
  }
  __device__ __forceinline__ void read(const char* base) {
    constexpr int STEP = HEAD == 2 ? 8 : 4;          // rows between two of the wave's row pairs
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      hv[q] = *reinterpret_cast<const f32x4*>(base + h_off + q * STEP * (GT * 4));
      if constexpr (HEAD == 2) dv[q] = *reinterpret_cast<const f32x4*>(base + d_off + q * STEP * 16);
      else dv[q][0] = *reinterpret_cast<const float*>(base + d_off + 12 + q * STEP * 16);
    }
  }
  __device__ __forceinline__ void fma() {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float d = dv[q][m];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[m][c] = fmaf(d, hv[q][c], acc[m][c]);
        sum[m] += d;
      }
  }
  __device__ __forceinline__ void store(const GemmPlan& G, float* __restrict__ ws, int chunk, int wave, int i, int kk) {
    constexpr int SLOTS = HEAD == 2 ? 4 : 2, N = HEAD == 2 ? 128 : 256;
    const GemmHead& Hd = G.head[HEAD - 1];
    const int slot = chunk * SLOTS + (HEAD == 2 ? wave : (wave & 1));
    const int n0 = HEAD == 2 ? 0 : 128 * (wave >> 1);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[m][c] += __shfl_xor(acc[m][c], 32);
      sum[m] += __shfl_xor(sum[m], 32);
      if (kk == 0) *reinterpret_cast<f32x4*>(ws + Hd.part_off + ((long long)slot * NM + m) * N + n0 + 4 * i) = acc[m];
      if (kk == 0 && i == 0 && n0 == 0) ws[Hd.bias_off + (long long)slot * NM + m] = sum[m];
    }
  }
};
template <>
struct HeadWork<0> {
  __device__ __forceinline__ void init(const GemmPlan&, const GemmWave&, int, int, int) {}
  __device__ __forceinline__ void read(const char*) {}
  __device__ __forceinline__ void fma() {}
  __device__ __forceinline__ void store(const GemmPlan&, float*, int, int, int, int) {}
};

#ifdef ANERF_EXP_STAGE_TIMING   // debug build only (tools/stage_timing_gemm.py)
__device__ unsigned long long* g_gemm_tbuf = nullptr;
#endif

template <int HEAD>
__device__ __forceinline__ void gemm_body(const GemmPlan& G, const GemmBlock& B, char* smem, float* __restrict__ ws,
                                          long long r0, int nst, int chunk, int wave, int lane) {
  const int i = lane & 31, kk = lane >> 5;
  const GemmWave& W = B.w[wave];
  const bool active = W.a_tile >= 0;
  const bool do_bias = active && W.bias_off >= 0;
  GemmLoader L;
  L.init(G, B, smem, r0, wave, lane, HEAD != 0);
  auto issue = [&](int stage, int slot) { L.issue(stage, slot); };

  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  f32x4 asum = {0.f, 0.f, 0.f, 0.f};
  HeadWork<HEAD> hw;
  hw.init(G, W, wave, i, kk);

  const unsigned a_off = W.a_tile * GTILE_BYTES + kk * (GT * 4) + i * 16;
  const unsigned b_off = W.b_tile * GTILE_BYTES + kk * (GT * 4) + i * 16;

  if (nst > 0) issue(0, 0);
  if (!active) {                                       // idle wave of a partially filled block: loads and barriers only
    for (int t = 0; t < nst; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 1 < nst) issue(t + 1, (t + 1) & 1);
    }
    return;
  }
  // The loop body below is straight-line code on purpose (the column sums are accumulated by every wave and only
  // stored by the ones that own a bias): any branch that touches it makes the compiler shuttle the 256 accumulators
  // between AGPRs and VGPRs at the merge points.
#ifdef ANERF_EXP_STAGE_TIMING
  unsigned long long* tb = g_gemm_tbuf;
  if (tb && blockIdx.x % 16 == 0 && lane == 0) tb += ((long long)(blockIdx.x / 16) * 4 + wave) * 3 * 128; else tb = nullptr;
#endif
  for (int t = 0; t < nst; ++t) {
#ifdef ANERF_EXP_STAGE_TIMING
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (tb && t < 128) { tb[3 * t] = t0; tb[3 * t + 1] = t1; tb[3 * t + 2] = t2; }
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef ANERF_EXP_GEMM_NOBARRIER   // ablation build only: results are wrong
    __syncthreads();                                   // stage t landed; everyone is done with stage t-1's slot
#endif
#endif
#ifdef ANERF_EXP_GEMM_BURST   // round-1 form: the whole next stage requested right behind the barrier
    if (t + 1 < nst) issue(t + 1, (t + 1) & 1);
#else
    // the next stage is requested a tile (two pieces) per MFMA group, groups 0..4 (+ the draw rows with group 5): the four
    // waves' 40 pieces no longer pile up in the CU's vector-memory queue behind the barrier (stage 9 280 -> 9 030 clocks of
    // 8 192; the operands still arrive in time: 8 clocks of vmcnt wait per stage, tools/stage_timing_gemm.py)
    const int nt_stage = t + 1 < nst ? t + 1 : t;      // (the last stage reloads itself into the idle slot: branch-free)
#endif
    const char* base = smem + (t & 1) * GSTAGE_BYTES;
    // Operands are read TWO row pairs ahead: the s_waitcnt in front of a group's MFMAs then only has to cover reads
    // issued a full group (1024 MFMA cycles) earlier and may leave the newest pair in flight (lgkmcnt(2)).  The
    // sched_barriers pin that order: left alone, the scheduler hoists the column-sum adds of later pairs above the
    // MFMAs and drags a wait for the freshly issued reads in front of them.
    constexpr int NS = GEMM_ROWS / 2;
    f32x4 av[NS], bv[NS];
    av[0] = *reinterpret_cast<const f32x4*>(base + a_off);
    bv[0] = *reinterpret_cast<const f32x4*>(base + b_off);
    av[1] = *reinterpret_cast<const f32x4*>(base + a_off + 2 * GT * 4);
    bv[1] = *reinterpret_cast<const f32x4*>(base + b_off + 2 * GT * 4);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 2 < NS) {
        av[s + 2] = *reinterpret_cast<const f32x4*>(base + a_off + (s + 2) * (2 * GT * 4));
        bv[s + 2] = *reinterpret_cast<const f32x4*>(base + b_off + (s + 2) * (2 * GT * 4));
      }
      if (s == NS - 2) hw.read(base);                  // behind the last operand reads: consumed after the stage's MFMAs
#ifndef ANERF_EXP_GEMM_BURST
      if (s < 5) L.issue_tile(nt_stage, (t + 1) & 1, s);
      if (HEAD != 0 && s == 5) L.issue_mini(nt_stage, (t + 1) & 1);
#endif
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][a], bv[s][c], acc[a][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // column sums (bias gradients) of the stage as ONE block of packed adds behind its MFMAs: 16 v_pk_add_f32 instead of
    // 32 v_add_f32 sprinkled between the groups (a VALU instruction inside the fp32 MFMA stream costs ~14 clocks, in a
    // block 4-8; stage 9350 -> clocks, tools/stage_timing_gemm.py); the head hosts' FMAs go in the same block
    f32x2 s_lo = {asum[0], asum[1]}, s_hi = {asum[2], asum[3]};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      s_lo += f32x2{av[s][0], av[s][1]};
      s_hi += f32x2{av[s][2], av[s][3]};
    }
    asum = f32x4{s_lo[0], s_lo[1], s_hi[0], s_hi[1]};
    hw.fma();
    __builtin_amdgcn_sched_barrier(0);
  }
  gemm_store(W, ws, chunk, acc, asum, do_bias, i, kk);
  hw.store(G, ws, chunk, wave, i, kk);
}

// ---- split-bf16 variant of the heavy body (anerf_weight_grads_b3): the products run on v_mfma_f32_32x32x16_bf16 with
// hi/lo-split operands (anerf_split.h).  A stage's 16 sample rows are exactly one k-step: lane (i, kh) reads rows
// 8kh..8kh+7 of its 4-column group with 8 ds_read_b128 per operand, splits the 8 values of each of the 4 blocks into
// (hi, lo) bf16x8 and issues 3 MFMAs per (a, c) block pair (48 per stage = 1536 matrix cycles).  To keep the matrix
// pipe fed the stages are software-pipelined through a 3-slot ring: while stage s is multiplied, the raw operands of
// stage s+1 (already landed -- same invariant as the forward kernel's pipe) are read and split, interleaved with the
// MFMAs chunk by chunk (bf16 MFMAs, unlike fp32 ones, do run beside VALU; see the loop).  Same accumulator layout, same
// epilogue, same deterministic reduction.  At ~1.4 ms per launch the kernel moves the same 5.5 GB of operand tiles as the fp32
// one does in 3.1 ms -- 3.9 TB/s through L2 / HBM -- which is what bounds it now: writing the interleave out (below) instead
// of requesting it changed the step time by < 1 %.
struct SplitOps {
  BOp A[4], B[4];
};

__device__ __forceinline__ void gemm_read_raw(const char* base, unsigned a_off, unsigned b_off, f32x4 (&ar)[8], f32x4 (&br)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    ar[t] = *reinterpret_cast<const f32x4*>(base + a_off + t * (GT * 4));
    br[t] = *reinterpret_cast<const f32x4*>(base + b_off + t * (GT * 4));
  }
}

__device__ __forceinline__ void gemm_split(const f32x4 (&ar)[8], const f32x4 (&br)[8], SplitOps& o, f32x4& asum) {
#pragma unroll
  for (int t = 0; t < 8; ++t) asum += ar[t];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    o.A[q] = split8(ar[0][q], ar[1][q], ar[2][q], ar[3][q], ar[4][q], ar[5][q], ar[6][q], ar[7][q]);
    o.B[q] = split8(br[0][q], br[1][q], br[2][q], br[3][q], br[4][q], br[5][q], br[6][q], br[7][q]);
  }
}

template <int HEAD>
__device__ __forceinline__ void gemm_body_b3(const GemmPlan& G, const GemmBlock& B, char* smem, float* __restrict__ ws,
                                             long long r0, int nst, int chunk, int wave, int lane) {
  const int i = lane & 31, kh = lane >> 5;
  const GemmWave& W = B.w[wave];
  const bool active = W.a_tile >= 0;
  const bool do_bias = active && W.bias_off >= 0;
  GemmLoader L;
  L.init(G, B, smem, r0, wave, lane, HEAD != 0);
  HeadWork<HEAD> hw;                                   // the hosted head rows stay fp32 FMAs (as they were fp32 MFMAs before)
  hw.init(G, W, wave, i, kh);
  // ring protocol: while stage s is consumed, stages s and s+1 have landed and s+2 is in flight
  if (nst > 0) L.issue(0, 0);
  if (nst > 1) L.issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (nst > 2) L.issue(2, 2);
  auto end_stage = [&](int s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage s+2 landed (this wave's share)
    __syncthreads();                                   // ... everybody's; nobody reads slot s % 3 any more
    if (s + 3 < nst) L.issue(s + 3, s % 3);
  };
  if (!active) {
    for (int s = 0; s < nst; ++s) end_stage(s);
    return;
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  f32x4 asum = {0.f, 0.f, 0.f, 0.f};
  const unsigned a_off = W.a_tile * GTILE_BYTES + 8 * kh * (GT * 4) + i * 16;
  const unsigned b_off = W.b_tile * GTILE_BYTES + 8 * kh * (GT * 4) + i * 16;

  SplitOps cur;
  {
    f32x4 ar[8], br[8];
    gemm_read_raw(smem, a_off, b_off, ar, br);
    gemm_split(ar, br, cur, asum);
  }
#ifdef ANERF_EXP_GEMM_B3_UNFENCED   // round-1 form: the interleave is only requested (sched_group_barrier), and not honoured
  for (int s = 0; s < nst; ++s) {
    SplitOps nxt = cur;
    f32x4 ar[8], br[8];
    const bool more = s + 1 < nst;
    if (more) gemm_read_raw(smem + ((s + 1) % 3) * GSTAGE_BYTES, a_off, b_off, ar, br);
    hw.read(smem + (s % 3) * GSTAGE_BYTES);
    hw.fma();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.A[a].lo, cur.B[c].hi, acc[a][c], 0, 0, 0);
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.A[a].hi, cur.B[c].lo, acc[a][c], 0, 0, 0);
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.A[a].hi, cur.B[c].hi, acc[a][c], 0, 0, 0);
      }
    if (more) gemm_split(ar, br, nxt, asum);
#pragma unroll
    for (int k = 0; k < 48; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      if (k < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
    }
    end_stage(s);
    cur = nxt;
  }
#else
  // The interleave is written out and fenced: a stage is eight chunks of 6 MFMAs (two (a, c) block pairs) + one split8 of
  // the NEXT stage's operands (~30 VALU, which the bf16 matrix pipe runs beside), a sched_barrier(0) behind each chunk.
  // Requested with sched_group_barriers only, the compiler emitted the 48 MFMAs first and the ~300 split instructions in a
  // row behind them (~5 100 clocks per stage for 1 536 matrix clocks).  Branch-free: the last stage splits whatever the
  // idle ring slot holds (never used) and its column sums are multiplied by 0.
  for (int s = 0; s < nst; ++s) {
    SplitOps nxt;
    f32x4 ar[8], br[8];
    const float mf = s + 1 < nst ? 1.f : 0.f;
    gemm_read_raw(smem + ((s + 1) % 3) * GSTAGE_BYTES, a_off, b_off, ar, br);
    hw.read(smem + (s % 3) * GSTAGE_BYTES);            // stage s's own slot: valid until end_stage(s)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int a = (2 * k + j) >> 2, c = (2 * k + j) & 3;
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.A[a].lo, cur.B[c].hi, acc[a][c], 0, 0, 0);
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.A[a].hi, cur.B[c].lo, acc[a][c], 0, 0, 0);
        acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur.A[a].hi, cur.B[c].hi, acc[a][c], 0, 0, 0);
      }
      const int q = k & 3;
      if (k < 4) {
        nxt.A[q] = split8(ar[0][q], ar[1][q], ar[2][q], ar[3][q], ar[4][q], ar[5][q], ar[6][q], ar[7][q]);
        asum[q] = fmaf(mf, ((ar[0][q] + ar[1][q]) + (ar[2][q] + ar[3][q])) + ((ar[4][q] + ar[5][q]) + (ar[6][q] + ar[7][q])), asum[q]);
      } else {
        nxt.B[q] = split8(br[0][q], br[1][q], br[2][q], br[3][q], br[4][q], br[5][q], br[6][q], br[7][q]);
      }
      if (k == 7) hw.fma();
      __builtin_amdgcn_sched_barrier(0);
    }
    end_stage(s);
    cur = nxt;
  }
#endif
  gemm_store(W, ws, chunk, acc, asum, do_bias, i, kh);
  hw.store(G, ws, chunk, wave, i, kh);
}

template <bool B3>
__global__ __launch_bounds__(256) void k_gemm_tn(const GemmPlan G, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // blockIdx -> (block job, row chunk).  Consecutive workgroup ids go round-robin over the 8 XCDs (each with its own
  // L2): within a full group of 8 chunks, chunk c of every job gets id = 8 * job + c % 8, so the jobs that share
  // activation rows share one L2.  No padding ids: every XCD gets the same number of blocks (an earlier version
  // padded the chunk count to a multiple of 8 with empty blocks -- XCD 0 then ran 3 chunks while the others ran 2,
  // and the kernel took two rounds instead of one).
  const int bid = blockIdx.x;
  const int full = (G.chunks_h / 8) * 8 * G.nheavy;          // ids covered by full groups of 8 chunks
  int job, chunk;
  if (bid < full) {
    const int grp = bid / (8 * G.nheavy), rem = bid - grp * (8 * G.nheavy);
    job = rem >> 3;
    chunk = grp * 8 + (rem & 7);
  } else {
    const int m = G.chunks_h & 7, k = bid - full;
    job = k / m;
    chunk = (G.chunks_h / 8) * 8 + k % m;
  }
  const long long r0 = (long long)chunk * G.rows_h;
  long long r1 = r0 + G.rows_h;
  if (r1 > G.p_pad) r1 = G.p_pad;
  const int nst = (int)((r1 - r0) / GEMM_ROWS);
  const GemmBlock& B = G.blk[job];
#ifdef ANERF_EXP_GEMM_NOHEAD   // ablation build only (tools/ablate.sh): head gradients are not computed
  const int head = 0;
#else
  const int head = job == G.head[0].blk ? 1 : (job == G.head[1].blk ? 2 : 0);      // block-uniform
#endif
  if constexpr (B3) {
    if (head == 1) gemm_body_b3<1>(G, B, smem, ws, r0, nst, chunk, wave, lane);
    else if (head == 2) gemm_body_b3<2>(G, B, smem, ws, r0, nst, chunk, wave, lane);
    else gemm_body_b3<0>(G, B, smem, ws, r0, nst, chunk, wave, lane);
  } else {
    if (head == 1) gemm_body<1>(G, B, smem, ws, r0, nst, chunk, wave, lane);
    else if (head == 2) gemm_body<2>(G, B, smem, ws, r0, nst, chunk, wave, lane);
    else gemm_body<0>(G, B, smem, ws, r0, nst, chunk, wave, lane);
  }
}

// Sum the chunk partials in index order and scatter into the gradient tensors.
__device__ __forceinline__ void reduce_dw_one(const GemmBatch& G, const float* __restrict__ ws, long long gid) {
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
    // six partials in flight at a time, added in chunk order (the same sequence of additions as a plain loop, whose 18+
    // dependent load -> add round trips per thread took 38 us per launch at 384 rays, 3 % of that step; 27 us now.  A float4
    // per thread was no faster: the kernel lives on threads in flight, not on bytes per thread)
    float s = 0.f;
    int c = 0;
    for (; c + 9 <= pr.chunks; c += 9) {      // (round 6: nine in flight -- 18 chunks = two batches; the additions keep their order)
      float v[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) v[k] = src[(long long)(c + k) * mn];
#pragma unroll
      for (int k = 0; k < 9; ++k) s += v[k];
    }
    for (; c + 6 <= pr.chunks; c += 6) {
      const float v0 = src[(long long)c * mn], v1 = src[(long long)(c + 1) * mn], v2 = src[(long long)(c + 2) * mn],
                  v3 = src[(long long)(c + 3) * mn], v4 = src[(long long)(c + 4) * mn], v5 = src[(long long)(c + 5) * mn];
      s += v0; s += v1; s += v2; s += v3; s += v4; s += v5;
    }
    for (; c < pr.chunks; ++c) s += src[(long long)c * mn];
    if (mrow >= pr.m_first && mrow < pr.m_first + pr.m_count) {
      const int col = pr.colmap ? pr.colmap[n] : n;
      float* d = pr.dst + (long long)(mrow - pr.m_first) * pr.dst_ld + pr.dst_col0 + col;
      if (pr.colscale) s *= pr.colscale[col];
      *d = G.accumulate ? *d + s : s;
    }
  } else {
    const int mrow = (int)(e - mn);
    const float* src = ws + pr.bias_off + mrow;
    float s = 0.f;
    for (int c = 0; c < pr.chunks; ++c) s += src[(long long)c * pr.M];
    if (mrow >= pr.bm_first && mrow < pr.bm_first + pr.bm_count) {
      float* d = pr.bias_dst + (mrow - pr.bm_first);
      *d = G.accumulate ? *d + s : s;
    }
  }
}
__global__ void k_reduce_dw(const GemmBatch G, const float* __restrict__ ws) {
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (gid >= G.total_out) return;
  reduce_dw_one(G, ws, gid);
}
// both network passes of a training step in one launch (the fine pass's partials wait in their own workspace region until the
// coarse pass's GEMM is done): the same per-element additions, one launch of 27 us instead of two
__global__ void k_reduce_dw2(const GemmBatch G1, const float* __restrict__ ws1, const GemmBatch G2, const float* __restrict__ ws2) {
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (gid < G1.total_out) reduce_dw_one(G1, ws1, gid);
  else if (gid - G1.total_out < G2.total_out) reduce_dw_one(G2, ws2, gid - G1.total_out);
}

int launch_reduce_dw2(const GemmBatch& G1, const float* ws1, const GemmBatch& G2, const float* ws2, hipStream_t st) {
  const long long tot = G1.total_out + G2.total_out;
  hipLaunchKernelGGL(k_reduce_dw2, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, G1, ws1, G2, ws2);
  return check_launch("k_reduce_dw2");
}

int launch_weight_grads(const GemmPlan& P, const GemmBatch& G, float* ws, bool b3, hipStream_t st, bool reduce) {
  static unsigned long long lds_set[2] = {};   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_gemm_tn<false>), GLDS_BYTES, &lds_set[0]);
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_gemm_tn<true>), GLDS3_BYTES, &lds_set[1]);
  const int grid = P.chunks_h * P.nheavy;
  if (b3) hipLaunchKernelGGL(k_gemm_tn<true>, dim3((unsigned)grid), dim3(256), GLDS3_BYTES, st, P, ws);
  else hipLaunchKernelGGL(k_gemm_tn<false>, dim3((unsigned)grid), dim3(256), GLDS_BYTES, st, P, ws);
  int rc = check_launch("k_gemm_tn");
  if (rc || !reduce) return rc;
  hipLaunchKernelGGL(k_reduce_dw, dim3((unsigned)((G.total_out + 255) / 256)), dim3(256), 0, st, G, (const float*)ws);
  return check_launch("k_reduce_dw");
}

}  // namespace anerf

#ifdef ANERF_EXP_STAGE_TIMING
extern "C" void anerf_debug_set_gemm_timing_buf(unsigned long long* p) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(anerf::g_gemm_tbuf), &p, sizeof(p));
}
#endif
