// anerf_mlp_bwd.hip -- backward kernels of the fused MLP (gfx950): k_mlp_bwd (backward-data on the W^T image) and
// k_mlp_bwd_in (input gradients for pose optimisation / frame codes).  Same register-resident transposed fp32-MFMA
// scheme as the forward (see anerf_mlp.hip / DESIGN.md 4.2).
#include "anerf_fwd_common.h"

namespace anerf {

// backward-data kernel: d(raw) -> d(pre-activation) of every layer, same register-resident transposed scheme
// with the W^T weight image (anerf_layout which=1).  Per layer l:  dh_{l-1} = W_l^T dz_l ;  dz_{l-1} = dh_{l-1}
// * [h_{l-1} > 0].  Writes dz0..dz7 [8][Ppad][256], dF [Ppad][256], dZv [Ppad][128] for the weight-gradient GEMMs
// (anerf_gemm.hip).  Autograd of NeRF.forward (core/networks/nerf.py:94-148) w.r.t. activations.
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
  const float* packed_t;   // W^T image
  const float* aux;        // natural-order head weights (forward aux)
  const float* draw;       // [P][4]
  const float* save_h;     // [8][Ppad][256]
  const float* save_g;     // [Ppad][128]
  float* dz;               // [8][Ppad][256]
  float* df;               // [Ppad][256]
  float* dzv;              // [Ppad][128]
  long long P, Ppad;
  int nstages;
};

// ---- helpers on the accumulator layout (register r of block nb, lane half h = feature 32nb + (r&3) + 8(r>>2) + 4h)
template <int NB>
__device__ __forceinline__ void load_mask(f32x4 (&mk)[32], const float* __restrict__ row_h) {
#ifdef ANERF_EXP_BWD_NOMASK   // ablation build only: masks are not loaded (results are wrong)
#pragma unroll
  for (int i = 0; i < 4 * NB; ++i) mk[i] = f32x4{1.f, 1.f, 1.f, 1.f};
  return;
#endif
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) mk[4 * nb + q] = *reinterpret_cast<const f32x4*>(row_h + 32 * nb + 8 * q);
}
// acc <- acc * [saved activation > 0], in place (the result is read directly as the next layer's MFMA B operands)
template <int NB>
__device__ __forceinline__ void mask_pass(f32x16 (&acc)[NB], const f32x4 (&mk)[32]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[nb][4 * q + 0] = mk[4 * nb + q].x > 0.f ? acc[nb][4 * q + 0] : 0.f;
      acc[nb][4 * q + 1] = mk[4 * nb + q].y > 0.f ? acc[nb][4 * q + 1] : 0.f;
      acc[nb][4 * q + 2] = mk[4 * nb + q].z > 0.f ? acc[nb][4 * q + 2] : 0.f;
      acc[nb][4 * q + 3] = mk[4 * nb + q].w > 0.f ? acc[nb][4 * q + 3] : 0.f;
    }
}
template <int NB>
__device__ __forceinline__ void store_acc(float* __restrict__ row_h, const f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
      *reinterpret_cast<f32x4*>(row_h + 32 * nb + 8 * q) = o;
    }
}
template <int NB>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// One layer of the backward chain:   out += W^T * m(prev),   m(prev) = prev * [saved activation of prev's layer > 0]
// (MASK) or prev itself (the feature gradient df, which has no activation).  `prev` holds RAW sums; the ReLU mask is applied
// where the values are consumed, stage by stage (stage s of the layer = block s of prev = 16 values per lane):
//   * top of stage s (right behind the barrier that ends stage s-1, in front of the weight pipe's re-issue): the block's
//     16 masked operands `bq` are formed from the accumulator block and the 4 mask quads `mk` fetched during stage s-1,
//     and the 4 mask quads of block s+1 are requested (of block 0 of THIS layer's output during the last stage, for the
//     next layer -- `next_mask_row_h`);
//   * every k-group multiplies its quad of `bq` and stores it: that float4 is the k-group's row piece of dz (or df);
// so mask loads and dz stores reach the CU's vector-memory path at most a few at a time (it takes ~100 cycles per 1 KiB
// wave-access; round 1 issued 32-load and 32-store bursts per layer and lost 8 % of the kernel to each), 16 + 16 VGPRs
// of mask / operand state replace the 128-register mask array, and nothing is waited for twice: the mask quads are
// complete at the stage barrier's `vmcnt(0)`.
// LAST: the final layer of the chain additionally collects the mask row of ITS output (h0) a block per stage into mk0.
// ------------------------------------------------------------------------------------------------
template <bool MASK>
__device__ __forceinline__ void form_operands(float (&bq)[16], const f32x16& blk, const f32x4 (&mk)[4]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (MASK) {
      bq[4 * q + 0] = mk[q].x > 0.f ? blk[4 * q + 0] : 0.f;
      bq[4 * q + 1] = mk[q].y > 0.f ? blk[4 * q + 1] : 0.f;
      bq[4 * q + 2] = mk[q].z > 0.f ? blk[4 * q + 2] : 0.f;
      bq[4 * q + 3] = mk[q].w > 0.f ? blk[4 * q + 3] : 0.f;
    } else {
      bq[4 * q + 0] = blk[4 * q + 0];
      bq[4 * q + 1] = blk[4 * q + 1];
      bq[4 * q + 2] = blk[4 * q + 2];
      bq[4 * q + 3] = blk[4 * q + 3];
    }
  }
}
__device__ __forceinline__ void load_quads(f32x4 (&mk)[4], const float* __restrict__ row_h_blk) {
#ifdef ANERF_EXP_BWD_NOMASK   // ablation build only: masks are not loaded (results are wrong)
#pragma unroll
  for (int q = 0; q < 4; ++q) mk[q] = f32x4{1.f, 1.f, 1.f, 1.f};
  return;
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) mk[q] = *reinterpret_cast<const f32x4*>(row_h_blk + 8 * q);
}

// KG0: index of the layer's first k-group inside its weight segment (a multiple of 4).  On entry `mk` holds the mask quads
// of prev's block 0 (MASK) and the pipe stands right behind a stage barrier with its re-issue still to do (`pending`),
// or -- first layer of the chain -- somewhere inside a running stage (`pending` false).
template <bool MASK, bool LAST, int KG0>
__device__ __forceinline__ void bwd_layer(Pipe3B& pipe, f32x16 (&out)[8], const f32x16 (&prev)[8], f32x4 (&mk)[4],
                                          const float* __restrict__ prev_mask_row_h, float* __restrict__ prev_dz_row_h,
                                          const float* __restrict__ next_mask_row_h, f32x4 (&mk0)[32], bool& pending) {
  float bq[16];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    // ONE block behind the stage barrier (fence in front, pins behind): hoisted piecemeal into the previous stage's k-groups --
    // where the scheduler put them -- the ~48 instructions cost the matrix pipe ~14 clocks each instead of ~6
    __builtin_amdgcn_sched_barrier(0);
    form_operands<MASK>(bq, prev[s], mk);
    // pin the operands in front of the re-issue: their VALU is free to move, and when hipcc scheduled it behind the (hidden)
    // LDS-DMA issue its counted vmcnt waits for the mask quads also waited for DMA pieces issued a moment before -- every
    // second stage of the trunk took 12 500 clocks instead of 9 600 (tools/stage_timing_bwd.py)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bq[i]));
    if (pending) pipe.stage_refill();
    pending = false;
    // (spreading the weight pipe's eight re-issue pieces over k-groups 0..2 as well was slower: tile 680k -> 693k clocks;
    // bringing the mask rows in by coalesced LDS-DMA -- 8 whole rows per instruction, XOR-swizzled chunks, 2 x 4 KiB per wave,
    // conflict-free read-back -- instead of these 32-row gathers is correct and exactly as fast: 688k, the wait moves to the
    // stage barrier's vmcnt(0))
    // the next block's mask quads: one load per k-group, next to its store (all four behind the barrier piled up with the
    // four waves' 32 LDS-DMA pieces in the CU's vector-memory queue: 10 % of the kernel, ablation in DESIGN 4.2)
    const float* nq = nullptr;
    if (s < 7) {
      if constexpr (MASK) nq = prev_mask_row_h + 32 * (s + 1);
    } else if (next_mask_row_h) {
      nq = next_mask_row_h;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kg = 4 * s + ks;
      const f32x4 o = {bq[4 * ks], bq[4 * ks + 1], bq[4 * ks + 2], bq[4 * ks + 3]};
#ifndef ANERF_EXP_BWD_NOSTORE   // ablation build only: results are not written
      *reinterpret_cast<f32x4*>(prev_dz_row_h + 8 * kg) = o;
#endif
#ifndef ANERF_EXP_BWD_NOMASK
      if (nq) mk[ks] = *reinterpret_cast<const f32x4*>(nq + 8 * ks);
      if constexpr (LAST) mk0[4 * s + ks] = *reinterpret_cast<const f32x4*>(next_mask_row_h + 32 * s + 8 * ks);   // mask row of `out` (h0)
#else
      mk[ks] = f32x4{1.f, 1.f, 1.f, 1.f};
      if constexpr (LAST) mk0[4 * s + ks] = f32x4{1.f, 1.f, 1.f, 1.f};
#endif
      __builtin_amdgcn_sched_barrier(0);     // the store / load go in front of the k-group's MFMAs
      kgroup<8, Pipe3B, false>(pipe, out, KG0 + kg, false, false, o.x, o.y, o.z, o.w);
    }
    pipe.stage_rendezvous();
    pending = true;
  }
}

#ifdef ANERF_EXP_STAGE_TIMING   // debug build only (tools/stage_timing_bwd.py)
__device__ unsigned long long* g_bwd_tbuf = nullptr;
#endif

__global__ __launch_bounds__(256) void k_mlp_bwd(const BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe3B pipe;
  pipe.init(A.packed_t, smem, wave, lane, A.nstages);
#ifdef ANERF_EXP_STAGE_TIMING
  const unsigned long long tt0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long* const tb = g_bwd_tbuf;
  if (tb && blockIdx.x % 97 == 0 && lane == 0) pipe.tbuf = tb + ((long long)(blockIdx.x / 97) * 4 + wave) * 3 * 128;
#endif
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  // every lane stores unconditionally to the clamped row pc: tail lanes (p >= P) recompute the last valid sample and rewrite
  // its rows with identical values -- no exec-mask changes inside the MFMA stream
  const long long pc = p < A.P ? p : A.P - 1;
  // head rows (w_c, w_alpha) -> LDS copy of the aux image, read back as float4 per 8-feature group
  float* aux_l = reinterpret_cast<float*>(smem + LDS_AUX_OFF);
  {   // by (hidden) LDS-DMA, as in k_mlp_fwd
    constexpr int NPIECE = AUX_FLOATS / 256;
    const char* ga = reinterpret_cast<const char*>(A.aux) + lane * 16;
#pragma unroll
    for (int k = 0; k < (NPIECE + 3) / 4; ++k) {
      const int piece = 4 * k + wave;
      if (piece < NPIECE) {
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + LDS_AUX_OFF)) + piece * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds0), "v"(ga + piece * 1024) : "memory", "m0");
      }
    }
    const int i = NPIECE * 64 + tid;
    if (i < AUX_FLOATS / 4) reinterpret_cast<f32x4*>(aux_l)[i] = reinterpret_cast<const f32x4*>(A.aux)[i];
  }
  const float* aux_h = aux_l + 4 * h;
  const f32x4 dr = *reinterpret_cast<const f32x4*>(A.draw + pc * 4);
  f32x4 mk0[32];     // view-layer mask (16 quads) at the start, the h0 mask row at the end
  load_mask<4>(mk0, A.save_g + pc * 128 + 4 * h);
  pipe.begin();   // barrier: aux visible, weight stages 0/1 landed
  pipe.prime();

  f32x16 accA[8], accB[8];   // ping-pong: a layer's output set (raw sums) is the next layer's operand set
  f32x16 accv[4];
  // ---- rgb head (VALU): dg = Wc^T dc ; dzv = dg * [g > 0]
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int o = 32 * nb + 8 * q;
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(aux_h + AUX_WC + o);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(aux_h + AUX_WC + 128 + o);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(aux_h + AUX_WC + 256 + o);
      accv[nb][4 * q + 0] = w0.x * dr.x + w1.x * dr.y + w2.x * dr.z;
      accv[nb][4 * q + 1] = w0.y * dr.x + w1.y * dr.y + w2.y * dr.z;
      accv[nb][4 * q + 2] = w0.z * dr.x + w1.z * dr.y + w2.z * dr.z;
      accv[nb][4 * q + 3] = w0.w * dr.x + w1.w * dr.y + w2.w * dr.z;
    }
  mask_pass<4>(accv, mk0);
  // ---- view layer, feature columns: df = Wv[:, :256]^T dzv      (16 k-groups = 4 stages over the 128 view units); every
  // k-group stores its own quad of dzv
  zero_acc<8>(accA);
  float* dzv_row_h = A.dzv + pc * 128 + 4 * h;
#pragma unroll
  for (int kg = 0; kg < 16; ++kg) {
    const f32x4 o = {accv[kg >> 2][4 * (kg & 3) + 0], accv[kg >> 2][4 * (kg & 3) + 1], accv[kg >> 2][4 * (kg & 3) + 2],
                     accv[kg >> 2][4 * (kg & 3) + 3]};
    *reinterpret_cast<f32x4*>(dzv_row_h + 8 * kg) = o;
    __builtin_amdgcn_sched_barrier(0);
    kgroup<8>(pipe, accA, kg, kg == 0, false, o.x, o.y, o.z, o.w);
  }
  // ---- feature layer + density head: dh7 = Wf^T df + w_alpha * dsigma (raw; its mask is applied by the layer below)
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 wa = *reinterpret_cast<const f32x4*>(aux_h + AUX_WA + 32 * nb + 8 * q);
      accB[nb][4 * q + 0] = wa.x * dr.w;
      accB[nb][4 * q + 1] = wa.y * dr.w;
      accB[nb][4 * q + 2] = wa.z * dr.w;
      accB[nb][4 * q + 3] = wa.w * dr.w;
    }
  const float* hrow = A.save_h + pc * 256 + 4 * h;
  float* zrow = A.dz + pc * 256 + 4 * h;
  const long long plane = A.Ppad * 256;
  f32x4 mk[4];
  bool pending = false;      // the view part above ended its last stage through kgroup's own end_stage_raw
  bwd_layer<false, false, 16>(pipe, accB, accA, mk, nullptr, A.df + pc * 256 + 4 * h, hrow + 7 * plane, mk0, pending);
  // ---- trunk: layer L consumes dh_L (masking it with h_L and storing dz_L) and produces dh_{L-1},  L = 7..1
  // (W_5: hidden columns only)
#pragma unroll 1
  for (int L = 7; L >= 3; L -= 2) {
    zero_acc<8>(accA);
    bwd_layer<true, false, 32>(pipe, accA, accB, mk, hrow + L * plane, zrow + L * plane, hrow + (L - 1) * plane, mk0, pending);
    zero_acc<8>(accB);
    bwd_layer<true, false, 32>(pipe, accB, accA, mk, hrow + (L - 1) * plane, zrow + (L - 1) * plane, hrow + (L - 2) * plane, mk0,
                               pending);
  }
  zero_acc<8>(accA);
  bwd_layer<true, true, 32>(pipe, accA, accB, mk, hrow + plane, zrow + plane, hrow, mk0, pending);
  // dz0 = dh0 * [h0 > 0]: no consumer in this kernel; the mask row was collected during the last layer
  mask_pass<8>(accA, mk0);
  store_acc<8>(zrow, accA);
#ifdef ANERF_EXP_STAGE_TIMING
  if (tid == 0 && tb) {
    unsigned long long* t = tb + 64 * 4 * 128 * 3 + 6 * (long long)blockIdx.x;
    t[0] = tt0; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr0; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg(63492); t[5] = __builtin_amdgcn_s_getreg(63508);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// input-gradient kernel (pose optimisation / frame codes only): gradients w.r.t. the ENCODED inputs, in stream
// column order, on the which=2 weight image:
//     dX'[p][432] = W0'^T dz0 + W5x'^T dz5          dU'[p][UW] = Wvu'^T dzv
// Output columns are produced 256 at a time (8 blocks); lane (m,h) receives exactly the columns whose forward
// B operands it generated (k-group 4*nb+q, half h), which is what k_encode_bwd consumes.
// ------------------------------------------------------------------------------------------------
struct BwdInArgs {
  const float* packed_i;
  const float* dz;     // [8][Ppad][256]
  const float* dzv;    // [Ppad][128]
  float* dx;           // [Ppad][432]
  float* du;           // [Ppad][UW]
  long long P, Ppad;
  int nstages, uw;
};

// One segment of the contraction: NKG k-groups (a multiple of 4 = whole stages) whose B operands are the float4 quads of a
// gradient row (quad kg at src_row_h + 8 kg), into `acc`.  Same discipline as bwd_layer: the stage's four quads `cur` were
// requested during the previous stage (complete at its barrier) and are pinned in front of the weight pipe's re-issue;
// every k-group requests one quad of the next stage (of `next_src_row_h`, the next segment's row, at the end) and writes SPK
// float4 of the PREVIOUS output group (`outv`, columns out_c0.. of out_row_h, width out_w) -- no load or store bursts.
// NBU: 32-column blocks of the output group that exist (a narrow last group skips the MFMAs of the others).
template <int NKG, int SPK, int NBU = 8>
__device__ __forceinline__ void bwd_in_segment(Pipe3B& pipe, f32x16 (&acc)[8], f32x4 (&cur)[4], const float* __restrict__ src_row_h,
                                               const float* __restrict__ next_src_row_h, const float (&outv)[128],
                                               float* __restrict__ out_row_h, int out_c0, int out_w, bool& pending) {
#pragma unroll
  for (int s = 0; s < NKG / 4; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(cur[i]));
    if (pending) pipe.stage_refill();
    pending = false;
    const float* nq = s < NKG / 4 - 1 ? src_row_h + 32 * (s + 1) : next_src_row_h;
    f32x4 nxt[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int kg = 4 * s + ks;
      if (nq) nxt[ks] = *reinterpret_cast<const f32x4*>(nq + 8 * ks);
      else nxt[ks] = cur[ks];
#pragma unroll
      for (int t = 0; t < SPK; ++t) {
        const int j = kg * SPK + t;            // quad of the previous group: block j >> 2, quad j & 3
        if (j < 32) {
          const int c = out_c0 + 8 * j;        // (+ 4h is in out_row_h)
          if (c < out_w)
            *reinterpret_cast<f32x4*>(out_row_h + c) = f32x4{outv[4 * j], outv[4 * j + 1], outv[4 * j + 2], outv[4 * j + 3]};
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      kgroup<8, Pipe3B, false, NBU>(pipe, acc, kg, false, false, cur[ks].x, cur[ks].y, cur[ks].z, cur[ks].w);
    }
    pipe.stage_rendezvous();
    pending = true;
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = nxt[i];
  }
}

// Round 2: on the 3-slot hidden-DMA pipe with the two-quarter fragment window (kgroup), operands and results one float4
// per k-group instead of 32-load / 32-store bursts per 256 columns (round 1: 2-slot pipe, bursts): 1.38 -> 1.24 ms per launch on the Mixamo step.
__global__ __launch_bounds__(256) void k_mlp_bwd_in(const BwdInArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe3B pipe;
  pipe.init(A.packed_i, smem, wave, lane, A.nstages);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const long long pc = p < A.P ? p : A.P - 1;      // tail lanes recompute the last valid row and rewrite it with identical values
  const float* z0 = A.dz + pc * 256 + 4 * h;
  const float* z5 = A.dz + (5 * A.Ppad + pc) * 256 + 4 * h;
  const float* zv = A.dzv + pc * 128 + 4 * h;
  float* dx = A.dx + pc * 432 + 4 * h;
  float* du = A.du + pc * A.uw + 4 * h;
  f32x4 cur[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const f32x4*>(z0 + 8 * i);
  pipe.begin();
  pipe.prime();
  f32x16 acc[8];
  float outv[128];
#pragma unroll
  for (int i = 0; i < 128; ++i) outv[i] = 0.f;
  bool pending = false;
  // ---- dX' columns 0..255: W0'^T dz0 + W5x'^T dz5
  zero_acc<8>(acc);
  bwd_in_segment<32, 0>(pipe, acc, cur, z0, z5, outv, dx, 0, 0, pending);
  bwd_in_segment<32, 0>(pipe, acc, cur, z5, z0, outv, dx, 0, 0, pending);
  take<8, false>(outv, acc);
  // ---- dX' columns 256..431 = 5.5 blocks: 6 of the group's 8 are multiplied (its first 32 k-groups write group 0)
  zero_acc<8>(acc);
  bwd_in_segment<32, 1, 6>(pipe, acc, cur, z0, z5, outv, dx, 0, 432, pending);
  bwd_in_segment<32, 0, 6>(pipe, acc, cur, z5, zv, outv, dx, 0, 0, pending);
  take<8, false>(outv, acc);
  // ---- dU' = Wvu'^T dzv, 256 columns at a time; group g's 16 k-groups write the group before it.  The last group is narrow in
  // every configuration (648 / 664 = 2 x 256 + 136 / 152: 5 blocks; 72: 3 blocks)
  const int ngu = (A.uw + 255) / 256;
  const int nbl = (A.uw - 256 * (ngu - 1) + 31) / 32;        // blocks of the last group
  auto u_group = [&](int g, const float* next_row, float* out_row, int out_c0, int out_w) {
    zero_acc<8>(acc);
    if (g + 1 < ngu || nbl > 5) bwd_in_segment<16, 2, 8>(pipe, acc, cur, zv, next_row, outv, out_row, out_c0, out_w, pending);
    else if (nbl > 3) bwd_in_segment<16, 2, 5>(pipe, acc, cur, zv, next_row, outv, out_row, out_c0, out_w, pending);
    else bwd_in_segment<16, 2, 3>(pipe, acc, cur, zv, next_row, outv, out_row, out_c0, out_w, pending);
    take<8, false>(outv, acc);
  };
  u_group(0, ngu > 1 ? zv : nullptr, dx, 256, 432);
#pragma unroll 1
  for (int g = 1; g < ngu; ++g) u_group(g, g + 1 < ngu ? zv : nullptr, du, 256 * (g - 1), A.uw);
  // the last group's columns
  {
    const int c0 = 256 * (ngu - 1);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int c = c0 + 8 * j;
      if (c < A.uw) *reinterpret_cast<f32x4*>(du + c) = f32x4{outv[4 * j], outv[4 * j + 1], outv[4 * j + 2], outv[4 * j + 3]};
    }
  }
}

int mlp_bwd_in_entry(const float* packed_i, const float* dz, const float* dzv, float* dx, float* du, long long P,
                     long long Ppad, int nstages, int uw, hipStream_t st) {
  BwdInArgs b;
  b.packed_i = packed_i; b.dz = dz; b.dzv = dzv; b.dx = dx; b.du = du; b.P = P; b.Ppad = Ppad; b.nstages = nstages; b.uw = uw;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = RING_SLOTS * STAGE_BYTES;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_in), (int)lds, &lds_set);
  hipLaunchKernelGGL(k_mlp_bwd_in, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd_in");
}

// ------------------------------------------------------------------------------------------------
// Round 6: k_mlp_bwd_in with the backward of the fused ENCODING as its column-group epilogue.
// The lane (sample m, half h) that receives a 256-column group of dX' / dU' is the lane whose forward twin generated those
// columns -- it owns joints j = 8 G + 4 h + t -- so it can apply the hand-derived backward of transform / norms / gates /
// sin-cos (k_encode_bwd's arithmetic) to the group while the values are still in registers, and keep only
//     dv[12] (distance) + dr[36] (unit bone direction) + de[36] (unit ray direction in bone space)      = 84 partial sums
// per lane across the five groups.  dX' / dU' (4.4 KB per sample written by this kernel and read again by k_encode_bwd: 1.2 GB
// per launch at 3072 rays) are never stored; what leaves the kernel is dY / dQ [P][72] (576 B per sample) for k_pose_reduce and
// the two quads of frame-code columns of dU' that k_code_rowsum reads.
// Register budget (k_mlp_bwd_in compiles to 228 VGPR + 128 AGPR, 0 scratch): the 84 sums live across the MFMA segments, where the
// allocator parks them in the 128 unused AGPRs (v_accvgpr_write / _read at the epilogue boundaries); an epilogue itself runs
// with the accumulators dead and needs, per joint quad G, ~50 transient registers (4 joints' bone-space state + the sin/cos
// chain) next to the 128 values of the group.  sin/cos: one exact evaluation per chain anchor (distance: bands 0 and 4, as the
// forward kernel; unit-vector components: band 0 without range reduction) and double-angle steps in between.
struct EncArgs {
  const float* rays; const float* z; const float* skts; const float* cut_v; const float* cut_d; const float* pnoise; const float* tau_dev;
  float* dY; float* dQ;
  long long skt_stride;
  int ray_stride, S, gate_bones, n_rays;
  float tau_v, tau_d;
};
// LDS behind the weight ring: rows 0..2 of the bone matrices of the <= 18 rays a 128-sample tile touches (as the forward kernel
// stages them) + the 2 x 24 cutoffs -- each of the five epilogues re-derives its joints' bone-space state from them, and at one wave
// per SIMD a dependent global load in front of every joint quad (15 per tile) was most of the epilogues' time
constexpr int ENC_BONES_OFF = RING_SLOTS * STAGE_BYTES;
constexpr int ENC_CUT_OFF = ENC_BONES_OFF + MAX_TILE_RAYS * 72 * 16;
constexpr int ENC_LDS_BYTES = ENC_CUT_OFF + 256;

// (no __restrict__ on the pointers of the epilogue functions: for a noalias read-only pointer the compiler may -- and did -- hoist all
// 36 bone-matrix loads of all five epilogues over the `asm volatile("" ::: "memory")` fences to the top of the kernel and spill them)
struct JointQuad { float v[4], iv[4], rh[12], e[12], iq[4]; };      // distance, 1 / distance, unit direction, unit ray direction, 1 / |R d|
// 1 / sqrt(x) from v_rsq_f32 (1 ulp) + one Newton step: 5 VALU against ~20 for sqrtf + an IEEE division -- joint_quad runs 15 times per tile
__device__ __forceinline__ float rsqrt_nr(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  return r * fmaf(-0.5f * x * r, r, 1.5f);
}
__device__ __forceinline__ void joint_quad(const f32x4* sk, int G, int h, float x0, float x1, float x2, float d0, float d1,
                                           float d2, JointQuad& q) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = 8 * G + 4 * h + t;
    const f32x4 r0 = sk[3 * j], r1 = sk[3 * j + 1], r2 = sk[3 * j + 2];      // (LDS: this ray's [24][3] float4 rows)
    const float y0 = r0.x * x0 + r0.y * x1 + r0.z * x2 + r0.w;
    const float y1 = r1.x * x0 + r1.y * x1 + r1.z * x2 + r1.w;
    const float y2 = r2.x * x0 + r2.y * x1 + r2.z * x2 + r2.w;
    const float n2 = fmaxf(y0 * y0 + y1 * y1 + y2 * y2, 1e-24f);      // (|y| clamped at 1e-12 as F.normalize / the forward do)
    const float inv = rsqrt_nr(n2);
    q.v[t] = n2 * inv;
    q.iv[t] = inv;
    q.rh[3 * t] = y0 * inv; q.rh[3 * t + 1] = y1 * inv; q.rh[3 * t + 2] = y2 * inv;
    const float q0 = r0.x * d0 + r0.y * d1 + r0.z * d2, q1 = r1.x * d0 + r1.y * d1 + r1.z * d2,
                q2 = r2.x * d0 + r2.y * d1 + r2.z * d2;
    const float qi = rsqrt_nr(fmaxf(q0 * q0 + q1 * q1 + q2 * q2, 1e-24f));
    q.iq[t] = qi;
    q.e[3 * t] = q0 * qi; q.e[3 * t + 1] = q1 * qi; q.e[3 * t + 2] = q2 * qi;
  }
}

// value of local k-group X (0..31), slot T of the finished group: accumulator block X >> 2, register 4 (X & 3) + T (take<>'s order)
#define OV(X, T) acc[(X) >> 2][4 * ((X) & 3) + (T)]
template <int KG0, int NK>
__device__ __forceinline__ void enc_x_group(const f32x16 (&acc)[8], float (&dv)[12], const f32x4* sk, int h, float x0,
                                            float x1, float x2, float tau_v, const float* cut_v, int gate_bones,
                                            float* dY_row, int& anchor) {
  auto has = [](int kg) { return kg >= KG0 && kg < KG0 + NK; };
#pragma unroll
  for (int G = 0; G < 3; ++G) {
    // one joint quad at a time: without the fences the scheduler hoists all 36 bone-matrix loads and the three quads' transcendental
    // chains to the top of the epilogue (latency hiding it does not need here) and the kernel spills
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    // `anchor` is an opaque zero that every region's loads are addressed through and that the previous region's results are tied
    // to (asm below): fences alone order memory operations only -- the optimiser had gathered all three regions' loads in front of
    // all three regions' arithmetic (and the scheduler then moved them up into the MFMA segment, where they spilled)
    asm volatile("" : "+v"(anchor) : "v"(OV(G < NK ? G : 0, 0)));
    JointQuad q;
    joint_quad(sk + anchor, G, h, x0, x1, x2, 0.f, 0.f, 0.f, q);
    float wv[4], wvp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wv[t] = cutoff_gate(tau_v, q.v[t], (cut_v + anchor)[8 * G + 4 * h + t]);
      wvp[t] = -tau_v * wv[t] * (1.f - wv[t]);
    }
    if (has(G)) {
#pragma unroll
      for (int t = 0; t < 4; ++t) dv[4 * G + t] += OV(G - KG0, t) * (wv[t] + q.v[t] * wvp[t]);
    }
    // which bands of this joint quad lie in the group: chain anchors at f = 0 and f = 4
    int fmin = 99, fmax = -1;
#pragma unroll
    for (int f = 0; f < 7; ++f)
      if (has(3 + 6 * f + G) || has(6 + 6 * f + G)) { fmin = f < fmin ? f : fmin; fmax = f; }
    float s[4], c[4];
#pragma unroll
    for (int f = 0; f < 7; ++f) {
      if (f > fmax || fmax < 0) continue;
      const int anchor = fmin >= 4 ? 4 : 0;
      if (f < anchor) continue;
      const float F = (float)(1 << f);
      if (f == 0 || f == 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) sincos_f32(q.v[t] * F, s[t], c[t]);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float s2 = 2.f * s[t] * c[t], c2 = fmaf(-2.f * s[t], s[t], 1.f);
          s[t] = s2; c[t] = c2;
        }
      }
      const int ks = 3 + 6 * f + G, kc = 6 + 6 * f + G;
      if (has(ks)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dv[4 * G + t] += OV(ks - KG0, t) * (F * c[t] * wv[t] + s[t] * wvp[t]);
      }
      if (has(kc)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) dv[4 * G + t] += OV(kc - KG0, t) * (-F * s[t] * wv[t] + c[t] * wvp[t]);
      }
    }
    if (has(45 + 3 * G)) {      // (the three bone-direction k-groups of a joint quad lie in the same group)
      float oy[12];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        // flat index 4 g + t' of the quad's 12 direction values = 3 (joint in quad) + component
        float d_r[3];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int i = 3 * t + cc;
          d_r[cc] = OV(45 + 3 * G + i / 4 - KG0, (i & 3));
        }
        float dot = d_r[0] * q.rh[3 * t] + d_r[1] * q.rh[3 * t + 1] + d_r[2] * q.rh[3 * t + 2];
        if (gate_bones) {   // cutoff_bones: the network saw r * w(v) -- d r = w * d(r w); the gate's slope adds (d(r w) . r) w' to d v
          dv[4 * G + t] += dot * wvp[t];
          dot *= wv[t];
          d_r[0] *= wv[t]; d_r[1] *= wv[t]; d_r[2] *= wv[t];
        }
        const float iv = q.iv[t];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) oy[3 * t + cc] = (d_r[cc] - dot * q.rh[3 * t + cc]) * iv;
      }
#ifndef ANERF_EXP_ENC_NOSTORE   // ablation build only (tools/ablate.sh): what do the parked partial sums cost?  results are wrong
#pragma unroll
      for (int w4 = 0; w4 < 3; ++w4)
        *reinterpret_cast<f32x4*>(dY_row + 3 * (8 * G + 4 * h) + 4 * w4) = f32x4{oy[4 * w4], oy[4 * w4 + 1], oy[4 * w4 + 2], oy[4 * w4 + 3]};
#else
      asm volatile("" :: "v"(oy[0]), "v"(oy[5]), "v"(oy[11]));
#endif
    }
    asm volatile("" : "+v"(anchor) : "v"(dv[4 * G]), "v"(dv[4 * G + 1]), "v"(dv[4 * G + 2]), "v"(dv[4 * G + 3]));
  }
}

// dU' k-groups KU0 .. KU0 + NK - 1 (stream order: 9 band + 3 G + g for the 1 + 2 LD direction bands, then CODE / 8 frame-code k-groups).
// The raw sums d e of a joint quad travel between the groups in the sample's dQ row (FIRST writes, the others add); LAST finishes
// the quad: dQ = (de - (de . e) e) / |q|, dY = its parked direction share + dv r.
template <int KU0, int NK, int LD, int CODE, bool FIRST, bool LAST>
__device__ __forceinline__ void enc_u_group(const f32x16 (&acc)[8], float (&dv)[12], const f32x4* sk, int h, float x0,
                                            float x1, float x2, float d0, float d1, float d2, float tau_d,
                                            const float* cut_d, float* du_row_h, float* dY_row,
                                            float* dQ_row, int& anchor) {
  auto has = [](int ku) { return ku >= KU0 && ku < KU0 + NK; };
#pragma unroll
  for (int G = 0; G < 3; ++G) {
    __builtin_amdgcn_sched_barrier(0);     // (one joint quad at a time, see enc_x_group)
    asm volatile("" ::: "memory");
    asm volatile("" : "+v"(anchor) : "v"(OV(0, 0)));
    float de[12];
    if (FIRST) {
#pragma unroll
      for (int i = 0; i < 12; ++i) de[i] = 0.f;
    } else {
#pragma unroll
      for (int w4 = 0; w4 < 3; ++w4) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(dQ_row + anchor + 3 * (8 * G + 4 * h) + 4 * w4);
        de[4 * w4] = pv.x; de[4 * w4 + 1] = pv.y; de[4 * w4 + 2] = pv.z; de[4 * w4 + 3] = pv.w;
      }
    }
    f32x4 py[3];
    if (LAST) {
#pragma unroll
      for (int w4 = 0; w4 < 3; ++w4) py[w4] = *reinterpret_cast<const f32x4*>(dY_row + anchor + 3 * (8 * G + 4 * h) + 4 * w4);
    }
    JointQuad q;
    joint_quad(sk + anchor, G, h, x0, x1, x2, d0, d1, d2, q);
    float wd[4], wdp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wd[t] = cutoff_gate(tau_d, q.v[t], (cut_d + anchor)[8 * G + 4 * h + t]);
      wdp[t] = -tau_d * wd[t] * (1.f - wd[t]);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int k0 = 3 * G + g;
      if (has(k0)) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int i = 4 * g + t, a = i / 3;
          const float gt = OV(k0 - KU0, t);
          de[i] += gt * wd[a];
          dv[4 * G + a] += gt * q.e[i] * wdp[a];
        }
      }
      int fmax = -1;
#pragma unroll
      for (int f = 0; f < LD; ++f)
        if (has(9 * (1 + 2 * f) + k0) || has(9 * (2 + 2 * f) + k0)) fmax = f;
      float s[4], c[4];
#pragma unroll
      for (int f = 0; f < LD; ++f) {
        if (f > fmax) continue;
        const float F = (float)(1 << f);
        if (f == 0) {
#pragma unroll
          for (int t = 0; t < 4; ++t) sincos_unit_f32(q.e[4 * g + t], s[t], c[t]);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float s2 = 2.f * s[t] * c[t], c2 = fmaf(-2.f * s[t], s[t], 1.f);
            s[t] = s2; c[t] = c2;
          }
        }
        const int ks = 9 * (1 + 2 * f) + k0, kc = 9 * (2 + 2 * f) + k0;
        if (has(ks)) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int i = 4 * g + t, a = i / 3;
            const float gs = OV(ks - KU0, t);
            de[i] += gs * c[t] * F * wd[a];
            dv[4 * G + a] += gs * s[t] * wdp[a];
          }
        }
        if (has(kc)) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int i = 4 * g + t, a = i / 3;
            const float gc = OV(kc - KU0, t);
            de[i] -= gc * s[t] * F * wd[a];
            dv[4 * G + a] += gc * c[t] * wdp[a];
          }
        }
      }
    }
    if (!LAST) {
#ifndef ANERF_EXP_ENC_NOSTORE
#pragma unroll
      for (int w4 = 0; w4 < 3; ++w4)
        *reinterpret_cast<f32x4*>(dQ_row + 3 * (8 * G + 4 * h) + 4 * w4) = f32x4{de[4 * w4], de[4 * w4 + 1], de[4 * w4 + 2], de[4 * w4 + 3]};
#else
      asm volatile("" :: "v"(de[0]), "v"(de[5]), "v"(de[11]));
#endif
    } else {   // through the norms: q -> e, and the distance part of y -> (v, r)   (k_encode_bwd's closing block)
      float oy[12], oq[12];
      const float pyf[12] = {py[0].x, py[0].y, py[0].z, py[0].w, py[1].x, py[1].y, py[1].z, py[1].w, py[2].x, py[2].y, py[2].z, py[2].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float dote = de[3 * t] * q.e[3 * t] + de[3 * t + 1] * q.e[3 * t + 1] + de[3 * t + 2] * q.e[3 * t + 2];
        const float iq = q.iq[t];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          oy[3 * t + cc] = fmaf(dv[4 * G + t], q.rh[3 * t + cc], pyf[3 * t + cc]);
          oq[3 * t + cc] = (de[3 * t + cc] - dote * q.e[3 * t + cc]) * iq;
        }
      }
#pragma unroll
      for (int w4 = 0; w4 < 3; ++w4) {
        *reinterpret_cast<f32x4*>(dY_row + 3 * (8 * G + 4 * h) + 4 * w4) = f32x4{oy[4 * w4], oy[4 * w4 + 1], oy[4 * w4 + 2], oy[4 * w4 + 3]};
        *reinterpret_cast<f32x4*>(dQ_row + 3 * (8 * G + 4 * h) + 4 * w4) = f32x4{oq[4 * w4], oq[4 * w4 + 1], oq[4 * w4 + 2], oq[4 * w4 + 3]};
      }
    }
    asm volatile("" : "+v"(anchor) : "v"(dv[4 * G]), "v"(dv[4 * G + 1]), "v"(dv[4 * G + 2]), "v"(dv[4 * G + 3]), "v"(de[0]), "v"(de[11]));
  }
  // frame-code columns (the last CODE / 8 k-groups of dU'): k_code_rowsum reads them from du
#pragma unroll
  for (int j = 0; j < CODE / 8; ++j) {
    const int ku = 9 * (1 + 2 * LD) + j;
    if (has(ku)) *reinterpret_cast<f32x4*>(du_row_h + 8 * ku) = f32x4{OV(ku - KU0, 0), OV(ku - KU0, 1), OV(ku - KU0, 2), OV(ku - KU0, 3)};
  }
}

template <int LD, int CODE>
__global__ __launch_bounds__(256) void k_mlp_bwd_in_enc(const BwdInArgs A, const EncArgs E) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe3B pipe;
  pipe.init(A.packed_i, smem, wave, lane, A.nstages);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const long long pc = p < A.P ? p : A.P - 1;      // tail lanes recompute the last valid row and rewrite it with identical values
  const float* z0 = A.dz + pc * 256 + 4 * h;
  const float* z5 = A.dz + (5 * A.Ppad + pc) * 256 + 4 * h;
  const float* zv = A.dzv + pc * 128 + 4 * h;
  constexpr int UW = 72 * (1 + 2 * LD) + CODE, NKU = UW / 8;
  float* du = A.du + pc * UW + 4 * h;
  // the sample's geometry (k_encode_bwd's prologue)
  const long long ray = div_samples(pc, E.S);
  const float* rp = E.rays + ray * E.ray_stride;
  const float zz = E.z[pc];
  const float d0 = rp[3], d1 = rp[4], d2 = rp[5];
  float x0 = fmaf(d0, zz, rp[0]), x1 = fmaf(d1, zz, rp[1]), x2 = fmaf(d2, zz, rp[2]);
  if (E.pnoise) {
    x0 += E.pnoise[3 * pc]; x1 += E.pnoise[3 * pc + 1]; x2 += E.pnoise[3 * pc + 2];
  }
  // stage the bone matrices of this tile's rays + the cutoffs (visible behind pipe.begin()'s barrier)
  const long long tile_p0 = (long long)blockIdx.x * TILE;
  const long long ray0 = div_samples(tile_p0, E.S);
  long long ray1 = div_samples(tile_p0 + TILE - 1, E.S);
  if (ray1 > E.n_rays - 1) ray1 = E.n_rays - 1;
  {
    f32x4* bw = reinterpret_cast<f32x4*>(smem + ENC_BONES_OFF);
    const int n_stage = (int)(ray1 - ray0 + 1) * 72;
    for (int i = tid; i < n_stage; i += 256) {
      const int ri = i / 72, rem = i - ri * 72, j = rem / 3, row = rem - 3 * j;
      bw[i] = *reinterpret_cast<const f32x4*>(E.skts + (ray0 + ri) * E.skt_stride + j * 16 + row * 4);
    }
    float* cw = reinterpret_cast<float*>(smem + ENC_CUT_OFF);
    if (tid < 24) cw[tid] = E.cut_v[tid];
    else if (tid < 48) cw[tid] = E.cut_d[tid - 24];
  }
  const f32x4* sk = reinterpret_cast<const f32x4*>(smem + ENC_BONES_OFF) + (int)(ray - ray0) * 72;
  const float* cutv_l = reinterpret_cast<const float*>(smem + ENC_CUT_OFF);
  const float* cutd_l = cutv_l + 24;
  const float tau_v = E.tau_dev ? E.tau_dev[0] : E.tau_v, tau_d = E.tau_dev ? E.tau_dev[1] : E.tau_d;
  int anchor = 0;    // see enc_x_group
  float dv[12];      // the one set of partial sums that stays in registers; the direction sums travel in the dY / dQ rows
#pragma unroll
  for (int i = 0; i < 12; ++i) dv[i] = 0.f;
  float* dYr = E.dY + pc * 72;
  float* dQr = E.dQ + pc * 72;
  f32x4 cur[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cur[i] = *reinterpret_cast<const f32x4*>(z0 + 8 * i);
  pipe.begin();
  pipe.prime();
  f32x16 acc[8];
  const float outv[128] = {};     // (bwd_in_segment's store operand: unused with SPK = 0)
  bool pending = false;
  float* none = nullptr;
  // ---- dX' columns 0..255 (k-groups 0..31)
  zero_acc<8>(acc);
  bwd_in_segment<32, 0>(pipe, acc, cur, z0, z5, outv, none, 0, 0, pending);
  bwd_in_segment<32, 0>(pipe, acc, cur, z5, z0, outv, none, 0, 0, pending);
  __builtin_amdgcn_sched_barrier(0);     // the epilogue reads the accumulators in place (a 128-register copy would spill)
  enc_x_group<0, 32>(acc, dv, sk, h, x0, x1, x2, tau_v, cutv_l, E.gate_bones, dYr, anchor);
  // ---- dX' columns 256..431 (k-groups 32..53)
  zero_acc<8>(acc);
  bwd_in_segment<32, 0, 6>(pipe, acc, cur, z0, z5, outv, none, 0, 0, pending);
  bwd_in_segment<32, 0, 6>(pipe, acc, cur, z5, zv, outv, none, 0, 0, pending);
  __builtin_amdgcn_sched_barrier(0);     // the epilogue reads the accumulators in place (a 128-register copy would spill)
  enc_x_group<32, 22>(acc, dv, sk, h, x0, x1, x2, tau_v, cutv_l, E.gate_bones, dYr, anchor);
  // ---- dU' = Wvu'^T dzv, 256 columns (32 k-groups) at a time
  if constexpr (NKU <= 32) {          // multires_views = 0: one narrow group (72 columns: 3 blocks)
    zero_acc<8>(acc);
    bwd_in_segment<16, 0, 3>(pipe, acc, cur, zv, nullptr, outv, none, 0, 0, pending);
    __builtin_amdgcn_sched_barrier(0);     // the epilogue reads the accumulators in place (a 128-register copy would spill)
    enc_u_group<0, NKU, LD, CODE, true, true>(acc, dv, sk, h, x0, x1, x2, d0, d1, d2, tau_d, cutd_l, du, dYr, dQr, anchor);
  } else {
    zero_acc<8>(acc);
    bwd_in_segment<16, 0, 8>(pipe, acc, cur, zv, zv, outv, none, 0, 0, pending);
    __builtin_amdgcn_sched_barrier(0);     // the epilogue reads the accumulators in place (a 128-register copy would spill)
    enc_u_group<0, 32, LD, CODE, true, false>(acc, dv, sk, h, x0, x1, x2, d0, d1, d2, tau_d, cutd_l, du, dYr, dQr, anchor);
    zero_acc<8>(acc);
    bwd_in_segment<16, 0, 8>(pipe, acc, cur, zv, zv, outv, none, 0, 0, pending);
    __builtin_amdgcn_sched_barrier(0);     // the epilogue reads the accumulators in place (a 128-register copy would spill)
    enc_u_group<32, 32, LD, CODE, false, false>(acc, dv, sk, h, x0, x1, x2, d0, d1, d2, tau_d, cutd_l, du, dYr, dQr, anchor);
    zero_acc<8>(acc);
    bwd_in_segment<16, 0, 5>(pipe, acc, cur, zv, nullptr, outv, none, 0, 0, pending);     // 136 / 152 columns: 5 blocks
    __builtin_amdgcn_sched_barrier(0);     // the epilogue reads the accumulators in place (a 128-register copy would spill)
    enc_u_group<64, NKU - 64, LD, CODE, false, true>(acc, dv, sk, h, x0, x1, x2, d0, d1, d2, tau_d, cutd_l, du, dYr, dQr, anchor);
  }
}

#undef OV

int mlp_bwd_in_enc_entry(int ld, int code, const float* packed_i, const float* dz, const float* dzv, float* du, long long P, long long Ppad,
                         int nstages, const float* rays, int ray_stride, const float* z, const float* skts, long long skt_stride,
                         float tau_v, float tau_d, const float* cut_v, const float* cut_d, int S, float* dY, float* dQ, const float* pnoise,
                         int gate_bones, const float* tau_dev, int n_rays, hipStream_t st) {
  BwdInArgs b;
  b.packed_i = packed_i; b.dz = dz; b.dzv = dzv; b.dx = nullptr; b.du = du; b.P = P; b.Ppad = Ppad; b.nstages = nstages;
  b.uw = 72 * (1 + 2 * ld) + code;
  EncArgs e;
  e.rays = rays; e.z = z; e.skts = skts; e.cut_v = cut_v; e.cut_d = cut_d; e.pnoise = pnoise; e.tau_dev = tau_dev; e.dY = dY; e.dQ = dQ;
  e.skt_stride = skt_stride; e.ray_stride = ray_stride; e.S = S; e.gate_bones = gate_bones; e.tau_v = tau_v; e.tau_d = tau_d;
  e.n_rays = n_rays;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = ENC_LDS_BYTES;
  static unsigned long long lds_set[3] = {};   // per-device bits, see ensure_dynamic_lds
  if (ld == 4 && code == 16) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_in_enc<4, 16>), (int)lds, &lds_set[0]);
    hipLaunchKernelGGL((k_mlp_bwd_in_enc<4, 16>), dim3((unsigned)nblk), dim3(256), lds, st, b, e);
  } else if (ld == 4 && code == 0) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_in_enc<4, 0>), (int)lds, &lds_set[1]);
    hipLaunchKernelGGL((k_mlp_bwd_in_enc<4, 0>), dim3((unsigned)nblk), dim3(256), lds, st, b, e);
  } else if (ld == 0 && code == 0) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_in_enc<0, 0>), (int)lds, &lds_set[2]);
    hipLaunchKernelGGL((k_mlp_bwd_in_enc<0, 0>), dim3((unsigned)nblk), dim3(256), lds, st, b, e);
  } else {
    return set_error(ANERF_E_CONFIG, "mlp_bwd_in_enc: multires_views must be 0 or 4, framecode_ch 0 or 16");
  }
  return check_launch("k_mlp_bwd_in_enc");
}

int mlp_bwd_entry(const float* packed_t, const float* aux, const float* draw, const AnerfSaved* sv, float* dz, float* df,
                  float* dzv, long long P, int nstages, hipStream_t st) {
  BwdArgs b;
  b.packed_t = packed_t; b.aux = aux; b.draw = draw; b.save_h = sv->h; b.save_g = sv->g;
  b.dz = dz; b.df = df; b.dzv = dzv; b.P = P; b.Ppad = sv->p_pad; b.nstages = nstages;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = LDS_BONES_OFF;   // 3-slot weight ring + the aux copy (no bone staging in the backward)
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd), (int)lds, &lds_set);
  hipLaunchKernelGGL(k_mlp_bwd, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd");
}

}  // namespace anerf

#ifdef ANERF_EXP_STAGE_TIMING
extern "C" void anerf_debug_set_bwd_timing_buf(unsigned long long* p) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(anerf::g_bwd_tbuf), &p, sizeof(p));
}
#endif
