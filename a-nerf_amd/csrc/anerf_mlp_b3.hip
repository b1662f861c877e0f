// anerf_mlp_b3.hip -- "bf16x3" variant of the fused encode+MLP forward kernel (render / eval path) for gfx950.
//
// Same dataflow as k_mlp_fwd (anerf_mlp.hip): 128-sample tile, 4 waves x 32 samples, layers computed transposed,
// activations register-resident across all layers, weights streamed L2 -> LDS through the 3-slot ring.  The GEMMs
// run on the bf16 matrix cores with error-compensated operands: every fp32 value x is split x = hi + lo (two
// bf16, round-to-nearest; |x - hi - lo| <= 2^-18 |x|) and each product uses three MFMAs
//     W x ~= Whi*Xhi + Whi*Xlo + Wlo*Xhi          (v_mfma_f32_32x32x16_bf16, fp32 accumulation)
// i.e. ~2^-17 relative error per product instead of bf16's 2^-9: enough for the path's 1e-4 RGB bar (tests), at
// ~5x the fp32-MFMA rate (3 x 32 cycles per K=16 instead of 8 x 64).  Weights are split once by k_pack_b3; the
// activations are split when a k-step's B operands are formed (8 values per lane: 4 v_cvt_pk + 8 sub).
// A k-step = 16 contraction indices: lane half h supplies 8; for hidden layers these are accumulator registers
// 8s..8s+7 of block nbk (features 32nbk+16s + (e&3)+8(e>>2)+4h), so as in the fp32 kernel nothing is permuted.
// Bias, accumulation, ReLU, heads and the encoding stay fp32.
// Reference ops replaced: identical to anerf_mlp.hip (core/encoders.py, core/cutoff_embedder.py,
// core/networks/nerf.py:94-148).
#include <type_traits>
#include "anerf_fwd_common.h"
#include "anerf_split.h"

namespace anerf {
#ifdef ANERF_EXP_STAGE_TIMING
extern float* g_tile_timing_buf;   // anerf_mlp.hip (debug build only)
#endif

// Compiler-scheduled k-step (training forward and the backward kernels: with their extra live state -- saved-row
// stores, 128 ReLU-mask values -- the two fragment buffers of the pipelined form below cost more in spills than they gain).
// NBU <= NB: only the first NBU blocks are read and multiplied (narrow last column group of k_mlp_bwd_in_b3; the stage keeps its
// NB-block layout).
template <int NB, int NBU = NB>
__device__ __forceinline__ void kstep_plain(Pipe3& pipe, f32x16 (&acc)[NB], int ks, bool last, const BOp& b) {
  constexpr int KPS = STAGE_FRAGS / (2 * NB);   // k-steps per 32-fragment stage
  constexpr int NPF = NB < 4 ? NB : 4;          // blocks whose fragments are prefetched across the stage barrier
  const int kk = ks % KPS;
  bf16x8 ah[NB], al[NB];
#pragma unroll
  for (int nb = 0; nb < NBU; ++nb) {
    if (kk == 0 && ks != 0 && nb < NPF) {
      ah[nb] = __builtin_bit_cast(bf16x8, pipe.pref[2 * nb]);
      al[nb] = __builtin_bit_cast(bf16x8, pipe.pref[2 * nb + 1]);
    } else {
      ah[nb] = *reinterpret_cast<const bf16x8*>(pipe.smem + pipe.cur + ((kk * NB + nb) * 2) * FRAG_BYTES);
      al[nb] = *reinterpret_cast<const bf16x8*>(pipe.smem + pipe.cur + ((kk * NB + nb) * 2 + 1) * FRAG_BYTES);
    }
  }
  if (kk == KPS - 1 && !last) {
#pragma unroll
    for (int i = 0; i < 2 * NPF; ++i) pipe.pref[i] = *reinterpret_cast<const f32x4*>(pipe.smem + pipe.nxt + i * FRAG_BYTES);
  }
#pragma unroll
  for (int nb = 0; nb < NBU; ++nb) {
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[nb], b.hi, acc[nb], 0, 0, 0);
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nb], b.lo, acc[nb], 0, 0, 0);
    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nb], b.hi, acc[nb], 0, 0, 0);
  }
  if (kk == KPS - 1 || last) pipe.end_stage();
}

// One k-step (16 contraction indices) against NB 32-row feature blocks: per block a (hi, lo) fragment pair from
// LDS and three MFMAs.  ks: k-step index relative to the segment start; last: final k-step of the segment.
//
// The fragment reads are software-pipelined by hand.  Blocks are consumed in groups of 4 (8 fragments = 32 registers,
// 12 MFMAs = 384 matrix cycles); pipe.pref always holds the FIRST group of the NEXT k-step (same stage, or -- before the
// stage barrier -- the first fragments of the next ring slot), and a group's 8 ds_read_b128 are issued between the head
// (4 MFMAs) and the tail (8 MFMAs) of the group before it, so the s_waitcnt lgkmcnt(0) the compiler puts in front of a
// group's first MFMA only ever waits for reads that had 256 matrix cycles to land.  Left to itself the compiler emits
// "2 reads, s_waitcnt lgkmcnt(0), 3 MFMAs" per block through one 8-register buffer (the kernel is at the register
// limit), which exposes the LDS latency behind every block.  The sched_barrier fences pin the order of MFMA and DS
// instructions only (mask 0x6: VALU / SALU -- the next operand split, the encoding -- may move across them).
template <int NB, bool PIPE = false, int NBU = NB>
__device__ __forceinline__ void kstep(Pipe3& pipe, f32x16 (&acc)[NB], int ks, bool last, const BOp& b) {
  if constexpr (!PIPE) {
    kstep_plain<NB, NBU>(pipe, acc, ks, last, b);
    return;
  }
  static_assert(NBU == NB || !PIPE, "the hand-pipelined form multiplies every block");
  static_assert(NB == 8 || NB == 4, "groups of 4 feature blocks");
  constexpr int KPS = STAGE_FRAGS / (2 * NB);   // k-steps per 32-fragment stage
  const int kk = ks % KPS;
  auto read_group = [&](bf16x8 (&f)[8], unsigned slot_off, int kq, int nb0) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      f[i] = *reinterpret_cast<const bf16x8*>(pipe.smem + slot_off + ((kq * NB + nb0) * 2 + i) * FRAG_BYTES);
  };
  auto head = [&](const bf16x8 (&f)[8], int nb0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[nb0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 * i + 1], b.hi, acc[nb0 + i], 0, 0, 0);
  };
  auto tail = [&](const bf16x8 (&f)[8], int nb0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[nb0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 * i], b.lo, acc[nb0 + i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[nb0 + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2 * i], b.hi, acc[nb0 + i], 0, 0, 0);
  };
  auto lookahead = [&]() {   // first group of the next k-step -> pipe.pref
    if (last) return;
    bf16x8 n[8];
    if (kk == KPS - 1) read_group(n, pipe.nxt, 0, 0);
    else read_group(n, pipe.cur, kk + 1, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) pipe.pref[i] = __builtin_bit_cast(f32x4, n[i]);
  };
  bf16x8 ga[8];
  if (ks == 0) {            // segment start: nothing was looked ahead
    read_group(ga, pipe.cur, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) ga[i] = __builtin_bit_cast(bf16x8, pipe.pref[i]);
  }
  __builtin_amdgcn_sched_barrier(0x6);
  head(ga, 0);
  __builtin_amdgcn_sched_barrier(0x6);
  if constexpr (NB == 8) {
    bf16x8 gb[8];
    read_group(gb, pipe.cur, kk, 4);
    __builtin_amdgcn_sched_barrier(0x6);
    tail(ga, 0);
    __builtin_amdgcn_sched_barrier(0x6);
    head(gb, 4);
    __builtin_amdgcn_sched_barrier(0x6);
    lookahead();
    __builtin_amdgcn_sched_barrier(0x6);
    tail(gb, 4);
  } else {
    lookahead();
    __builtin_amdgcn_sched_barrier(0x6);
    tail(ga, 0);
  }
  if (kk == KPS - 1 || last) pipe.end_stage();
}

// 16 k-steps whose operands are the previous layer's 256 outputs (bias inside, ReLU already applied in place)
template <int NB, int KS0, bool PIPE = false>
__device__ __forceinline__ void hidden_part_b3(Pipe3& pipe, f32x16 (&acc)[NB], const f32x16 (&prev)[8], bool last) {
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int nbk = ks >> 1, r0 = 8 * (ks & 1);
    const BOp b = split8(prev[nbk][r0], prev[nbk][r0 + 1], prev[nbk][r0 + 2], prev[nbk][r0 + 3], prev[nbk][r0 + 4],
                         prev[nbk][r0 + 5], prev[nbk][r0 + 6], prev[nbk][r0 + 7]);
    kstep<NB, PIPE>(pipe, acc, KS0 + ks, last && ks == 15, b);
  }
}

// Render kernel: a finished layer's 128 values per lane as packed (hi, lo) bf16 pairs = the next layer's B operands, produced
// in ONE fenced block (v_accvgpr_read, optional ReLU, 6 conversion instructions per pair), as take<> does for the fp32
// kernel.  Splitting each k-step's 8 values next to its MFMAs (hidden_part_b3) puts ~36 VALU instructions into every
// k-step; the stage clocks (tools/stage_timing.py B3=1) show hidden-layer stages of 2 440-2 500 clocks for 1 536 matrix
// clocks with exactly that difference -- and neither the LDS wait (a full k-step of read-ahead: no change) nor the barrier
// (raw: no change) in it.  (The training forward keeps the in-stream split: with its row stores the 128 operand registers
// spill.)
struct POps {
  unsigned hi[64], lo[64];   // pair j = values 2j, 2j+1 of the layer's accumulator order (block j >> 3, registers 2(j&7), +1)
};
template <bool RELU>
__device__ __forceinline__ void take_split(POps& o, const f32x16 (&acc)[8]) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float a = RELU ? relu_i(acc[nb][r]) : acc[nb][r], b = RELU ? relu_i(acc[nb][r + 1]) : acc[nb][r + 1];
      split2(a, b, o.hi[8 * nb + (r >> 1)], o.lo[8 * nb + (r >> 1)]);
    }
#pragma unroll
  for (int j = 0; j < 64; ++j) asm volatile("" : "+v"(o.hi[j]), "+v"(o.lo[j]));
  __builtin_amdgcn_sched_barrier(0);
}
// 16 k-steps on pre-split operands: k-step ks uses pairs 4ks .. 4ks+3
template <int NB, int KS0, bool PIPE = false>
__device__ __forceinline__ void hidden_part_b3p(Pipe3& pipe, f32x16 (&acc)[NB], const POps& p, bool last) {
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    BOp b;
    b.hi = __builtin_bit_cast(bf16x8, u32x4{p.hi[4 * ks], p.hi[4 * ks + 1], p.hi[4 * ks + 2], p.hi[4 * ks + 3]});
    b.lo = __builtin_bit_cast(bf16x8, u32x4{p.lo[4 * ks], p.lo[4 * ks + 1], p.lo[4 * ks + 2], p.lo[4 * ks + 3]});
    kstep<NB, PIPE>(pipe, acc, KS0 + ks, last && ks == 15, b);
  }
}

// emit N/8 k-steps from a value array (N a multiple of 8).  save != nullptr (training forward): the lane's values are
// also written to save[8 * (ks0 - ks_base + k) .. +7] -- its half of the saved X' / U' row, in value order (see
// anerf_build_perm_tables_b3); only the first `nreal` values of the segment are real, the rest is zero padding.
// SAVE is a compile-time switch and every lane stores (tail lanes of the last tile are clamped to the last valid
// sample and rewrite its values), so there is no exec-mask juggling in the middle of a stage.
template <int NB, int N, bool SAVE = false, bool PIPE = false>
__device__ __forceinline__ void emit(Pipe3& pipe, f32x16 (&acc)[NB], int ks0, int ks_last, const float (&val)[N],
                                     float* __restrict__ save = nullptr, int ks_base = 0, int nreal = 1 << 30) {
#pragma unroll
  for (int k = 0; k < N / 8; ++k) {
    if constexpr (SAVE) {
      // HAZARD (measured, gfx950): the stores are pinned in FRONT of the split of the same values.  Scheduled freely,
      // the compiler sinks them to the last use of val[] and the next ds_read_b128 (a weight fragment) is issued into
      // the store's data registers right behind it: the saved U' then carried fragment bits in a few hundred elements
      // per launch and the rgb logits were off by 1e-5..1e-4, different elements every run (first seen with a per-lane
      // `if (save)` around the stores, still there with unconditional stores, gone with this sched_barrier).  With the
      // split (16 VALU) between a store and the reuse of its registers the store has read its data.
      const int i0 = 8 * (ks0 - ks_base + k);
      if (i0 + 3 < nreal) *reinterpret_cast<f32x4*>(save + i0) = f32x4{val[8 * k], val[8 * k + 1], val[8 * k + 2], val[8 * k + 3]};
      if (i0 + 7 < nreal) *reinterpret_cast<f32x4*>(save + i0 + 4) = f32x4{val[8 * k + 4], val[8 * k + 5], val[8 * k + 6], val[8 * k + 7]};
      __builtin_amdgcn_sched_barrier(0);
    }
    const BOp b = split8(val[8 * k], val[8 * k + 1], val[8 * k + 2], val[8 * k + 3], val[8 * k + 4], val[8 * k + 5],
                         val[8 * k + 6], val[8 * k + 7]);
    kstep<NB, PIPE>(pipe, acc, ks0 + k, ks0 + k == ks_last, b);
  }
}

// The lane's 216 x-values, band-major [15 bands x 12 owned joints][36 bone-direction components], as 27 k-steps:
// 7 band pairs (24 values = 3 k-steps each) then cos_6 (12) + directions (36) = 6 k-steps.
template <int LV, bool SAVE = false, bool PIPE = false>
__device__ __forceinline__ void x_part_b3(Pipe3& pipe, f32x16 (&acc)[8], const float (&v)[12], const float (&wv)[12],
                                          const float (&rh)[36], bool last, float* __restrict__ xsave = nullptr) {
  static_assert(LV == 7, "band pairing below is written for multires = 7");
  constexpr int KS_LAST = 26;
  // (gs, c) = (w sin a, cos a): precise every 4th band, double-angle steps in between -- the same arithmetic as the fp32
  // kernel's x_part (anerf_mlp.hip), so both kernels feed identical values to the network
  float gs[12], cb[12];
#pragma unroll
  for (int p = 0; p < LV; ++p) {
    float val[24];
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      val[a] = (p == 0 ? v[a] : cb[a]) * wv[a];             // band 2p: raw (p = 0) or cos_{p-1}
      if (p % 4 == 0) {
        float s;
        sincos_f32(v[a] * (float)(1 << p), s, cb[a]);
        gs[a] = s * wv[a];
      } else {
        const float t = cb[a] + cb[a];
        gs[a] = gs[a] * t;
        cb[a] = fmaf(t, cb[a], -1.0f);
      }
      val[12 + a] = gs[a];                                   // band 2p+1: sin_p
    }
    emit<8, 24, SAVE, PIPE>(pipe, acc, 3 * p, last ? KS_LAST : -1, val, xsave);
  }
  float val[48];
#pragma unroll
  for (int a = 0; a < 12; ++a) val[a] = cb[a] * wv[a];       // band 14: cos_6
#pragma unroll
  for (int i = 0; i < 36; ++i) val[12 + i] = rh[i];
  emit<8, 48, SAVE, PIPE>(pipe, acc, 3 * LV, last ? KS_LAST : -1, val, xsave);
}

// TRAIN: also save what the (fp32) backward needs -- h0..h7, f, g row-major exactly as k_mlp_fwd<TRAIN> does, and the
// density-net / view-net inputs X', U' in THIS kernel's value order: row = [half 0's values | half 1's values]
// (216 + 216 and NU + NU floats: same plane shapes as the fp32 path, different column permutation).
template <int NB>
__device__ __forceinline__ void store_rows(float* __restrict__ row_h, const f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
      *reinterpret_cast<f32x4*>(row_h + 32 * nb + 8 * q) = o;
    }
}

template <int LV, int LD, int CODE, bool TRAIN>
__global__ __launch_bounds__(256) void k_mlp_fwd_b3(const MlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;

  Pipe3 pipe;
  pipe.init(A.packed, smem, wave, lane, A.nstages);
#ifdef ANERF_EXP_STAGE_TIMING
  const unsigned long long tt0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
  if (A.tbuf && blockIdx.x % 997 == 0 && lane == 0)
    pipe.tbuf = A.tbuf + ((long long)(blockIdx.x / 997) * 4 + wave) * 3 * 128;
#endif

  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  float tau_v = A.tau_v, tau_d = A.tau_d;   // TRAIN: optionally from the device-resident step block (see k_mlp_fwd)
  if constexpr (TRAIN) {
    if (A.tau_dev) {
      tau_v = A.tau_dev[0];
      tau_d = A.tau_dev[1];
    }
  }
#ifdef ANERF_EXP_B3_NOHSAVE
  const bool save = false;
#else
  const bool save = TRAIN && valid;
#endif
#ifdef ANERF_EXP_B3_NOXSAVE
  constexpr bool XS = false;
#else
  constexpr bool XS = TRAIN;
#endif
  auto HROW = [&](int l) { return A.save_h + ((long long)l * A.Ppad + p) * 256 + 4 * h; };

  float* aux_l = reinterpret_cast<float*>(smem + LDS_AUX_OFF);
  for (int i = tid; i < AUX_FLOATS / 4; i += 256)
    reinterpret_cast<f32x4*>(aux_l)[i] = reinterpret_cast<const f32x4*>(A.aux)[i];
  const float* aux_h = aux_l + 4 * h;

  float dray[3];
  const f32x4* bones4 = reinterpret_cast<const f32x4*>(smem + LDS_BONES_OFF);
  const long long tile_p0 = (long long)blockIdx.x * TILE;
  const long long ray0 = div_samples(tile_p0, A.S);
  long long ray1 = div_samples(tile_p0 + TILE - 1, A.S);
  if (ray1 > A.N - 1) ray1 = A.N - 1;
  const long long ray = div_samples(pc, A.S);
  const int n_stage_rays = A.skt_stride == 0 ? 1 : (int)(ray1 - ray0 + 1);
  const int lr = A.skt_stride == 0 ? 0 : (int)(ray - ray0);
  {
    f32x4* bw = reinterpret_cast<f32x4*>(smem + LDS_BONES_OFF);
    for (int i = tid; i < n_stage_rays * 72; i += 256) {
      const int ri = i / 72, rem = i - ri * 72, j = rem / 3, row = rem - 3 * j;
      bw[i] = *reinterpret_cast<const f32x4*>(A.skts + (ray0 + ri) * A.skt_stride + j * 16 + row * 4);
    }
  }
  const float* rp = A.rays + ray * A.ray_stride;
  const float z = A.z[pc];
  dray[0] = rp[3];
  dray[1] = rp[4];
  dray[2] = rp[5];
  float x0 = fmaf(dray[0], z, rp[0]), x1 = fmaf(dray[1], z, rp[1]), x2 = fmaf(dray[2], z, rp[2]);
  if (A.pnoise) {   // wave-uniform: ray_noise_std > 0 (raycasters.py:660)
    x0 += A.pnoise[3 * pc];
    x1 += A.pnoise[3 * pc + 1];
    x2 += A.pnoise[3 * pc + 2];
  }
  pipe.begin();
  // Skeleton-relative features of the sample for the lane half's 12 joints.  Evaluated THREE times (layer 0, the skip
  // layer, the view layer's gates) from the 3 position registers + the bone matrices in LDS instead of keeping 60 values
  // alive across the hidden layers: ~400 VALU per re-evaluation against 16 k-steps x 24 MFMAs per layer, and the
  // registers go to the weight-fragment reads in flight (the kernel sits at the 512-register limit).
  auto encode_x = [&](float (&v)[12], float (&wv)[12], float (&rh)[36]) {
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const int j = 8 * (a >> 2) + 4 * h + (a & 3);
      const f32x4 r0 = bones4[(lr * 24 + j) * 3 + 0], r1 = bones4[(lr * 24 + j) * 3 + 1], r2 = bones4[(lr * 24 + j) * 3 + 2];
      const float y0 = r0.x * x0 + r0.y * x1 + r0.z * x2 + r0.w;
      const float y1 = r1.x * x0 + r1.y * x1 + r1.z * x2 + r1.w;
      const float y2 = r2.x * x0 + r2.y * x1 + r2.z * x2 + r2.w;
      const float n = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
      const float inv = rcp_nr(fmaxf(n, 1e-12f));
      v[a] = n;
      wv[a] = cutoff_gate(tau_v, n, A.cut_v[j]);
      const float gb = A.gate_bones ? inv * wv[a] : inv;      // cutoff_bones: r_j * w_j
      rh[3 * a + 0] = y0 * gb;
      rh[3 * a + 1] = y1 * gb;
      rh[3 * a + 2] = y2 * gb;
    }
  };

  constexpr int DIMD = 72 * (1 + 2 * LD);
  constexpr int KSX = 27;   // k-steps of the x part
  constexpr bool PP = !TRAIN;  // hand-pipelined k-steps in the render kernel only (see kstep)
  f32x16 accA[8], accB[8];

  // ---- layer 0
  init_bias<8>(accA, aux_h + AUX_B0);
  {
    float v[12], wv[12], rh[36];
    encode_x(v, wv, rh);
    x_part_b3<LV, XS, PP>(pipe, accA, v, wv, rh, true, XS ? A.save_x + pc * (24 * (1 + 2 * LV) + 72) + h * (12 * (1 + 2 * LV) + 36) : nullptr);
  }
  f32x16 accv[4];
  float sigma_raw;
  if constexpr (!TRAIN) {
    // Render: a finished layer is handed over as pre-split operands in one block (take_split; the ReLU inside)
    POps ops;
    auto hidden = [&](auto ks0, f32x16 (&next)[8], f32x16 (&prev)[8], bool relu, const float* bias) __attribute__((always_inline)) {
      constexpr int KS0 = decltype(ks0)::value;
      if (bias) init_bias<8>(next, bias);       // (nullptr: already initialised -- the skip layer, whose x part came first)
      if (relu) take_split<true>(ops, prev); else take_split<false>(ops, prev);
      hidden_part_b3p<8, KS0, true>(pipe, next, ops, true);
    };
    using K0 = std::integral_constant<int, 0>;
#pragma unroll 1
    for (int L = 1; L <= 3; L += 2) {
      hidden(K0{}, accB, accA, true, aux_h + AUX_B0 + 256 * L);
      hidden(K0{}, accA, accB, true, aux_h + AUX_B0 + 256 * (L + 1));
    }
    // layer 5 (skip): x re-encoded from the position; h4 waits in its accumulators until the x part is done
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
    init_bias<8>(accB, aux_h + AUX_B0 + 256 * 5);
    {
      float v[12], wv[12], rh[36];
      encode_x(v, wv, rh);
      x_part_b3<LV, false, PP>(pipe, accB, v, wv, rh, false);
    }
    hidden(std::integral_constant<int, KSX>{}, accB, accA, true, nullptr);
    hidden(K0{}, accA, accB, true, aux_h + AUX_B0 + 256 * 6);
    hidden(K0{}, accB, accA, true, aux_h + AUX_B0 + 256 * 7);
    relu_pass<8>(accB);                       // h7 in place: the density head reads it
    sigma_raw = head_dot<8>(accB, aux_h + AUX_WA) + aux_l[AUX_BA];
    hidden(K0{}, accA, accB, false, aux_h + AUX_BF);   // feature layer (h7 is already activated)
    init_bias<4>(accv, aux_h + AUX_BV);
    take_split<false>(ops, accA);             // the feature: no activation
    hidden_part_b3p<4, 0, true>(pipe, accv, ops, false);
  } else {
    relu_pass<8>(accA);
    if (save) store_rows<8>(HROW(0), accA);
    // ---- layers 1..4
  #pragma unroll 1
    for (int L = 1; L <= 3; L += 2) {
      init_bias<8>(accB, aux_h + AUX_B0 + 256 * L);
      hidden_part_b3<8, 0, PP>(pipe, accB, accA, true);
      relu_pass<8>(accB);
      if (save) store_rows<8>(HROW(L), accB);
      init_bias<8>(accA, aux_h + AUX_B0 + 256 * (L + 1));
      hidden_part_b3<8, 0, PP>(pipe, accA, accB, true);
      relu_pass<8>(accA);
      if (save) store_rows<8>(HROW(L + 1), accA);
    }
    // ---- layer 5 (skip): x re-encoded from the position (opaque so that the compiler does not keep layer 0's values alive)
    asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
    init_bias<8>(accB, aux_h + AUX_B0 + 256 * 5);
    {
      float v[12], wv[12], rh[36];
      encode_x(v, wv, rh);
      x_part_b3<LV, false, PP>(pipe, accB, v, wv, rh, false);
    }
    hidden_part_b3<8, KSX, PP>(pipe, accB, accA, true);
    relu_pass<8>(accB);
    if (save) store_rows<8>(HROW(5), accB);
    // ---- layers 6, 7
    init_bias<8>(accA, aux_h + AUX_B0 + 256 * 6);
    hidden_part_b3<8, 0, PP>(pipe, accA, accB, true);
    relu_pass<8>(accA);
    if (save) store_rows<8>(HROW(6), accA);
    init_bias<8>(accB, aux_h + AUX_B0 + 256 * 7);
    hidden_part_b3<8, 0, PP>(pipe, accB, accA, true);
    relu_pass<8>(accB);
    if (save) store_rows<8>(HROW(7), accB);
    sigma_raw = head_dot<8>(accB, aux_h + AUX_WA) + aux_l[AUX_BA];
    // ---- feature layer
    init_bias<8>(accA, aux_h + AUX_BF);
    hidden_part_b3<8, 0, PP>(pipe, accA, accB, true);
    if (save) store_rows<8>(A.save_f + p * 256 + 4 * h, accA);
    // ---- view layer: [feature (16 k-steps); D bands; code; zero padding to a multiple of 8 values per lane]
    init_bias<4>(accv, aux_h + AUX_BV);
    hidden_part_b3<4, 0, PP>(pipe, accv, accA, false);
  }
  constexpr int NU = 36 * (1 + 2 * LD) + CODE / 2;          // values per lane
  constexpr int NUP = (NU + 7) / 8 * 8;
  constexpr int KSV_LAST = 16 + NUP / 8 - 1;
  float* usave = XS ? A.save_u + pc * (2 * NU) + h * NU : nullptr;
  float e[36], wd[12];
  asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2));
#pragma unroll
  for (int a = 0; a < 12; ++a) {
    const int j = 8 * (a >> 2) + 4 * h + (a & 3);
    const f32x4 r0 = bones4[(lr * 24 + j) * 3 + 0], r1 = bones4[(lr * 24 + j) * 3 + 1], r2 = bones4[(lr * 24 + j) * 3 + 2];
    const float p0 = r0.x * x0 + r0.y * x1 + r0.z * x2 + r0.w;            // joint distance again (gate of the directions)
    const float p1 = r1.x * x0 + r1.y * x1 + r1.z * x2 + r1.w;
    const float p2 = r2.x * x0 + r2.y * x1 + r2.z * x2 + r2.w;
    const float vn = sqrtf(p0 * p0 + p1 * p1 + p2 * p2);
    const float y0 = r0.x * dray[0] + r0.y * dray[1] + r0.z * dray[2];
    const float y1 = r1.x * dray[0] + r1.y * dray[1] + r1.z * dray[2];
    const float y2 = r2.x * dray[0] + r2.y * dray[1] + r2.z * dray[2];
    const float inv = rcp_nr(fmaxf(sqrtf(y0 * y0 + y1 * y1 + y2 * y2), 1e-12f));
    e[3 * a + 0] = y0 * inv;
    e[3 * a + 1] = y1 * inv;
    e[3 * a + 2] = y2 * inv;
    wd[a] = cutoff_gate(tau_d, vn, A.cut_d[j]);
  }
  float gse[36], cbe[36];
#pragma unroll
  for (int pq = 0; pq < LD; ++pq) {       // band pairs (2pq, 2pq+1) = (raw | cos_{pq-1}, sin_pq): 72 values = 9 k-steps
    float val[72];
#pragma unroll
    for (int i = 0; i < 36; ++i) {
      val[i] = (pq == 0 ? e[i] : cbe[i]) * wd[i / 3];
      if (pq % 4 == 0) {
        float s;
        if (pq == 0) sincos_unit_f32(e[i], s, cbe[i]);
        else sincos_f32(e[i] * (float)(1 << pq), s, cbe[i]);
        gse[i] = s * wd[i / 3];
      } else {
        const float t = cbe[i] + cbe[i];
        gse[i] = gse[i] * t;
        cbe[i] = fmaf(t, cbe[i], -1.0f);
      }
      val[36 + i] = gse[i];
    }
    emit<4, 72, XS, PP>(pipe, accv, 16 + 9 * pq, KSV_LAST, val, usave, 16, NU);
  }
  {
    // last band (cos_{LD-1}, or the raw band when LD == 0) + frame code + zero padding
    constexpr int NT = NUP - 72 * LD;
    float val[NT];
#pragma unroll
    for (int i = 0; i < 36; ++i) val[i] = (LD == 0 ? e[i] : cbe[i]) * wd[i / 3];
    if constexpr (CODE > 0) {
      int ci = (int)A.cam[ray];
      ci = ci < 0 ? 0 : (ci >= A.n_codes ? A.n_codes - 1 : ci);
      const float* crow = A.codes + (long long)ci * CODE + 8 * h;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(crow), c1 = *reinterpret_cast<const f32x4*>(crow + 4);
      val[36] = c0.x; val[37] = c0.y; val[38] = c0.z; val[39] = c0.w;
      val[40] = c1.x; val[41] = c1.y; val[42] = c1.z; val[43] = c1.w;
    }
#pragma unroll
    for (int i = NU - 72 * LD; i < NT; ++i) val[i] = 0.f;
    emit<4, NT, XS, PP>(pipe, accv, 16 + 9 * LD, KSV_LAST, val, usave, 16, NU);
  }
  relu_pass<4>(accv);
  if (save) store_rows<4>(A.save_g + p * 128 + 4 * h, accv);
  const float c0 = head_dot<4>(accv, aux_h + AUX_WC + 0) + aux_l[AUX_BC + 0];
  const float c1 = head_dot<4>(accv, aux_h + AUX_WC + 128) + aux_l[AUX_BC + 1];
  const float c2 = head_dot<4>(accv, aux_h + AUX_WC + 256) + aux_l[AUX_BC + 2];
  if (valid && h == 0) {
    f32x4 o = {c0, c1, c2, sigma_raw};
    *reinterpret_cast<f32x4*>(A.raw + p * 4) = o;
  }
#ifdef ANERF_EXP_STAGE_TIMING   // every tile: start / end clocks + where it ran, behind the stage records (as k_mlp_fwd)
  if (tid == 0 && A.tbuf) {
    unsigned long long* t = A.tbuf + 64 * 4 * 128 * 3 + 6 * (long long)blockIdx.x;
    t[0] = tt0; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr0; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg(63492); t[5] = __builtin_amdgcn_s_getreg(63508);
  }
#endif
}

template <int LV, int LD, int CODE, bool TRAIN>
static int launch_b3(const MlpArgs& a, hipStream_t st) {
  const long long nblk = (a.P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  if (a.P > 0xFFFFFF00ll) return set_error(ANERF_E_SHAPE, "more than 2^32 - 256 samples in one call");   // div_samples is 32-bit
  const size_t lds = LDS_BONES_OFF + MAX_TILE_RAYS * 72 * 16;
  auto kern = k_mlp_fwd_b3<LV, LD, CODE, TRAIN>;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &lds_set);
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
  return check_launch("k_mlp_fwd_b3");
}

// sv == nullptr: render (no activations saved); else the training forward (anerf_mlp_raw_train_b3)
int mlp_b3_entry(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays, int ray_stride,
                 const float* z, const float* skts, long long skt_stride, const float* cam, const float* codes, int n_codes,
                 float tau_v, float tau_d, const float* cut_v, const float* cut_d, long long P, int N, int S, int nstages,
                 float* raw, const AnerfSaved* sv, hipStream_t st, const float* pnoise, const float* tau_dev) {
  MlpArgs a;
  a.pnoise = pnoise;
  a.tau_dev = tau_dev;
  a.packed = packed; a.aux = aux; a.rays = rays; a.z = z; a.skts = skts; a.cam = cam; a.codes = codes;
  a.cut_v = cut_v; a.cut_d = cut_d; a.x = nullptr; a.raw = raw; a.P = P; a.Ppad = P; a.skt_stride = skt_stride;
  a.S = S; a.N = N; a.ray_stride = ray_stride; a.n_codes = n_codes; a.x_width = 0; a.nstages = nstages;
  a.tau_v = tau_v; a.tau_d = tau_d; a.gate_bones = cfg->cutoff_bones;
  a.save_h = a.save_f = a.save_g = a.save_x = a.save_u = nullptr;
#ifdef ANERF_EXP_STAGE_TIMING
  a.tbuf = reinterpret_cast<unsigned long long*>(g_tile_timing_buf);
#endif
  if (sv) {
    a.save_h = sv->h; a.save_f = sv->f; a.save_g = sv->g; a.save_x = sv->x; a.save_u = sv->u; a.Ppad = sv->p_pad;
  }
  const int ld = cfg->multires_views, cd = cfg->framecode_ch;
  if (cfg->multires != 7) return set_error(ANERF_E_CONFIG, "multires must be 7");
  if (sv) {
    if (ld == 4 && cd == 0) return launch_b3<7, 4, 0, true>(a, st);
    if (ld == 4 && cd == 16) return launch_b3<7, 4, 16, true>(a, st);
    if (ld == 0 && cd == 0) return launch_b3<7, 0, 0, true>(a, st);
  } else {
    if (ld == 4 && cd == 0) return launch_b3<7, 4, 0, false>(a, st);
    if (ld == 4 && cd == 16) return launch_b3<7, 4, 16, false>(a, st);
    if (ld == 0 && cd == 0) return launch_b3<7, 0, 0, false>(a, st);
  }
  return set_error(ANERF_E_CONFIG, "unsupported (multires_views, framecode_ch); built: (4,0) (4,16) (0,0)");
}

// ------------------------------------------------------------------------------------------------
// k_mlp_bwd_b3 -- backward-data of the MLP on split-bf16 MFMAs (anerf_mlp_backward_b3): the structure of k_mlp_bwd
// (anerf_mlp_bwd.hip: ping-pong accumulator sets read directly as the next layer's B operands, ReLU-mask row
// prefetched per layer, in-place mask pass, one continuous weight segment) with the k-steps of this file: the W^T image
// `which = 4` holds (hi, lo) fragment pairs, dz values are split in registers.  Same inputs / outputs as k_mlp_bwd.
// ------------------------------------------------------------------------------------------------
struct BwdArgs3 {
  const float* packed_t;   // W^T image, which = 4
  const float* aux;
  const float* draw;       // [P][4]
  const float* save_h;     // [8][Ppad][256]
  const float* save_g;     // [Ppad][128]
  float* dz;               // [8][Ppad][256]
  float* df;               // [Ppad][256]
  float* dzv;              // [Ppad][128]
  long long P, Ppad;
  int nstages;
};

template <int NB>
__device__ __forceinline__ void load_mask3(f32x4 (&mk)[32], const float* __restrict__ row_h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) mk[4 * nb + q] = *reinterpret_cast<const f32x4*>(row_h + 32 * nb + 8 * q);
}
template <int NB>
__device__ __forceinline__ void mask_pass3(f32x16 (&acc)[NB], const f32x4 (&mk)[32]) {
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
__device__ __forceinline__ void zero_acc3(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
}

__device__ __forceinline__ void bwd_layer_b3(Pipe3& pipe, f32x16 (&out)[8], const f32x16 (&prev)[8], f32x4 (&mk)[32],
                                             const float* __restrict__ mask_row_h, float* __restrict__ dz_row_h, bool valid,
                                             bool last) {
  load_mask3<8>(mk, mask_row_h);
  zero_acc3<8>(out);
  hidden_part_b3<8, 16>(pipe, out, prev, last);     // KS0 = 16: any non-zero multiple of the k-steps per stage
  mask_pass3<8>(out, mk);
  if (valid) store_rows<8>(dz_row_h, out);
}

__global__ __launch_bounds__(256) void k_mlp_bwd_b3(const BwdArgs3 A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe3 pipe;
  pipe.init(A.packed_t, smem, wave, lane, A.nstages);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  float* aux_l = reinterpret_cast<float*>(smem + LDS_AUX_OFF);
  for (int i = tid; i < AUX_FLOATS / 4; i += 256)
    reinterpret_cast<f32x4*>(aux_l)[i] = reinterpret_cast<const f32x4*>(A.aux)[i];
  const float* aux_h = aux_l + 4 * h;
  const f32x4 dr = *reinterpret_cast<const f32x4*>(A.draw + pc * 4);
  f32x4 mk[32];
  load_mask3<4>(mk, A.save_g + pc * 128 + 4 * h);
  pipe.begin();

  f32x16 accA[8], accB[8];
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
  mask_pass3<4>(accv, mk);
  if (valid) store_rows<4>(A.dzv + p * 128 + 4 * h, accv);
  // ---- view layer, feature columns: df = Wv[:, :256]^T dzv      (8 k-steps over the 128 view units)
  load_mask3<8>(mk, A.save_h + (7 * A.Ppad + pc) * 256 + 4 * h);
  zero_acc3<8>(accA);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const int nbk = ks >> 1, r0 = 8 * (ks & 1);
    const BOp b = split8(accv[nbk][r0], accv[nbk][r0 + 1], accv[nbk][r0 + 2], accv[nbk][r0 + 3], accv[nbk][r0 + 4],
                         accv[nbk][r0 + 5], accv[nbk][r0 + 6], accv[nbk][r0 + 7]);
    kstep<8>(pipe, accA, ks, false, b);
  }
  if (valid) store_rows<8>(A.df + p * 256 + 4 * h, accA);
  // ---- feature layer + density head: dh7 = Wf^T df + w_alpha * dsigma ; dz7 = dh7 * [h7 > 0]
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
  hidden_part_b3<8, 8>(pipe, accB, accA, false);
  mask_pass3<8>(accB, mk);
  if (valid) store_rows<8>(A.dz + (7 * A.Ppad + p) * 256 + 4 * h, accB);
  // ---- trunk: dz_{l-1} = (W_l^T dz_l) * [h_{l-1} > 0],  l = 7..1   (W_5: hidden columns only)
  const float* hrow = A.save_h + pc * 256 + 4 * h;
  float* zrow = A.dz + p * 256 + 4 * h;
  const long long plane = A.Ppad * 256;
#pragma unroll 1
  for (int L = 7; L >= 3; L -= 2) {
    bwd_layer_b3(pipe, accA, accB, mk, hrow + (L - 1) * plane, zrow + (L - 1) * plane, valid, false);
    bwd_layer_b3(pipe, accB, accA, mk, hrow + (L - 2) * plane, zrow + (L - 2) * plane, valid, false);
  }
  bwd_layer_b3(pipe, accA, accB, mk, hrow, zrow, valid, true);
}

int mlp_bwd_b3_entry(const float* packed_t, const float* aux, const float* draw, const AnerfSaved* sv, float* dz, float* df,
                     float* dzv, long long P, int nstages, hipStream_t st) {
  BwdArgs3 b;
  b.packed_t = packed_t; b.aux = aux; b.draw = draw; b.save_h = sv->h; b.save_g = sv->g;
  b.dz = dz; b.df = df; b.dzv = dzv; b.P = P; b.Ppad = sv->p_pad; b.nstages = nstages;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = LDS_BONES_OFF;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_b3), (int)lds, &lds_set);
  hipLaunchKernelGGL(k_mlp_bwd_b3, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd_b3");
}

// ------------------------------------------------------------------------------------------------
// k_mlp_bwd_in_b3 -- input gradients (pose refinement / frame codes) on split-bf16 MFMAs (anerf_input_grads_b3):
//     dX'[p][432] = W0'^T dz0 + W5x'^T dz5          dU'[p][UW] = Wvu'^T dzv        (stream column order, as k_mlp_bwd_in)
// on the which = 5 image.  The B operands come from memory: lane (m, hh) needs, for k-step ks, the two float4
// dz[16 ks + 4 hh ..+3] and dz[16 ks + 8 + 4 hh ..+3] of its sample's row -- the dz0 and dz5 rows (2 x 128 registers) are
// fetched once per tile and kept for both 256-column output groups; dzv reuses dz0's registers afterwards.
// ------------------------------------------------------------------------------------------------
struct BwdInArgs3 {
  const float* packed_i;
  const float* dz;     // [8][Ppad][256]
  const float* dzv;    // [Ppad][128]
  float* dx;           // [Ppad][432]
  float* du;           // [Ppad][UW]
  long long P, Ppad;
  int nstages, uw;
};

template <int NKS>
__device__ __forceinline__ void load_dz_row(f32x4 (&d)[32], const float* __restrict__ row_h) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    d[2 * ks] = *reinterpret_cast<const f32x4*>(row_h + 16 * ks);
    d[2 * ks + 1] = *reinterpret_cast<const f32x4*>(row_h + 16 * ks + 8);
  }
}

template <int NKS, int NBU = 8>
__device__ __forceinline__ void contract_row(Pipe3& pipe, f32x16 (&acc)[8], const f32x4 (&d)[32], int ks0, bool last) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const BOp b = split8(d[2 * ks].x, d[2 * ks].y, d[2 * ks].z, d[2 * ks].w, d[2 * ks + 1].x, d[2 * ks + 1].y,
                         d[2 * ks + 1].z, d[2 * ks + 1].w);
    kstep<8, false, NBU>(pipe, acc, ks0 + ks, last && ks == NKS - 1, b);
  }
}

__device__ __forceinline__ void store_cols3(float* __restrict__ row, const f32x16 (&acc)[8], int c0, int w, int h) {
#pragma unroll
  for (int nb = 0; nb < 8; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 32 * nb + 8 * q + 4 * h;
      if (c < w) {
        f32x4 o = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
        *reinterpret_cast<f32x4*>(row + c) = o;
      }
    }
}

__global__ __launch_bounds__(256) void k_mlp_bwd_in_b3(const BwdInArgs3 A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  Pipe3 pipe;
  pipe.init(A.packed_i, smem, wave, lane, A.nstages);
  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  f32x4 d0[32], d5[32];
  load_dz_row<16>(d0, A.dz + pc * 256 + 4 * h);
  load_dz_row<16>(d5, A.dz + (5 * A.Ppad + pc) * 256 + 4 * h);
  pipe.begin();
  f32x16 acc[8];
  const int ngu = (A.uw + 255) / 256;
  // one continuous weight segment: k-step numbering only matters modulo the k-steps per stage (2)
  zero_acc3<8>(acc);
  contract_row<16>(pipe, acc, d0, 0, false);
  contract_row<16>(pipe, acc, d5, 16, false);
  if (valid) store_cols3(A.dx + p * 432, acc, 0, 432, h);
  zero_acc3<8>(acc);                                         // columns 256..431 = 5.5 blocks: 6 of the 8 are multiplied
  contract_row<16, 6>(pipe, acc, d0, 16, false);
  contract_row<16, 6>(pipe, acc, d5, 16, false);
  if (valid) store_cols3(A.dx + p * 432, acc, 256, 432, h);
  load_dz_row<8>(d0, A.dzv + pc * 128 + 4 * h);
  const int nbl = (A.uw - 256 * (ngu - 1) + 31) / 32;         // blocks of the (narrow) last group: 5 (648 / 664) or 3 (72)
#pragma unroll 1
  for (int gi = 0; gi < ngu; ++gi) {
    zero_acc3<8>(acc);
    const bool lastg = gi == ngu - 1;
    if (!lastg || nbl > 5) contract_row<8>(pipe, acc, d0, 16, lastg);
    else if (nbl > 3) contract_row<8, 5>(pipe, acc, d0, 16, true);
    else contract_row<8, 3>(pipe, acc, d0, 16, true);
    if (valid) store_cols3(A.du + p * A.uw, acc, 256 * gi, A.uw, h);
  }
}

int mlp_bwd_in_b3_entry(const float* packed_i, const float* dz, const float* dzv, float* dx, float* du, long long P,
                        long long Ppad, int nstages, int uw, hipStream_t st) {
  BwdInArgs3 b;
  b.packed_i = packed_i; b.dz = dz; b.dzv = dzv; b.dx = dx; b.du = du; b.P = P; b.Ppad = Ppad; b.nstages = nstages; b.uw = uw;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = LDS_BONES_OFF;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_in_b3), (int)lds, &lds_set);
  hipLaunchKernelGGL(k_mlp_bwd_in_b3, dim3((unsigned)nblk), dim3(256), lds, st, b);
  return check_launch("k_mlp_bwd_in_b3");
}

// ------------------------------------------------------------------------------------------------
// parameter gather + hi/lo split into the bf16x3 weight image (which = 3).  Table entry per 16-bit element:
// bit 29 = part (0 hi, 1 lo), bits 24..28 = tensor id, bits 0..23 = element offset; -1 = zero.
// ------------------------------------------------------------------------------------------------
__global__ void k_pack_b3(AnerfNetParams P, const int32_t* __restrict__ table, long long n, unsigned short* __restrict__ out) {
  const float* tens[24];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    tens[i] = P.w[i];
    tens[12 + i] = P.b[i];
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int32_t e = table[i];
    unsigned short val = 0;
    if (e >= 0) {
      const int part = (e >> 29) & 1, id = (e >> 24) & 31, off = e & 0xFFFFFF;
      const float* src = P.w[0];
#pragma unroll
      for (int k = 0; k < 24; ++k)
        if (id == k) src = tens[k];
      float x = src[off] * sched_scale(P, id, off);
      asm volatile("" : "+v"(x));   // the ROUNDED fp32 product is what gets split: no contraction of the multiply into `x - hi` below
      const __bf16 hi = (__bf16)x;
      const __bf16 r = part ? (__bf16)(x - (float)hi) : hi;
      val = __builtin_bit_cast(unsigned short, r);
    }
    out[i] = val;
  }
}

int launch_pack_b3(const AnerfNetParams* P, const int32_t* table, long long n, void* out, hipStream_t st) {
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_pack_b3, dim3(blocks), dim3(256), 0, st, *P, table, n, (unsigned short*)out);
  return check_launch("k_pack_b3");
}

}  // namespace anerf
