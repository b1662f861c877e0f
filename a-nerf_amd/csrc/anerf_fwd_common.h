// anerf_fwd_common.h -- pieces shared by the forward kernels (fp32 k_mlp_fwd, bf16x3 k_mlp_fwd_b3): LDS layout,
// 3-slot weight-stream pipe, bias init / ReLU pass / head dot on the accumulator sets, kernel argument block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf_dev.h"

namespace anerf {

constexpr int RING_SLOTS = 3;
constexpr int LDS_AUX_OFF = RING_SLOTS * STAGE_BYTES;             // biases + head rows (AUX_FLOATS floats)
constexpr int LDS_AUX_BYTES = (AUX_FLOATS * 4 + 255) / 256 * 256;
constexpr int LDS_BONES_OFF = LDS_AUX_OFF + LDS_AUX_BYTES;        // MAX_TILE_RAYS x 24 x 3 float4

// ------------------------------------------------------------------------------------------------
// weight-stream pipe: global -> LDS ring (3 slots x 32 KiB), all 4 waves cooperate.  Invariant while stage s is
// being consumed: stages s and s+1 are complete and visible to every wave; stage s+2 is in flight.
// Tried and rejected on the bf16x3 kernel (short ~1500-cycle stages, where more lead would help):
//   * 4 slots + counted `s_waitcnt vmcnt(8)` (keep the newest stage pending): stale fragments -- LDS-DMA
//     completions are not ordered the way a counted wait needs; parity tests caught it;
//   * 4 slots + per-wave-pair stage ownership (16 glds per owner): also produced stale fragments;
//   * 4 slots with plain vmcnt(0): correct but 6 % slower than 3 slots (same one-stage lead, worse allocation).
// Round 2, fp32 kernels: a 4-slot ring with ONE wait + barrier per two stages (rendezvous in front of the pair's last
// k-group, LDS-DMA of the pair after next issued there): correct (all parity tests) but slower -- render kernel 219.5 vs
// 216.0 ms, training forward 3.80 vs 3.63 ms (16-deep DMA bursts, 20-30 spilled VGPRs); the 3-slot pipe stays.
// Round 2, fp32 training forward: issuing a stage's eight pieces two per k-group instead of all behind the barrier
// (to keep the vector-memory path free for the activation stores) -- the weight stream turned out not to interfere
// with the stores at all (ablation without weight loads: stores cost the same 0.24 ms), and hipcc either re-clusters
// the pieces or, when they are fenced with sched_barrier masks, spills thousands of registers.
// ------------------------------------------------------------------------------------------------
// STAGGER (fp32 kernels, round 3): a stage's refill is not issued by all four waves right behind the stage barrier but by wave
// w behind quarter w of the stage's FIRST k-group (kgroup), i.e. 512 matrix clocks apart.  The CU has ONE texture-address
// unit: 32 wave-instructions (4 waves x 8 pieces of 1 KiB) arriving at the same moment queue up in front of it, the last wave
// starts its MFMAs ~400 clocks late and the next barrier waits for it.  tools/probe/mfma_probe_f32.hip, the stage of these
// kernels without any VALU work (8 192 matrix clocks): MFMA + reads 8 231, + burst behind the barrier 8 601, + staggered by wave
// 8 375 clocks; one piece per MFMA 8 921 and 2 pieces per k-group 8 609 are worse (every vector-memory instruction between
// two MFMAs costs far more than its issue slot), one burst of 8 per wave with nobody else at the unit is the cheapest form.
template <bool ASMDMA, bool STAGGER = false>
struct Pipe3T {
  static constexpr bool kStagger = STAGGER;
  int refill_stage = -1;   // STAGGER: stage this wave still has to request into refill_slot (-1: none)
  int refill_slot = 0;
  int r_sel = 7;           // STAGGER: this wave's index while a refill is pending, else 7 (matches no quarter)
  unsigned r_lds = 0;      // ... LDS address of the MIDDLE of this wave's 8 fragments in the slot being refilled
  const char* r_g = nullptr;   // ... global address of the middle of this wave's 8 fragments of the stage being requested
  const char* gsrc;   // wave-UNIFORM source of this wave's first fragment of stage 0 (lane l adds lane16: saddr + voffset form)
  char* smem;
  unsigned wave_dst;  // wave-uniform LDS byte offset of this wave's 8 fragments inside a stage
  unsigned lane16;    // lane * 16
  unsigned cur;       // lane-relative LDS byte offset of the stage being consumed
  unsigned nxt;       // ... of the following stage
  int slot;           // ring slot of the stage being consumed
  int stage;          // stage being consumed
  int nstages;
  int wave;           // wave index inside the workgroup (wave-uniform)
  f32x4 pref[8];      // split-bf16 kernels: fragments of the next stage's first k-group, loaded before the stage barrier
  f32x4 a[4];         // fp32 kernels: two-quarter window of weight fragments (see kgroup)
#ifdef ANERF_EXP_STAGE_TIMING   // debug build only (tools/stage_timing.py): per-wave arrive / leave clocks of every stage barrier
  unsigned long long* tbuf = nullptr;
#endif

  __device__ __forceinline__ void issue(int s, int sl) {
#ifdef ANERF_EXP_NOGLDS   // ablation build only (tools/ablate.sh): never load weights
    (void)s; (void)sl; return;
#endif
    const char* g = gsrc + (size_t)s * STAGE_BYTES;
    char* l = smem + sl * STAGE_BYTES + wave_dst;
    if constexpr (ASMDMA) {
      // fp32 kernels: the LDS-DMA is issued through inline asm so that hipcc does not see it.  A pending
      // `global_load_lds` is booked by the compiler's wait-count pass as a FLAT access that may touch LDS, and while one is
      // pending every LDS wait it emits is a full `s_waitcnt lgkmcnt(0)` -- with a DMA in flight all the time that meant
      // every k-group drained its freshly issued fragment reads (kgroup).  Ordering against the LDS reads is the ring
      // protocol's (vmcnt(0) + barrier, both asm with a memory clobber), exactly as before.
      // saddr + 32-bit lane offset + immediate: the instruction offset is added to the global AND the LDS address, so four
      // pieces share one M0 / one scalar base (8 pieces = 2 x (s_mov m0 + 4 DMA) and no VALU; the 64-bit-vaddr form needed a
      // v_lshl_add_u64 + s_mov per piece, ~50 instructions per stage inside the MFMA stream)
      const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)l);
#pragma unroll
      for (int half = 0; half < 2; ++half)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072"
                     :: "s"(lds0 + half * 4 * FRAG_BYTES), "v"(lane16), "s"(g + half * 4 * FRAG_BYTES) : "memory", "m0");
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + i * FRAG_BYTES + lane16), (lds_ptr_t)(l + i * FRAG_BYTES), 16, 0, 0);
    }
  }
  // the burst issue's two statements (same operands, same clobbers), executed by wave 0 only: EXEC is cleared for the others
  __device__ __forceinline__ void issue_wave0(int s, int sl) {
#ifdef ANERF_EXP_NOGLDS
    (void)s; (void)sl; return;
#endif
    const char* g = gsrc + (size_t)s * STAGE_BYTES;
    char* l = smem + sl * STAGE_BYTES + wave_dst;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)l);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      unsigned long long save_exec, is_mine;
      asm volatile("s_mov_b64 %0, exec\n\t"
                   "v_cmp_eq_u32_e64 %1, %5, 0\n\t"
                   "s_nop 3\n\t"
                   "s_mov_b64 exec, %1\n\t"
                   "s_cbranch_execz 1f\n\t"
                   "s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %3, %4\n\t"
                   "global_load_lds_dwordx4 %3, %4 offset:1024\n\t"
                   "global_load_lds_dwordx4 %3, %4 offset:2048\n\t"
                   "global_load_lds_dwordx4 %3, %4 offset:3072\n"
                   "1:\n\t"
                   "s_mov_b64 exec, %0"
                   : "=&s"(save_exec), "=&s"(is_mine)
                   : "s"(lds0 + half * 4 * FRAG_BYTES), "v"(lane16), "s"(g + half * 4 * FRAG_BYTES), "s"(wave_dst) : "memory", "m0");
    }
  }
  __device__ __forceinline__ void set_offsets() {
    cur = lane16 + slot * STAGE_BYTES;
    nxt = lane16 + (slot == RING_SLOTS - 1 ? 0 : slot + 1) * STAGE_BYTES;
  }
  __device__ __forceinline__ void init(const float* packed, char* smem_, int wave, int lane, int nstages_) {
    gsrc = reinterpret_cast<const char*>(packed) + wave * (8 * FRAG_BYTES);
    smem = smem_;
    wave_dst = wave * (8 * FRAG_BYTES);
    lane16 = lane * 16;
    slot = 0;
    stage = 0;
    nstages = nstages_;
    this->wave = wave;
    set_offsets();
    issue(0, 0);
    if (nstages > 1) issue(1, 1);
  }
  // after everybody's prologue LDS writes: stages 0 and 1 landed; start stage 2
  __device__ __forceinline__ void begin() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nstages > 2) issue(2, 2);
  }
  // fp32 kernels, once after begin(): fragments of the very first k-group
  __device__ __forceinline__ void prime() {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(smem + cur + j * FRAG_BYTES);   // quarters 0, 1 (NB = 8)
  }
  // fp32 kernels: end of a stage WITHOUT draining the LDS queue.  The reads still in flight at this point are the
  // prefetched fragments of the next k-group; they target the NEXT ring slot, not the one this barrier hands back to the
  // weight loader, so they may cross it (the compiler tracks them and waits where their values are used).
  __device__ __forceinline__ void end_stage_raw() {
    stage_rendezvous();
    stage_refill();
  }
  // the two halves of end_stage_raw, for callers that want to run code between them (k_mlp_bwd forms the next stage's
  // masked operands there: whatever it loaded with ordinary global loads is complete at that point, and the compiler's
  // counted waits for those loads must not find the freshly issued LDS-DMA pieces queued behind them)
  __device__ __forceinline__ void stage_rendezvous() {
    // one asm block, "memory"-clobbered: neither the compiler's memory operations (the LDS-DMA issue!) nor LDS reads may
    // move across it, and no LDS wait is added
#ifdef ANERF_EXP_STAGE_TIMING
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_barrier" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (tbuf) { tbuf[3 * stage] = t0; tbuf[3 * stage + 1] = t1; tbuf[3 * stage + 2] = t2; }
    return;
#endif
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // stages +1 / +2 landed for everybody; slot `slot` is free
  }
  __device__ __forceinline__ void stage_refill() {
    if constexpr (STAGGER) {
      refill_stage = stage + RING_SLOTS < nstages ? stage + RING_SLOTS : -1;
      refill_slot = slot;
      // operands of the staggered request, once per stage (4 SGPRs live instead of 7 recomputed at each of the four sites: the
      // fused kernels run out of SGPRs otherwise and spill them into VGPRs).  Addresses of the MIDDLE of the wave's 8 KiB: the
      // instruction's signed 13-bit offset (-4096 .. +3072) reaches all eight pieces from one base, for the global and the
      // LDS address alike.
      r_sel = refill_stage >= 0 ? wave : 7;
      r_g = gsrc + (size_t)(refill_stage >= 0 ? refill_stage : 0) * STAGE_BYTES + 4 * FRAG_BYTES;
      r_lds = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + slot * STAGE_BYTES + wave_dst)) + 4 * FRAG_BYTES;
      // wave 0's turn is right here, behind the barrier, and in EXACTLY the statement form of the burst issue it replaces (two
      // "memory"-clobbering statements with its operand lists, issue_wave0): with any other form at this point -- or none, as
      // in the no-weight-loads ablation build -- the fused forward kernels allocate differently and spill ~430 VGPRs; hipcc's
      // schedule of that 250 KB straight-line kernel hangs on such details
      if (stage + RING_SLOTS < nstages) issue_wave0(stage + RING_SLOTS, slot);
    } else {
      if (stage + RING_SLOTS < nstages) issue(stage + RING_SLOTS, slot);
    }
    slot = slot == RING_SLOTS - 1 ? 0 : slot + 1;
    ++stage;
    set_offsets();
  }
  // STAGGER: called by kgroup at the top of k-group Q * KPS / 4 of a stage (Q = 1..3).  The slot being refilled was consumed in the
  // stage before (freed by its barrier); the pieces have the rest of this stage (>= 2 000 matrix clocks) to land before its
  // vmcnt(0).  The wave test is a branch INSIDE the asm block: a C++ `if` around the issue would split the stage's straight-line
  // block and cost the fused kernels 9-14 spilled VGPRs.
  template <int Q>
  __device__ __forceinline__ void staggered_issue() {
    constexpr int q = Q;
#ifdef ANERF_EXP_NOGLDS
    return;
#endif
    // Three properties of this asm statement decide between 0 and > 1000 spilled VGPRs in the fused kernels (tested one by one,
    // compile-only): NO "memory" clobber -- the pieces go to the slot consumed in the stage before, which nothing the compiler
    // knows of touches until the next stage barrier, and asm volatile statements keep their order among themselves --; NO
    // "scc" clobber (an "scc" clobber in the middle of a stage's region alone, on an asm that does nothing else, costs 700+
    // spills); and NO "vcc" clobber (fine in the backward kernels, but the fused forward kernels keep lane masks of the
    // encoding in VCC / SGPR pairs and run out of SGPRs).  So the wave test is a VALU compare into a scratch SGPR pair that
    // becomes EXEC: s_cbranch_execz skips the requests, EXEC is restored behind them.
    unsigned long long save_exec, is_mine;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "v_cmp_eq_u32_e64 %1, %2, %3\n\t"          // all lanes: this wave's turn?  (VALU compare into an SGPR pair: no SCC, no VCC)
                 "s_nop 3\n\t"                              // (inline asm is invisible to the hazard recognizer: VALU-written SGPR -> SALU)
                 "s_mov_b64 exec, %1\n\t"
                 "s_cbranch_execz 1f\n\t"
                 "s_mov_b32 m0, %4\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:-4096\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:-3072\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:-2048\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:-1024\n\t"
                 "global_load_lds_dwordx4 %5, %6\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:1024\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:2048\n\t"
                 "global_load_lds_dwordx4 %5, %6 offset:3072\n"
                 "1:\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(save_exec), "=&s"(is_mine) : "s"(r_sel), "n"(Q), "s"(r_lds), "v"(lane16), "s"(r_g) : "m0");
    if (q == 3) r_sel = 7;
    if (q == 3) refill_stage = -1;
  }
  // end of the stage being consumed: its slot is refilled with stage+3
  __device__ __forceinline__ void end_stage() {
    if constexpr (ASMDMA) {   // hidden-DMA pipes never drain the LDS queue at the stage barrier (see end_stage_raw)
      end_stage_raw();
      return;
    }
#ifdef ANERF_EXP_STAGE_TIMING
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stages +1 / +2 has landed
#ifdef ANERF_EXP_STAGE_TIMING
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
#ifndef ANERF_EXP_NOBARRIER   // ablation build only: results are wrong without the barrier
    __syncthreads();                                    // everybody's has; nobody reads slot `slot` any more
#endif
#ifdef ANERF_EXP_STAGE_TIMING
    if (tbuf) { tbuf[3 * stage] = t0; tbuf[3 * stage + 1] = t1; tbuf[3 * stage + 2] = __builtin_amdgcn_s_memtime(); }
#endif
    if (stage + RING_SLOTS < nstages) issue(stage + RING_SLOTS, slot);
    slot = slot == RING_SLOTS - 1 ? 0 : slot + 1;
    ++stage;
    set_offsets();
  }
};
// split-bf16 kernels.  Round 2, in three steps on the render kernel (stage = 48 MFMAs = 1 536 matrix clocks; clocks from
// tools/stage_timing.py B3=1, throughput from tools/microbench_mlp.py --b3):
//   builtin LDS-DMA + __syncthreads, operands split next to their MFMAs       stage 2 440   413-418 TFLOP/s algorithmic
//   + hidden DMA / raw barrier alone: the barrier's LDS drain (420 clocks) moves in front of the MFMAs, 15-21 VGPRs spill   419
//   + a layer's operands pre-split in one block (take_split, anerf_mlp_b3.hip), visible DMA                2 240   418-424
//   + both                                                                                                 2 090   435-437
// (a full k-step of fragment read-ahead on top, 16 reads at once instead of kstep's interleaved groups: 2 330, slower).
// The other split-bf16 kernels gain 2-3 % from the hidden DMA as well (training step 9.34 -> 9.10 ms).
// ANERF_EXP_B3_VISIBLE_DMA rebuilds the round-1 pipe.
#ifdef ANERF_EXP_B3_VISIBLE_DMA
using Pipe3 = Pipe3T<false>;
#else
using Pipe3 = Pipe3T<true>;
#endif
// fp32 forward kernels.  Round 3 built the staggered refill into them (it needs wave 0's share in the burst issue's exact statement
// form at the barrier, issue_wave0, and the other waves' asm without "memory" / "scc" / "vcc" clobbers: any other form spills
// hundreds of VGPRs) and measured it same-box against the burst (tools/microbench_mlp.py, 65 536 rays x 64 samples): fused 143.1-
// 143.6 TFLOP/s with the burst, 140.7-141.3 staggered (sites at k-groups 1/2/3, 1/2/2 or 1/1/1 alike); pre-encoded 143.6 vs 139.9-
// 141.1.  The probe's 2.6 % does not survive the real kernels: three more asm sites per stage (+10 % code in a kernel that is
// 250 KB of straight-line code), a VALU compare + EXEC juggling at each inside the fp32 MFMA stream.  -DANERF_EXP_STAGGER rebuilds it.
#ifdef ANERF_EXP_STAGGER
using Pipe3F = Pipe3T<true, true>;
#else
using Pipe3F = Pipe3T<true, false>;
#endif
// fp32 backward kernels (k_mlp_bwd, k_mlp_bwd_in): they compile with the staggered refill without a spill, but run 4 % SLOWER
// with it (k_mlp_bwd 1.71 -> 1.79 / 2.26 -> 2.33 ms per launch of the 3072-ray step, k_mlp_bwd_in 1.05 -> 1.08): their k-groups
// carry ordinary vector loads (mask / operand quads) whose compiler-counted `s_waitcnt vmcnt(n)` also count the hidden DMA
// pieces issued behind them, so a staggered burst in the middle of a stage turns those waits into waits for the DMA.  The
// probe's 2.6 % (8 601 -> 8 375 clocks per stage) stays on the table: -DANERF_EXP_STAGGER builds the variant.
#ifdef ANERF_EXP_STAGGER
using Pipe3B = Pipe3T<true, true>;
#else
using Pipe3B = Pipe3T<true, false>;
#endif

// acc[nb][r] <- bias[n(nb,r,h)] from the LDS copy of the natural-order bias vector (bias_h = vector + 4h)
template <int NB>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[NB], const float* bias_h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias_h + 32 * nb + 8 * q);
      acc[nb][4 * q + 0] = b.x;
      acc[nb][4 * q + 1] = b.y;
      acc[nb][4 * q + 2] = b.z;
      acc[nb][4 * q + 3] = b.w;
    }
}

// in-place ReLU of a finished layer (one VALU pass; measured 0.7 % of a layer, vs 4.3 % when the max is
// interleaved with the consuming MFMAs -- tools/probe/mfma_probe2.hip)
// max(x, 0) as ONE instruction: v_max_i32 on the bit pattern (negative floats are negative integers; -0 -> +0).
// fmaxf compiles to two -- hipcc first canonicalises the MFMA result (v_max x, x) because it cannot prove it is not a
// signalling NaN.
__device__ __forceinline__ float relu_i(float x) {
  const int i = __float_as_int(x);
  return __int_as_float(i > 0 ? i : 0);
}
template <int NB>
__device__ __forceinline__ void relu_pass(f32x16 (&acc)[NB]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = relu_i(acc[nb][r]);
}
// fp32 forward: a finished layer's accumulator set -> its (ReLU'd) values as 128 VGPRs = the next layer's B operands, in ONE
// fenced VALU pass.  Measured per stage with s_memtime (tools/stage_timing.py): when the compiler is left to apply the
// ReLU where the values are consumed (it did so for every second hidden layer: v_accvgpr_read + 2 v_max per operand in
// front of its k-group) each of those instructions costs ~14 clocks inside the MFMA stream, against 4-8 in a block.
template <int NB, bool RELU>
__device__ __forceinline__ void take(float (&hb)[128], const f32x16 (&acc)[NB]) {
  __builtin_amdgcn_sched_barrier(0);   // (the first reads would otherwise be hoisted between the layer's last MFMAs)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) hb[16 * nb + r] = RELU ? relu_i(acc[nb][r]) : acc[nb][r];
  // pin the pass here: the empty asm defines each value at this point (MachineSink would otherwise move every v_max into
  // the stage block that consumes it -- each stage is a basic block of its own), the fence keeps the scheduler from mixing
  // the pass with the next layer's first MFMAs
#pragma unroll
  for (int i = 0; i < 16 * NB; ++i) asm volatile("" : "+v"(hb[i]));
  __builtin_amdgcn_sched_barrier(0);
}
template <int NB>
__device__ __forceinline__ float head_dot_v(const float (&hb)[128], const float* wrow_h) {
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wrow_h + 32 * nb + 8 * q);
      s = fmaf(hb[16 * nb + 4 * q + 0], w.x, s);
      s = fmaf(hb[16 * nb + 4 * q + 1], w.y, s);
      s = fmaf(hb[16 * nb + 4 * q + 2], w.z, s);
      s = fmaf(hb[16 * nb + 4 * q + 3], w.w, s);
    }
  return s + __shfl_xor(s, 32);
}

// dot of the lane's 16*NB activation values with a natural-order weight row (LDS), summed over both halves
template <int NB>
__device__ __forceinline__ float head_dot(const f32x16* acc, const float* wrow_h) {
  float s = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(wrow_h + 32 * nb + 8 * q);
      s = fmaf(acc[nb][4 * q + 0], w.x, s);
      s = fmaf(acc[nb][4 * q + 1], w.y, s);
      s = fmaf(acc[nb][4 * q + 2], w.z, s);
      s = fmaf(acc[nb][4 * q + 3], w.w, s);
    }
  return s + __shfl_xor(s, 32);
}

// One k-group (8 contraction indices: 4 from each lane half = 4 MFMA k-steps, "quarters") against NB 32-row feature blocks.
// kg: k-group index relative to the segment (layer) start; last: final k-group of the segment (segments are padded to
// whole stages).  Both fold at compile time.
// Fragment pipeline: the weight image is quarter-major (anerf_capi.hip): fragments q * (NB/4) .. + NB/4 - 1 of a k-group
// carry exactly the A operands of quarter q, so they are dead as soon as that quarter's MFMAs are issued.  pipe.a[] is a
// two-quarter window (4 fragment registers sets = 16 VGPRs): when a call starts it holds quarters 0 and 1 of this k-group;
// after quarter q's MFMAs its slot (q & 1) is refilled with quarter q + 2 -- of this k-group (q = 0, 1) or quarters 0 / 1 of
// the NEXT one (q = 2, 3) -- i.e. every read is issued one full quarter (512 MFMA cycles) before its data is used.  The
// next k-group lies in the next ring slot when this one ends a stage; that slot is complete by the ring invariant, so the
// reads may run ahead of the barrier.  (Round 1 read all NB fragments between two k-groups and paid the LDS latency of
// four lock-step waves x 8 KiB there, ~420 times per tile; a full-k-group window (32 VGPRs) made the render kernel spill.)
// AUTO_END = false: the caller ends the stage itself (pipe.stage_rendezvous() ... pipe.stage_refill()) after the k-group
// for which `boundary` holds.
// NBU <= NB: only the first NBU output blocks are multiplied (a narrow last column group of k_mlp_bwd_in: the stage keeps its
// NB-block layout and fragment reads, the MFMAs of the blocks past the edge are simply not issued).
template <int NB, class PIPE, bool AUTO_END = true, int NBU = NB>
__device__ __forceinline__ void kgroup(PIPE& pipe, f32x16 (&acc)[NB], int kg, bool first, bool last, float b0, float b1,
                                       float b2, float b3) {
  static_assert(NB == 8 || NB == 4, "kgroup: 8 or 4 output blocks");
  static_assert(NBU >= 1 && NBU <= NB, "kgroup: used blocks");
  constexpr int KPS = STAGE_FRAGS / NB;  // k-groups per stage
  constexpr int FPQ = NB / 4;            // fragments per quarter
  const int ks = kg % KPS;
  const bool boundary = ks == KPS - 1 || last;
  const char* csrc = pipe.smem + pipe.cur + ks * NB * FRAG_BYTES;                                   // this k-group
  const char* nsrc = boundary ? pipe.smem + pipe.nxt : csrc + NB * FRAG_BYTES;                      // the next one
  (void)first;
  // STAGGER: wave w requests the stage's refill in front of k-group w * KPS / 4 of the stage (see Pipe3T): between k-groups, not
  // between the quarters of one (the probe's other staggered form, same gain).
  if constexpr (PIPE::kStagger) {
    // wave 0 requested its share right behind the stage barrier (stage_refill); wave Q's site is the top of k-group Q * KPS / 4.
    // A segment's last stage may be short (layer 0 / 5: 54 / 86 k-groups of 4 per stage): its final k-group (`last`) is also the
    // site of every wave whose own one does not exist in this stage.
    constexpr int QK = KPS / 4;
#ifndef ANERF_STAGGER_S1      // site (in quarters of a stage) of waves 1, 2, 3: experiment knobs
#define ANERF_STAGGER_S1 1
#define ANERF_STAGGER_S2 2
#define ANERF_STAGGER_S3 3
#endif
    constexpr int K1 = ANERF_STAGGER_S1 * QK, K2 = ANERF_STAGGER_S2 * QK, K3 = ANERF_STAGGER_S3 * QK;
    if (ks == K1 || (last && ks < K1)) pipe.template staggered_issue<1>();
    if (ks == K2 || (last && ks < K2)) pipe.template staggered_issue<2>();
    if (ks == K3 || (last && ks < K3)) pipe.template staggered_issue<3>();
  }
  const float b[4] = {b0, b1, b2, b3};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int sl = (q & 1) * FPQ;
#pragma unroll
    for (int nb = 0; nb < NBU; ++nb)
      acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(pipe.a[sl + (nb >> 2)][nb & 3], b[q], acc[nb], 0, 0, 0);
    const char* src = q < 2 ? csrc + (q + 2) * FPQ * FRAG_BYTES : nsrc + (q - 2) * FPQ * FRAG_BYTES;
#pragma unroll
    for (int g = 0; g < FPQ; ++g) pipe.a[sl + g] = *reinterpret_cast<const f32x4*>(src + g * FRAG_BYTES);
    // fence (only VALU / SALU may cross): left alone, the scheduler sinks these reads to just in front of their first use
    __builtin_amdgcn_sched_barrier(0x6);
  }
  if constexpr (AUTO_END) {
    if (boundary) pipe.end_stage_raw();
  }
}

// one float4 of a saved-activation row (ablation build, tools/ablate.sh: NOSAVE drops the store)
__device__ __forceinline__ void save_quad(float* dst, f32x4 q) {
#if defined(ANERF_EXP_NOSAVE)
  (void)dst; (void)q;
#else
  *reinterpret_cast<f32x4*>(dst) = q;
#endif
}
constexpr int SAVE_QUAD_STRIDE = 8;

// 32 k-groups whose B operands are the previous layer's 256 outputs, read straight from its accumulator set
// `prev` (bias added by the accumulator init, ReLU already applied in place): no copy, no VALU in the MFMA stream.
// SAVE (training forward): `prev` is also written to its row of the saved-activation plane by the layer that CONSUMES
// it, one float4 quad per k-group (quad j = the k-group's own B operands = features 8j+4h..+3 -> save_row_h + 8j).
// A CU's vector-memory path moves ~10 B/clk (MI355X_MICROARCH.md), i.e. ~100 cycles per 1 KiB wave-store, and a wave
// that issues into a busy path stalls -- with one wave per SIMD that is lost MFMA time.  Measured on the 3072 x 80
// training forward (tools/microbench_train_fwd.py; no stores at all: 3.40 ms): 32 stores after the layer 3.85 ms, a
// stage's 4 stores behind its barrier 3.95 ms, one store per k-group 3.65 ms; waiting for the stores is NOT the cost
// (a counted vmcnt that leaves them in flight changes nothing), neither is their scatter (lane-linear 1 KiB stores: same).
template <int NB, int KG0, bool SAVE = false, class PIPE>
__device__ __forceinline__ void hidden_part(PIPE& pipe, f32x16 (&acc)[NB], const f32x16 (&prev)[8], bool first,
                                            bool last, float* __restrict__ save_row_h = nullptr) {
  constexpr int KPS = STAGE_FRAGS / NB;
#pragma unroll
  for (int kg = 0; kg < 32; ++kg) {
    if constexpr (SAVE) {
      const f32x4 o = {prev[kg >> 2][4 * (kg & 3) + 0], prev[kg >> 2][4 * (kg & 3) + 1], prev[kg >> 2][4 * (kg & 3) + 2],
                       prev[kg >> 2][4 * (kg & 3) + 3]};
      save_quad(save_row_h + SAVE_QUAD_STRIDE * kg, o);
      __builtin_amdgcn_sched_barrier(0);   // in front of this k-group's MFMAs (the scheduler would sink it behind them)
    }
    kgroup<NB>(pipe, acc, KG0 + kg, first && kg == 0, last && kg == 31, prev[kg >> 2][4 * (kg & 3) + 0],
               prev[kg >> 2][4 * (kg & 3) + 1], prev[kg >> 2][4 * (kg & 3) + 2], prev[kg >> 2][4 * (kg & 3) + 3]);
  }
}

// the same with the previous layer's outputs held as 128 VGPRs (take<>): prev[4 kg .. 4 kg + 3] are k-group kg's operands
template <int NB, int KG0, bool SAVE = false, class PIPE>
__device__ __forceinline__ void hidden_part_v(PIPE& pipe, f32x16 (&acc)[NB], const float (&prev)[128], bool first,
                                              bool last, float* __restrict__ save_row_h = nullptr) {
#pragma unroll
  for (int kg = 0; kg < 32; ++kg) {
    if constexpr (SAVE) {
      const f32x4 o = {prev[4 * kg + 0], prev[4 * kg + 1], prev[4 * kg + 2], prev[4 * kg + 3]};
      save_quad(save_row_h + SAVE_QUAD_STRIDE * kg, o);
      __builtin_amdgcn_sched_barrier(0);   // in front of this k-group's MFMAs (the scheduler would sink it behind them)
    }
    kgroup<NB>(pipe, acc, KG0 + kg, first && kg == 0, last && kg == 31, prev[4 * kg + 0], prev[4 * kg + 1],
               prev[4 * kg + 2], prev[4 * kg + 3]);
  }
}

struct MlpArgs {
  const float* packed;
  const float* aux;
  const float* rays;
  const float* z;
  const float* skts;
  const float* cam;
  const float* codes;
  const float* cut_v;
  const float* cut_d;
  const float* x;  // PRE
  const float* pnoise;   // [P][3] additive offset of the sample points (ray_noise_std > 0, raycasters.py:660) or nullptr
  float* raw;
  // TRAIN: saved activations, row-major planes with Ppad rows (rows >= P are never written)
  float* save_h;   // [8][Ppad][256]  h0..h7 (post-ReLU)
  float* save_f;   // [Ppad][256]     feature (no activation)
  float* save_g;   // [Ppad][128]     view-layer output (post-ReLU)
  float* save_x;   // [Ppad][432]     x in stream column order
  float* save_u;   // [Ppad][UW]      view inputs (D, code) in stream column order
  long long P;
  long long Ppad;
  long long skt_stride;
  int S, N, ray_stride, n_codes, x_width, nstages;
  float tau_v, tau_d;
  const float* tau_dev;   // ABI revision 6, TRAIN kernels only: {tau_v, tau_d} in device memory (AnerfStepBlock) or nullptr = the two above
  int gate_bones;   // --cutoff_bones: the bone-direction block r is gated by the distance gate as well (raycasters.py:54-57)
#ifdef ANERF_EXP_STAGE_TIMING
  unsigned long long* tbuf;   // debug build only: per-stage clocks (tools/stage_timing.py)
#endif
};

}  // namespace anerf
