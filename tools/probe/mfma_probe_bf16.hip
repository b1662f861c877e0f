// mfma_probe_bf16.hip -- where is the roof of the split-bf16 render kernel (k_mlp_fwd_b3)?  Timing experiment only.
// The hidden-layer stage of that kernel = 48 x v_mfma_f32_32x32x16_bf16 per wave (2 k-steps x 8 blocks x 3 products, 1 536
// matrix clocks), fed by 32 ds_read_b128 per wave (the (hi, lo) fragment pairs: 32 KiB per wave, the same 32 KiB for all four
// waves) and by the L2 -> LDS weight stream (32 KiB per stage and workgroup).  Variants, same launch shape as the kernel
// (256 threads, 1 workgroup per CU through a 131 KiB LDS allocation, one wave per SIMD):
//   V0  MFMAs only, operands in registers                      -> the matrix pipe's own sustained rate and the clock it holds
//   V1  + the stage's 32 fragment reads (LDS -> registers)     -> + LDS read traffic: 4 waves x 32 KiB per stage
//   V2  + the LDS-DMA weight stream (3-slot ring, one barrier per stage) -> the kernel's stage without any VALU work
// Reported per variant: ms, executed bf16 TFLOP/s, matrix clocks per stage at the measured clock, GHz held (s_memtime, a
// 100 MHz-independent core-clock counter, against s_memrealtime, 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe_bf16.hip -o /tmp/mfma_probe_bf16 && /tmp/mfma_probe_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

constexpr int STAGE_BYTES = 32768, FRAG = 1024, SLOTS = 3, SLOTS4 = 4;

// DMA: 0 none, 1 compiler-visible builtin + __syncthreads, 2 hidden (inline asm, as Pipe3T<true>) + raw "vmcnt(0); s_barrier"
// READS: 0 none (operands stay in registers), 1 a k-step's 16 fragments read right in front of its MFMAs (compiler-scheduled),
//        2 double-buffered: the NEXT k-step's 16 fragments are requested before the current k-step's 24 MFMAs (64 VGPRs of read-ahead),
//        3 the kernel's grouping: 4 blocks (8 fragments) ahead, requested between the head (4 MFMAs) and tail (8) of a group
template <int DMA, int READS, int NOMM = 0>
__global__ __launch_bounds__(256) void k_probe(const char* __restrict__ wstream0, long long stream_bytes, int stages, float* out,
                                               unsigned long long* clocks, int replicas, long long replica_stride, int stage_skew) {
  // replicas > 1: workgroup i streams copy (i % replicas) of the image (another address range -> other L2 channels at any instant);
  // stage_skew: workgroup i starts its walk (i % 8) * stage_skew stages into the image
  const char* wstream = wstream0 + (long long)(blockIdx.x % replicas) * replica_stride;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < SLOTS * STAGE_BYTES / 4; i += 256) ((unsigned*)smem)[i] = 0x3c003c00u + (i & 255);   // small bf16 values
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  bf16x8 bh, bl;
  f32x4 fa[16], fb[16];     // fragment buffers: [2 b] = hi of block b, [2 b + 1] = lo
#pragma unroll
  for (int e = 0; e < 8; ++e) { bh[e] = (short)(0x3c00 + lane); bl[e] = (short)(0x3800 + e); }
#pragma unroll
  for (int i = 0; i < 16; ++i) fa[i] = fb[i] = f32x4{1.f, 1.f, 1.f, 1.f};
  const unsigned lane16 = lane * 16;
  auto issue = [&](long long goff, int sl) __attribute__((always_inline)) {
    const char* g = wstream + goff + wave * 8 * FRAG;
    char* l = smem + sl * STAGE_BYTES + wave * 8 * FRAG;
    if (DMA == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(g + i * FRAG + lane16), (lds_ptr_t)(l + i * FRAG), 16, 0, 0);
    } else if (DMA >= 2 && DMA <= 4) {
      const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)l);
#pragma unroll
      for (int half = 0; half < (DMA == 3 ? 1 : 2); ++half)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072"
                     :: "s"(lds0 + half * 4 * FRAG), "v"(lane16), "s"(g + half * 4 * FRAG) : "memory", "m0");
    }
  };
  // one piece (1 KiB per wave) of a stage, hidden from the compiler
  auto piece = [&](long long goff, int sl, int i) __attribute__((always_inline)) {
    const char* g = wstream + goff + wave * 8 * FRAG + i * FRAG;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + sl * STAGE_BYTES + wave * 8 * FRAG + i * FRAG));
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(lds0), "v"(lane16), "s"(g) : "memory", "m0");
  };
  auto rd = [&](f32x4 (&f)[16], const char* base, int ks, int b0, int nb) __attribute__((always_inline)) {
#pragma unroll
    for (int b = b0; b < b0 + nb; ++b) {
      f[2 * b] = *reinterpret_cast<const f32x4*>(base + ((ks * 8 + b) * 2) * FRAG);
      f[2 * b + 1] = *reinterpret_cast<const f32x4*>(base + ((ks * 8 + b) * 2 + 1) * FRAG);
    }
  };
  auto mm = [&](const f32x4 (&f)[16], int b) __attribute__((always_inline)) {
    if (NOMM) {     // no matrix work: keep the fragment loads alive with one VALU op each (how long do the LDS accesses alone take?)
      acc[b][0] += f[2 * b][0];
      acc[b][1] += f[2 * b + 1][0];
      return;
    }
    const bf16x8 ah = __builtin_bit_cast(bf16x8, f[2 * b]), al = __builtin_bit_cast(bf16x8, f[2 * b + 1]);
    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[b], 0, 0, 0);
    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[b], 0, 0, 0);
    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[b], 0, 0, 0);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int slot = 0;
  long long goff = 0;
  if (DMA && DMA != 5 && DMA < 9) {
    const long long g0 = (long long)((blockIdx.x % 8) * stage_skew) * STAGE_BYTES;
    issue(g0, 0); issue(g0 + STAGE_BYTES, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(g0 + 2LL * STAGE_BYTES, 2);
    goff = g0 + 3LL * STAGE_BYTES;
  }
  if (DMA == 6) {
    // 4-slot ring, two stages of lead: stage st consumes slot st % 4; during stage st the pieces of stage st + 3 go to slot
    // (st + 3) % 4 one at a time (behind every 6th MFMA); at the end of stage st only the pieces issued DURING st may remain
    // pending (counted vmcnt(8)): those of stage st + 2 have had a whole extra stage.  TIMING ONLY -- round 1 found counted waits
    // on LDS-DMA unsafe (stale data); this measures what such a pipe would be worth.
    for (int s0 = 0; s0 < 3; ++s0)
      for (int i = 0; i < 8; ++i) piece((long long)s0 * STAGE_BYTES, s0, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    long long go = 3LL * STAGE_BYTES;
    int sl = 0;
    if (READS == 3) rd(fa, smem + lane16, 0, 0, 4);
    if (READS == 4) {
      // two groups of read-ahead: three 8-fragment buffers g0 g1 g2 rotate over the 4 groups of a stage (12 groups = 3 stages
      // per full rotation, so the loop body below is written for 3 consecutive stages)
      f32x4 g[3][8];
      auto rdg = [&](f32x4 (&d)[8], const char* base, int grp) __attribute__((always_inline)) {   // group grp of a stage: k-step grp >> 1, blocks 4 (grp & 1) ..
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int b = 4 * (grp & 1) + j, ks = grp >> 1;
          d[2 * j] = *reinterpret_cast<const f32x4*>(base + ((ks * 8 + b) * 2) * FRAG);
          d[2 * j + 1] = *reinterpret_cast<const f32x4*>(base + ((ks * 8 + b) * 2 + 1) * FRAG);
        }
      };
      auto mmg = [&](const f32x4 (&d)[8], int j, int b) __attribute__((always_inline)) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, d[2 * j]), al = __builtin_bit_cast(bf16x8, d[2 * j + 1]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[b], 0, 0, 0);
      };
      rdg(g[0], smem + lane16, 0);
      rdg(g[1], smem + lane16, 1);
      for (int st = 0; st < stages; st += 3) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {          // global group index within the 3-stage body
          const int stg = u >> 2, grp = u & 3, buf = u % 3;
          const int csl = (sl + stg) & 3, fill = (sl + stg + 3) & 3;
          const int u2 = u + 2, stg2 = u2 >> 2, grp2 = u2 & 3;
          const char* src2 = smem + ((sl + stg2) & 3) * STAGE_BYTES + lane16;
          const int b0 = 4 * (grp & 1);
          (void)csl;
          mmg(g[buf], 0, b0);
          __builtin_amdgcn_sched_barrier(0);
          rdg(g[(u + 2) % 3], src2, grp2);
          piece(go, fill, 2 * grp);
          __builtin_amdgcn_sched_barrier(0);
          mmg(g[buf], 1, b0 + 1);
          __builtin_amdgcn_sched_barrier(0);
          piece(go, fill, 2 * grp + 1);
          __builtin_amdgcn_sched_barrier(0);
          mmg(g[buf], 2, b0 + 2);
          mmg(g[buf], 3, b0 + 3);
          if (grp == 3) {
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            go += STAGE_BYTES;
            if (go + STAGE_BYTES > stream_bytes) go = 0;
          }
        }
        sl = (sl + 3) & 3;
      }
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[b][r];
      out[blockIdx.x * 256 + threadIdx.x] = s;
      if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
      return;
    }
    for (int st = 0; st < stages; ++st) {
      const char* cur = smem + sl * STAGE_BYTES + lane16;
      const int nsl = (sl + 1) & 3, fill = (sl + 3) & 3;
      const char* nxt = smem + nsl * STAGE_BYTES + lane16;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int b0 = (gq & 1) * 4;
        if (gq < 2) mm(fa, b0); else mm(fb, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (READS == 3) {
          if (gq == 0) rd(fa, cur, 0, 4, 4);
          else if (gq == 1) rd(fb, cur, 1, 0, 4);
          else if (gq == 2) rd(fb, cur, 1, 4, 4);
          else rd(fa, nxt, 0, 0, 4);
        }
        piece(go, fill, 2 * gq);
        __builtin_amdgcn_sched_barrier(0);
        if (gq < 2) mm(fa, b0 + 1); else mm(fb, b0 + 1);
        __builtin_amdgcn_sched_barrier(0);
        piece(go, fill, 2 * gq + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (gq < 2) { mm(fa, b0 + 2); mm(fa, b0 + 3); } else { mm(fb, b0 + 2); mm(fb, b0 + 3); }
      }
      asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
      go += STAGE_BYTES;
      if (go + STAGE_BYTES > stream_bytes) go = 0;
      sl = nsl;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
    return;
  }
  if (DMA == 8) {
    // weights through REGISTERS instead of LDS-DMA: global_load_dwordx4 into 8 staging registers during stage s, ds_write_b128
    // into the freed slot at the top of stage s + 1 (LDS writes at the full 128 B/clk instead of the DMA's ~57)
    f32x4 stg[8];
    long long go = 3LL * STAGE_BYTES;
    int sl = 0;
    const char* gw = wstream + wave * 8 * FRAG + lane16;
#pragma unroll
    for (int i = 0; i < 8; ++i) stg[i] = *reinterpret_cast<const f32x4*>(gw + go + i * FRAG);
    if (READS == 3) rd(fa, smem + lane16, 0, 0, 4);
    for (int st = 0; st < stages; ++st) {
      const char* cur = smem + sl * STAGE_BYTES + lane16;
      const int nsl = sl == 2 ? 0 : sl + 1, fill = sl == 0 ? 2 : sl - 1;
      const char* nxt = smem + nsl * STAGE_BYTES + lane16;
      // top of the stage: last stage's staged registers -> the slot freed by the barrier; request the next stage's
      char* dstl = smem + fill * STAGE_BYTES + wave * 8 * FRAG + lane16;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(dstl + i * FRAG) = stg[i];
      go += STAGE_BYTES;
      if (go + STAGE_BYTES > stream_bytes) go = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) stg[i] = *reinterpret_cast<const f32x4*>(gw + go + i * FRAG);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int b0 = (gq & 1) * 4;
        if (gq < 2) mm(fa, b0); else mm(fb, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (READS == 3) {
          if (gq == 0) rd(fa, cur, 0, 4, 4);
          else if (gq == 1) rd(fb, cur, 1, 0, 4);
          else if (gq == 2) rd(fb, cur, 1, 4, 4);
          else rd(fa, nxt, 0, 0, 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (gq < 2) { mm(fa, b0 + 1); mm(fa, b0 + 2); mm(fa, b0 + 3); }
        else { mm(fb, b0 + 1); mm(fb, b0 + 2); mm(fb, b0 + 3); }
      }
      __syncthreads();
      sl = nsl;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[b][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += stg[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
    return;
  }
  if (DMA == 9 || DMA == 10) {
    // PARTIAL LDS BYPASS (round 6): 8 of the stage's 32 fragments -- the last group, k-step 1 / blocks 4..7 -- do not
    // go through LDS at all: every wave fetches them itself with 8 x global_load_dwordx4 (L2 -> A-operand VGPRs, 32 staging
    // registers), issued right behind the stage barrier IN FRONT of the LDS-DMA pieces, so that `s_waitcnt vmcnt(6)` before the
    // group's MFMAs waits for exactly these loads (vector-memory operations return in issue order).  The LDS-DMA ring carries the
    // other 24 fragments (6 pieces per wave instead of 8); LDS work per stage: 24 x 4 KiB of reads + 24 KiB of DMA writes instead
    // of 32 x 4 + 32.  DMA == 10: the same with the direct loads' group FIRST in the NEXT stage (a whole stage of latency cover,
    // 64 staging registers: does the L2 latency matter?).
    f32x4 stg[8];
    auto issue6 = [&](long long go, int sl) __attribute__((always_inline)) {
      const char* g = wstream + go + wave * 6 * FRAG;
      const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + sl * STAGE_BYTES + wave * 6 * FRAG));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:3072"
                   :: "s"(lds0), "v"(lane16), "s"(g) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:1024"
                   :: "s"(lds0 + 4 * FRAG), "v"(lane16), "s"(g + 4 * FRAG) : "memory", "m0");
    };
    f32x4 stg2[8];                                         // DMA == 10: the second staging set
    auto direct_to = [&](f32x4 (&d)[8], long long go) __attribute__((always_inline)) {
      const char* g = wstream + go + 24 * FRAG;           // the same 8 KiB for all four waves (they hit in L1 / L2)
      asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                   "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                   "global_load_dwordx4 %3, %4, %5 offset:3072"
                   : "=v"(d[0]), "=v"(d[1]), "=v"(d[2]), "=v"(d[3]) : "v"(lane16), "s"(g) : "memory");
      asm volatile("global_load_dwordx4 %0, %4, %5\n\t"
                   "global_load_dwordx4 %1, %4, %5 offset:1024\n\t"
                   "global_load_dwordx4 %2, %4, %5 offset:2048\n\t"
                   "global_load_dwordx4 %3, %4, %5 offset:3072"
                   : "=v"(d[4]), "=v"(d[5]), "=v"(d[6]), "=v"(d[7]) : "v"(lane16), "s"(g + 4 * FRAG) : "memory");
    };
    auto direct = [&](long long go) __attribute__((always_inline)) { direct_to(stg, go); };
    auto mms_of = [&](const f32x4 (&d)[8], int j, int b) __attribute__((always_inline)) {
      if (NOMM) { acc[b][0] += d[2 * j][0]; acc[b][1] += d[2 * j + 1][0]; return; }
      const bf16x8 ah = __builtin_bit_cast(bf16x8, d[2 * j]), al = __builtin_bit_cast(bf16x8, d[2 * j + 1]);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[b], 0, 0, 0);
    };
    auto mms = [&](int j, int b) __attribute__((always_inline)) { mms_of(stg, j, b); };
    const long long wrap = stream_bytes - STAGE_BYTES;
    if (DMA == 10) {
      // one WHOLE stage of lead for the direct loads: two staging sets alternate (64 VGPRs -- 9 more than k_mlp_fwd_b3 has free);
      // the set a stage's last group consumes was requested at the top of the PREVIOUS stage and is complete at that stage's
      // closing vmcnt(0): no counted wait in front of the group at all
      issue6(0, 0); issue6(STAGE_BYTES, 1);
      direct_to(stg, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      direct_to(stg2, STAGE_BYTES);
      issue6(2LL * STAGE_BYTES, 2);
      long long go = 3LL * STAGE_BYTES, dgo = 2LL * STAGE_BYTES;
      int sl = 0;
      rd(fa, smem + lane16, 0, 0, 4);
      auto body = [&](f32x4 (&use)[8]) __attribute__((always_inline)) {
        const char* cur = smem + sl * STAGE_BYTES + lane16;
        const int nsl = sl == SLOTS - 1 ? 0 : sl + 1;
        const char* nxt = smem + nsl * STAGE_BYTES + lane16;
        mm(fa, 0);
        __builtin_amdgcn_sched_barrier(0);
        rd(fa, cur, 0, 4, 4);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa, 1); mm(fa, 2); mm(fa, 3);
        mm(fa, 4);
        __builtin_amdgcn_sched_barrier(0);
        rd(fb, cur, 1, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        mm(fa, 5); mm(fa, 6); mm(fa, 7);
        mm(fb, 0); mm(fb, 1); mm(fb, 2); mm(fb, 3);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+v"(use[0]), "+v"(use[1]), "+v"(use[2]), "+v"(use[3]), "+v"(use[4]), "+v"(use[5]), "+v"(use[6]), "+v"(use[7]));
        mms_of(use, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        rd(fa, nxt, 0, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        mms_of(use, 1, 5); mms_of(use, 2, 6); mms_of(use, 3, 7);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        direct_to(use, dgo);                       // this set again in two stages
        issue6(go, sl);
        go += STAGE_BYTES;
        if (go > wrap) go = 0;
        dgo += STAGE_BYTES;
        if (dgo > wrap) dgo = 0;
        sl = nsl;
      };
      for (int st = 0; st < stages; st += 2) {
        body(stg);
        body(stg2);
      }
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      float s = 0.f;
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[b][r];
      out[blockIdx.x * 256 + threadIdx.x] = s;
      if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
      return;
    }
    issue6(0, 0); issue6(STAGE_BYTES, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    direct(0);
    issue6(2LL * STAGE_BYTES, 2);
    long long go = 3LL * STAGE_BYTES, dgo = STAGE_BYTES;     // go: the stage the freed slot is refilled with; dgo: the NEXT stage's direct part
    int sl = 0;
    rd(fa, smem + lane16, 0, 0, 4);
    for (int st = 0; st < stages; ++st) {
      const char* cur = smem + sl * STAGE_BYTES + lane16;
      const int nsl = sl == SLOTS - 1 ? 0 : sl + 1;
      const char* nxt = smem + nsl * STAGE_BYTES + lane16;
      // group 0: k-step 0, blocks 0..3 (fa[0..7]); request group 1
      mm(fa, 0);
      __builtin_amdgcn_sched_barrier(0);
      rd(fa, cur, 0, 4, 4);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa, 1); mm(fa, 2); mm(fa, 3);
      // group 1: k-step 0, blocks 4..7; request group 2
      mm(fa, 4);
      __builtin_amdgcn_sched_barrier(0);
      rd(fb, cur, 1, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa, 5); mm(fa, 6); mm(fa, 7);
      // group 2: k-step 1, blocks 0..3; NO request: group 3 comes from the staging registers
      mm(fb, 0); mm(fb, 1); mm(fb, 2); mm(fb, 3);
      __builtin_amdgcn_sched_barrier(0);
      // group 3: k-step 1, blocks 4..7 straight from L2; request the NEXT stage's group 0
      asm volatile("s_waitcnt vmcnt(6)" : "+v"(stg[0]), "+v"(stg[1]), "+v"(stg[2]), "+v"(stg[3]), "+v"(stg[4]), "+v"(stg[5]), "+v"(stg[6]), "+v"(stg[7]) :: "memory");
      mms(0, 4);
      __builtin_amdgcn_sched_barrier(0);
      rd(fa, nxt, 0, 0, 4);
      __builtin_amdgcn_sched_barrier(0);
      mms(1, 5); mms(2, 6); mms(3, 7);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      direct(dgo);
      issue6(go, sl);
      go += STAGE_BYTES;
      if (go > wrap) go = 0;
      dgo += STAGE_BYTES;
      if (dgo > wrap) dgo = 0;
      sl = nsl;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[b][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
    return;
  }
  if (READS == 2) rd(fa, smem + lane16, 0, 0, 8);
  if (READS == 3) rd(fa, smem + lane16, 0, 0, 4);
  for (int st = 0; st < stages; ++st) {
    const char* cur = smem + slot * STAGE_BYTES + lane16;
    const int nslot = slot == SLOTS - 1 ? 0 : slot + 1;
    const char* nxt = smem + nslot * STAGE_BYTES + lane16;
    if (READS == 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        rd(fa, cur, ks, 0, 8);
#pragma unroll
        for (int b = 0; b < 8; ++b) mm(fa, b);
      }
    } else if (READS == 2) {
      rd(fb, cur, 1, 0, 8);                       // k-step 1 under k-step 0's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) mm(fa, b);
      __builtin_amdgcn_sched_barrier(0);
      rd(fa, nxt, 0, 0, 8);                       // next stage's k-step 0 (complete by the ring invariant) under k-step 1's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 8; ++b) mm(fb, b);
    } else if (READS == 3) {
      // groups of 4 blocks: [ks 0: b 0-3] [ks 0: b 4-7] [ks 1: b 0-3] [ks 1: b 4-7]; group g+1 is requested behind group g's first block
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ks = gq >> 1, b0 = (gq & 1) * 4;
        if (gq < 2) mm(fa, b0); else mm(fb, b0);
        __builtin_amdgcn_sched_barrier(0);
        // the NEXT group's fragments go to the other half of the buffer
        if (gq == 0) rd(fa, cur, 0, 4, 4);
        else if (gq == 1) rd(fb, cur, 1, 0, 4);
        else if (gq == 2) rd(fb, cur, 1, 4, 4);
        else rd(fa, nxt, 0, 0, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (gq < 2) { mm(fa, b0 + 1); mm(fa, b0 + 2); mm(fa, b0 + 3); }
        else { mm(fb, b0 + 1); mm(fb, b0 + 2); mm(fb, b0 + 3); }
        (void)ks;
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int b = 0; b < 8; ++b) mm(fa, b);
    }
    if (DMA == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else if (DMA == 2 || DMA == 3) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (DMA == 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (DMA == 5) {
      asm volatile("s_barrier" ::: "memory");
    } else {
      asm volatile("" ::: "memory");
    }
    if (DMA && DMA != 5) {
      issue(goff, slot);
      goff += STAGE_BYTES;
      if (goff + STAGE_BYTES > stream_bytes) goff = 0;
    }
    slot = nslot;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[b][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int DMA, int READS, int NOMM = 0>
void run(const char* name, const char* wstream, long long stream_bytes, float* out, unsigned long long* clocks, int blocks, int stages,
         int replicas = 1, long long replica_stride = 0, int stage_skew = 0) {
  const int lds = 4 * STAGE_BYTES + 1024;   // the kernel's allocation: one workgroup per CU
  hipFuncSetAttribute((const void*)k_probe<DMA, READS, NOMM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<DMA, READS, NOMM>), dim3(blocks), dim3(256), lds, 0, wstream, stream_bytes, 64, out, clocks, replicas, replica_stride, stage_skew);
  hipDeviceSynchronize();
  float best = 1e30f;
  double ghz = 0, cps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<DMA, READS, NOMM>), dim3(blocks), dim3(256), lds, 0, wstream, stream_bytes, stages, out, clocks, replicas, replica_stride, stage_skew);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2 * 1024];
    hipMemcpy(h, clocks, sizeof(unsigned long long) * 2 * (blocks < 1024 ? blocks : 1024), hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    const int nb = blocks < 1024 ? blocks : 1024;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cps = c / nb / stages; }     // s_memrealtime ticks at 100 MHz
  }
  const double rounds = (double)((blocks + 255) / 256);
  const double flops = (double)blocks * 4 * stages * 48.0 * 2.0 * 32 * 32 * 16;
  printf("%-58s %9.3f ms  %8.1f TFLOP/s executed bf16 = %5.3f of 2500   stage %7.0f core clocks   %5.3f GHz   (%g rounds)\n", name, best,
         flops / best / 1e9, flops / best / 1e9 / 2500.0, cps, ghz, rounds);
}

int main() {
  const long long stream_bytes = 107LL * STAGE_BYTES;      // one weight image (3.5 MB, L2-resident as in the kernel)
  char* w; float* out; unsigned long long* clocks;
  const long long rstride = stream_bytes + 16 * STAGE_BYTES + 4096 + 256;     // copies at another channel phase
  hipMalloc(&w, 8 * rstride + 16 * STAGE_BYTES); hipMemset(w, 0x3c, 8 * rstride + 16 * STAGE_BYTES);
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clocks, 4096 * 16);
  const int blocks = 256 * 8, stages = 2000;               // 8 rounds of one workgroup per CU, ~3.2 M matrix clocks each
  run<0, 0>("MFMA only (registers)", w, stream_bytes, out, clocks, blocks, stages);
  run<0, 3>("+ reads one 4-block group ahead (kernel's grouping)", w, stream_bytes, out, clocks, blocks, stages);
  run<2, 0>("hidden DMA + raw barrier, no reads", w, stream_bytes, out, clocks, blocks, stages);
  run<2, 3>("hidden DMA + raw barrier, reads one group ahead", w, stream_bytes, out, clocks, blocks, stages);
  run<2, 0>("  same, no reads, 8 image copies (workgroup i -> copy i % 8)", w, stream_bytes, out, clocks, blocks, stages, 8, rstride, 0);
  run<2, 0>("  same, no reads, one image, workgroups start 13 stages apart (i % 8)", w, stream_bytes, out, clocks, blocks, stages, 1, 0, 13);
  run<2, 0>("  same, no reads, 8 copies AND skewed starts", w, stream_bytes, out, clocks, blocks, stages, 8, rstride, 13);
  run<2, 3>("  reads one group ahead, 8 copies AND skewed starts", w, stream_bytes, out, clocks, blocks, stages, 8, rstride, 13);
  run<2, 0>("hidden DMA + raw barrier, no reads, 64 workgroups only (1 CU in 4)", w, stream_bytes, out, clocks, 64, stages);
  run<2, 0>("hidden DMA + raw barrier, no reads, 128 workgroups", w, stream_bytes, out, clocks, 128, stages);
  run<6, 0>("4-slot ring, pieces spread over the stage, counted vmcnt(8) [timing only], no reads", w, stream_bytes, out, clocks, blocks, stages);
  run<6, 3>("4-slot ring, pieces spread, counted vmcnt(8) [timing only], reads one group ahead", w, stream_bytes, out, clocks, blocks, stages);
  run<6, 4>("4-slot ring, pieces spread, counted vmcnt(8) [timing only], reads TWO groups ahead", w, stream_bytes, out, clocks, blocks, 1998);
  run<0, 3, 1>("NO MFMA: fragment reads only (128 KiB / stage and CU)", w, stream_bytes, out, clocks, blocks, stages);
  run<6, 0, 1>("NO MFMA: LDS-DMA only, spread, 4-slot (32 KiB / stage and CU)", w, stream_bytes, out, clocks, blocks, stages);
  run<6, 3, 1>("NO MFMA: fragment reads + LDS-DMA", w, stream_bytes, out, clocks, blocks, stages);
  run<2, 0, 1>("NO MFMA: LDS-DMA only, burst + vmcnt(0) + barrier, 3-slot", w, stream_bytes, out, clocks, blocks, stages);
  run<8, 0>("weights through registers (global_load + ds_write), no reads", w, stream_bytes, out, clocks, blocks, stages);
  run<8, 3>("weights through registers (global_load + ds_write), reads one group ahead", w, stream_bytes, out, clocks, blocks, stages);
  run<3, 0>("hidden DMA HALF volume (16 KiB / stage) + raw barrier, no reads", w, stream_bytes, out, clocks, blocks, stages);
  run<4, 0>("hidden DMA, vmcnt(0) but NO barrier, no reads", w, stream_bytes, out, clocks, blocks, stages);
  run<5, 0>("barrier only (no DMA), no reads", w, stream_bytes, out, clocks, blocks, stages);
  run<0, 0>("MFMA only again", w, stream_bytes, out, clocks, blocks, stages);
  // round 6: partial LDS bypass -- 8 of 32 fragments per stage straight from L2 into A-operand VGPRs, 24 through the LDS-DMA ring
  run<2, 3>("[r6] today's structure again: hidden DMA + raw barrier, reads one group ahead", w, stream_bytes, out, clocks, blocks, stages);
  run<9, 3>("[r6] PARTIAL BYPASS: 24 fragments via LDS-DMA ring + 8 via global_load_dwordx4 per wave", w, stream_bytes, out, clocks, blocks, stages);
  run<9, 3, 1>("[r6] NO MFMA: partial bypass memory work only (24 reads x 4 waves + 24 KiB DMA + 4 x 8 KiB direct)", w, stream_bytes, out, clocks, blocks, stages);
  run<10, 3>("[r6] PARTIAL BYPASS, direct loads a WHOLE stage ahead (two staging sets, 64 VGPRs)", w, stream_bytes, out, clocks, blocks, stages);
  run<10, 3, 1>("[r6] NO MFMA: the same memory work, direct loads a whole stage ahead", w, stream_bytes, out, clocks, blocks, stages);
  return 0;
}
