// mfma_probe_f32.hip -- issue-order experiment for v_mfma_f32_32x32x2_f32 (timing only): the fp32 kernels' pure-MFMA loop
// runs at ~66.2 clocks per MFMA (8 470 per 128, r01 probes) against the nominal 64; the bf16 32x32x16 loop of
// mfma_probe_bf16.hip -- THREE dependent MFMAs on one accumulator in a row -- runs at 32.2 of 32.  Does the order in which a
// k-group's 32 MFMAs (4 k-steps x 8 accumulator blocks) are issued matter?
//   ORDER 0  k-step outer, block inner (the kernels' order: 8 independent accumulators round robin)
//   ORDER 1  block outer, k-step inner (4 dependent MFMAs on one accumulator, then the next block)
//   ORDER 2  pairs: 2 dependent MFMAs per block, two sweeps over the blocks
// operands: A from 16 VGPRs (4 k-steps x 4-block fragments, as the kernels' window), B = 4 values.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe_f32.hip -o /tmp/mfma_probe_f32 && /tmp/mfma_probe_f32
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ __launch_bounds__(256) void k_probe(float* out, int kgroups, unsigned long long* clocks) {
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  f32x4 a[8];      // a[2 q + g][e]: A operand of k-step q, block 4 g + e
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f32x4{1e-3f * lane, 2e-3f, 3e-3f * i, 4e-3f};
  float bq[4] = {0.5f, 0.25f, 0.125f, 1e-2f * lane};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int kg = 0; kg < kgroups; ++kg) {
    if (ORDER == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * q + (nb >> 2)][nb & 3], bq[q], acc[nb], 0, 0, 0);
    } else if (ORDER == 1) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * q + (nb >> 2)][nb & 3], bq[q], acc[nb], 0, 0, 0);
    } else {
#pragma unroll
      for (int qq = 0; qq < 2; ++qq)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
          for (int q = 2 * qq; q < 2 * qq + 2; ++q)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * q + (nb >> 2)][nb & 3], bq[q], acc[nb], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
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

// ---- the fp32 kernels' stage: 4 k-groups x 32 MFMAs (8 192 matrix clocks), 3-slot 32 KiB weight ring, one "vmcnt(0); s_barrier"
// per stage, fragment reads through the kernels' two-quarter window.  How much does the LDS-DMA re-issue cost, and does it
// matter WHEN each wave issues its 8 pieces?
//   ISSUE 0  no DMA at all (reads + barrier only)
//   ISSUE 1  all four waves right behind the barrier (the kernels today: 32 wave-instructions hit the CU's one texture-address
//            unit at the same moment)
//   ISSUE 2  staggered: wave w issues its 8 pieces behind quarter w of the stage's FIRST k-group (512 matrix clocks apart)
//   ISSUE 3  staggered over k-groups: wave w behind k-group w's first quarter (2 048 clocks apart)
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int STAGE_BYTES = 32768, FRAG = 1024;
template <int ISSUE, int READS, int STORE = 0>
__global__ __launch_bounds__(256) void k_stage(const char* __restrict__ wstream, long long stream_bytes, int stages, float* out,
                                               unsigned long long* clocks, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 3 * STAGE_BYTES / 4; i += 256) ((float*)smem)[i] = 1e-3f * (i & 31);
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  f32x4 a[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
  const unsigned lane16 = lane * 16;
  auto issue = [&](long long goff, int sl) __attribute__((always_inline)) {
    const char* g = wstream + goff + wave * 8 * FRAG;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + sl * STAGE_BYTES + wave * 8 * FRAG));
#pragma unroll
    for (int half = 0; half < 2; ++half)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:3072"
                   :: "s"(lds0 + half * 4 * FRAG), "v"(lane16), "s"(g + half * 4 * FRAG) : "memory", "m0");
  };
  // STORE: the training forward's saved-activation stores -- one float4 per lane and k-group into a row-major [sample][256] plane
  // (lane (m, h): row m, 16 bytes at column 8 kg + 4 h): 1 all four waves in front of the k-group (today), 2 wave w behind quarter
  // w of the k-group (EXEC masks prepared once per wave: no compare, no branch at the sites), 3 like 2 but only two sites
  // (waves 0, 1 behind quarter 0, waves 2, 3 behind quarter 2)
  float* srow = sink + ((long long)blockIdx.x * 128 + wave * 32 + (lane & 31)) * 256 + 4 * (lane >> 5);
  unsigned mq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned m = __builtin_amdgcn_readfirstlane((STORE == 3 ? (wave >> 1) == (q >> 1) && (q & 1) == 0 : wave == q) ? ~0u : 0u);
    mq[q] = m;
  }
  const f32x4 sv = {1.f * lane, 2.f, 3.f, 4.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int slot = 0;
  long long goff = 0;
  if (ISSUE) {
    issue(0, 0); issue(STAGE_BYTES, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(2LL * STAGE_BYTES, 2);
    goff = 3LL * STAGE_BYTES;
  }
  if (READS) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(smem + lane16 + j * FRAG);
  }
  for (int st = 0; st < stages; ++st) {
    const char* cur = smem + slot * STAGE_BYTES + lane16;
    const int nslot = slot == 2 ? 0 : slot + 1, fill = slot == 0 ? 2 : slot - 1;      // `fill`: consumed in the previous stage
    const char* nxt = smem + nslot * STAGE_BYTES + lane16;
    if (ISSUE == 1 && st > 0) issue(goff, fill);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const char* csrc = cur + ks * 8 * FRAG;
      const char* nsrc = ks == 3 ? nxt : csrc + 8 * FRAG;
      if (STORE == 1) {
        *reinterpret_cast<f32x4*>(srow + 8 * ((st * 4 + ks) & 31)) = sv;
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sl = (q & 1) * 2;
        if (ISSUE == 4 && st > 0 && ks == 0 && wave == q) {
          // this wave's quarter: ONE piece behind each of the quarter's 8 MFMAs (m0 set once per 4 pieces; the immediate offset is
          // added to the global and the LDS address)
          const char* g = wstream + goff + wave * 8 * FRAG;
          const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + fill * STAGE_BYTES + wave * 8 * FRAG));
#pragma unroll
          for (int nb = 0; nb < 8; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl + (nb >> 2)][nb & 3], 0.5f + q, acc[nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((nb & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds0 + (nb >> 2) * 4 * FRAG) : "memory", "m0");
            if ((nb & 3) == 0) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(lane16), "s"(g + (nb >> 2) * 4 * FRAG) : "memory", "m0");
            if ((nb & 3) == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(lane16), "s"(g + (nb >> 2) * 4 * FRAG) : "memory", "m0");
            if ((nb & 3) == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(lane16), "s"(g + (nb >> 2) * 4 * FRAG) : "memory", "m0");
            if ((nb & 3) == 3) asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(lane16), "s"(g + (nb >> 2) * 4 * FRAG) : "memory", "m0");
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl + (nb >> 2)][nb & 3], 0.5f + q, acc[nb], 0, 0, 0);
        }
        if (READS) {
          const char* src = q < 2 ? csrc + (q + 2) * 2 * FRAG : nsrc + (q - 2) * 2 * FRAG;
          a[sl] = *reinterpret_cast<const f32x4*>(src);
          a[sl + 1] = *reinterpret_cast<const f32x4*>(src + FRAG);
        }
        if (ISSUE == 4 && st > 0 && ks == 0 && wave == q) {
          // one piece behind each of the NEXT quarter's... (see below: handled inside the MFMA loop of the following quarter)
        }
        if ((ISSUE == 5 || ISSUE == 6) && st > 0 && wave == q && ks < (ISSUE == 5 ? 4 : 2)) {
          // ISSUE 5: 2 pieces behind quarter w of EVERY k-group; ISSUE 6: 4 pieces behind quarter w of k-groups 0 and 1
          const int per = ISSUE == 5 ? 2 : 4;
          const char* g = wstream + goff + wave * 8 * FRAG + ks * per * FRAG;
          const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + fill * STAGE_BYTES + wave * 8 * FRAG + ks * per * FRAG));
          if (per == 2)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024"
                         :: "s"(lds0), "v"(lane16), "s"(g) : "memory", "m0");
          else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072"
                         :: "s"(lds0), "v"(lane16), "s"(g) : "memory", "m0");
        }
        if (STORE == 2 || (STORE == 3 && (q & 1) == 0)) {
          float* pp = srow + 8 * ((st * 4 + ks) & 31);
          unsigned tmp;
          asm volatile("v_readfirstlane_b32 %0, %1\n\ts_nop 3\n\ts_mov_b32 exec_lo, %0\n\ts_mov_b32 exec_hi, %0\n\tglobal_store_dwordx4 %2, %3, off\n\ts_mov_b64 exec, -1"
                       : "=&s"(tmp) : "v"(mq[q]), "v"(pp), "v"(sv) : "memory");
        }
        if (ISSUE == 2 && st > 0 && ks == 0 && wave == q) issue(goff, fill);
        if (ISSUE == 3 && st > 0 && q == 0 && wave == ks) issue(goff, fill);
        __builtin_amdgcn_sched_barrier(0x6);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (ISSUE && st > 0) {
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

template <int ISSUE, int READS, int STORE = 0>
void run_stage(const char* name, const char* w, long long stream_bytes, float* out, unsigned long long* clocks, int blocks, int stages, float* sink = nullptr) {
  hipFuncSetAttribute((const void*)k_stage<ISSUE, READS, STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131584);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stage<ISSUE, READS, STORE>), dim3(blocks), dim3(256), 131584, 0, w, stream_bytes, 16, out, clocks, sink);
  hipDeviceSynchronize();
  float best = 1e30f; double ghz = 0, cps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_stage<ISSUE, READS, STORE>), dim3(blocks), dim3(256), 131584, 0, w, stream_bytes, stages, out, clocks, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2 * 1024];
    const int nb = blocks < 1024 ? blocks : 1024;
    hipMemcpy(h, clocks, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cps = c / nb / stages; }
  }
  const double flops = (double)blocks * 4 * stages * 128.0 * 2.0 * 32 * 32 * 2;
  printf("%-62s %9.3f ms  %7.1f TFLOP/s = %5.3f of 157.3   stage %6.0f clocks (8192 nominal)   %5.3f GHz\n", name, best,
         flops / best / 1e9, flops / best / 1e9 / 157.3, cps, ghz);
}

template <int ORDER>
void run(const char* name, float* out, unsigned long long* clocks, int blocks, int kgroups) {
  hipFuncSetAttribute((const void*)k_probe<ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 131584);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_probe<ORDER>), dim3(blocks), dim3(256), 131584, 0, out, 64, clocks);
  hipDeviceSynchronize();
  float best = 1e30f; double ghz = 0, cpm = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_probe<ORDER>), dim3(blocks), dim3(256), 131584, 0, out, kgroups, clocks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2 * 1024];
    const int nb = blocks < 1024 ? blocks : 1024;
    hipMemcpy(h, clocks, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cpm = c / nb / kgroups / 32.0; }
  }
  const double flops = (double)blocks * 4 * kgroups * 32.0 * 2.0 * 32 * 32 * 2;
  printf("%-62s %9.3f ms  %7.1f TFLOP/s = %5.3f of 157.3   %6.2f clocks / MFMA   %5.3f GHz\n", name, best, flops / best / 1e9,
         flops / best / 1e9 / 157.3, cpm, ghz);
}

int main() {
  float* out; unsigned long long* clocks;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clocks, 4096 * 16);
  const int blocks = 256 * 4, kgroups = 4000;
  run<0>("k-step outer, block inner (8 independent accumulators)", out, clocks, blocks, kgroups);
  run<1>("block outer, k-step inner (4 dependent MFMAs per accumulator)", out, clocks, blocks, kgroups);
  run<2>("2 dependent MFMAs per block, two sweeps", out, clocks, blocks, kgroups);
  run<0>("k-step outer again", out, clocks, blocks, kgroups);
  const long long stream_bytes = 107LL * STAGE_BYTES;
  char* w; hipMalloc(&w, stream_bytes + STAGE_BYTES); hipMemset(w, 0, stream_bytes + STAGE_BYTES);
  const int stages = 600;
  run_stage<0, 0>("stage: MFMA + barrier only", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<0, 1>("stage: + fragment reads (two-quarter window)", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<1, 1>("stage: + DMA, all waves right behind the barrier (today)", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<2, 1>("stage: + DMA, wave w behind quarter w of k-group 0", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<3, 1>("stage: + DMA, wave w behind k-group w's first quarter", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<4, 1>("stage: + DMA, wave w in quarter w of k-group 0, one piece per MFMA", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<5, 1>("stage: + DMA, 2 pieces behind quarter w of every k-group", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<6, 1>("stage: + DMA, 4 pieces behind quarter w of k-groups 0 and 1", w, stream_bytes, out, clocks, blocks, stages);
  float* sink; hipMalloc(&sink, (size_t)blocks * 128 * 256 * 4);
  run_stage<1, 1, 0>("stage: burst DMA + reads, no stores (reference)", w, stream_bytes, out, clocks, blocks, stages, sink);
  run_stage<1, 1, 1>("stage: + one float4 store per k-group, all waves at once", w, stream_bytes, out, clocks, blocks, stages, sink);
  run_stage<1, 1, 2>("stage: + stores staggered: wave w behind quarter w (EXEC masks)", w, stream_bytes, out, clocks, blocks, stages, sink);
  run_stage<1, 1, 3>("stage: + stores at two sites (waves 0,1 / waves 2,3)", w, stream_bytes, out, clocks, blocks, stages, sink);
  run_stage<1, 0>("stage: DMA behind the barrier, no reads", w, stream_bytes, out, clocks, blocks, stages);
  run_stage<2, 0>("stage: DMA staggered by quarter, no reads", w, stream_bytes, out, clocks, blocks, stages);
  return 0;
}
