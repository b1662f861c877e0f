// mfma_probe_h16.hip -- the HALF-TIME tile (DESIGN.md section 9 / EXPERIMENTS.md "fine-pass tail"), timing only: a 64-sample tile whose four
// waves carry 16 samples each on v_mfma_f32_16x16x4_f32.  Same FLOPs per sample and the same weight bytes per stage as the 128-sample
// kernels (a stage = 32 KiB of weights = 256 outputs x 32 contraction rows), so a wave issues 128 MFMAs of 32 clocks = 4 096 clocks per
// stage instead of 128 x 64 = 8 192: if the REAL stage stays near 0.5 x the 128-sample kernels' 8 700 clocks, half tiles fill what whole
// tiles leave idle -- the coarse pass of a 384-ray shard (192 tiles on 256 CUs -> 384 half tiles = 1.5 rounds of half time = 0.75 x) and the
// last half round of the 3072-ray fine pass (7.5 -> 7 + 0.5 x).
// Layout assumed (the transposed, register-resident scheme carries over): C tile 16 outputs x 16 samples = f32x4 per lane (lane = sample
// n = lane & 15, row group lane >> 4); a finished layer's values ARE the next layer's B operands (k-step (blk, i) contracts features
// 16 blk + 4 (lane >> 4) + i), the weight image is re-ordered to match (its own pack table); A operand = 1 float per lane and MFMA, so one
// ds_read_b128 feeds 4 MFMAs and a stage needs the same 32 fragment reads per wave as today -- in half the time.
//   VAR 0  MFMAs + barrier only (operands in registers)
//   VAR 1  + the 32 weight-fragment reads per stage, one k-group (8 fragments) ahead
//   VAR 2  + LDS-DMA re-issue of the consumed ring slot behind the barrier (the full pipe of the kernels)
//   VAR 3  as 2 + the layer hand-over every 8th stage (64 v_max on the bit patterns, fenced, as take<> does)
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe_h16.hip -o tools/probe/mfma_probe_h16 && tools/probe/mfma_probe_h16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int STAGE_BYTES = 32768, FRAG = 1024;

template <int VAR, int NW>
__global__ __launch_bounds__(64 * NW) void k_h16(const char* __restrict__ wstream, long long stream_bytes, int stages, float* out,
                                                 unsigned long long* clocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 3 * STAGE_BYTES / 4; i += 64 * NW) ((float*)smem)[i] = 1e-3f * (i & 31);
  __syncthreads();
  f32x4 acc[16];         // 256 outputs x 16 samples: 16 blocks of 16 outputs
  float hb[64];          // the previous layer's outputs = B operands: 64 k-steps' worth (one float per k-step and lane)
#pragma unroll
  for (int b = 0; b < 16; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 64; ++r) hb[r] = 1e-2f * (r + 1) + 1e-4f * lane;
  const unsigned lane16 = lane * 16;
  auto issue = [&](long long goff, int sl) __attribute__((always_inline)) {
    constexpr int PER = 32 / NW;              // fragments per wave and stage: 8 (4 waves) or 4 (8 waves)
    const char* g = wstream + goff + wave * PER * FRAG;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + sl * STAGE_BYTES + wave * PER * FRAG));
#pragma unroll
    for (int half = 0; half < PER / 4; ++half)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:3072"
                   :: "s"(lds0 + half * 4 * FRAG), "v"(lane16), "s"(g + half * 4 * FRAG) : "memory", "m0");
  };
  // a k-group = 8 fragments = 32 MFMAs: fragment q of k-group kg feeds output blocks 2 q, 2 q + 1 with 2 k-steps each
  f32x4 a[2][8];
  auto load_kg = [&](int buf, const char* stage_base, int kg) __attribute__((always_inline)) {
    if (VAR >= 1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) a[buf][q] = *reinterpret_cast<const f32x4*>(stage_base + (kg * 8 + q) * FRAG + lane16);
    }
  };
#pragma unroll
  for (int buf = 0; buf < 2; ++buf)
#pragma unroll
    for (int q = 0; q < 8; ++q) a[buf][q] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int slot = 0;
  long long goff = 0;
  if (VAR >= 2) {
    issue(0, 0); issue(STAGE_BYTES, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(2LL * STAGE_BYTES, 2);
    goff = 3LL * STAGE_BYTES;
  }
  load_kg(0, smem, 0);
  for (int st = 0; st < stages; ++st) {
    const char* cur = smem + slot * STAGE_BYTES;
    const int nslot = slot == 2 ? 0 : slot + 1, fill = slot == 0 ? 2 : slot - 1;
    const char* nxt = smem + nslot * STAGE_BYTES;
    if (VAR >= 2 && st > 0) issue(goff, fill);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int cb = kg & 1, nb = cb ^ 1;
      if (kg < 3) load_kg(nb, cur, kg + 1);
      else load_kg(nb, nxt, 0);               // (the next stage's slot is complete: it was waited for at the previous barrier)
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)           // 4 MFMAs per fragment: blocks 2 q + (e >> 1), k-step 2 kg' + (e & 1) of the stage's 8
          acc[2 * q + (e >> 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][q][e], hb[(16 * kg + 2 * q + (e & 1)) & 63],
                                                                        acc[2 * q + (e >> 1)], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0x6);
    }
    if (VAR >= 3 && (st & 7) == 7) {          // layer hand-over: the accumulators become the next layer's B operands
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < 16; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          int v = __builtin_bit_cast(int, acc[b][c]);
          asm volatile("v_max_i32 %0, %1, 0" : "=v"(v) : "v"(v));
          hb[4 * b + c] = __builtin_bit_cast(float, v);
          acc[b][c] = 1e-3f * c;
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (VAR >= 2 && st > 0) {
      goff += STAGE_BYTES;
      if (goff + STAGE_BYTES > stream_bytes) goff = 0;
    }
    slot = nslot;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 16; ++b) s += acc[b][0] + acc[b][1] + acc[b][2] + acc[b][3];
  out[blockIdx.x * 64 * NW + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}


// reference on the same box: the SAME pipe (ring, fragment reads a k-group ahead, DMA re-issue, barrier) with the kernels' 32x32x2 MFMAs --
// 128 per stage = 8 192 clocks nominal, the 128-sample tile's stage
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_ref32(const char* __restrict__ wstream, long long stream_bytes, int stages, float* out,
                                               unsigned long long* clocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 3 * STAGE_BYTES / 4; i += 256) ((float*)smem)[i] = 1e-3f * (i & 31);
  __syncthreads();
  f32x16 acc[8];
  float hb[32];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 32; ++r) hb[r] = 1e-2f * (r + 1) + 1e-4f * lane;
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
  f32x4 a[2][8];
#pragma unroll
  for (int q = 0; q < 8; ++q) a[0][q] = *reinterpret_cast<const f32x4*>(smem + q * FRAG + lane16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int slot = 0;
  issue(0, 0); issue(STAGE_BYTES, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  issue(2LL * STAGE_BYTES, 2);
  long long goff = 3LL * STAGE_BYTES;
  for (int st = 0; st < stages; ++st) {
    const char* cur = smem + slot * STAGE_BYTES;
    const int nslot = slot == 2 ? 0 : slot + 1, fill = slot == 0 ? 2 : slot - 1;
    const char* nxt = smem + nslot * STAGE_BYTES;
    if (st > 0) issue(goff, fill);
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const int cb = kg & 1, nb = cb ^ 1;
      const char* src = kg < 3 ? cur + (kg + 1) * 8 * FRAG : nxt;
#pragma unroll
      for (int q = 0; q < 8; ++q) a[nb][q] = *reinterpret_cast<const f32x4*>(src + q * FRAG + lane16);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][q][e], hb[(4 * kg + e) & 31], acc[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0x6);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st > 0) {
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

template <typename K>
void run_k(K kern, const char* name, const char* w, long long stream_bytes, float* out, unsigned long long* clocks, int blocks, int stages,
           int nominal) {
  const int lds = 3 * STAGE_BYTES;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, w, stream_bytes, 16, out, clocks);
  (void)hipDeviceSynchronize();
  float best = 1e30f; double ghz = 0, cps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, w, stream_bytes, stages, out, clocks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * 1024);
    const int nb = blocks < 1024 ? blocks : 1024;
    (void)hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cps = c / nb / stages; }
  }
  printf("%-100s %8.3f ms   stage %6.0f clocks (%d nominal)   %5.3f GHz  [%s]\n", name, best, cps, nominal, ghz,
         hipGetErrorString(hipGetLastError()));
}

template <int VAR, int NW>
void run(const char* name, const char* w, long long stream_bytes, float* out, unsigned long long* clocks, int blocks, int stages) {
  const int lds = 3 * STAGE_BYTES;
  auto kern = k_h16<VAR, NW>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds, 0, w, stream_bytes, 16, out, clocks);
  (void)hipDeviceSynchronize();
  float best = 1e30f; double ghz = 0, cps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NW), lds, 0, w, stream_bytes, stages, out, clocks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * 1024);
    const int nb = blocks < 1024 ? blocks : 1024;
    (void)hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cps = c / nb / stages; }
  }
  printf("%-100s %8.3f ms   stage %6.0f clocks (%d nominal)   %5.3f GHz  [%s]\n", name, best, cps, NW == 4 ? 4096 : 8192, ghz,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out; unsigned long long* clocks;
  (void)hipMalloc(&out, 4096 * 512 * 4); (void)hipMalloc(&clocks, 4096 * 16);
  const int blocks = 256 * 4, stages = 856;      // 107 stages = one pass through the network's weight image; 8 of them
  const long long stream_bytes = 107LL * STAGE_BYTES;
  char* w; (void)hipMalloc(&w, stream_bytes + STAGE_BYTES); (void)hipMemset(w, 0, stream_bytes + STAGE_BYTES);
  run_k(k_ref32, "REFERENCE 128-sample stage on this box: 128 x v_mfma_f32_32x32x2_f32, fragment reads, DMA re-issue", w, stream_bytes, out, clocks,
        blocks, stages / 2, 8192);
  run<0, 4>("half-time stage: 128 x v_mfma_f32_16x16x4_f32 + barrier, operands in registers", w, stream_bytes, out, clocks, blocks, stages);
  run<1, 4>("+ 32 weight-fragment reads (ds_read_b128) per stage, one k-group ahead", w, stream_bytes, out, clocks, blocks, stages);
  run<2, 4>("+ LDS-DMA re-issue of the consumed slot behind the barrier (3-slot ring, 32 KiB per stage)", w, stream_bytes, out, clocks, blocks, stages);
  run<3, 4>("+ layer hand-over every 8th stage (64 v_max, fenced)", w, stream_bytes, out, clocks, blocks, stages);
  // the same 16-sample waves, EIGHT per tile = two per SIMD sharing one ring: a 128-sample tile at occupancy 2 (a stage is now 2 x 4 096
  // clocks of MFMAs per SIMD; what one wave waits for -- barrier, DMA issue, hand-over -- the other can fill)
  run<0, 8>("8 x 16-sample waves (2 per SIMD): MFMAs + barrier", w, stream_bytes, out, clocks, blocks, stages / 2);
  run<1, 8>("8 waves: + fragment reads (every wave reads every fragment: 64 KiB per SIMD and stage)", w, stream_bytes, out, clocks, blocks, stages / 2);
  run<2, 8>("8 waves: + LDS-DMA re-issue (4 KiB per wave)", w, stream_bytes, out, clocks, blocks, stages / 2);
  run<3, 8>("8 waves: + layer hand-over every 8th stage", w, stream_bytes, out, clocks, blocks, stages / 2);
  return 0;
}
