// mfma_probe_fs.hip -- the FEATURE-SPLIT stage, timing only: can a 96-sample tile whose four
// waves each compute 64 of a layer's 256 outputs run a stage in < 0.8 x the 128-sample tile's stage?  (Then 24 576 coarse samples of a
// 384-ray shard = 256 tiles = one round on all CUs at < 0.8 x the time of today's 192 tiles on 192 CUs.)
// Stage = 4 k-groups of the existing weight image (32 KiB, 3-slot ring, LDS-DMA re-issue behind the barrier as in the kernels).  Per
// k-group a wave reads 4 weight fragments (k-step q: fragment 2 q + (wave >> 1); it uses 2 of the 4 packed blocks) and 3 B quads
// (one per 32-sample group) from a double-buffered LDS slab, and issues 4 k-steps x 2 blocks x 3 groups = 24 MFMAs (1 536 clocks;
// stage: 96 MFMAs = 6 144 clocks).  Before the stage barrier ONE wave (the owner of the next 32 features) publishes its block: 48 x
// v_max + 12 ds_write_b128 into the other slab half.
//   VAR 0  MFMAs + barrier only (operands in registers)
//   VAR 1  + weight fragments and B quads read one k-group ahead, + DMA re-issue          (no publishing)
//   VAR 2  + the publishing wave's 48 v_max + 12 ds_write_b128 in front of the barrier      (the full stage)
//   VAR 3  as 2, but the publish is spread: 12 v_max + 3 ds_write_b128 behind each k-group of the PREVIOUS stage
// Reference: the 128-sample stage of mfma_probe_f32.hip (8 601 clocks per 8 192) on the same box is printed first.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe_fs.hip -o tools/probe/mfma_probe_fs && tools/probe/mfma_probe_fs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int STAGE_BYTES = 32768, FRAG = 1024;
constexpr int SLAB_ROW = 36 * 4;                    // 32 features + 4 pad floats per sample row: conflict-free ds_read_b128
constexpr int SLAB_GROUP = 32 * SLAB_ROW;           // one 32-sample group
constexpr int SLAB_BYTES = 3 * SLAB_GROUP;          // 13 824 B per stage
constexpr int SLAB_OFF = 3 * STAGE_BYTES;

template <int VAR>
__global__ __launch_bounds__(256) void k_fs(const char* __restrict__ wstream, long long stream_bytes, int stages, float* out,
                                            unsigned long long* clocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < (3 * STAGE_BYTES + 2 * SLAB_BYTES) / 4; i += 256) ((float*)smem)[i] = 1e-3f * (i & 31);
  __syncthreads();
  f32x16 acc[2][3];      // this wave's 2 feature blocks x 3 sample groups
  f32x16 prevb;          // (publisher) a finished block of the previous layer, one group: 16 values per lane; 3 groups = 48
  f32x16 prev[3];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][g][r] = 0.f;
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) prev[g][r] = 1e-2f * (r + 1) + 1e-4f * lane + g;
  (void)prevb;
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
  // B quad of k-group ks, sample group g for lane (m, h): slab[g][m][8 ks + 4 h ..]
  const unsigned b_lane = (lane & 31) * SLAB_ROW + (lane >> 5) * 16;
  f32x4 a[2][4], bq[2][3];     // double-buffered: k-group's 4 weight fragments + 3 B quads
  const int e0 = 2 * (wave & 1);
  const unsigned a_frag = (wave >> 1) * FRAG;
  auto load_kg = [&](int buf, const char* stage_base, const char* slab, int ks) __attribute__((always_inline)) {
    if (VAR >= 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) a[buf][q] = *reinterpret_cast<const f32x4*>(stage_base + ks * 8 * FRAG + q * 2 * FRAG + a_frag + lane16);
#pragma unroll
      for (int g = 0; g < 3; ++g) bq[buf][g] = *reinterpret_cast<const f32x4*>(slab + g * SLAB_GROUP + b_lane + ks * 32);
    }
  };
#pragma unroll
  for (int buf = 0; buf < 2; ++buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) a[buf][q] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
#pragma unroll
    for (int g = 0; g < 3; ++g) bq[buf][g] = f32x4{0.5f, 0.25f, 0.125f, 1e-2f * lane};
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int slot = 0;
  long long goff = 0;
  if (VAR >= 1) {
    issue(0, 0); issue(STAGE_BYTES, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue(2LL * STAGE_BYTES, 2);
    goff = 3LL * STAGE_BYTES;
  }
  load_kg(0, smem, smem + SLAB_OFF, 0);
  for (int st = 0; st < stages; ++st) {
    const char* cur = smem + slot * STAGE_BYTES;
    const int nslot = slot == 2 ? 0 : slot + 1, fill = slot == 0 ? 2 : slot - 1;
    const char* nxt = smem + nslot * STAGE_BYTES;
    const char* slab_cur = smem + SLAB_OFF + (st & 1) * SLAB_BYTES;
    char* slab_nxt = smem + SLAB_OFF + ((st + 1) & 1) * SLAB_BYTES;
    const bool publisher = wave == ((st + 1) & 3);       // owner of the NEXT stage's 32 features
    if (VAR >= 1 && st > 0) issue(goff, fill);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cb = ks & 1, nb = cb ^ 1;
      // next k-group's operands one k-group ahead (of this stage, or k-group 0 of the next stage: its ring slot is complete, its slab
      // half is NOT before the barrier -- so the B quads of a stage's first k-group are read right behind the barrier)
      if (ks < 3) load_kg(nb, cur, slab_cur, ks + 1);
      else if (VAR >= 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) a[nb][q] = *reinterpret_cast<const f32x4*>(nxt + q * 2 * FRAG + a_frag + lane16);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int g = 0; g < 3; ++g)
            acc[b][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cb][q][e0 + b], bq[cb][g][q], acc[b][g], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0x6);
      if (VAR == 3 && publisher) {      // a quarter of the publish behind every k-group
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            int v = __builtin_bit_cast(int, prev[g][4 * ks + c]);
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v) : "v"(v));
            o[c] = __builtin_bit_cast(float, v);
          }
          *reinterpret_cast<f32x4*>(slab_nxt + g * SLAB_GROUP + b_lane + ks * 32) = o;
        }
      }
    }
    if (VAR == 2 && publisher) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
          f32x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            int v = __builtin_bit_cast(int, prev[g][4 * kq + c]);
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v) : "v"(v));
            o[c] = __builtin_bit_cast(float, v);
          }
          *reinterpret_cast<f32x4*>(slab_nxt + g * SLAB_GROUP + b_lane + kq * 32) = o;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (VAR >= 1) {      // the next stage's first B quads: their slab half is complete now
#pragma unroll
      for (int g = 0; g < 3; ++g) bq[0][g] = *reinterpret_cast<const f32x4*>(slab_nxt + g * SLAB_GROUP + b_lane);
      if (st > 0) {
        goff += STAGE_BYTES;
        if (goff + STAGE_BYTES > stream_bytes) goff = 0;
      }
    }
    slot = nslot;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[b][g][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int VAR>
void run(const char* name, const char* w, long long stream_bytes, float* out, unsigned long long* clocks, int blocks, int stages) {
  const int lds = 3 * STAGE_BYTES + 2 * SLAB_BYTES;
  (void)hipFuncSetAttribute((const void*)k_fs<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fs<VAR>), dim3(blocks), dim3(256), lds, 0, w, stream_bytes, 16, out, clocks);
  (void)hipDeviceSynchronize();
  float best = 1e30f; double ghz = 0, cps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_fs<VAR>), dim3(blocks), dim3(256), lds, 0, w, stream_bytes, stages, out, clocks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * 1024);
    const int nb = blocks < 1024 ? blocks : 1024;
    (void)hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cps = c / nb / stages; }
  }
  printf("%-92s %8.3f ms   stage %6.0f clocks (6144 nominal; 128-sample kernels: 8700 per 8192)   %5.3f GHz  [%s]\n", name, best, cps, ghz,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float* out; unsigned long long* clocks;
  (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&clocks, 4096 * 16);
  const int blocks = 256 * 4, stages = 600;
  const long long stream_bytes = 107LL * STAGE_BYTES;
  char* w; (void)hipMalloc(&w, stream_bytes + STAGE_BYTES); (void)hipMemset(w, 0, stream_bytes + STAGE_BYTES);
  run<0>("feature-split stage: 96 MFMAs + barrier, operands in registers", w, stream_bytes, out, clocks, blocks, stages);
  run<1>("+ weight fragments / B quads from LDS one k-group ahead, DMA re-issue behind the barrier", w, stream_bytes, out, clocks, blocks, stages);
  run<2>("+ one wave publishes its block in front of the barrier (48 v_max + 12 ds_write_b128)", w, stream_bytes, out, clocks, blocks, stages);
  run<3>("+ the publish spread over the previous stage's k-groups (12 v_max + 3 ds_write_b128 each)", w, stream_bytes, out, clocks, blocks, stages);
  run<1>("no publishing again", w, stream_bytes, out, clocks, blocks, stages);
  return 0;
}
