// mfma_probe_ballot.hip -- ReLU masks of the backward-data kernel as LANE BALLOTS in SGPRs, timing + a
// functional check of the scalar-store round trip.  Stage = the fp32 kernels' stage (4 k-groups x 32 MFMAs = 8 192 matrix
// clocks, 3-slot 32 KiB weight ring refilled by LDS-DMA behind the barrier, two-quarter fragment window), plus what k_mlp_bwd
// does per stage for its masks and results:
//   MASK 0  nothing (the forward's clean stage: 8 601 in the round-3 probe)
//   MASK 1  TODAY: one global_load_dwordx4 mask quad (32 rows x 16 B gather out of a [sample][256] plane) + one
//           global_store_dwordx4 (the dz quad) per k-group; at the stage top 16 x (v_cmp_gt_f32 + v_cndmask) form the operands
//   MASK 2  BALLOTS: the stage's 16 masks are 16 SGPR pairs (one bit per lane) fetched with 2 x s_load_dwordx8 + ... (4 loads of
//           4 pairs) during the previous stage; stage top: s_waitcnt lgkmcnt(0) (SMEM returns out of order: only a full drain
//           is safe, and it drains the fragment reads in flight too) + 16 x v_cndmask_b32 with an SGPR-pair mask; the dz store
//           per k-group stays.  No vector load in the MFMA stream.
//   MASK 3  as 2 without the dz stores (what the loads alone cost)
//   MASK 4  as 1 without the mask loads (stores + 16 cndmask only): the ablation "no mask loads"
// Producer side (the forward's layer hand-over, once per 8 stages): TAKE 0 = 128 x v_max_i32 (today), TAKE 1 = 128 x (v_max_i32 +
// v_cmp_gt_i32_e64 into an SGPR pair) + 64 x s_store_dwordx4 (two ballots each) + one s_dcache_wb at the end of the kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe_ballot.hip -o /tmp/mfma_probe_ballot && /tmp/mfma_probe_ballot
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef u64 u64x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr int STAGE_BYTES = 32768, FRAG = 1024;

template <int MASK, int TAKE>
__global__ __launch_bounds__(256) void k_stage(const char* __restrict__ wstream, long long stream_bytes, int stages, float* out,
                                               unsigned long long* clocks, float* plane, float* dzp, u64* ballots) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 3 * STAGE_BYTES / 4; i += 256) ((float*)smem)[i] = 1e-3f * (i & 31);
  __syncthreads();
  f32x16 acc[8];
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  f32x16 prev;      // the raw block the stage's operands are formed from
#pragma unroll
  for (int r = 0; r < 16; ++r) prev[r] = 1e-2f * (r + 1) + 1e-4f * lane;
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
  // rows of this lane in the [sample][256] planes (lane (m, h): row m of the wave's 32, 16 bytes at column 8 kg + 4 h)
  const long long row = (long long)blockIdx.x * 128 + wave * 32 + (lane & 31);
  const float* mrow = plane + row * 256 + 4 * (lane >> 5);
  float* zrow = dzp + row * 256 + 4 * (lane >> 5);
  // ballots of this wave: [stage & 63][16] u64 (wave-uniform address)
  const u64* bal = ballots + ((long long)blockIdx.x * 4 + wave) * 64 * 16;
  f32x4 mk[4] = {f32x4{1, 1, 1, 1}, f32x4{1, -1, 1, 1}, f32x4{1, 1, -1, 1}, f32x4{-1, 1, 1, 1}};
  u64x4 sm[4];      // 16 SGPR pairs = the stage's 16 ballots
  if (MASK == 2 || MASK == 3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(sm[j]) : "s"(bal), "n"(64 * 0) : "memory");
  }
  float bq[16];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  int slot = 0;
  long long goff = 0;
  issue(0, 0); issue(STAGE_BYTES, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  issue(2LL * STAGE_BYTES, 2);
  goff = 3LL * STAGE_BYTES;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(smem + lane16 + j * FRAG);
  for (int st = 0; st < stages; ++st) {
    const char* cur = smem + slot * STAGE_BYTES + lane16;
    const int nslot = slot == 2 ? 0 : slot + 1, fill = slot == 0 ? 2 : slot - 1;
    const char* nxt = smem + nslot * STAGE_BYTES + lane16;
    // ---- stage top: the block's 16 masked operands, one fenced block in front of the weight re-issue
    __builtin_amdgcn_sched_barrier(0);
    if (MASK == 1 || MASK == 4) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bq[4 * q + 0] = mk[q].x > 0.f ? prev[4 * q + 0] : 0.f;
        bq[4 * q + 1] = mk[q].y > 0.f ? prev[4 * q + 1] : 0.f;
        bq[4 * q + 2] = mk[q].z > 0.f ? prev[4 * q + 2] : 0.f;
        bq[4 * q + 3] = mk[q].w > 0.f ? prev[4 * q + 3] : 0.f;
      }
    } else if (MASK == 2 || MASK == 3) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sm[0]), "+s"(sm[1]), "+s"(sm[2]), "+s"(sm[3]));
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(bq[i]) : "v"(prev[i]), "s"(sm[i >> 2][i & 3]));
      // the next stage's ballots: requested now, a whole stage ahead (scalar cache -> L2, not the vector-memory path)
      const u64* nb = bal + ((st + 1) & 63) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(sm[j]) : "s"(nb), "n"(0) : "memory");
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) bq[i] = prev[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bq[i]));
    if (st > 0) issue(goff, fill);
    // ---- the forward's layer hand-over every 8th stage (producer side of the ballots)
    if (TAKE >= 0 && (st & 7) == 7) {
      __builtin_amdgcn_sched_barrier(0);
      if (TAKE == 0) {
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float x = acc[b][r];
            asm volatile("" : "+v"(x));
            int v = __builtin_bit_cast(int, x);
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v) : "v"(v));
            acc[b][r] = __builtin_bit_cast(float, v);
          }
      } else if (TAKE == 1) {
        u64* dst = ballots + (((long long)blockIdx.x * 4 + wave) * 64 + ((st >> 3) & 7) * 8) * 16;   // 128 ballots = 1 KiB
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            float x0 = acc[b][r], x1 = acc[b][r + 1];
            asm volatile("" : "+v"(x0), "+v"(x1));
            int v0 = __builtin_bit_cast(int, x0), v1 = __builtin_bit_cast(int, x1);
            u64 m0, m1;
            asm volatile("v_cmp_gt_i32_e64 %0, %1, 0" : "=s"(m0) : "v"(v0));
            asm volatile("v_cmp_gt_i32_e64 %0, %1, 0" : "=s"(m1) : "v"(v1));
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v0) : "v"(v0));
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v1) : "v"(v1));
            const u64x2 mm = {m0, m1};
            asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(mm), "s"(dst), "n"(0) : "memory");
            dst += 2;
            acc[b][r] = __builtin_bit_cast(float, v0);
            acc[b][r + 1] = __builtin_bit_cast(float, v1);
          }
      } else if (TAKE == 2) {          // the compares alone (ballots dropped): what the VALU side costs
#pragma unroll
        for (int b = 0; b < 8; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float x = acc[b][r];
            asm volatile("" : "+v"(x));
            int v = __builtin_bit_cast(int, x);
            u64 m;
            asm volatile("v_cmp_gt_i32_e64 %0, %1, 0" : "=s"(m) : "v"(v));
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v) : "v"(v));
            asm volatile("" :: "s"(m));
            acc[b][r] = __builtin_bit_cast(float, v);
          }
      } else if (TAKE == 3) {          // batched: 16 compares into 32 SGPRs, then 8 scalar stores back to back
        u64* dst = ballots + (((long long)blockIdx.x * 4 + wave) * 64 + ((st >> 3) & 7) * 8) * 16;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          u64 m[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float x = acc[b][r];
            asm volatile("" : "+v"(x));
            int v = __builtin_bit_cast(int, x);
            asm volatile("v_cmp_gt_i32_e64 %0, %1, 0" : "=s"(m[r]) : "v"(v));
            asm volatile("v_max_i32 %0, %1, 0" : "=v"(v) : "v"(v));
            acc[b][r] = __builtin_bit_cast(float, v);
          }
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const u64x2 mm = {m[r], m[r + 1]};
            asm volatile("s_store_dwordx4 %0, %1, %2" :: "s"(mm), "s"(dst), "n"(0) : "memory");
            dst += 2;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const char* csrc = cur + ks * 8 * FRAG;
      const char* nsrc = ks == 3 ? nxt : csrc + 8 * FRAG;
      const f32x4 o = {bq[4 * ks], bq[4 * ks + 1], bq[4 * ks + 2], bq[4 * ks + 3]};
      const int kg = (4 * st + ks) & 31;
      if (MASK == 1 || MASK == 2 || MASK == 4) *reinterpret_cast<f32x4*>(zrow + 8 * kg) = o;
      if (MASK == 1) mk[ks] = *reinterpret_cast<const f32x4*>(mrow + 8 * ((kg + 4) & 31));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sl = (q & 1) * 2;
        const float bv = q == 0 ? o.x : q == 1 ? o.y : q == 2 ? o.z : o.w;
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[sl + (nb >> 2)][nb & 3], bv, acc[nb], 0, 0, 0);
        const char* src = q < 2 ? csrc + (q + 2) * 2 * FRAG : nsrc + (q - 2) * 2 * FRAG;
        a[sl] = *reinterpret_cast<const f32x4*>(src);
        a[sl + 1] = *reinterpret_cast<const f32x4*>(src + FRAG);
        __builtin_amdgcn_sched_barrier(0x6);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (st > 0) {
      goff += STAGE_BYTES;
      if (goff + STAGE_BYTES > stream_bytes) goff = 0;
    }
    slot = nslot;
  }
  if (TAKE == 1 || TAKE == 3) asm volatile("s_dcache_wb" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[b][r];
#pragma unroll
  for (int i = 0; i < 16; ++i) s += bq[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = t1 - t0; clocks[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MASK, int TAKE>
void run_stage(const char* name, const char* w, long long stream_bytes, float* out, unsigned long long* clocks, int blocks, int stages,
               float* plane, float* dzp, u64* ballots) {
  hipFuncSetAttribute((const void*)k_stage<MASK, TAKE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131584);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k_stage<MASK, TAKE>), dim3(blocks), dim3(256), 131584, 0, w, stream_bytes, 16, out, clocks, plane, dzp, ballots);
  hipDeviceSynchronize();
  float best = 1e30f; double ghz = 0, cps = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_stage<MASK, TAKE>), dim3(blocks), dim3(256), 131584, 0, w, stream_bytes, stages, out, clocks, plane, dzp, ballots);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * 1024);
    const int nb = blocks < 1024 ? blocks : 1024;
    hipMemcpy(h.data(), clocks, sizeof(unsigned long long) * 2 * nb, hipMemcpyDeviceToHost);
    double c = 0, r = 0;
    for (int i = 0; i < nb; ++i) { c += (double)h[2 * i]; r += (double)h[2 * i + 1]; }
    if (ms < best) { best = ms; ghz = c / r * 0.1; cps = c / nb / stages; }
  }
  const double flops = (double)blocks * 4 * stages * 128.0 * 2.0 * 32 * 32 * 2;
  printf("%-78s %9.3f ms  %6.1f TF = %5.3f   stage %6.0f clocks   %5.3f GHz\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.3, cps, ghz);
}

// functional check: lanes write ballots with s_store_dwordx4 (+ s_dcache_wb), a SECOND kernel reads them with s_load and applies
// v_cndmask from the SGPR pair; compared with the plain per-lane select on the host
__global__ void k_put(const float* x, u64* masks, int n) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u64* dst = masks + ((long long)blockIdx.x * 4 + wave) * n;
  for (int i = 0; i < n; i += 2) {
    const int v0 = __builtin_bit_cast(int, x[(((long long)blockIdx.x * 4 + wave) * n + i) * 64 + lane]);
    const int v1 = __builtin_bit_cast(int, x[(((long long)blockIdx.x * 4 + wave) * n + i + 1) * 64 + lane]);
    u64 m0, m1;
    asm volatile("v_cmp_gt_i32_e64 %0, %1, 0" : "=s"(m0) : "v"(v0));
    asm volatile("v_cmp_gt_i32_e64 %0, %1, 0" : "=s"(m1) : "v"(v1));
    const u64x2 mm = {m0, m1};
    u64* du = dst + i;
    asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(mm), "s"(du) : "memory");
  }
  asm volatile("s_dcache_wb" ::: "memory");
}
__global__ void k_get(const float* y, const u64* masks, float* out, int n) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u64* src = masks + ((long long)blockIdx.x * 4 + wave) * n;
  for (int i = 0; i < n; i += 4) {
    u64x4 m;
    const u64* s = src + i;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(m) : "s"(s) : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long idx = (((long long)blockIdx.x * 4 + wave) * n + i + j) * 64 + lane;
      float r;
      asm volatile("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(y[idx]), "s"(m[j]));
      out[idx] = r;
    }
  }
}

int main() {
  // ---- functional round trip
  {
    const int blocks = 64, n = 128;
    const size_t N = (size_t)blocks * 4 * n * 64;
    std::vector<float> hx(N), hy(N), ho(N);
    unsigned s = 12345;
    for (size_t i = 0; i < N; ++i) {
      s = s * 1664525u + 1013904223u;
      const int k = (s >> 8) % 7;
      hx[i] = k == 0 ? 0.f : k == 1 ? -0.f : k == 2 ? 1e-42f : k == 3 ? -1e-42f : ((int)(s >> 12) % 2001 - 1000) * 1e-3f;
      hy[i] = 1.f + (float)(i % 977);
    }
    float *x, *y, *o; u64* m;
    hipMalloc(&x, N * 4); hipMalloc(&y, N * 4); hipMalloc(&o, N * 4); hipMalloc(&m, (size_t)blocks * 4 * n * 8);
    hipMemcpy(x, hx.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(y, hy.data(), N * 4, hipMemcpyHostToDevice);
    hipMemset(m, 0xff, (size_t)blocks * 4 * n * 8);
    hipLaunchKernelGGL(k_put, dim3(blocks), dim3(256), 0, 0, x, m, n);
    hipLaunchKernelGGL(k_get, dim3(blocks), dim3(256), 0, 0, y, m, o, n);
    hipDeviceSynchronize();
    hipMemcpy(ho.data(), o, N * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < N; ++i) bad += ho[i] != (hx[i] > 0.f ? hy[i] : 0.f);
    printf("scalar-store round trip (s_store_dwordx4 + s_dcache_wb -> next kernel's s_load_dwordx8 -> v_cndmask from an SGPR pair), "
           "%zu values incl. +-0 and denormals: %zu wrong  [%s]\n", N, bad, hipGetErrorString(hipGetLastError()));
  }
  float* out; unsigned long long* clocks;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clocks, 4096 * 16);
  const int blocks = 256 * 4, stages = 600;
  const long long stream_bytes = 107LL * STAGE_BYTES;
  char* w; hipMalloc(&w, stream_bytes + STAGE_BYTES); hipMemset(w, 0, stream_bytes + STAGE_BYTES);
  float *plane, *dzp; u64* ballots;
  hipMalloc(&plane, (size_t)blocks * 128 * 256 * 4); hipMalloc(&dzp, (size_t)blocks * 128 * 256 * 4);
  hipMemset(plane, 0x3f, (size_t)blocks * 128 * 256 * 4);
  hipMalloc(&ballots, (size_t)blocks * 4 * 64 * 16 * 8 + 4096); hipMemset(ballots, 0x55, (size_t)blocks * 4 * 64 * 16 * 8 + 4096);
#define RUN(M, T, NAME) run_stage<M, T>(NAME, w, stream_bytes, out, clocks, blocks, stages, plane, dzp, ballots)
  RUN(0, -1, "stage: MFMA + DMA + fragment reads (clean)");
  RUN(1, -1, "bwd TODAY: 1 mask-quad gather + 1 dz store per k-group, 16 x (cmp + cndmask) at the top");
  RUN(4, -1, "bwd ablation: no mask loads (stores + selects stay)");
  RUN(2, -1, "bwd BALLOTS: 4 x s_load_dwordx8 a stage ahead, lgkmcnt(0) + 16 x v_cndmask(SGPR pair), dz stores");
  RUN(3, -1, "bwd BALLOTS without the dz stores");
  RUN(1, -1, "bwd TODAY again");
  RUN(0, 0, "fwd hand-over TODAY every 8th stage: 128 x v_max_i32");
  RUN(0, 1, "fwd hand-over BALLOTS: 128 x (v_max_i32 + v_cmp_e64) + 64 x s_store_dwordx4, s_dcache_wb at the end");
  RUN(0, 2, "fwd hand-over: the 128 v_cmp_e64 alone (ballots dropped)");
  RUN(0, 3, "fwd hand-over BALLOTS batched: 16 x v_cmp, then 8 x s_store_dwordx4 back to back");
  RUN(0, 0, "fwd hand-over TODAY again");
  return 0;
}
