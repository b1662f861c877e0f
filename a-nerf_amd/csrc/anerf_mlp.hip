// anerf_mlp.hip -- fused "encode + 11-layer MLP" forward kernel for gfx950 (MI355X, CDNA4).
//
// Replaces, per sample point, the reference op sequence
//   transform_batch_pts / transform_batch_rays      core/encoders.py:8-37
//   RelDistEncoder / VecNormEncoder                 core/encoders.py:110-122,181-193
//   CutoffEmbedder._embed (x2) + identity embedder  core/cutoff_embedder.py:111-174
//   run_network cat + NeRF.forward                  core/raycasters.py:557-577, core/networks/nerf.py:94-148
// The 1080-wide encoding and the 256-wide activations never exist in HBM.
//
// Execution model (see DESIGN.md for the derivation)
//   * workgroup = 4 waves = one 128-sample tile; every wave owns 32 samples for the whole network.
//   * each layer is computed TRANSPOSED on the fp32 matrix cores: D^T[n][m] = sum_k W[n][k] * act[m][k] with
//     v_mfma_f32_32x32x2_f32, A = weight fragment (from LDS), B = activation (a register), D = 32 features x
//     32 samples.  Lane l holds sample m = l&31; lanes 0-31 / 32-63 hold the two k-halves.  The accumulator layout
//     of layer i (feature n = 32nb + (r&3) + 8(r>>2) + 4(l>>5) in register r of block nb) IS the B-operand layout
//     of layer i+1, so activations stay in registers across all layers: no LDS round trip, no barrier for them.
//   * a layer's accumulators (AGPRs) start from its bias (LDS copy, read straight into the AGPRs); when the layer is done
//     its 128 values per lane are taken into VGPRs in ONE fenced block (v_accvgpr_read + v_max_i32: the ReLU) and those
//     VGPRs are the next layer's B operands -- no VALU inside the MFMA stream of the hidden layers (a VALU instruction
//     there costs ~14 clocks of matrix time, in a block 4-8: fwd_common.h, take<>).
//   * weights are pre-packed (k_pack) into a linear stream of 1 KiB MFMA fragments in consumption order and streamed
//     L2 -> LDS with global_load_lds_dwordx4 through a 3 x 32 KiB ring, issued two stages ahead; one barrier per
//     stage (= 8192 MFMA cycles per wave); fragments are read a full quarter of a k-group ahead (two-quarter window,
//     kgroup), across the stage barrier too, so no LDS-read latency is exposed.
//   * the encoding is produced in registers just in time as B operands: lane half h owns joints
//     {j : ((j>>2)&1) == h}; bone matrices of the tile's rays are staged in LDS.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "anerf_fwd_common.h"

namespace anerf {
#if defined(ANERF_EXP_TILE_TIMING) || defined(ANERF_EXP_STAGE_TIMING)
float* g_tile_timing_buf = nullptr;   // set through anerf_debug_set_timing_buf (debug build only)
#endif

// TRAIN: row-major store of a finished layer: features 32nb+8q+4h .. +3 of `row`
template <int NB>
__device__ __forceinline__ void store_act(float* __restrict__ row, const f32x16* acc, int h) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {acc[nb][4 * q], acc[nb][4 * q + 1], acc[nb][4 * q + 2], acc[nb][4 * q + 3]};
      save_quad(row + 32 * nb + 8 * q + 4 * h, o);
    }
}

// ------------------------------------------------------------------------------------------------
// the x-part (432 = 360 distance-PE + 72 bone-direction channels) as 54 k-groups.
// Encoded just in time (PRE = false) or read from a pre-encoded row (PRE = true, NeRF.forward seam).
// STORE: also write the 4 operands of every k-group to xsave[8*kg + 4h .. +3] (stream column order X').
// ------------------------------------------------------------------------------------------------
template <int LV, bool PRE, bool STORE>
__device__ __forceinline__ void x_part(Pipe3F& pipe, f32x16 (&acc)[8], const float (&v)[12], const float (&wv)[12],
                                       const float (&rh)[36], const float* __restrict__ xrow, int h,
                                       float* __restrict__ xsave, bool last) {
  constexpr int NKG = 3 * (1 + 2 * LV) + 9;
  auto KG = [&](int kg, float b0, float b1, float b2, float b3) __attribute__((always_inline)) {
    if constexpr (STORE) {
      const f32x4 o = {b0, b1, b2, b3};
      save_quad(xsave + 8 * kg + 4 * h, o);
      __builtin_amdgcn_sched_barrier(0);
    }
    kgroup<8>(pipe, acc, kg, kg == 0, last && kg == NKG - 1, b0, b1, b2, b3);
  };
  if constexpr (PRE) {
#pragma unroll
    for (int kg = 0; kg < 3 * (1 + 2 * LV); ++kg) {
      const float* c = xrow + 8 * kg + 4 * h;
      KG(kg, c[0], c[1], c[2], c[3]);
    }
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      float b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int a = (4 * g + t) / 3, c = (4 * g + t) % 3;
        b[t] = xrow[24 * (1 + 2 * LV) + 3 * (8 * (a >> 2) + (a & 3)) + c + 12 * h];
      }
      KG(3 * (1 + 2 * LV) + g, b[0], b[1], b[2], b[3]);
    }
  } else {
#pragma unroll
    for (int g = 0; g < 3; ++g)
      KG(g, v[4 * g] * wv[4 * g], v[4 * g + 1] * wv[4 * g + 1], v[4 * g + 2] * wv[4 * g + 2],
         v[4 * g + 3] * wv[4 * g + 3]);
    // bands f = 0..LV-1 of the gated sin/cos(2^f v): precise evaluation every 4th band, double-angle steps in between on the
    // pair (gs, c) = (w sin a, cos a):  w sin 2a = gs * 2c,  cos 2a = 2c * c - 1  -- the gate rides along in gs, so a step is 3
    // operations + 1 for the gated cosine, and values are processed in pairs (v_pk_add / v_pk_mul / v_pk_fma_f32: packed fp32
    // runs at twice the scalar VALU rate, and every VALU instruction issued into the fp32 MFMA stream is lost MFMA time).
    // Angle error doubles per step: <= 8 x (1 ulp + 6e-8) ~ 1.3e-6 after three steps.
    f32x2 gs[6], cc[6], w2[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w2[i] = f32x2{wv[2 * i], wv[2 * i + 1]};
#pragma unroll
    for (int f = 0; f < LV; ++f) {
      __builtin_amdgcn_sched_barrier(0);     // a band's values as one block in front of its six k-groups (see the view layer)
      if (f % 4 == 0) {
#pragma unroll
        for (int a = 0; a < 12; ++a) {
          float s, c;
          sincos_f32(v[a] * (float)(1 << f), s, c);
          gs[a >> 1][a & 1] = s * wv[a];
          cc[a >> 1][a & 1] = c;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const f32x2 t = cc[i] + cc[i];
          gs[i] = gs[i] * t;
          cc[i] = t * cc[i] - 1.0f;
        }
      }
      f32x2 gc[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) gc[i] = cc[i] * w2[i];
#pragma unroll
      for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(gs[i]), "+v"(gc[i]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 3; ++g) KG(3 + 6 * f + g, gs[2 * g][0], gs[2 * g][1], gs[2 * g + 1][0], gs[2 * g + 1][1]);
#pragma unroll
      for (int g = 0; g < 3; ++g) KG(6 + 6 * f + g, gc[2 * g][0], gc[2 * g][1], gc[2 * g + 1][0], gc[2 * g + 1][1]);
    }
#pragma unroll
    for (int g = 0; g < 9; ++g)
      KG(3 * (1 + 2 * LV) + g, rh[4 * g], rh[4 * g + 1], rh[4 * g + 2], rh[4 * g + 3]);
  }
}

// ------------------------------------------------------------------------------------------------
// PRE (NeRF.forward seam, pre-encoded input rows): a run of k-groups whose B operands are float4 quads of the sample's input
// row, with the discipline of k_mlp_bwd_in (anerf_mlp_bwd.hip): the quads of a stage are requested one per k-group during the
// stage BEFORE (each lane gathers 16 bytes of its own row: 32 rows per wave-load, ~100 cycles of the CU's vector-memory path
// apiece, never in bursts), are complete at that stage's barrier (vmcnt(0)), and are pinned in front of the weight pipe's
// re-issue so that the compiler's counted waits for them never find freshly issued LDS-DMA pieces queued behind them.
// Round 2 loaded every quad as four scalar dwords right in front of its k-group (alignment unknown to hipcc) and waited there:
// 130 TFLOP/s against 143 for the fused kernel.
// NB output blocks -> KPS = 32 / NB k-groups per stage.  KG0: index of the run's first k-group inside its weight segment (a
// multiple of KPS); NKG k-groups, quad of k-group i at addr(i).  ENDS: the run ends its weight segment (the last k-group
// closes the stage); otherwise the caller continues the segment with more k-groups.  `pending`: the pipe stands behind a
// stage barrier whose re-issue is still to do.
// ------------------------------------------------------------------------------------------------
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));    // rows are only dword-aligned when x_width is odd (frame-code column)
template <int NB, int KG0, int NKG, bool ENDS, class ADDR>
__device__ __forceinline__ void pre_run(Pipe3F& pipe, f32x16 (&acc)[NB], ADDR addr, bool& pending) {
  constexpr int KPS = STAGE_FRAGS / NB;
  static_assert(KG0 % KPS == 0, "pre_run starts on a stage boundary");
  f32x4 cur[KPS], nxt[KPS];
#pragma unroll
  for (int i = 0; i < KPS; ++i) cur[i] = i < NKG ? f32x4(*reinterpret_cast<const f32x4_u*>(addr(i))) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < (NKG + KPS - 1) / KPS; ++s) {
#pragma unroll
    for (int i = 0; i < KPS; ++i) asm volatile("" : "+v"(cur[i]));
    if (pending) pipe.stage_refill();
    pending = false;
#pragma unroll
    for (int ks = 0; ks < KPS; ++ks) {
      const int i = KPS * s + ks;
      if (i < NKG) {
        if (i + KPS < NKG) nxt[ks] = f32x4(*reinterpret_cast<const f32x4_u*>(addr(i + KPS)));
        else nxt[ks] = cur[ks];
        __builtin_amdgcn_sched_barrier(0);     // the request goes in front of the k-group's MFMAs
        kgroup<NB, Pipe3F, false>(pipe, acc, KG0 + i, false, ENDS && i == NKG - 1, cur[ks].x, cur[ks].y, cur[ks].z, cur[ks].w);
      }
    }
    const bool stage_done = KPS * (s + 1) <= NKG || ENDS;      // a trailing partial stage is closed only when the segment ends
    if (stage_done) {
      pipe.stage_rendezvous();
      pending = true;
    }
#pragma unroll
    for (int i = 0; i < KPS; ++i) cur[i] = nxt[i];
  }
}

// MODE 0: rays + depths -> raw [P,4].   MODE 1 (density query, raycasters.py:597-648): points A.z = pts [P,3] under ONE
// shared pose -> sigma logit [P]; only the trunk (layers 0..7 + alpha head) runs, the stream stops after layer 7.
template <int LV, int LD, int CODE, bool PRE, bool TRAIN, int MODE = 0>
__global__ __launch_bounds__(256) void k_mlp_fwd(const MlpArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;

#ifdef ANERF_EXP_TILE_TIMING   // debug build only: (start, end) wall-clock stamps of every tile in A.save_h (8 B each)
  const unsigned long long tk0 = wall_clock64();
#endif
  Pipe3F pipe;
  pipe.init(A.packed, smem, wave, lane, A.nstages);
#ifdef ANERF_EXP_STAGE_TIMING
  const unsigned long long tt0 = __builtin_amdgcn_s_memtime(), tr0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef ANERF_EXP_STAGE_TIMING   // every 997th tile: [tile][wave][stage][3] clocks in A.save_u
  if (A.tbuf && blockIdx.x % 997 == 0 && lane == 0)
    pipe.tbuf = A.tbuf + ((long long)(blockIdx.x / 997) * 4 + wave) * 3 * 128;
#endif

  const long long p = (long long)blockIdx.x * TILE + wave * 32 + m;
  const bool valid = p < A.P;
  const long long pc = valid ? p : A.P - 1;
  // gate temperatures: kernel arguments, or (training step under a captured hipGraph, ABI revision 6) two floats of the
  // device-resident step block -- wave-uniform scalar loads; the render kernels (TRAIN = false) keep the arguments
  float tau_v = A.tau_v, tau_d = A.tau_d;
  if constexpr (TRAIN) {
    if (A.tau_dev) {
      tau_v = A.tau_dev[0];
      tau_d = A.tau_dev[1];
    }
  }
  // TRAIN: every lane stores its saved activations unconditionally -- tail lanes (p >= P) computed on the clamped sample
  // pc = P - 1 and so rewrite that row with identical values -- which keeps exec-mask changes out of the MFMA stream.
  const long long ps = pc;

  // ---- biases / head rows -> LDS (natural order), read back as float4 per k-group
  // (LDS-DMA, 1 KiB per wave-instruction, complete at pipe.begin()'s vmcnt(0) like the first weight stages: the former
  // load -> ds_write loop was three dependent round trips at the head of every tile; the last 2 float4 go the old way)
  float* aux_l = reinterpret_cast<float*>(smem + LDS_AUX_OFF);
  {
    constexpr int NPIECE = AUX_FLOATS / 256;   // whole 1 KiB pieces (12)
    const char* ga = reinterpret_cast<const char*>(A.aux) + lane * 16;
#pragma unroll
    for (int k = 0; k < (NPIECE + 3) / 4; ++k) {
      const int piece = 4 * k + wave;
      if (piece < NPIECE) {   // hidden from hipcc like the weight pipe's DMA (a visible one would turn every LDS wait into lgkmcnt(0))
        const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((lds_ptr_t)(smem + LDS_AUX_OFF)) + piece * 1024;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds0), "v"(ga + piece * 1024) : "memory", "m0");
      }
    }
    const int i = NPIECE * 64 + tid;
    if (i < AUX_FLOATS / 4) reinterpret_cast<f32x4*>(aux_l)[i] = reinterpret_cast<const f32x4*>(A.aux)[i];
  }
  const float* aux_h = aux_l + 4 * h;   // this lane half's float4 inside every 8-feature group

  float v[12], wv[12], rh[36];
  float dray[3] = {0.f, 0.f, 0.f};
  int lr = 0;
  long long ray = 0;
  const float* xrow = nullptr;
  const f32x4* bones4 = reinterpret_cast<const f32x4*>(smem + LDS_BONES_OFF);

  if constexpr (PRE) {
    xrow = A.x + pc * A.x_width;
#pragma unroll
    for (int a = 0; a < 12; ++a) v[a] = wv[a] = 0.f;
#pragma unroll
    for (int a = 0; a < 36; ++a) rh[a] = 0.f;
    pipe.begin();
    pipe.prime();
  } else {
    // ---- stage the bone matrices (rows 0..2 of each 4x4 world->bone matrix) of this tile's rays in LDS
    const long long tile_p0 = (long long)blockIdx.x * TILE;
    const long long ray0 = div_samples(tile_p0, A.S);
    long long ray1 = div_samples(tile_p0 + TILE - 1, A.S);
    if (ray1 > A.N - 1) ray1 = A.N - 1;
    ray = MODE == 1 ? 0 : div_samples(pc, A.S);
    const int n_stage_rays = (A.skt_stride == 0 || MODE == 1) ? 1 : (int)(ray1 - ray0 + 1);
    lr = (A.skt_stride == 0 || MODE == 1) ? 0 : (int)(ray - ray0);
    f32x4* bw = reinterpret_cast<f32x4*>(smem + LDS_BONES_OFF);
    for (int i = tid; i < n_stage_rays * 72; i += 256) {
      const int ri = i / 72, rem = i - ri * 72, j = rem / 3, row = rem - 3 * j;
      bw[i] = *reinterpret_cast<const f32x4*>(A.skts + (ray0 + ri) * A.skt_stride + j * 16 + row * 4);
    }
    float x0, x1, x2;
    if constexpr (MODE == 1) {
      x0 = A.z[3 * pc];
      x1 = A.z[3 * pc + 1];
      x2 = A.z[3 * pc + 2];
    } else {
      const float* rp = A.rays + ray * A.ray_stride;
      const float z = A.z[pc];
      x0 = fmaf(rp[3], z, rp[0]);
      x1 = fmaf(rp[4], z, rp[1]);
      x2 = fmaf(rp[5], z, rp[2]);
      if (A.pnoise) {   // wave-uniform: ray_noise_std > 0, training only
        x0 += A.pnoise[3 * pc];
        x1 += A.pnoise[3 * pc + 1];
        x2 += A.pnoise[3 * pc + 2];
      }
    }
    pipe.begin();   // barrier: aux + bones visible, weight stages 0/1 landed
    pipe.prime();
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const int j = 8 * (a >> 2) + 4 * h + (a & 3);
      const f32x4 r0 = bones4[(lr * 24 + j) * 3 + 0], r1 = bones4[(lr * 24 + j) * 3 + 1],
                  r2 = bones4[(lr * 24 + j) * 3 + 2];
      const float y0 = r0.x * x0 + r0.y * x1 + r0.z * x2 + r0.w;
      const float y1 = r1.x * x0 + r1.y * x1 + r1.z * x2 + r1.w;
      const float y2 = r2.x * x0 + r2.y * x1 + r2.z * x2 + r2.w;
      const float n = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
      const float inv = rcp_nr(fmaxf(n, 1e-12f));
      v[a] = n;
      wv[a] = cutoff_gate(tau_v, n, A.cut_v[j]);
      const float gb = A.gate_bones ? inv * wv[a] : inv;      // cutoff_bones: r_j * w_j (bone embedder = the distance gate)
      rh[3 * a + 0] = y0 * gb;
      rh[3 * a + 1] = y1 * gb;
      rh[3 * a + 2] = y2 * gb;
    }
  }

  constexpr int DIMX = 24 * (1 + 2 * LV) + 72;
  constexpr int DIMD = 72 * (1 + 2 * LD);
  constexpr int UW = DIMD + CODE;
  constexpr int KGX = DIMX / 8;
  f32x16 accA[8], accB[8];   // ping-pong accumulator sets (AGPRs): bias + sums of the layer being computed / of the previous one
  float hb[128];             // the previous layer's outputs, ReLU applied = this layer's B operands (VGPRs, take<>)

  // ---- layer 0: x(432) -> A
  // TRAIN: h_l is saved by the layer that CONSUMES it (hidden_part_v<.., true>: one quad per k-group)
  float* hsave = TRAIN ? A.save_h + ps * 256 + 4 * h : nullptr;      // this lane's quads in plane 0; plane l at + l * plane
  const long long plane = TRAIN ? A.Ppad * 256 : 0;
  init_bias<8>(accA, aux_h + AUX_B0);
  bool pre_pending = false;       // PRE: the pipe stands behind a stage barrier, re-issue still to do (pre_run)
  // PRE: k-group kg's quad of the input row.  Distance-PE channels: columns 8 kg + 4 h ..; bone directions (stream order =
  // joint-owned 3-vectors): 12 consecutive floats per joint quad and lane half
  auto xaddr = [&](int kg) __attribute__((always_inline)) -> const float* {
    constexpr int NV = 3 * (1 + 2 * LV);
    return kg < NV ? xrow + 8 * kg + 4 * h : xrow + 24 * (1 + 2 * LV) + 24 * ((kg - NV) / 3) + 12 * h + 4 * ((kg - NV) % 3);
  };
  if constexpr (PRE) {
    pre_run<8, 0, KGX, true>(pipe, accA, xaddr, pre_pending);
    if (pre_pending) pipe.stage_refill();
    pre_pending = false;
  } else {
    x_part<LV, PRE, TRAIN>(pipe, accA, v, wv, rh, xrow, h, TRAIN ? A.save_x + ps * DIMX : nullptr, true);
  }
  // ---- layers 1..4: A -> B -> A -> B -> A
#pragma unroll 1
  for (int L = 1; L <= 3; L += 2) {
    init_bias<8>(accB, aux_h + AUX_B0 + 256 * L);   // (LDS reads in flight under the VALU pass)
    take<8, true>(hb, accA);
    hidden_part_v<8, 0, TRAIN>(pipe, accB, hb, true, true, hsave + (L - 1) * plane);
    init_bias<8>(accA, aux_h + AUX_B0 + 256 * (L + 1));
    take<8, true>(hb, accB);
    hidden_part_v<8, 0, TRAIN>(pipe, accA, hb, true, true, hsave + L * plane);
  }
  // ---- layer 5: [x(432); h4(256)] -> B   (skip connection: x is re-encoded, never stored; h4 waits in its accumulator
  // set and is taken when the x part is done).  The asm makes v/wv opaque so the compiler re-derives the sin/cos products
  // here instead of keeping 168 of them live (spilled to scratch) since layer 0.
  if constexpr (!PRE) {
#pragma unroll
    for (int a = 0; a < 12; ++a) asm volatile("" : "+v"(v[a]), "+v"(wv[a]));
  }
  init_bias<8>(accB, aux_h + AUX_B0 + 256 * 5);
  if constexpr (PRE) {
    pre_run<8, 0, KGX, false>(pipe, accB, xaddr, pre_pending);      // 54 k-groups: ends mid-stage, the hidden part continues it
    if (pre_pending) pipe.stage_refill();
    pre_pending = false;
  } else {
    x_part<LV, PRE, false>(pipe, accB, v, wv, rh, xrow, h, nullptr, false);
  }
  take<8, true>(hb, accA);
  hidden_part_v<8, KGX, TRAIN>(pipe, accB, hb, false, true, hsave + 4 * plane);
  // ---- layers 6, 7: B -> A -> B
  init_bias<8>(accA, aux_h + AUX_B0 + 256 * 6);
  take<8, true>(hb, accB);
  hidden_part_v<8, 0, TRAIN>(pipe, accA, hb, true, true, hsave + 5 * plane);
  init_bias<8>(accB, aux_h + AUX_B0 + 256 * 7);
  take<8, true>(hb, accA);
  hidden_part_v<8, 0, TRAIN>(pipe, accB, hb, true, true, hsave + 6 * plane);
  take<8, true>(hb, accB);
  // ---- density head (VALU): sigma_raw = w_alpha . h7 + b_alpha
  const float sigma_raw = head_dot_v<8>(hb, aux_h + AUX_WA) + aux_l[AUX_BA];
  if constexpr (MODE == 1) {
    if (valid && h == 0) A.raw[p] = sigma_raw;
    return;
  }
  // ---- feature layer (no activation on its output): B -> A
  init_bias<8>(accA, aux_h + AUX_BF);
  hidden_part_v<8, 0, TRAIN>(pipe, accA, hb, true, true, hsave + 7 * plane);
  take<8, false>(hb, accA);
  // ---- view layer: [feature(256); D(72*(1+2LD)); code(CODE)] -> 128 units
  f32x16 accv[4];
  constexpr int NKGU = UW / 8;
  init_bias<4>(accv, aux_h + AUX_BV);
  hidden_part_v<4, 0, TRAIN>(pipe, accv, hb, true, false, TRAIN ? A.save_f + ps * 256 + 4 * h : nullptr);
  float* usave = TRAIN ? A.save_u + ps * UW + 4 * h : nullptr;
  auto KGV = [&](int kgu, float b0, float b1, float b2, float b3) __attribute__((always_inline)) {
    if constexpr (TRAIN) {
      const f32x4 o = {b0, b1, b2, b3};
      save_quad(usave + 8 * kgu, o);
      __builtin_amdgcn_sched_barrier(0);
    }
    kgroup<4>(pipe, accv, 32 + kgu, false, kgu == NKGU - 1, b0, b1, b2, b3);
  };
  if constexpr (PRE) {
    // view-direction PE: band b, joint quad g / 3: 12 consecutive floats per lane half, as for the bone directions
    auto uaddr = [&](int kgu) __attribute__((always_inline)) -> const float* {
      return xrow + DIMX + 72 * (kgu / 9) + 24 * ((kgu % 9) / 3) + 12 * h + 4 * ((kgu % 9) % 3);
    };
    constexpr int NKD = DIMD / 8;       // 81 (9 with multires_views = 0)
    pre_run<4, 32, NKD, CODE == 0>(pipe, accv, uaddr, pre_pending);
    if (pre_pending) pipe.stage_refill();
    pre_pending = false;
  } else {
    // per-ray unit direction in each owned bone frame, gated per sample by the distance gate (tau_d, cut_d).  The ray
    // direction is re-read here (12 bytes per lane, L2 hits) rather than kept in three VGPRs across the eight trunk layers,
    // where both accumulator sets fill the AGPR file and the allocator spilled them to scratch (HBM traffic per tile).
    if constexpr (MODE == 0) {
      const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      long long pe = (long long)blockIdx.x * TILE + wave * 32 + (lane_e & 31);
      pe = pe < A.P ? pe : A.P - 1;
      const float* rp = A.rays + div_samples(pe, A.S) * A.ray_stride;
      dray[0] = rp[3];
      dray[1] = rp[4];
      dray[2] = rp[5];
    }
    float e[36], wd[12];
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      const int j = 8 * (a >> 2) + 4 * h + (a & 3);
      const f32x4 r0 = bones4[(lr * 24 + j) * 3 + 0], r1 = bones4[(lr * 24 + j) * 3 + 1],
                  r2 = bones4[(lr * 24 + j) * 3 + 2];
      const float y0 = r0.x * dray[0] + r0.y * dray[1] + r0.z * dray[2];
      const float y1 = r1.x * dray[0] + r1.y * dray[1] + r1.z * dray[2];
      const float y2 = r2.x * dray[0] + r2.y * dray[1] + r2.z * dray[2];
      const float inv = rcp_nr(fmaxf(sqrtf(y0 * y0 + y1 * y1 + y2 * y2), 1e-12f));
      e[3 * a + 0] = y0 * inv;
      e[3 * a + 1] = y1 * inv;
      e[3 * a + 2] = y2 * inv;
      wd[a] = cutoff_gate(tau_d, v[a], A.cut_d[j]);
    }
    f32x2 wd2[18];   // the gate of every direction component, in operand pairs
#pragma unroll
    for (int i = 0; i < 18; ++i) wd2[i] = f32x2{wd[(2 * i) / 3], wd[(2 * i + 1) / 3]};
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      const f32x2 o0 = f32x2{e[4 * g], e[4 * g + 1]} * wd2[2 * g], o1 = f32x2{e[4 * g + 2], e[4 * g + 3]} * wd2[2 * g + 1];
      KGV(g, o0[0], o0[1], o1[0], o1[1]);
    }
    // gated sin / cos(2^f e) as in x_part: (gs, c) pairs, precise at f = 0 (|e| <= 1: no range reduction), packed
    // double-angle steps after that.  A band's 72 values are produced in ONE fenced block in front of its 18 k-groups (the
    // feature operands are dead by now, registers are free): sprinkled over the k-groups each of these instructions cost
    // the matrix pipe ~14 clocks, in a block ~6.
    f32x2 gse[18], cce[18];
#pragma unroll
    for (int f = 0; f < LD; ++f) {
      f32x2 gce[18];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        if (f % 4 == 0) {
          float s0, c0, s1, c1;
          if (f == 0) {
            sincos_unit_f32(e[2 * i], s0, c0);
            sincos_unit_f32(e[2 * i + 1], s1, c1);
          } else {
            sincos_f32(e[2 * i] * (float)(1 << f), s0, c0);
            sincos_f32(e[2 * i + 1] * (float)(1 << f), s1, c1);
          }
          gse[i] = f32x2{s0, s1} * wd2[i];
          cce[i] = f32x2{c0, c1};
        } else {
          const f32x2 tt = cce[i] + cce[i];
          gse[i] = gse[i] * tt;
          cce[i] = tt * cce[i] - 1.0f;
        }
        gce[i] = cce[i] * wd2[i];
      }
#pragma unroll
      for (int i = 0; i < 18; ++i) asm volatile("" : "+v"(gse[i]), "+v"(gce[i]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 9; ++g)
        KGV(9 * (1 + 2 * f) + g, gse[2 * g][0], gse[2 * g][1], gse[2 * g + 1][0], gse[2 * g + 1][1]);
#pragma unroll
      for (int g = 0; g < 9; ++g)
        KGV(9 * (2 + 2 * f) + g, gce[2 * g][0], gce[2 * g][1], gce[2 * g + 1][0], gce[2 * g + 1][1]);
    }
  }
  if constexpr (CODE > 0) {
    float fidx;
    if constexpr (PRE) fidx = xrow[DIMX + DIMD];
    else {   // the ray index is re-derived (see the note at the final store)
      const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      long long pe = (long long)blockIdx.x * TILE + wave * 32 + (lane_e & 31);
      pe = pe < A.P ? pe : A.P - 1;
      fidx = A.cam[div_samples(pe, A.S)];
    }
    int ci = (int)fidx;
    ci = ci < 0 ? 0 : (ci >= A.n_codes ? A.n_codes - 1 : ci);
    const float* crow = A.codes + (long long)ci * CODE;
#pragma unroll
    for (int g = 0; g < CODE / 8; ++g) {
      const f32x4 c4 = *reinterpret_cast<const f32x4*>(crow + 8 * g + 4 * h);
      KGV(9 * (1 + 2 * LD) + g, c4.x, c4.y, c4.z, c4.w);
    }
  }
  relu_pass<4>(accv);
  if constexpr (TRAIN) store_act<4>(A.save_g + ps * 128, accv, h);
  // ---- rgb head (VALU)
  const float c0 = head_dot<4>(accv, aux_h + AUX_WC + 0) + aux_l[AUX_BC + 0];
  const float c1 = head_dot<4>(accv, aux_h + AUX_WC + 128) + aux_l[AUX_BC + 1];
  const float c2 = head_dot<4>(accv, aux_h + AUX_WC + 256) + aux_l[AUX_BC + 2];
  {
    // the output position is re-derived here (mbcnt = lane id) instead of keeping `p` live through the whole kernel: that
    // one long-lived VGPR pair was spilled to scratch, 2 KB of extra HBM traffic per tile
    const int lane_e = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const long long pe = (long long)blockIdx.x * TILE + wave * 32 + (lane_e & 31);
    if (pe < A.P && lane_e < 32) {
      f32x4 o = {c0, c1, c2, sigma_raw};
      *reinterpret_cast<f32x4*>(A.raw + pe * 4) = o;
    }
  }
#ifdef ANERF_EXP_STAGE_TIMING   // every tile: start / end clocks + where it ran, behind the stage records ([64][4][128][3])
  if (tid == 0 && A.tbuf) {
    unsigned long long* t = A.tbuf + 64 * 4 * 128 * 3 + 6 * (long long)blockIdx.x;
    t[0] = tt0; t[1] = __builtin_amdgcn_s_memtime(); t[2] = tr0; t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = __builtin_amdgcn_s_getreg(63492); t[5] = __builtin_amdgcn_s_getreg(63508);
  }
#endif
#ifdef ANERF_EXP_TILE_TIMING
  if (tid == 0 && A.save_u) {
    unsigned long long* t = reinterpret_cast<unsigned long long*>(A.save_u) + 2 * (long long)blockIdx.x;
    t[0] = tk0;
    t[1] = wall_clock64();
  }
#endif
}

template <int LV, int LD, int CODE, bool PRE, bool TRAIN>
static int launch(const MlpArgs& a, hipStream_t st) {
  const long long nblk = (a.P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  const size_t lds = LDS_BONES_OFF + (PRE ? 0 : MAX_TILE_RAYS * 72 * 16);
  auto kern = k_mlp_fwd<LV, LD, CODE, PRE, TRAIN>;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &lds_set);
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
  return check_launch("k_mlp_fwd");
}

int mlp_dispatch(const AnerfConfig* cfg, const MlpArgs& a, bool pre, bool train, hipStream_t st) {
  const int lv = cfg->multires, ld = cfg->multires_views, cd = cfg->framecode_ch;
  if (lv != 7) return set_error(ANERF_E_CONFIG, "multires must be 7");
  if (a.P > 0xFFFFFF00ll) return set_error(ANERF_E_SHAPE, "more than 2^32 - 256 samples in one call");   // div_samples is 32-bit
  if (pre && train) return set_error(ANERF_E_CONFIG, "training forward needs the fused (not pre-encoded) path");
#define ANERF_CASE(LD_, CD_)                                               \
  if (ld == LD_ && cd == CD_) {                                            \
    if (pre) return launch<7, LD_, CD_, true, false>(a, st);               \
    return train ? launch<7, LD_, CD_, false, true>(a, st) : launch<7, LD_, CD_, false, false>(a, st); \
  }
  ANERF_CASE(4, 0)
  ANERF_CASE(4, 16)
  ANERF_CASE(0, 0)
#undef ANERF_CASE
  return set_error(ANERF_E_CONFIG, "unsupported (multires_views, framecode_ch); built: (4,0) (4,16) (0,0)");
}

// density query: pts [P,3] under one pose -> sigma logit [P]
int mlp_density_entry(const float* packed, const float* aux, const float* pts, const float* skts, float tau_v,
                      const float* cut_v, long long P, int nstages_trunk, float* sigma, hipStream_t st, int gate_bones) {
  MlpArgs a;
  memset(&a, 0, sizeof(a));   // (pnoise = nullptr)
  a.packed = packed; a.aux = aux; a.z = pts; a.skts = skts; a.cut_v = cut_v; a.cut_d = cut_v; a.raw = sigma;
  a.P = P; a.Ppad = P; a.skt_stride = 0; a.S = 1; a.N = 1; a.nstages = nstages_trunk; a.tau_v = tau_v; a.tau_d = tau_v; a.gate_bones = gate_bones;
  const long long nblk = (P + TILE - 1) / TILE;
  if (nblk <= 0) return ANERF_OK;
  if (P > 0xFFFFFF00ll) return set_error(ANERF_E_SHAPE, "more than 2^32 - 256 points in one call");
  const size_t lds = LDS_BONES_OFF + MAX_TILE_RAYS * 72 * 16;
  auto kern = k_mlp_fwd<7, 0, 0, false, false, 1>;
  static unsigned long long lds_set = 0;   // per-device bits, see ensure_dynamic_lds
  ensure_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &lds_set);
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, a);
  return check_launch("k_mlp_fwd<density>");
}

int mlp_raw_entry(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays, int ray_stride,
                  const float* z, const float* skts, long long skt_stride, const float* cam, const float* codes,
                  int n_codes, float tau_v, float tau_d, const float* cut_v, const float* cut_d, const float* x,
                  int x_width, long long P, int N, int S, int nstages, float* raw, bool pre, const AnerfSaved* sv,
                  hipStream_t st, const float* pnoise, const float* tau_dev) {
  MlpArgs a;
  a.pnoise = pnoise;
  a.tau_dev = tau_dev;
  a.packed = packed; a.aux = aux; a.rays = rays; a.z = z; a.skts = skts; a.cam = cam; a.codes = codes;
  a.cut_v = cut_v; a.cut_d = cut_d; a.x = x; a.raw = raw; a.P = P; a.skt_stride = skt_stride;
  a.S = S; a.N = N; a.ray_stride = ray_stride; a.n_codes = n_codes; a.x_width = x_width; a.nstages = nstages;
  a.tau_v = tau_v; a.tau_d = tau_d; a.gate_bones = cfg->cutoff_bones;
  a.save_h = a.save_f = a.save_g = a.save_x = a.save_u = nullptr;
#if defined(ANERF_EXP_TILE_TIMING)
  a.save_u = g_tile_timing_buf;
#endif
#if defined(ANERF_EXP_STAGE_TIMING)
  a.tbuf = reinterpret_cast<unsigned long long*>(g_tile_timing_buf);
#endif
  a.Ppad = P;
  if (sv) {
    a.save_h = sv->h; a.save_f = sv->f; a.save_g = sv->g; a.save_x = sv->x; a.save_u = sv->u; a.Ppad = sv->p_pad;
  }
  return mlp_dispatch(cfg, a, pre, sv != nullptr, st);
}

}  // namespace anerf

#if defined(ANERF_EXP_TILE_TIMING) || defined(ANERF_EXP_STAGE_TIMING)
extern "C" void anerf_debug_set_timing_buf(float* p) { anerf::g_tile_timing_buf = p; }
#endif
