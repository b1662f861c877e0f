// anerf_pose.hip -- backward of the fused encoding (A4-A6) for pose optimisation and frame codes:
//   k_encode_bwd   dX'/dU' (stream column order, from k_mlp_bwd_in) -> d y_j (bone-space point) and d q_j
//                  (q_j = R_j d, bone-space ray direction) per sample; autograd of core/encoders.py:8-37,110-122,
//                  181-193 + core/cutoff_embedder.py:111-174 restated by hand
//   k_pose_reduce  sum over the samples of a ray: d skts[n][j][r][:] = (sum dy_r x^T + (sum dq_r) d^T | sum dy_r)
//                  (deterministic: one thread per (ray, joint, row), no atomics)
//   k_code_rowsum / k_code_reduce  d codes[c] += sum over the rays with cam_idx = c and their samples of dU'[code columns]
// VALU / HBM-bound helpers; only run when skts or frame codes require gradients (Mixamo config).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf_dev.h"

namespace anerf {

// One thread per (sample, lane half h, joint quad G): it owns joints j = 8G + 4h + t (t = 0..3), i.e. k-group column t of
// k-groups g == G (mod 3) of the x part and direction components 12G..12G+11 of the lane half -- a third of what one
// lane of the forward kernel encodes.  (One thread per sample-half held 12 joints' worth of state: 256 VGPRs + 210
// AGPRs of spill space, one wave per SIMD, every strided gradient load exposed; the quad version needs < 128 registers.)
template <int LD>
__global__ __launch_bounds__(256) void k_encode_bwd(const float* __restrict__ dx, const float* __restrict__ du, int uw,
                                                    const float* __restrict__ rays, int ray_stride,
                                                    const float* __restrict__ z, const float* __restrict__ skts,
                                                    long long skt_stride, float tau_v, float tau_d,
                                                    const float* __restrict__ cut_v, const float* __restrict__ cut_d,
                                                    long long P, int S, float* __restrict__ dY, float* __restrict__ dQ,
                                                    const float* __restrict__ pnoise, int gate_bones,
                                                    const float* __restrict__ tau_dev) {
  constexpr int LV = 7;
  if (tau_dev) {   // ABI revision 6: the step block's {tau_v, tau_d} (a captured training step), else the arguments
    tau_v = tau_dev[0];
    tau_d = tau_dev[1];
  }
  const long long gid = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const long long p = gid / 6;
  const int sub = (int)(gid - p * 6);
  const int h = sub & 1, G = sub >> 1;
  if (p >= P) return;
  const long long ray = p / S;
  const float* rp = rays + ray * ray_stride;
  const float zz = z[p];
  const float d0 = rp[3], d1 = rp[4], d2 = rp[5];
  float x0 = fmaf(d0, zz, rp[0]), x1 = fmaf(d1, zz, rp[1]), x2 = fmaf(d2, zz, rp[2]);
  if (pnoise) {   // ray_noise_std > 0 (raycasters.py:660): the sample points carry an additive offset
    x0 += pnoise[3 * p]; x1 += pnoise[3 * p + 1]; x2 += pnoise[3 * p + 2];
  }
  const float* sk = skts + ray * skt_stride;
  float v[4], wv[4], wvp[4], wd[4], wdp[4], rh[12], e[12], qn[4], dv[4], dr[12], de[12];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = 8 * G + 4 * h + t;
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(sk + j * 16), r1 = *reinterpret_cast<const f32x4*>(sk + j * 16 + 4),
                r2 = *reinterpret_cast<const f32x4*>(sk + j * 16 + 8);
    const float y0 = r0.x * x0 + r0.y * x1 + r0.z * x2 + r0.w;
    const float y1 = r1.x * x0 + r1.y * x1 + r1.z * x2 + r1.w;
    const float y2 = r2.x * x0 + r2.y * x1 + r2.z * x2 + r2.w;
    const float n = sqrtf(y0 * y0 + y1 * y1 + y2 * y2);
    const float inv = 1.f / fmaxf(n, 1e-12f);
    v[t] = n;
    rh[3 * t] = y0 * inv; rh[3 * t + 1] = y1 * inv; rh[3 * t + 2] = y2 * inv;
    const float q0 = r0.x * d0 + r0.y * d1 + r0.z * d2, q1 = r1.x * d0 + r1.y * d1 + r1.z * d2,
                q2 = r2.x * d0 + r2.y * d1 + r2.z * d2;
    const float qq = sqrtf(q0 * q0 + q1 * q1 + q2 * q2);
    const float qi = 1.f / fmaxf(qq, 1e-12f);
    qn[t] = qq;
    e[3 * t] = q0 * qi; e[3 * t + 1] = q1 * qi; e[3 * t + 2] = q2 * qi;
    wv[t] = cutoff_gate(tau_v, n, cut_v[j]);
    wvp[t] = -tau_v * wv[t] * (1.f - wv[t]);
    wd[t] = cutoff_gate(tau_d, n, cut_d[j]);
    wdp[t] = -tau_d * wd[t] * (1.f - wd[t]);
    dv[t] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) de[i] = 0.f;
  // ---- x part (stream order: k-group kg holds joints 8 (kg % 3) + 4 h + t): raw, then sin / cos per band, then directions
  const float* gx = dx + p * 432 + 4 * h;
  {
    const f32x4 Gr = *reinterpret_cast<const f32x4*>(gx + 8 * G);
    const float Gt[4] = {Gr.x, Gr.y, Gr.z, Gr.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) dv[t] += Gt[t] * (wv[t] + v[t] * wvp[t]);
  }
#pragma unroll
  for (int f = 0; f < LV; ++f) {
    const f32x4 Gs = *reinterpret_cast<const f32x4*>(gx + 8 * (3 + 6 * f + G));
    const f32x4 Gc = *reinterpret_cast<const f32x4*>(gx + 8 * (6 + 6 * f + G));
    const float gs[4] = {Gs.x, Gs.y, Gs.z, Gs.w}, gc[4] = {Gc.x, Gc.y, Gc.z, Gc.w};
    const float F = (float)(1 << f);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float s, c;
      sincos_f32(v[t] * F, s, c);
      dv[t] += gs[t] * (F * c * wv[t] + s * wvp[t]) + gc[t] * (-F * s * wv[t] + c * wvp[t]);
    }
  }
  // bone-direction components 12 G .. 12 G + 11 of this half = k-groups 45 + 3 G .. + 2
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const f32x4 Gd = *reinterpret_cast<const f32x4*>(gx + 8 * (45 + 3 * G + g));
    dr[4 * g] = Gd.x; dr[4 * g + 1] = Gd.y; dr[4 * g + 2] = Gd.z; dr[4 * g + 3] = Gd.w;
  }
  // ---- view part: raw + LD sin/cos bands of the 12 owned direction components, gated by wd(v)
  const float* gu = du + p * uw + 4 * h;
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const f32x4 Gr = *reinterpret_cast<const f32x4*>(gu + 8 * (3 * G + g));
    const float Gt[4] = {Gr.x, Gr.y, Gr.z, Gr.w};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = 4 * g + t, a = i / 3;
      de[i] += Gt[t] * wd[a];
      dv[a] += Gt[t] * e[i] * wdp[a];
    }
  }
#pragma unroll
  for (int f = 0; f < LD; ++f)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const f32x4 Gs = *reinterpret_cast<const f32x4*>(gu + 8 * (9 * (1 + 2 * f) + 3 * G + g));
      const f32x4 Gc = *reinterpret_cast<const f32x4*>(gu + 8 * (9 * (2 + 2 * f) + 3 * G + g));
      const float gs[4] = {Gs.x, Gs.y, Gs.z, Gs.w}, gc[4] = {Gc.x, Gc.y, Gc.z, Gc.w};
      const float F = (float)(1 << f);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = 4 * g + t, a = i / 3;
        float s, c;
        sincos_f32(e[i] * F, s, c);
        de[i] += (gs[t] * c - gc[t] * s) * F * wd[a];
        dv[a] += (gs[t] * s + gc[t] * c) * wdp[a];
      }
    }
  // ---- through the norms: y -> (v, r),  q -> e
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = 8 * G + 4 * h + t;
    float dot = dr[3 * t] * rh[3 * t] + dr[3 * t + 1] * rh[3 * t + 1] + dr[3 * t + 2] * rh[3 * t + 2];
    if (gate_bones) {   // cutoff_bones: the network saw r * w(v) -- d r = w * d(r w), and the gate's slope adds (d(r w) . r) w' to d v
      dv[t] += dot * wvp[t];
      dot *= wv[t];
      dr[3 * t] *= wv[t]; dr[3 * t + 1] *= wv[t]; dr[3 * t + 2] *= wv[t];
    }
    const float iv = 1.f / fmaxf(v[t], 1e-12f);
    const float dote = de[3 * t] * e[3 * t] + de[3 * t + 1] * e[3 * t + 1] + de[3 * t + 2] * e[3 * t + 2];
    const float iq = 1.f / fmaxf(qn[t], 1e-12f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dY[p * 72 + 3 * j + c] = dv[t] * rh[3 * t + c] + (dr[3 * t + c] - dot * rh[3 * t + c]) * iv;
      dQ[p * 72 + 3 * j + c] = (de[3 * t + c] - dote * e[3 * t + c]) * iq;
    }
  }
}

// One block of 320 threads per ray: thread (g, l) = (sample group g < 4, column l < 72 = (joint, row r < 3)) sums the ray's samples
// s = g, g + 4, ... in order; the four group sums are combined in the fixed order (g0 + g1) + (g2 + g3) through LDS (round 6: a
// 4-way split of what was ONE 64..80-step dependent chain per thread -- the kernel is latency-bound, 27 -> ~8 us at 384 rays).
// Not accumulating (first network pass of a step): threads 288..311 also write row 3 of their joint's 4 x 4 block (= 0: the
// homogeneous row has no gradient), so the caller needs no zero fill of dskts.
__global__ __launch_bounds__(320) void k_pose_reduce(const float* __restrict__ dY, const float* __restrict__ dQ,
                                                     const float* __restrict__ rays, int ray_stride,
                                                     const float* __restrict__ z, int n, int S, int accumulate,
                                                     float* __restrict__ dskts, const float* __restrict__ pnoise) {
  __shared__ float sh[4][72][5];
  const int ray = blockIdx.x, t = threadIdx.x;
  if (ray >= n) return;
  const int g = t / 72, l = t - 72 * g;
  if (g < 4) {
    const float* rp = rays + (long long)ray * ray_stride;
    const float o0 = rp[0], o1 = rp[1], o2 = rp[2], d0 = rp[3], d1 = rp[4], d2 = rp[5];
    float R0 = 0.f, R1 = 0.f, R2 = 0.f, T = 0.f, Q = 0.f;
    for (int s = g; s < S; s += 4) {
      const long long p = (long long)ray * S + s;
      const float zz = z[p];
      const float dy = dY[p * 72 + l], dq = dQ[p * 72 + l];
      float x0 = fmaf(d0, zz, o0), x1 = fmaf(d1, zz, o1), x2 = fmaf(d2, zz, o2);
      if (pnoise) {
        x0 += pnoise[3 * p]; x1 += pnoise[3 * p + 1]; x2 += pnoise[3 * p + 2];
      }
      R0 = fmaf(dy, x0, R0);
      R1 = fmaf(dy, x1, R1);
      R2 = fmaf(dy, x2, R2);
      T += dy;
      Q += dq;
    }
    sh[g][l][0] = R0; sh[g][l][1] = R1; sh[g][l][2] = R2; sh[g][l][3] = T; sh[g][l][4] = Q;
  }
  __syncthreads();
  if (t < 72) {
    const float* rp = rays + (long long)ray * ray_stride;
    const float d0 = rp[3], d1 = rp[4], d2 = rp[5];
    float v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (sh[0][t][k] + sh[1][t][k]) + (sh[2][t][k] + sh[3][t][k]);
    const int j = t / 3, r = t - 3 * j;
    float* o = dskts + ((long long)ray * 24 + j) * 16 + r * 4;
    const float g0 = fmaf(v[4], d0, v[0]), g1 = fmaf(v[4], d1, v[1]), g2 = fmaf(v[4], d2, v[2]);
    if (accumulate) {   // second network pass of a step (anerf_backward): same sum autograd forms from two overwriting calls
      o[0] += g0; o[1] += g1; o[2] += g2; o[3] += v[3];
    } else {
      o[0] = g0; o[1] = g1; o[2] = g2; o[3] = v[3];
    }
  } else if (!accumulate && t >= 288 && t < 312) {
    float* o = dskts + ((long long)ray * 24 + (t - 288)) * 16 + 12;
    o[0] = 0.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
  }
}

// Frame-code gradients in two fixed-order stages (bit-reproducible; an earlier version added per-ray sums with float
// atomics, the only run-to-run nondeterminism of the training step):
//   k_code_rowsum  one wave per ray, 4 sample groups x 16 columns: rowsum[ray][lane] = sum over the ray's samples of
//                  dU'[code column lane] (each group sums samples k = g mod 4, groups combined by two fixed shuffles)
//   k_code_reduce  one block per frame code c (exits at once when no ray of the batch carries c): thread (slot, lane)
//                  adds rowsum[r][lane] of the rays r = slot, slot + 64, ... with cam_idx = c, the 64 slot sums are then
//                  added in slot order; one writer per element of dcodes.
__global__ __launch_bounds__(256) void k_code_rowsum(const float* __restrict__ du, int uw, int n, int S,
                                                     float* __restrict__ rowsum) {
  const int l64 = threadIdx.x & 63, lane = l64 & 15, grp = l64 >> 4;      // 4 sample groups x 16 code columns
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;
  float s = 0.f;
  for (int k = grp; k < S; k += 4) s += du[((long long)ray * S + k) * uw + (uw - 16) + lane];
  s += __shfl_xor(s, 16);      // (g0 + g1), (g2 + g3): fixed pairing, same result in both lanes of a pair
  s += __shfl_xor(s, 32);
  if (grp == 0) rowsum[ray * 16 + lane] = s;
}

__device__ __forceinline__ int code_of(const float* __restrict__ cam, int r, int n_codes) {
  const int ci = (int)cam[r];
  return ci < 0 ? 0 : (ci >= n_codes ? n_codes - 1 : ci);
}

constexpr int CODE_SLOTS = 64;
__global__ __launch_bounds__(16 * CODE_SLOTS) void k_code_reduce(const float* __restrict__ rowsum, const float* __restrict__ cam,
                                                                 int n, int n_codes, float* __restrict__ dcodes) {
  __shared__ float sh[CODE_SLOTS][16];
  const int c = blockIdx.x, lane = threadIdx.x & 15, slot = threadIdx.x >> 4;
  int any = 0;
  for (int r = threadIdx.x; r < n; r += 16 * CODE_SLOTS) any |= code_of(cam, r, n_codes) == c;
  if (!__syncthreads_or(any)) return;
  float s = 0.f;
  for (int r = slot; r < n; r += CODE_SLOTS) {
    const float v = rowsum[r * 16 + lane];
    s += code_of(cam, r, n_codes) == c ? v : 0.f;
  }
  sh[slot][lane] = s;
  __syncthreads();
  if (slot == 0) {
    float t = 0.f;
    for (int q = 0; q < CODE_SLOTS; ++q) t += sh[q][lane];
    dcodes[c * 16 + lane] += t;
  }
}

int launch_encode_bwd(int ld, const float* dx, const float* du, int uw, const float* rays, int ray_stride, const float* z,
                      const float* skts, long long skt_stride, float tau_v, float tau_d, const float* cut_v,
                      const float* cut_d, int n, int S, float* dY, float* dQ, float* dskts, bool accumulate, hipStream_t st,
                      const float* pnoise, int gate_bones, const float* tau_dev) {
  const long long P = (long long)n * S;
  const unsigned blocks = (unsigned)((6 * P + 255) / 256);
  if (ld == 4)
    hipLaunchKernelGGL(k_encode_bwd<4>, dim3(blocks), dim3(256), 0, st, dx, du, uw, rays, ray_stride, z, skts, skt_stride,
                       tau_v, tau_d, cut_v, cut_d, P, S, dY, dQ, pnoise, gate_bones, tau_dev);
  else
    hipLaunchKernelGGL(k_encode_bwd<0>, dim3(blocks), dim3(256), 0, st, dx, du, uw, rays, ray_stride, z, skts, skt_stride,
                       tau_v, tau_d, cut_v, cut_d, P, S, dY, dQ, pnoise, gate_bones, tau_dev);
  int rc = check_launch("k_encode_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(k_pose_reduce, dim3(n), dim3(320), 0, st, (const float*)dY, (const float*)dQ, rays, ray_stride, z, n, S,
                     accumulate ? 1 : 0, dskts, pnoise);
  return check_launch("k_pose_reduce");
}

int launch_pose_reduce(const float* dY, const float* dQ, const float* rays, int ray_stride, const float* z, int n, int S, float* dskts,
                       bool accumulate, hipStream_t st, const float* pnoise) {
  hipLaunchKernelGGL(k_pose_reduce, dim3(n), dim3(320), 0, st, dY, dQ, rays, ray_stride, z, n, S, accumulate ? 1 : 0, dskts, pnoise);
  return check_launch("k_pose_reduce");
}

int launch_code_reduce(const float* du, int uw, const float* cam, int n, int S, int n_codes, float* rowsum, float* dcodes,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_code_rowsum, dim3((n + 3) / 4), dim3(256), 0, st, du, uw, n, S, rowsum);
  int rc = check_launch("k_code_rowsum");
  if (rc) return rc;
  hipLaunchKernelGGL(k_code_reduce, dim3(n_codes), dim3(16 * CODE_SLOTS), 0, st, (const float*)rowsum, cam, n, n_codes, dcodes);
  return check_launch("k_code_reduce");
}

}  // namespace anerf
