// anerf_aux.hip -- the small per-ray kernels around the fused MLP (gfx950):
//   k_ray_bounds   get_near_far_in_cylinder            core/utils/ray_utils.py:292-344
//   k_coarse_z     sample_from_lineseg                 core/utils/ray_utils.py:204-251
//   k_composite    NeRF.raw2outputs                    core/networks/nerf.py:150-205
//   k_importance   isample_from_lineseg + sample_pdf   core/utils/ray_utils.py:157-201,255-289
//   k_pack         parameter gather into the packed weight images
// All are HBM/latency-bound byte movers: coalesced loads, one wavefront per ray where a scan is needed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf_dev.h"

namespace anerf {

// ------------------------------------------------------------------------------------------------
__global__ void k_pack(AnerfNetParams P, const int32_t* __restrict__ table, long long n, float* __restrict__ out) {
  const float* tens[24];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    tens[i] = P.w[i];
    tens[12 + i] = P.b[i];
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int32_t e = table[i];
    float val = 0.f;
    if (e >= 0) {
      const int id = e >> 24, off = e & 0xFFFFFF;
      const float* src = P.w[0];
#pragma unroll
      for (int k = 0; k < 24; ++k)
        if (id == k) src = tens[k];
      val = src[off] * sched_scale(P, id, off);
    }
    out[i] = val;
  }
}

// ------------------------------------------------------------------------------------------------
// A2.  One thread per ray.  stats = {sum_near, sum_far, cnt_near, cnt_far} over non-NaN rows (for the fallback), as four
// 64-bit integers: the sums in 2^-32 fixed point, so that the cross-wave accumulation (integer atomics) is exact and
// order-independent -- the fallback value, and with it every z of a missed ray, is bit-reproducible.
constexpr double STATS_FIX = 4294967296.0;   // 2^32
// near / far of one ray: the ray/circle intersection in the ground plane (ray_utils.py:292-326); NaN when the ray misses the circle
__device__ __forceinline__ void ray_bounds_of(const float* __restrict__ r, const float* __restrict__ c, float& nn, float& ff) {
  const float ox = r[0], oz = r[2], dx = r[3], dz = r[5], near = r[6], far = r[7];
  const float pnx = fmaf(dx, near, ox), pnz = fmaf(dz, near, oz);
  const float pfx = fmaf(dx, far, ox), pfz = fmaf(dz, far, oz);
  const float ncx = c[0] - pnx, ncz = c[1] - pnz;
  const float nfx = pfx - pnx, nfz = pfz - pnz;
  const float nf_len = sqrtf(nfx * nfx + nfz * nfz);
  const float scale = sqrtf(dx * dx + dz * dz);
  const float cross = ncx * nfz - ncz * nfx;
  const float dist = fabsf(cross) / nf_len;
  const float Q = sqrtf(c[2] * c[2] - dist * dist);   // NaN when the ray misses the circle
  const float K = (ncx * nfx + ncz * nfz) / nf_len;
  const float inside = (Q < K) ? 1.f : 0.f;
  nn = near + inside * (K - Q) / scale;
  ff = near + (K + Q) / scale;
}

__global__ void k_ray_bounds(const float* __restrict__ rays, int ray_stride, const float* __restrict__ cyls, int cyl_stride, int n,
                             float* __restrict__ near_far, unsigned long long* __restrict__ stats) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float sn = 0.f, sf = 0.f, cn = 0.f, cf = 0.f;
  if (i < n) {
    float nn, ff;
    ray_bounds_of(rays + (long long)i * ray_stride, cyls + (long long)i * cyl_stride, nn, ff);   // cyl_stride 5, or 0: one shared cylinder
    near_far[2 * i + 0] = nn;
    near_far[2 * i + 1] = ff;
    if (nn == nn) { sn = nn; cn = 1.f; }
    if (ff == ff) { sf = ff; cf = 1.f; }
  }
  // wave reduce (fixed shuffle tree), one integer atomic per wave and statistic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sn += __shfl_xor(sn, o);
    sf += __shfl_xor(sf, o);
    cn += __shfl_xor(cn, o);
    cf += __shfl_xor(cf, o);
  }
  if ((threadIdx.x & 63) == 0 && (cn + cf) > 0.f) {
    atomicAdd(stats + 0, (unsigned long long)__double2ll_rn((double)sn * STATS_FIX));   // two's complement: signed sums work
    atomicAdd(stats + 1, (unsigned long long)__double2ll_rn((double)sf * STATS_FIX));
    atomicAdd(stats + 2, (unsigned long long)cn);
    atomicAdd(stats + 3, (unsigned long long)cf);
  }
}

// A3.  One thread per sample; applies the NaN fallback (mean of the call's valid rows, else the placeholder).
__global__ void k_coarse_z(const float* __restrict__ near_far, const unsigned long long* __restrict__ stats,
                           const float* __restrict__ rays, int ray_stride, int n, int S,
                           const float* __restrict__ t_rand, int lindisp, float* __restrict__ z_out,
                           float* __restrict__ near_far_fixed) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= (long long)n * S) return;
  const int ray = (int)(idx / S), s = (int)(idx - (long long)ray * S);
  float nn = near_far[2 * ray], ff = near_far[2 * ray + 1];
  if (!(nn == nn) || !(ff == ff)) {
    // torch.where(isnan(Q)) rows: both replaced (ray_utils.py:331-342)
    const float* r = rays + (long long)ray * ray_stride;
    nn = stats[2] > 0 ? (float)((double)(long long)stats[0] / STATS_FIX / (double)stats[2]) : r[6];
    ff = stats[3] > 0 ? (float)((double)(long long)stats[1] / STATS_FIX / (double)stats[3]) : r[7];
  }
  if (s == 0 && near_far_fixed) {
    near_far_fixed[2 * ray] = nn;
    near_far_fixed[2 * ray + 1] = ff;
  }
  auto zat = [&](int k) -> float {
    const float t = (float)k / (float)(S - 1);
    return lindisp ? 1.f / (1.f / nn * (1.f - t) + 1.f / ff * t) : nn * (1.f - t) + ff * t;
  };
  float z = zat(s);
  if (t_rand) {
    const float lo = s == 0 ? z : 0.5f * (z + zat(s - 1));
    const float hi = s == S - 1 ? z : 0.5f * (zat(s + 1) + z);
    z = lo + (hi - lo) * t_rand[idx];
  }
  z_out[idx] = z;
}

// A2 + A3 in ONE launch for the caster-call sizes of training and chunked rendering (n <= BOUNDS_Z_MAX_RAYS; round 6: the stats
// zero fill + k_ray_bounds + k_coarse_z were three dispatches on the ~4.7 us launch floor).  One thread per sample computes ITS
// ray's bounds (the same inlined arithmetic) -- no near/far array, no grid-wide dependency -- unless one of the block's rays misses
// the cylinder: only then does the block need the call-wide NaN-mean statistics, and it recomputes them itself, in exactly
// k_ray_bounds' form (64-ray groups aligned to multiples of 64, the same shuffle tree per group, the same 2^-32 fixed-point
// conversion per group; integer sums are order-independent), so the fallback values are bit-identical to the two-kernel route.
constexpr int BOUNDS_Z_MAX_RAYS = 16384;
__global__ __launch_bounds__(256) void k_bounds_z(const float* __restrict__ rays, int ray_stride, const float* __restrict__ cyls,
                                                  int cyl_stride, int n, int S, const float* __restrict__ t_rand, int lindisp,
                                                  float* __restrict__ z_out) {
  __shared__ long long sh_s[4][2];
  __shared__ unsigned long long sh_c[4][2];
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  const bool valid = idx < (long long)n * S;
  const int ray = valid ? (int)(idx / S) : 0, s = valid ? (int)(idx - (long long)ray * S) : 0;
  float nn = 0.f, ff = 0.f;
  if (valid) ray_bounds_of(rays + (long long)ray * ray_stride, cyls + (long long)ray * cyl_stride, nn, ff);
  const int bad = valid && (!(nn == nn) || !(ff == ff));
  if (__syncthreads_or(bad)) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long long a0 = 0, a1 = 0;
    unsigned long long a2 = 0, a3 = 0;
    for (int g = wave; g < (n + 63) / 64; g += 4) {
      const int i = g * 64 + lane;
      float sn = 0.f, sf = 0.f, cn = 0.f, cf = 0.f;
      if (i < n) {
        float bn, bf;
        ray_bounds_of(rays + (long long)i * ray_stride, cyls + (long long)i * cyl_stride, bn, bf);
        if (bn == bn) { sn = bn; cn = 1.f; }
        if (bf == bf) { sf = bf; cf = 1.f; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        sn += __shfl_xor(sn, o);
        sf += __shfl_xor(sf, o);
        cn += __shfl_xor(cn, o);
        cf += __shfl_xor(cf, o);
      }
      if ((cn + cf) > 0.f) {
        a0 += __double2ll_rn((double)sn * STATS_FIX);
        a1 += __double2ll_rn((double)sf * STATS_FIX);
        a2 += (unsigned long long)cn;
        a3 += (unsigned long long)cf;
      }
    }
    if (lane == 0) {
      sh_s[wave][0] = a0; sh_s[wave][1] = a1; sh_c[wave][0] = a2; sh_c[wave][1] = a3;
    }
    __syncthreads();
    if (bad) {   // torch.where(isnan(Q)) rows: both replaced (ray_utils.py:331-342)
      const long long t0 = sh_s[0][0] + sh_s[1][0] + sh_s[2][0] + sh_s[3][0], t1 = sh_s[0][1] + sh_s[1][1] + sh_s[2][1] + sh_s[3][1];
      const unsigned long long c0 = sh_c[0][0] + sh_c[1][0] + sh_c[2][0] + sh_c[3][0], c1 = sh_c[0][1] + sh_c[1][1] + sh_c[2][1] + sh_c[3][1];
      const float* r = rays + (long long)ray * ray_stride;
      nn = c0 > 0 ? (float)((double)t0 / STATS_FIX / (double)c0) : r[6];
      ff = c1 > 0 ? (float)((double)t1 / STATS_FIX / (double)c1) : r[7];
    }
  }
  if (!valid) return;
  auto zat = [&](int k) -> float {
    const float t = (float)k / (float)(S - 1);
    return lindisp ? 1.f / (1.f / nn * (1.f - t) + 1.f / ff * t) : nn * (1.f - t) + ff * t;
  };
  float z = zat(s);
  if (t_rand) {
    const float lo = s == 0 ? z : 0.5f * (z + zat(s - 1));
    const float hi = s == S - 1 ? z : 0.5f * (zat(s + 1) + z);
    z = lo + (hi - lo) * t_rand[idx];
  }
  z_out[idx] = z;
}

// ------------------------------------------------------------------------------------------------
// A10.  One wavefront per ray; lane i owns the contiguous samples [i*C, (i+1)*C), C = ceil(S/64) <= 8.
// Transmittance = exclusive product scan: in-lane running product + wave-level shuffle scan.
__device__ __forceinline__ float density_act(int act, float x, float shift) {
  if (act == 0) return fmaxf(x, 0.f);
  const float y = x - shift;            // F.softplus(beta=1, threshold=20)
  return y > 20.f ? y : log1pf(expf(y));
}

__global__ __launch_bounds__(256) void k_composite(const float* __restrict__ raw, const float* __restrict__ z,
                                                   const float* __restrict__ rays, int ray_stride,
                                                   const float* __restrict__ noise, int n, int S, int act,
                                                   float inv_B, float shift, float* __restrict__ rgb_map,
                                                   float* __restrict__ disp_map, float* __restrict__ acc_map,
                                                   float* __restrict__ weights, float* __restrict__ alpha_out,
                                                   float* __restrict__ depth_map) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;
  const int C = (S + 63) >> 6;
  const float* rp = rays + (long long)ray * ray_stride;
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const long long base = (long long)ray * S;
  float al[8], cr[8], cg[8], cb[8], zz[8];
  float prod = 1.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int s = lane * C + c;
    al[c] = 0.f; cr[c] = cg[c] = cb[c] = 0.f; zz[c] = 0.f;
    if (c < C && s < S) {
      const f32x4 rw = *reinterpret_cast<const f32x4*>(raw + (base + s) * 4);
      const float zc = z[base + s];
      const float delta = (s == S - 1 ? 1e10f : (z[base + s + 1] - zc)) * dn;
      float pre = rw.w * inv_B;
      if (noise) pre += noise[base + s];
      const float a = 1.f - expf(-density_act(act, pre, shift) * delta);
      al[c] = a;
      zz[c] = zc;
      cr[c] = 1.002f / (1.f + expf(-rw.x)) - 0.001f;
      cg[c] = 1.002f / (1.f + expf(-rw.y)) - 0.001f;
      cb[c] = 1.002f / (1.f + expf(-rw.z)) - 0.001f;
      prod *= (1.f - a + 1e-10f);
    }
  }
  // inclusive product scan of per-lane products, then shift to exclusive
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float up = __shfl_up(incl, o);
    if (lane >= o) incl *= up;
  }
  float T = __shfl_up(incl, 1);
  if (lane == 0) T = 1.f;
  float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int s = lane * C + c;
    if (c < C && s < S) {
      const float w = al[c] * T;
      weights[base + s] = w;
      alpha_out[base + s] = al[c];
      sr = fmaf(w, cr[c], sr);
      sg = fmaf(w, cg[c], sg);
      sb = fmaf(w, cb[c], sb);
      sd = fmaf(w, zz[c], sd);
      sa += w;
      T *= (1.f - al[c] + 1e-10f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sr += __shfl_xor(sr, o);
    sg += __shfl_xor(sg, o);
    sb += __shfl_xor(sb, o);
    sd += __shfl_xor(sd, o);
    sa += __shfl_xor(sa, o);
  }
  if (lane == 0) {
    rgb_map[3 * ray + 0] = sr;
    rgb_map[3 * ray + 1] = sg;
    rgb_map[3 * ray + 2] = sb;
    float disp = 1.f / fmaxf(1e-10f, sd / (sa + 1e-10f));
    if (fabsf(sa) <= 1e-8f) disp = 0.f;          // torch.isclose(sum_w, 0): atol 1e-8 (+ rtol * 0)
    disp_map[ray] = disp;
    acc_map[ray] = fminf(sa, 1.f);
    if (depth_map) depth_map[ray] = sd;
  }
}

// ------------------------------------------------------------------------------------------------
// Backward of A10 (autograd of nerf.py:150-205 w.r.t. raw).  One wavefront per ray, same sample ownership as the
// forward; recomputes alpha/T/w from raw, then
//   g_w_i   = g_rgb . c_i + g_depth z_i + g_A (+ g_weights_i)
//   g_a_k   = g_w_k T_k - (sum_{i>k} g_w_i w_i) / (1 - a_k + 1e-10) (+ g_alpha_k)     [suffix scan over the ray]
//   d sigma = g_a * delta * (1 - a) * act'(pre) / B ;   d c_raw = g_rgb * w * 1.002 * s (1 - s)
__global__ __launch_bounds__(256) void k_composite_bwd(const float* __restrict__ raw, const float* __restrict__ z,
                                                       const float* __restrict__ rays, int ray_stride,
                                                       const float* __restrict__ noise, int n, int S, int act,
                                                       float inv_B, float shift, const float* __restrict__ g_rgb,
                                                       const float* __restrict__ g_acc, const float* __restrict__ g_disp,
                                                       const float* __restrict__ g_alpha,
                                                       const float* __restrict__ g_weights, float* __restrict__ draw) {
  const int lane = threadIdx.x & 63;
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= n) return;
  const int C = (S + 63) >> 6;
  const float* rp = rays + (long long)ray * ray_stride;
  const float dn = sqrtf(rp[3] * rp[3] + rp[4] * rp[4] + rp[5] * rp[5]);
  const long long base = (long long)ray * S;
  float al[8], sr[8], sg[8], sb[8], zz[8], dl[8], dact[8];
  float prod = 1.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int s = lane * C + c;
    al[c] = 0.f; sr[c] = sg[c] = sb[c] = 0.f; zz[c] = 0.f; dl[c] = 0.f; dact[c] = 0.f;
    if (c < C && s < S) {
      const f32x4 rw = *reinterpret_cast<const f32x4*>(raw + (base + s) * 4);
      const float zc = z[base + s];
      const float delta = (s == S - 1 ? 1e10f : (z[base + s + 1] - zc)) * dn;
      float pre = rw.w * inv_B;
      if (noise) pre += noise[base + s];
      const float a = 1.f - expf(-density_act(act, pre, shift) * delta);
      al[c] = a;
      zz[c] = zc;
      dl[c] = delta;
      dact[c] = act == 0 ? (pre > 0.f ? 1.f : 0.f) : 1.f / (1.f + expf(-(pre - shift)));
      sr[c] = 1.f / (1.f + expf(-rw.x));
      sg[c] = 1.f / (1.f + expf(-rw.y));
      sb[c] = 1.f / (1.f + expf(-rw.z));
      prod *= (1.f - a + 1e-10f);
    }
  }
  float incl = prod;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float up = __shfl_up(incl, o);
    if (lane >= o) incl *= up;
  }
  float T0 = __shfl_up(incl, 1);
  if (lane == 0) T0 = 1.f;
  // forward sums needed by the disparity / accumulation gradients
  float T = T0, sd = 0.f, sa = 0.f;
  float w[8], Ts[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    w[c] = al[c] * T;
    Ts[c] = T;
    sd = fmaf(w[c], zz[c], sd);
    sa += w[c];
    T *= (1.f - al[c] + 1e-10f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sd += __shfl_xor(sd, o);
    sa += __shfl_xor(sa, o);
  }
  const float gr = g_rgb[3 * ray], gg = g_rgb[3 * ray + 1], gb = g_rgb[3 * ray + 2];
  // acc = min(A, 1): torch.minimum routes the gradient to A when A < 1 (half of it on a tie)
  float gA = g_acc ? g_acc[ray] * (sa < 1.f ? 1.f : (sa == 1.f ? 0.5f : 0.f)) : 0.f;
  float gD = 0.f;
  if (g_disp) {
    const float q = sd / (sa + 1e-10f);
    if (q > 1e-10f && fabsf(sa) > 1e-8f) {
      const float gq = -g_disp[ray] / (q * q);
      gD = gq / (sa + 1e-10f);
      gA += -gq * sd / ((sa + 1e-10f) * (sa + 1e-10f));
    }
  }
  // g_w and the suffix sums R_k = sum_{i>k} g_w_i w_i
  float gw[8], loc = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int s = lane * C + c;
    gw[c] = 0.f;
    if (c < C && s < S) {
      const float cr = sr[c] * 1.002f - 0.001f, cg = sg[c] * 1.002f - 0.001f, cb = sb[c] * 1.002f - 0.001f;
      gw[c] = gr * cr + gg * cg + gb * cb + gD * zz[c] + gA + (g_weights ? g_weights[base + s] : 0.f);
      loc += gw[c] * w[c];
    }
  }
  float suf = loc;   // inclusive suffix sum over lanes
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float dn_ = __shfl_down(suf, o);
    if (lane + o < 64) suf += dn_;
  }
  float R = suf - loc;   // contributions of lanes > this one
  // walk this lane's samples from last to first
#pragma unroll
  for (int c = 7; c >= 0; --c) {
    const int s = lane * C + c;
    if (c < C && s < S) {
      const float u = 1.f - al[c] + 1e-10f;
      float ga = gw[c] * Ts[c] - R / u;
      if (g_alpha) ga += g_alpha[base + s];
      const float dsig = ga * dl[c] * (1.f - al[c]) * dact[c] * inv_B;
      f32x4 o;
      o.x = gr * w[c] * 1.002f * sr[c] * (1.f - sr[c]);
      o.y = gg * w[c] * 1.002f * sg[c] * (1.f - sg[c]);
      o.z = gb * w[c] * 1.002f * sb[c] * (1.f - sb[c]);
      o.w = dsig;
      *reinterpret_cast<f32x4*>(draw + (base + s) * 4) = o;
      R += gw[c] * w[c];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// A11.  One wavefront per ray, 4 rays per block.  LDS per wave: cdf[S-1], bins[S-1], cat[S+Ni].
__global__ __launch_bounds__(256) void k_importance(const float* __restrict__ z, const float* __restrict__ w, int n,
                                                    int S, int Ni, const float* __restrict__ u, int single_net,
                                                    float* __restrict__ z_samples, float* __restrict__ z_merged,
                                                    long long* __restrict__ sorted_idx) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ray = blockIdx.x * 4 + wv;
  const int T = S + Ni, nb = S - 1;       // nb bins / cdf entries
  float* cdf = lds + wv * (2 * nb + T);
  float* bins = cdf + nb;
  float* cat = bins + nb;
  if (ray >= n) return;
  const float* zr = z + (long long)ray * S;
  const float* wr = w + (long long)ray * S;
  // pdf weights p[i], i in [0, S-2): from weights[1:-1] (+ smoothing for single_net), + 1e-5
  float part = 0.f;
  for (int i = lane; i < S - 2; i += 64) {
    float pw;
    if (single_net) pw = 0.5f * (fmaxf(wr[i], wr[i + 1]) + fmaxf(wr[i + 1], wr[i + 2])) + 0.01f;
    else pw = wr[i + 1];
    pw += 1e-5f;
    cdf[i + 1] = pw;      // stash, normalised below
    part += pw;
  }
  for (int i = lane; i < nb; i += 64) bins[i] = 0.5f * (zr[i] + zr[i + 1]);
  for (int i = lane; i < S; i += 64) cat[i] = zr[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  __builtin_amdgcn_wave_barrier();
  // sequential cumulative sum of pdf (torch.cumsum order) by lane 0; S <= 512 so this is a few hundred adds
  if (lane == 0) {
    float run = 0.f;
    cdf[0] = 0.f;
    for (int i = 1; i < nb; ++i) {
      run += cdf[i] / part;
      cdf[i] = run;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int k = lane; k < Ni; k += 64) {
    float uu;
    if (u) uu = u[(long long)ray * Ni + k];
    else uu = Ni > 1 ? (float)k / (float)(Ni - 1) : 0.f;       // torch.linspace(0,1,Ni)
    // searchsorted(cdf, u, right=True): first index with cdf[idx] > u
    int lo = 0, hi = nb;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cdf[mid] <= uu) lo = mid + 1; else hi = mid;
    }
    const int below = lo - 1 > 0 ? lo - 1 : 0;
    const int above = lo < nb - 1 ? lo : nb - 1;
    const float c0 = cdf[below], c1 = cdf[above];
    float den = c1 - c0;
    if (den < 1e-5f) den = 1.f;
    const float t = (uu - c0) / den;
    const float zs = bins[below] + t * (bins[above] - bins[below]);
    z_samples[(long long)ray * Ni + k] = zs;
    cat[S + k] = zs;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // stable rank sort of cat[0..T): rank = #{j : cat[j] < cat[i]  or  (cat[j] == cat[i] and j < i)}
  for (int i = lane; i < T; i += 64) {
    const float ci = cat[i];
    int rank = 0;
    for (int j = 0; j < T; ++j) {
      const float cj = cat[j];
      rank += (cj < ci) || (cj == ci && j < i);
    }
    z_merged[(long long)ray * T + rank] = ci;
    if (sorted_idx) sorted_idx[(long long)ray * T + rank] = i;
  }
}

// ------------------------------------------------------------------------------------------------
// Caller-side frame helpers: per-pixel rays of a bbox (get_rays, core/utils/ray_utils.py:6-28, on the
// pixels kp_to_valid_rays selects, :83-136) written straight into the [N,11] ray batch render() assembles
// (core/trainer.py:116-135), and the background composite + scatter of render_path (run_nerf.py:118-131).
__global__ void k_gen_rays(int W, int x0, int y0, int bw, int bh, float fx, float fy, float cx, float cy,
                           const float* __restrict__ c2w /*[3,4] row-major*/, float near, float far,
                           float* __restrict__ ray_batch, long long* __restrict__ valid_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= bw * bh) return;
  const int py = y0 + i / bw, px = x0 + i % bw;
  const float dx = ((float)px - cx) / fx, dy = -((float)py - cy) / fy, dz = -1.f;
  const float d0 = dx * c2w[0] + dy * c2w[1] + dz * c2w[2];
  const float d1 = dx * c2w[4] + dy * c2w[5] + dz * c2w[6];
  const float d2 = dx * c2w[8] + dy * c2w[9] + dz * c2w[10];
  const float inv = 1.f / sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
  float* r = ray_batch + (long long)i * 11;
  r[0] = c2w[3]; r[1] = c2w[7]; r[2] = c2w[11];
  r[3] = d0; r[4] = d1; r[5] = d2;
  r[6] = near; r[7] = far;
  r[8] = d0 * inv; r[9] = d1 * inv; r[10] = d2 * inv;
  valid_idx[i] = (long long)py * W + px;
}

// rgb_img [H*W,3] must be pre-filled with the background; pixel valid_idx[i] <- rgb + (1-acc) * bg
__global__ void k_assemble(const float* __restrict__ rgb, const float* __restrict__ acc, const float* __restrict__ disp,
                           const long long* __restrict__ valid_idx, int n, float* __restrict__ rgb_img,
                           float* __restrict__ disp_img, float* __restrict__ acc_img) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long px = valid_idx[i];
  const float a = acc[i], t = 1.f - a;
#pragma unroll
  for (int c = 0; c < 3; ++c) rgb_img[px * 3 + c] = rgb[3 * i + c] + t * rgb_img[px * 3 + c];
  if (disp_img) disp_img[px] = disp[i];
  if (acc_img) acc_img[px] = a;
}

// ------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------
int launch_pack(const AnerfNetParams* P, const int32_t* table, long long n, float* out, hipStream_t st) {
  const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_pack, dim3(blocks), dim3(256), 0, st, *P, table, n, out);
  return check_launch("k_pack");
}

int launch_ray_bounds(const float* rays, int ray_stride, const float* cyls, int cyl_stride, int n, float* near_far, float* stats,
                      hipStream_t st) {
  if (hipMemsetAsync(stats, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return set_error(ANERF_E_LAUNCH, "memset stats");
  hipLaunchKernelGGL(k_ray_bounds, dim3((n + 255) / 256), dim3(256), 0, st, rays, ray_stride, cyls, cyl_stride, n, near_far,
                     reinterpret_cast<unsigned long long*>(stats));
  return check_launch("k_ray_bounds");
}

int bounds_z_max_rays() { return BOUNDS_Z_MAX_RAYS; }
int launch_bounds_z(const float* rays, int ray_stride, const float* cyls, int cyl_stride, int n, int S, const float* t_rand, int lindisp,
                    float* z, hipStream_t st) {
  const long long tot = (long long)n * S;
  hipLaunchKernelGGL(k_bounds_z, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, rays, ray_stride, cyls, cyl_stride, n, S, t_rand,
                     lindisp, z);
  return check_launch("k_bounds_z");
}

int launch_coarse_z(const float* near_far, const float* stats, const float* rays, int ray_stride, int n, int S,
                    const float* t_rand, int lindisp, float* z, float* nf_fixed, hipStream_t st) {
  const long long tot = (long long)n * S;
  hipLaunchKernelGGL(k_coarse_z, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, near_far,
                     reinterpret_cast<const unsigned long long*>(stats), rays, ray_stride, n, S, t_rand, lindisp, z, nf_fixed);
  return check_launch("k_coarse_z");
}

int launch_composite(const AnerfConfig* cfg, const float* raw, const float* z, const float* rays, int ray_stride,
                     const float* noise, int n, int S, float* rgb, float* disp, float* acc, float* weights,
                     float* alpha, float* depth, hipStream_t st) {
  hipLaunchKernelGGL(k_composite, dim3((n + 3) / 4), dim3(256), 0, st, raw, z, rays, ray_stride, noise, n, S,
                     cfg->density_act, 1.0f / cfg->density_scale, cfg->softplus_shift, rgb, disp, acc, weights, alpha,
                     depth);
  return check_launch("k_composite");
}

int launch_composite_bwd(const AnerfConfig* cfg, const float* raw, const float* z, const float* rays, int ray_stride,
                         const float* noise, int n, int S, const float* g_rgb, const float* g_acc, const float* g_disp,
                         const float* g_alpha, const float* g_weights, float* draw, hipStream_t st) {
  hipLaunchKernelGGL(k_composite_bwd, dim3((n + 3) / 4), dim3(256), 0, st, raw, z, rays, ray_stride, noise, n, S,
                     cfg->density_act, 1.0f / cfg->density_scale, cfg->softplus_shift, g_rgb, g_acc, g_disp, g_alpha,
                     g_weights, draw);
  return check_launch("k_composite_bwd");
}

int launch_gen_rays(int W, int x0, int y0, int bw, int bh, float fx, float fy, float cx, float cy, const float* c2w,
                    float near, float far, float* ray_batch, long long* valid_idx, hipStream_t st) {
  const int n = bw * bh;
  hipLaunchKernelGGL(k_gen_rays, dim3((n + 255) / 256), dim3(256), 0, st, W, x0, y0, bw, bh, fx, fy, cx, cy, c2w, near, far,
                     ray_batch, valid_idx);
  return check_launch("k_gen_rays");
}

int launch_assemble(const float* rgb, const float* acc, const float* disp, const long long* valid_idx, int n, float* rgb_img,
                    float* disp_img, float* acc_img, hipStream_t st) {
  hipLaunchKernelGGL(k_assemble, dim3((n + 255) / 256), dim3(256), 0, st, rgb, acc, disp, valid_idx, n, rgb_img, disp_img,
                     acc_img);
  return check_launch("k_assemble");
}

int launch_importance(const float* z, const float* w, int n, int S, int Ni, const float* u, int single_net,
                      float* zs, float* zm, long long* sidx, hipStream_t st) {
  const size_t lds = 4 * (size_t)(2 * (S - 1) + S + Ni) * sizeof(float);
  hipLaunchKernelGGL(k_importance, dim3((n + 3) / 4), dim3(256), lds, st, z, w, n, S, Ni, u, single_net, zs, zm, sidx);
  return check_launch("k_importance");
}

// single_net merge (raycasters.py:447-456, merge_samples :796-812): raw_f[n][k] = cat(raw_c, raw_is)[n][sorted_idx[n][k]]
__global__ void k_gather_raw(const float* __restrict__ raw_c, const float* __restrict__ raw_is, const long long* __restrict__ idx,
                             long long total, int S, int Ni, float* __restrict__ out) {
  const long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (g >= total) return;
  const long long n = g / (S + Ni);
  const long long j = idx[g];
  const f32x4 v = j < S ? *reinterpret_cast<const f32x4*>(raw_c + (n * S + j) * 4)
                        : *reinterpret_cast<const f32x4*>(raw_is + (n * Ni + (j - S)) * 4);
  *reinterpret_cast<f32x4*>(out + g * 4) = v;
}

int launch_gather_raw(const float* raw_c, const float* raw_is, const long long* idx, int n, int S, int Ni, float* out, hipStream_t st) {
  const long long total = (long long)n * (S + Ni);
  if (total == 0) return ANERF_OK;
  hipLaunchKernelGGL(k_gather_raw, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, raw_c, raw_is, idx, total, S, Ni, out);
  return check_launch("k_gather_raw");
}

// rows of 3 floats of the merged samples: out[n][k] = cat(a[n] (S rows), b[n] (Ni rows))[idx[n][k]]   (the sample offsets of
// ray_noise_std > 0 follow their samples through the sort of the importance resampling, raycasters.py:665-709)
__global__ void k_gather_rows3(const float* __restrict__ a, const float* __restrict__ b, const long long* __restrict__ idx, long long total,
                               int S, int Ni, float* __restrict__ out) {
  const long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (g >= total) return;
  const long long ray = g / (S + Ni);
  const long long k = idx[g];
  const float* src = k < S ? a + (ray * S + k) * 3 : b + (ray * Ni + (k - S)) * 3;
  out[3 * g] = src[0];
  out[3 * g + 1] = src[1];
  out[3 * g + 2] = src[2];
}

int launch_gather_rows3(const float* a, const float* b, const long long* idx, int n, int S, int Ni, float* out, hipStream_t st) {
  const long long total = (long long)n * (S + Ni);
  if (total == 0) return ANERF_OK;
  hipLaunchKernelGGL(k_gather_rows3, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, b, idx, total, S, Ni, out);
  return check_launch("k_gather_rows3");
}

}  // namespace anerf
