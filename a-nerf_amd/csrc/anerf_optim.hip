// anerf_optim.hip -- the scalar tail of a training step as three launches instead of ~120 tiny ones:
//   k_loss (+ k_loss_final): background composite + MSE/L1/Huber against the target colours for the fine and coarse
//       heads, the PSNR numerator, AND the gradient w.r.t. the rendered maps in the same pass
//       (core/trainer.py:353-380 _compute_nerf_loss, :8-60 img2mse / img2l1 / img2huber / mse2psnr);
//   k_adam: torch.optim.Adam's update (no amsgrad / weight decay, as trainer.py:173-183 builds it) over ONE flat
//       fp32 parameter buffer, fused with zero_grad and with the sum of squared gradients that
//       get_gradnorm (trainer.py:192-203) otherwise collects with 48 .item() syncs;
//   k_sumsq_final: fixed-order reduction of the per-block partial sums (deterministic).
// All HBM-bound streaming kernels: 16-byte accesses, grid-stride, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf.h"
#include "anerf_dev.h"

namespace anerf {

constexpr int RB = 256;   // threads per block of the reductions

__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];   // fixed order
}

// three block sums at once: the same shuffle tree and the same sh[0] + sh[1] + sh[2] + sh[3] order per value as three block_sum
// calls (bit-identical results), one pair of barriers instead of three
__device__ __forceinline__ void block_sum3(float& a, float& b, float& c, float (*sh3)[4]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
    c += __shfl_xor(c, o);
  }
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { sh3[0][w] = a; sh3[1][w] = b; sh3[2][w] = c; }
  __syncthreads();
  a = sh3[0][0] + sh3[0][1] + sh3[0][2] + sh3[0][3];
  b = sh3[1][0] + sh3[1][1] + sh3[1][2] + sh3[1][3];
  c = sh3[2][0] + sh3[2][1] + sh3[2][2] + sh3[2][3];
}

// One thread per ray.  partial[b][0..3] = sum of per-channel terms of (fine loss, coarse loss, fine squared error, 0)
__global__ __launch_bounds__(RB) void k_loss(const float* __restrict__ rgb, const float* __restrict__ acc,
                                             const float* __restrict__ rgb0, const float* __restrict__ acc0,
                                             const float* __restrict__ target, const float* __restrict__ bgs, int bg_stride,
                                             int n, int kind, float beta, float coarse_w, float* __restrict__ g_rgb,
                                             float* __restrict__ g_acc, float* __restrict__ g_rgb0,
                                             float* __restrict__ g_acc0, float* __restrict__ partial) {
  __shared__ float sh[4];
  const float inv = 1.0f / (3.0f * (float)n);
  // kind 0: d^2   1: |d|   2: smooth-L1 / Huber (F.smooth_l1_loss, trainer.py:57): |d| < beta ? d^2 / (2 beta) : |d| - beta/2
  auto term = [&](float d) {
    const float a = fabsf(d);
    return kind == 0 ? d * d : (kind == 1 || a >= beta) ? a - (kind == 2 ? 0.5f * beta : 0.f) : 0.5f * d * d / beta;
  };
  auto dterm = [&](float d) {
    const float a = fabsf(d);
    return kind == 0 ? 2.0f * d * inv
                     : (kind == 1 || a >= beta) ? (d > 0.f ? inv : (d < 0.f ? -inv : 0.f)) : d / beta * inv;
  };
  float lf = 0.f, lc = 0.f, se = 0.f;
  for (int r = blockIdx.x * RB + threadIdx.x; r < n; r += gridDim.x * RB) {
    float bg[3] = {0.f, 0.f, 0.f};
    if (bgs) {
      bg[0] = bgs[(long long)r * bg_stride + 0];
      bg[1] = bgs[(long long)r * bg_stride + 1];
      bg[2] = bgs[(long long)r * bg_stride + 2];
    }
    const float t0 = target[3 * r], t1 = target[3 * r + 1], t2 = target[3 * r + 2];
    const float tt[3] = {t0, t1, t2};
    {
      const float om = bgs ? 1.0f - acc[r] : 0.f;
      float ga = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = rgb[3 * r + c] + om * bg[c] - tt[c];
        se += d * d;
        lf += term(d);
        const float g = dterm(d);
        if (g_rgb) g_rgb[3 * r + c] = g;
        ga -= g * bg[c];
      }
      if (g_acc) g_acc[r] = ga;
    }
    if (rgb0) {
      const float om = bgs ? 1.0f - acc0[r] : 0.f;
      float ga = 0.f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float d = rgb0[3 * r + c] + om * bg[c] - tt[c];
        lc += term(d);
        const float g = coarse_w * dterm(d);
        if (g_rgb0) g_rgb0[3 * r + c] = g;
        ga -= g * bg[c];
      }
      if (g_acc0) g_acc0[r] = ga;
    }
  }
  lf = block_sum(lf, sh);
  lc = block_sum(lc, sh);
  se = block_sum(se, sh);
  if (threadIdx.x == 0) {
    partial[4 * blockIdx.x + 0] = lf;
    partial[4 * blockIdx.x + 1] = lc;
    partial[4 * blockIdx.x + 2] = se;
    partial[4 * blockIdx.x + 3] = 0.f;
  }
}

// Small batches (n <= LOSS_ONE_MAX rays, i.e. the 8-GPU shard of a 3072-ray step; round 6): k_loss + k_loss_final as ONE block that
// walks the same "virtual blocks" in turn -- thread t of virtual block b takes ray 256 b + t, the block sums use the same tree, the
// partials are then summed as k_loss_final sums them -- so the four outputs carry the SAME bits as the two-launch route.
constexpr int LOSS_ONE_MAX = 1024;
__global__ __launch_bounds__(RB) void k_loss_one(const float* __restrict__ rgb, const float* __restrict__ acc,
                                                 const float* __restrict__ rgb0, const float* __restrict__ acc0,
                                                 const float* __restrict__ target, const float* __restrict__ bgs, int bg_stride,
                                                 int n, int nblk, int kind, float beta, float coarse_w, float* __restrict__ g_rgb,
                                                 float* __restrict__ g_acc, float* __restrict__ g_rgb0,
                                                 float* __restrict__ g_acc0, float* __restrict__ out) {
  __shared__ float sh3[3][4];
  __shared__ float part[LOSS_ONE_MAX / RB][3];
  constexpr int NV = LOSS_ONE_MAX / RB;
  const float inv = 1.0f / (3.0f * (float)n);
  auto term = [&](float d) {
    const float a = fabsf(d);
    return kind == 0 ? d * d : (kind == 1 || a >= beta) ? a - (kind == 2 ? 0.5f * beta : 0.f) : 0.5f * d * d / beta;
  };
  auto dterm = [&](float d) {
    const float a = fabsf(d);
    return kind == 0 ? 2.0f * d * inv
                     : (kind == 1 || a >= beta) ? (d > 0.f ? inv : (d < 0.f ? -inv : 0.f)) : d / beta * inv;
  };
  // every virtual block's per-thread terms first (independent rays: their loads overlap), the block sums afterwards, in block order
  float lf[NV], lc[NV], se[NV];
#pragma unroll
  for (int vb = 0; vb < NV; ++vb) {
    lf[vb] = lc[vb] = se[vb] = 0.f;
    const int r = vb * RB + threadIdx.x;
    if (vb < nblk && r < n) {
      float bg[3] = {0.f, 0.f, 0.f};
      if (bgs) {
        bg[0] = bgs[(long long)r * bg_stride + 0];
        bg[1] = bgs[(long long)r * bg_stride + 1];
        bg[2] = bgs[(long long)r * bg_stride + 2];
      }
      const float tt[3] = {target[3 * r], target[3 * r + 1], target[3 * r + 2]};
      {
        const float om = bgs ? 1.0f - acc[r] : 0.f;
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = rgb[3 * r + c] + om * bg[c] - tt[c];
          se[vb] += d * d;
          lf[vb] += term(d);
          const float g = dterm(d);
          if (g_rgb) g_rgb[3 * r + c] = g;
          ga -= g * bg[c];
        }
        if (g_acc) g_acc[r] = ga;
      }
      if (rgb0) {
        const float om = bgs ? 1.0f - acc0[r] : 0.f;
        float ga = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float d = rgb0[3 * r + c] + om * bg[c] - tt[c];
          lc[vb] += term(d);
          const float g = coarse_w * dterm(d);
          if (g_rgb0) g_rgb0[3 * r + c] = g;
          ga -= g * bg[c];
        }
        if (g_acc0) g_acc0[r] = ga;
      }
    }
  }
#pragma unroll
  for (int vb = 0; vb < NV; ++vb) {
    if (vb < nblk) {          // (nblk is uniform over the block: every thread takes the same barriers)
      block_sum3(lf[vb], lc[vb], se[vb], sh3);
      if (threadIdx.x == 0) {
        part[vb][0] = lf[vb]; part[vb][1] = lc[vb]; part[vb][2] = se[vb];
      }
    }
  }
  __syncthreads();
  // k_loss_final's sums: thread i takes partials i, i + 256, ... (here nblk <= 4 < 256: thread i < nblk holds exactly one)
  float a = 0.f, b = 0.f, c = 0.f;
  if ((int)threadIdx.x < nblk) {
    a += part[threadIdx.x][0]; b += part[threadIdx.x][1]; c += part[threadIdx.x][2];
  }
  block_sum3(a, b, c, sh3);
  if (threadIdx.x == 0) {
    out[1] = a * inv;
    out[2] = b * inv;
    out[0] = a * inv + coarse_w * (b * inv);
    out[3] = c * inv;
  }
}

// out[0] = total loss, out[1] = fine loss, out[2] = coarse loss (unweighted), out[3] = fine MSE (PSNR = -10 log10)
__global__ __launch_bounds__(RB) void k_loss_final(const float* __restrict__ partial, int nblk, int n, float coarse_w,
                                                   float* __restrict__ out) {
  __shared__ float sh[4];
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblk; i += RB) {
    a += partial[4 * i];
    b += partial[4 * i + 1];
    c += partial[4 * i + 2];
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  c = block_sum(c, sh);
  if (threadIdx.x == 0) {
    const float inv = 1.0f / (3.0f * (float)n);
    out[1] = a * inv;
    out[2] = b * inv;
    out[0] = a * inv + coarse_w * (b * inv);
    out[3] = c * inv;
  }
}

// torch.optim.Adam single-tensor update (torch/optim/adam.py, capturable = False, maximize = False):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(RB) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, long long n4, long long n, float step_size,
                                             float b1, float b2, float sqrt_bc2, float eps, float gscale, int zero,
                                             float* __restrict__ partial, const AnerfStepBlock* __restrict__ blk, int grp) {
  __shared__ float sh[4];
  if (blk) {   // ABI revision 6: this group's step size / bias correction / gradient scale from the device-resident step block
    step_size = blk->adam_step_size[grp];
    sqrt_bc2 = blk->adam_sqrt_bc2[grp];
    gscale = blk->adam_grad_scale[grp];
  }
  float ss = 0.f;
  for (long long i = blockIdx.x * (long long)RB + threadIdx.x; i < n4; i += (long long)gridDim.x * RB) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i], gg = reinterpret_cast<f32x4*>(g)[i];
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * gscale;
      ss += gk * gk;
      mm[k] = mm[k] + (gk - mm[k]) * (1.0f - b1);                 // lerp form, as torch's exp_avg.lerp_(grad, 1-b1)
      vv[k] = vv[k] * b2 + (1.0f - b2) * gk * gk;                 // mul_(b2).addcmul_(g, g, 1-b2)
      const float denom = sqrtf(vv[k]) / sqrt_bc2 + eps;
      pp[k] = pp[k] - step_size * (mm[k] / denom);                // addcdiv_(m, denom, -step_size)
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (zero) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4*>(g)[i] = z;
    }
  }
  // tail (n not a multiple of 4): first block, first threads
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) {
    const long long i = 4 * n4 + threadIdx.x;
    const float gk = g[i] * gscale;
    ss += gk * gk;
    const float mk = m[i] + (gk - m[i]) * (1.0f - b1);
    const float vk = v[i] * b2 + (1.0f - b2) * gk * gk;
    m[i] = mk;
    v[i] = vk;
    p[i] = p[i] - step_size * (mk / (sqrtf(vk) / sqrt_bc2 + eps));
    if (zero) g[i] = 0.f;
  }
  if (partial) {
    ss = block_sum(ss, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = ss;
  }
}

// out[0] = total_norm = sqrt(sum g^2), out[1] = avg_norm = sqrt(sum g^2 / n_tensors)   (trainer.py:192-203)
__global__ __launch_bounds__(RB) void k_sumsq_final(const float* __restrict__ partial, int nblk, int n_tensors,
                                                    float* __restrict__ out) {
  __shared__ float sh[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < nblk; i += RB) a += partial[i];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) {
    out[0] = sqrtf(a);
    out[1] = sqrtf(a / (float)(n_tensors > 0 ? n_tensors : 1));
  }
}

}  // namespace anerf

using namespace anerf;

extern "C" {

int anerf_loss_blocks(int32_t n_rays) {
  int b = (n_rays + RB - 1) / RB;
  return b < 1 ? 1 : (b > 1024 ? 1024 : b);
}

int anerf_loss(const float* rgb, const float* acc, const float* rgb0, const float* acc0, const float* target,
               const float* bgs, int32_t bg_stride, int32_t n_rays, int32_t loss_type, float huber_beta,
               float coarse_weight, float* out4, float* g_rgb, float* g_acc, float* g_rgb0, float* g_acc0,
               float* partials, void* stream) {
  if (loss_type < 0 || loss_type > 2) return set_error(ANERF_E_CONFIG, "loss: loss_type must be 0 (MSE), 1 (L1) or 2 (Huber)");
  if (loss_type == 2 && !(huber_beta >= 0.f)) return set_error(ANERF_E_CONFIG, "loss: huber_beta must be >= 0");
  if (n_rays < 0 || (bgs && bg_stride != 0 && bg_stride < 3)) return set_error(ANERF_E_SHAPE, "loss: n_rays >= 0, bg_stride 0 or >= 3");
  if (n_rays == 0) return ANERF_OK;
  if (!rgb || !target || !out4 || !partials || (bgs && !acc) || (rgb0 && bgs && !acc0))
    return set_error(ANERF_E_NULL, "loss: NULL pointer");
  const int nblk = anerf_loss_blocks(n_rays);
  hipStream_t st = (hipStream_t)stream;
  if (n_rays <= LOSS_ONE_MAX) {   // one launch, the same bits (k_loss_one)
    hipLaunchKernelGGL(k_loss_one, dim3(1), dim3(RB), 0, st, rgb, acc, rgb0, acc0, target, bgs, (int)bg_stride, (int)n_rays, nblk,
                       (int)loss_type, huber_beta, rgb0 ? coarse_weight : 0.f, g_rgb, g_acc, g_rgb0, g_acc0, out4);
    return check_launch("k_loss_one");
  }
  hipLaunchKernelGGL(k_loss, dim3(nblk), dim3(RB), 0, st, rgb, acc, rgb0, acc0, target, bgs, (int)bg_stride, (int)n_rays,
                     (int)loss_type, huber_beta, coarse_weight, g_rgb, g_acc, g_rgb0, g_acc0, partials);
  int rc = check_launch("k_loss");
  if (rc) return rc;
  hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(RB), 0, st, (const float*)partials, nblk, (int)n_rays,
                     rgb0 ? coarse_weight : 0.f, out4);
  return check_launch("k_loss_final");
}

int anerf_adam_blocks(int64_t n) {
  int64_t b = (n / 4 + RB - 1) / RB;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int anerf_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, int32_t step, float grad_scale, int32_t zero_grads, int32_t n_tensors,
                    float* partials, float* norms2, void* stream) {
  if (n < 0 || step < 1) return set_error(ANERF_E_SHAPE, "adam: n >= 0 and step >= 1 (1-based)");
  if (n == 0) return ANERF_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || (norms2 && !partials)) return set_error(ANERF_E_NULL, "adam: NULL pointer");
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
    return set_error(ANERF_E_SHAPE, "adam: buffers must be 16-byte aligned");
  // bias corrections in double, as Python does for non-capturable Adam
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), sqrt_bc2 = (float)sqrt(bc2);
  const int nblk = anerf_adam_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam, dim3(nblk), dim3(RB), 0, st, params, grads, exp_avg, exp_avg_sq, (long long)(n / 4), (long long)n,
                     step_size, beta1, beta2, sqrt_bc2, eps, grad_scale, (int)zero_grads, norms2 ? partials : nullptr,
                     (const AnerfStepBlock*)nullptr, 0);
  int rc = check_launch("k_adam");
  if (rc || !norms2) return rc;
  hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(RB), 0, st, (const float*)partials, nblk, (int)n_tensors, norms2);
  return check_launch("k_sumsq_final");
}

int anerf_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                        const AnerfStepBlock* block, int32_t group, int32_t zero_grads, int32_t n_tensors, float* partials,
                        float* norms2, void* stream) {
  if (n < 0 || group < 0 || group >= ANERF_MAX_ADAM_GROUPS) return set_error(ANERF_E_SHAPE, "adam_dev: n >= 0, 0 <= group < 4");
  if (n == 0) return ANERF_OK;
  if (!block || !params || !grads || !exp_avg || !exp_avg_sq || (norms2 && !partials)) return set_error(ANERF_E_NULL, "adam_dev: NULL pointer");
  if ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) != 0)
    return set_error(ANERF_E_SHAPE, "adam_dev: buffers must be 16-byte aligned");
  const int nblk = anerf_adam_blocks(n);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam, dim3(nblk), dim3(RB), 0, st, params, grads, exp_avg, exp_avg_sq, (long long)(n / 4), (long long)n,
                     0.f, beta1, beta2, 1.f, eps, 1.f, (int)zero_grads, norms2 ? partials : nullptr, block, (int)group);
  int rc = check_launch("k_adam");
  if (rc || !norms2) return rc;
  hipLaunchKernelGGL(k_sumsq_final, dim3(1), dim3(RB), 0, st, (const float*)partials, nblk, (int)n_tensors, norms2);
  return check_launch("k_sumsq_final");
}

}  // extern "C"
