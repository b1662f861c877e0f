// anerf_step.hip -- the per-iteration host glue around the caster call as single launches (ABI revision 3, gfx950):
//   k_pack_multi    every stale weight image of a step in one launch           (was 4-6 x k_pack / k_pack_b3)
//   k_rand_fill     t_rand / u / raw noise / point noise of a caster call      (torch.rand x2 + torch.randn x2 + mul x2;
//                                                                               ray_utils.py:171-180,240-246, nerf.py:176-182)
//   k_ray_batch     render()'s ray-batch assembly                               (core/trainer.py:116-135: ones_like x2, norm, div, cat)
//   k_cyl_bbox      projected cylinder boxes of all frames of a render_path     (skeleton_utils.py:607-690, ray_utils.py:83-136)
// HBM / latency-bound byte movers; at 384 rays per rank (the 8-GPU shard of the 3072-ray batch) the launches they replace
// were ~0.3 ms of a 2.6 ms step.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdint.h>
#include "anerf.h"
#include "anerf_dev.h"

namespace anerf {

// ------------------------------------------------------------------------------------------------
struct PackJobs {
  AnerfPackJob j[ANERF_MAX_PACK_JOBS];
};

__global__ void k_pack_multi(PackJobs J) {
  const AnerfPackJob& job = J.j[blockIdx.y];
  const float* tens[24];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    tens[i] = job.params.w[i];
    tens[12 + i] = job.params.b[i];
  }
  const long long n = job.n;
  const int32_t* __restrict__ table = job.table;
  // fp32 images, four elements per thread and iteration (round 6: 16-byte table loads and image stores, four gathers in flight --
  // the kernel is a latency-bound byte mover, 15-18 us per step at one element per iteration); the same value per element
  long long done = 0;
  if (job.kind == 0 && (((uintptr_t)table | (uintptr_t)job.out) & 15) == 0) {
    const long long n4 = n / 4;
    for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
      const int4 e4 = reinterpret_cast<const int4*>(table)[q];
      const int32_t ee[4] = {e4.x, e4.y, e4.z, e4.w};
      float xx[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int32_t e = ee[u];
        float x = 0.f;
        if (e >= 0) {
          const int id = (e >> 24) & 31, off = e & 0xFFFFFF;
          const float* src = tens[0];
#pragma unroll
          for (int k = 0; k < 24; ++k)
            if (id == k) src = tens[k];
          x = src[off] * sched_scale(job.params, id, off);
        }
        xx[u] = x;
      }
      reinterpret_cast<f32x4*>(job.out)[q] = f32x4{xx[0], xx[1], xx[2], xx[3]};
    }
    done = 4 * n4;
  }
  for (long long i = done + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int32_t e = table[i];
    float x = 0.f;
    int part = 0;
    if (e >= 0) {
      const int id = (e >> 24) & 31, off = e & 0xFFFFFF;
      part = (e >> 29) & 1;
      const float* src = tens[0];
#pragma unroll
      for (int k = 0; k < 24; ++k)
        if (id == k) src = tens[k];
      x = src[off] * sched_scale(job.params, id, off);
      asm volatile("" : "+v"(x));   // the ROUNDED fp32 product is what gets split: no contraction of the multiply into `x - hi` below
    }
    if (job.kind == 0) {
      static_cast<float*>(job.out)[i] = x;
    } else {   // hi / lo split of the bf16x3 images (k_pack_b3)
      const __bf16 hi = (__bf16)x;
      const __bf16 r = part ? (__bf16)(x - (float)hi) : hi;
      static_cast<unsigned short*>(job.out)[i] = e >= 0 ? __builtin_bit_cast(unsigned short, r) : (unsigned short)0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter (c0..c3), key (k0, k1)
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

struct RandJobs {
  AnerfRandJob j[ANERF_MAX_RAND_JOBS];
};

// thread t of job blockIdx.y produces elements 4t .. 4t+3: counter = (t lo, t hi, job, offset lo), key = seed ^ (offset hi)
__global__ void k_rand_fill(RandJobs J, uint64_t seed, uint64_t offset, const AnerfStepBlock* __restrict__ blk) {
  if (blk) {   // ABI revision 6: (seed, offset) of a captured training step live in the device-resident step block
    seed = blk->rng_seed;
    offset += blk->rng_offset;   // `offset` carries the index of this fill inside the iteration (0, 1, ...: one per caster call)
  }
  const AnerfRandJob& job = J.j[blockIdx.y];
  const long long quads = (job.n + 3) / 4;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < quads; t += (long long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)t, (uint32_t)((unsigned long long)t >> 32), (uint32_t)blockIdx.y, (uint32_t)offset};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32));
    float v[4];
    if (job.kind == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (float)(c[k] >> 8) * 5.9604644775390625e-8f;          // 24 bits -> [0, 1)
    } else {
      // Box-Muller on (0, 1] x [0, 1): two pairs -> four normals
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float u1 = ((float)(c[2 * k] >> 8) + 1.0f) * 5.9604644775390625e-8f;              // (0, 1]
        const float u2 = (float)(c[2 * k + 1] >> 8) * 5.9604644775390625e-8f;
        const float r = sqrtf(-2.0f * logf(u1)) * job.scale;
        float s, co;
        sincospif(2.0f * u2, &s, &co);
        v[2 * k] = r * co;
        v[2 * k + 1] = r * s;
      }
    }
    const long long base = 4 * t;
    if (base + 3 < job.n && (((uintptr_t)job.out & 15) == 0)) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(job.out + base) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (base + k < job.n) job.out[base + k] = v[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------
__global__ void k_ray_batch(const float* __restrict__ ro, const float* __restrict__ rd, int n, float near, float far, int stride,
                            float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ox = ro[3 * i], oy = ro[3 * i + 1], oz = ro[3 * i + 2];
  const float dx = rd[3 * i], dy = rd[3 * i + 1], dz = rd[3 * i + 2];
  float* r = out + (long long)i * stride;
  r[0] = ox; r[1] = oy; r[2] = oz;
  r[3] = dx; r[4] = dy; r[5] = dz;
  r[6] = near; r[7] = far;
  if (stride >= 11) {
    const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);    // torch.norm(rays_d, dim=-1): sqrt of the fp32 sum of squares
    r[8] = dx / nrm; r[9] = dy / nrm; r[10] = dz / nrm;
  }
}

// ------------------------------------------------------------------------------------------------
// One block of 128 threads per frame, thread p < 100 projects one cap point (50 per cap), all in double like the reference's
// numpy code; the inverse of the (sign-flipped) camera matrix is the cofactor inverse of its affine 3 x 4 part.
__global__ __launch_bounds__(128) void k_cyl_bbox(const double* __restrict__ cyls, const double* __restrict__ c2ws,
                                                  const double* __restrict__ hwf, const int32_t* __restrict__ off,
                                                  const double* __restrict__ circle, int32_t* __restrict__ bbox) {
  __shared__ double sx[128], sy[128], bx[128], by[128];
  const int f = blockIdx.x, p = threadIdx.x;
  const double* cyl = cyls + 5 * f;
  const double* M = c2ws + 12 * f;
  // sw = [c0, -c1, -c2, t]; w2c = inv(sw): rotation part A^-1, translation -A^-1 t
  const double a00 = M[0], a01 = -M[1], a02 = -M[2], a10 = M[4], a11 = -M[5], a12 = -M[6], a20 = M[8], a21 = -M[9], a22 = -M[10];
  const double tx = M[3], ty = M[7], tz = M[11];
  const double c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const double c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
  const double c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
  const double det = a00 * c00 + a01 * c10 + a02 * c20;
  double px = 1e300, py = 1e300, qx = -1e300, qy = -1e300;
  if (p < 100) {
    const int k = p % 50;
    const double x = cyl[0] + circle[2 * k] * cyl[2], z = cyl[1] + circle[2 * k + 1] * cyl[2], y = p < 50 ? cyl[3] : cyl[4];
    const double dx = x - tx, dy = y - ty, dz = z - tz;
    const double cx = (c00 * dx + c01 * dy + c02 * dz) / det;
    const double cy = (c10 * dx + c11 * dy + c12 * dz) / det;
    const double cz = (c20 * dx + c21 * dy + c22 * dz) / det;
    px = qx = hwf[4 * f + 2] * cx / cz;
    py = qy = hwf[4 * f + 3] * cy / cz;
  }
  sx[p] = px; sy[p] = py; bx[p] = qx; by[p] = qy;
  __syncthreads();
  for (int o = 64; o > 0; o >>= 1) {
    if (p < o) {
      sx[p] = fmin(sx[p], sx[p + o]); sy[p] = fmin(sy[p], sy[p + o]);
      bx[p] = fmax(bx[p], bx[p + o]); by[p] = fmax(by[p], by[p + o]);
    }
    __syncthreads();
  }
  if (p == 0) {
    const int H = (int)hwf[4 * f], W = (int)hwf[4 * f + 1];
    auto clip = [](long long v, int lo, int hi) { return (int)(v < lo ? lo : (v > hi ? hi : v)); };
    bbox[4 * f + 0] = clip((long long)floor(sx[0]) + off[2 * f], 0, W - 1);
    bbox[4 * f + 1] = clip((long long)floor(sy[0]) + off[2 * f + 1], 0, H - 1);
    bbox[4 * f + 2] = clip((long long)ceil(bx[0]) + off[2 * f], 0, W - 1);
    bbox[4 * f + 3] = clip((long long)ceil(by[0]) + off[2 * f + 1], 0, H - 1);
  }
}

// ------------------------------------------------------------------------------------------------
// ABI revision 6: the step block is written by ONE thread from a by-value copy of its new contents (kernel arguments: no
// staging buffer the host could overwrite while an earlier write is still queued).  mask bit g = Adam group g is (re)written.
__global__ void k_step_block_write(AnerfStepBlock* __restrict__ dst, AnerfStepBlock v, unsigned mask) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  dst->rng_seed = v.rng_seed;
  dst->rng_offset = v.rng_offset;
  dst->tau_v = v.tau_v;
  dst->tau_d = v.tau_d;
  for (int g = 0; g < ANERF_MAX_ADAM_GROUPS; ++g)
    if (mask & (1u << g)) {
      dst->adam_step_size[g] = v.adam_step_size[g];
      dst->adam_sqrt_bc2[g] = v.adam_sqrt_bc2[g];
      dst->adam_grad_scale[g] = v.adam_grad_scale[g];
    }
}

}  // namespace anerf

using namespace anerf;

extern "C" {

int anerf_pack_params_multi(const AnerfPackJob* jobs, int32_t n_jobs, void* stream) {
  if (n_jobs == 0) return ANERF_OK;
  if (!jobs) return set_error(ANERF_E_NULL, "pack_multi: jobs is NULL");
  if (n_jobs < 0 || n_jobs > ANERF_MAX_PACK_JOBS) return set_error(ANERF_E_SHAPE, "pack_multi: 0 <= n_jobs <= 8");
  PackJobs J;
  long long nmax = 0;
  for (int j = 0; j < ANERF_MAX_PACK_JOBS; ++j) {
    J.j[j] = jobs[j < n_jobs ? j : 0];
    if (j >= n_jobs) continue;
    const AnerfPackJob& q = jobs[j];
    if (!q.table || !q.out || q.n < 0) return set_error(ANERF_E_NULL, "pack_multi: NULL table / out");
    if (q.kind != 0 && q.kind != 1) return set_error(ANERF_E_CONFIG, "pack_multi: kind must be 0 (float) or 1 (bf16 hi/lo)");
    for (int i = 0; i < 12; ++i)
      if (!q.params.w[i] || !q.params.b[i]) return set_error(ANERF_E_NULL, "pack_multi: NULL tensor");
    if (!sched_ok(q.params)) return set_error(ANERF_E_SHAPE, "pack_multi: sched_dim_x must be 432 and sched_dim_u 72 / 648 / 664 (the encoded widths of the supported layouts; trunk width 256)");
    if (q.n > nmax) nmax = q.n;
  }
  if (nmax == 0) return ANERF_OK;
  const int blocks = (int)((nmax + 255) / 256 < 1024 ? (nmax + 255) / 256 : 1024);
  hipLaunchKernelGGL(k_pack_multi, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, J);
  return check_launch("k_pack_multi");
}

static int rand_fill_impl(const AnerfRandJob* jobs, int32_t n_jobs, uint64_t seed, uint64_t offset, const AnerfStepBlock* blk,
                          void* stream) {
  if (n_jobs == 0) return ANERF_OK;
  if (!jobs) return set_error(ANERF_E_NULL, "rand_fill: jobs is NULL");
  if (n_jobs < 0 || n_jobs > ANERF_MAX_RAND_JOBS) return set_error(ANERF_E_SHAPE, "rand_fill: 0 <= n_jobs <= 6");
  RandJobs J;
  long long nmax = 0;
  for (int j = 0; j < ANERF_MAX_RAND_JOBS; ++j) {
    J.j[j] = jobs[j < n_jobs ? j : 0];
    if (j >= n_jobs) continue;
    if (jobs[j].n < 0 || (jobs[j].n > 0 && !jobs[j].out)) return set_error(ANERF_E_NULL, "rand_fill: NULL out");
    if (jobs[j].kind != 0 && jobs[j].kind != 1) return set_error(ANERF_E_CONFIG, "rand_fill: kind must be 0 (uniform) or 1 (normal)");
    if (jobs[j].n > nmax) nmax = jobs[j].n;
  }
  if (nmax == 0) return ANERF_OK;
  const long long quads = (nmax + 3) / 4;
  const int blocks = (int)((quads + 255) / 256 < 2048 ? (quads + 255) / 256 : 2048);
  hipLaunchKernelGGL(k_rand_fill, dim3(blocks, n_jobs), dim3(256), 0, (hipStream_t)stream, J, seed, offset, blk);
  return check_launch("k_rand_fill");
}

int anerf_rand_fill(const AnerfRandJob* jobs, int32_t n_jobs, uint64_t seed, uint64_t offset, void* stream) {
  return rand_fill_impl(jobs, n_jobs, seed, offset, nullptr, stream);
}

int anerf_rand_fill_dev(const AnerfRandJob* jobs, int32_t n_jobs, const AnerfStepBlock* block, int32_t call_index, void* stream) {
  if (!block) return set_error(ANERF_E_NULL, "rand_fill_dev: block is NULL");
  if (call_index < 0) return set_error(ANERF_E_SHAPE, "rand_fill_dev: call_index >= 0");
  return rand_fill_impl(jobs, n_jobs, 0, (uint64_t)call_index, block, stream);
}

int anerf_step_block_write(AnerfStepBlock* block, const AnerfStepValues* v, void* stream) {
  if (!block || !v) return set_error(ANERF_E_NULL, "step_block_write: NULL pointer");
  if (((uintptr_t)block & 15) != 0) return set_error(ANERF_E_SHAPE, "step_block_write: the block must be 16-byte aligned");
  if (v->n_groups < 0 || v->n_groups > ANERF_MAX_ADAM_GROUPS) return set_error(ANERF_E_SHAPE, "step_block_write: 0 <= n_groups <= 4");
  AnerfStepBlock b;
  memset(&b, 0, sizeof(b));
  b.rng_seed = v->rng_seed;
  b.rng_offset = v->rng_offset;
  b.tau_v = v->tau_v;
  b.tau_d = v->tau_d;
  unsigned mask = 0;
  for (int g = 0; g < v->n_groups; ++g) {
    if (v->adam_step[g] <= 0) continue;            // this group does not step this iteration
    // bias corrections in double, exactly as anerf_adam_step computes them from (lr, step)
    const double bc1 = 1.0 - pow((double)v->beta1[g], (double)v->adam_step[g]);
    const double bc2 = 1.0 - pow((double)v->beta2[g], (double)v->adam_step[g]);
    b.adam_step_size[g] = (float)((double)v->lr[g] / bc1);
    b.adam_sqrt_bc2[g] = (float)sqrt(bc2);
    b.adam_grad_scale[g] = v->grad_scale[g];
    mask |= 1u << g;
  }
  hipLaunchKernelGGL(k_step_block_write, dim3(1), dim3(64), 0, (hipStream_t)stream, block, b, mask);
  return check_launch("k_step_block_write");
}

int anerf_make_ray_batch(const float* rays_o, const float* rays_d, int32_t n_rays, float near, float far, int32_t out_stride,
                         float* ray_batch, void* stream) {
  if (n_rays == 0) return ANERF_OK;
  if (!rays_o || !rays_d || !ray_batch) return set_error(ANERF_E_NULL, "make_ray_batch: NULL pointer");
  if (n_rays < 0 || (out_stride != 8 && out_stride != 11)) return set_error(ANERF_E_SHAPE, "make_ray_batch: out_stride must be 8 or 11");
  hipLaunchKernelGGL(k_ray_batch, dim3((n_rays + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, n_rays, near, far,
                     out_stride, ray_batch);
  return check_launch("k_ray_batch");
}

int anerf_cyl_bbox(const double* cyls, const double* c2ws, const double* hwf, const int32_t* off, const double* circle,
                   int32_t n_frames, int32_t* bbox, void* stream) {
  if (n_frames == 0) return ANERF_OK;
  if (!cyls || !c2ws || !hwf || !off || !circle || !bbox) return set_error(ANERF_E_NULL, "cyl_bbox: NULL pointer");
  if (n_frames < 0) return set_error(ANERF_E_SHAPE, "cyl_bbox: n_frames < 0");
  hipLaunchKernelGGL(k_cyl_bbox, dim3(n_frames), dim3(128), 0, (hipStream_t)stream, cyls, c2ws, hwf, off, circle, bbox);
  return check_launch("k_cyl_bbox");
}

}  // extern "C"
