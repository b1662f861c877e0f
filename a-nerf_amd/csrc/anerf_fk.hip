// anerf_fk.hip -- forward kinematics of the SMPL skeleton and its backward:
//   PoseOptLayer.calculate_kinematic / get_kinematic_chain_T + unrolled_kinematic_chain (core/pose_opt.py:372-445,
//   482-566): per pose, axis-angle [24,3] or 6D-rotation [24,6] bones + pelvis [3] + rest pose [24,3]  ->  local rotations `rots`, joint-to-
//   world matrices `l2ws`, world-to-bone matrices `skts` (= inverse(l2ws), what the ray-march kernels consume) and
//   joint locations `kp`.  The reference runs ~40 small batched ops forward and ~100 backward per call; here it is one
//   launch each way.  The rotation map is pytorch3d's axis_angle_to_matrix (third-party, absent from this image:
//   pytorch3d/transforms/rotation_conversions.py, the version the reference's README installs -- py38_cu102_pyt190
//   wheels = v0.6.x), restated from its published source:
//     axis_angle_to_quaternion: th = |a|, q = (cos(th/2), a * k), k = sin(th/2)/th, or 0.5 - th^2/48 when th < 1e-6
//     quaternion_to_matrix:     R = I + (2 / q.q) * B(q)
//   Latency-bound scalar work: one 32-thread block per pose, thread j = joint j, the chain evaluated level by level
//   of the SMPL tree (9 levels) through LDS; it exists to take ~140 tiny launches off the pose-refinement step, not to
//   chase a roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "anerf.h"
#include "anerf_dev.h"

namespace anerf {

constexpr int NJ = 24;
// SMPLSkeleton.joint_trees (core/utils/skeleton_utils.py:98-104), root_id = 0
__device__ __constant__ int kParent[NJ] = {0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};

struct Quat { float r, i, j, k, th, kf; };

__device__ __forceinline__ Quat aa_to_quat(const float* a) {
  Quat q;
  q.th = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const float hf = 0.5f * q.th;
  q.kf = q.th < 1e-6f ? 0.5f - q.th * q.th / 48.f : sinf(hf) / q.th;
  q.r = cosf(hf);
  q.i = a[0] * q.kf;
  q.j = a[1] * q.kf;
  q.k = a[2] * q.kf;
  return q;
}

__device__ __forceinline__ void quat_to_mat(const Quat& q, float* R) {
  const float two_s = 2.0f / (q.r * q.r + q.i * q.i + q.j * q.j + q.k * q.k);
  R[0] = 1.f - two_s * (q.j * q.j + q.k * q.k);
  R[1] = two_s * (q.i * q.j - q.k * q.r);
  R[2] = two_s * (q.i * q.k + q.j * q.r);
  R[3] = two_s * (q.i * q.j + q.k * q.r);
  R[4] = 1.f - two_s * (q.i * q.i + q.k * q.k);
  R[5] = two_s * (q.j * q.k - q.i * q.r);
  R[6] = two_s * (q.i * q.k - q.j * q.r);
  R[7] = two_s * (q.j * q.k + q.i * q.r);
  R[8] = 1.f - two_s * (q.i * q.i + q.j * q.j);
}

// d(loss)/dR -> d(loss)/d(axis-angle)
__device__ __forceinline__ void aa_backward(const float* a, const float* dR, float* da) {
  const Quat q = aa_to_quat(a);
  const float n = q.r * q.r + q.i * q.i + q.j * q.j + q.k * q.k, two_s = 2.0f / n;
  // R = I + two_s * B
  const float B[9] = {-(q.j * q.j + q.k * q.k), q.i * q.j - q.k * q.r, q.i * q.k + q.j * q.r,
                      q.i * q.j + q.k * q.r, -(q.i * q.i + q.k * q.k), q.j * q.k - q.i * q.r,
                      q.i * q.k - q.j * q.r, q.j * q.k + q.i * q.r, -(q.i * q.i + q.j * q.j)};
  float d_two_s = 0.f, dB[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    d_two_s += dR[e] * B[e];
    dB[e] = dR[e] * two_s;
  }
  float dr = -q.k * dB[1] + q.j * dB[2] + q.k * dB[3] - q.i * dB[5] - q.j * dB[6] + q.i * dB[7];
  float di = q.j * dB[1] + q.k * dB[2] + q.j * dB[3] - 2.f * q.i * dB[4] - q.r * dB[5] + q.k * dB[6] + q.r * dB[7] - 2.f * q.i * dB[8];
  float dj = -2.f * q.j * dB[0] + q.i * dB[1] + q.r * dB[2] + q.i * dB[3] + q.k * dB[5] - q.r * dB[6] + q.k * dB[7] - 2.f * q.j * dB[8];
  float dk = -2.f * q.k * dB[0] - q.r * dB[1] + q.i * dB[2] + q.r * dB[3] - 2.f * q.k * dB[4] + q.j * dB[5] + q.i * dB[6] + q.j * dB[7];
  const float dn = -2.0f / (n * n) * d_two_s;      // two_s = 2 / n
  dr += 2.f * q.r * dn;
  di += 2.f * q.i * dn;
  dj += 2.f * q.j * dn;
  dk += 2.f * q.k * dn;
  // q = (cos(th/2), a * k(th))
  const float hf = 0.5f * q.th;
  float dth = -0.5f * sinf(hf) * dr;
  const float dkf = a[0] * di + a[1] * dj + a[2] * dk;
  const float dk_dth = q.th < 1e-6f ? -q.th / 24.f : (0.5f * cosf(hf) * q.th - sinf(hf)) / (q.th * q.th);
  dth += dkf * dk_dth;
  const float inv = q.th > 0.f ? 1.0f / q.th : 0.f;   // d|a|/da = a/|a| (0 at the origin, as torch.norm's backward)
  da[0] = q.kf * di + dth * a[0] * inv;
  da[1] = q.kf * dj + dth * a[1] * inv;
  da[2] = q.kf * dk + dth * a[2] * inv;
}

// 6D rotation parameters (opt_rot6d, 6 of the reference's 8 configs; Zhou et al. 2019): rot6d_to_rotmat
// (core/utils/skeleton_utils.py:420-436).  x = R[:, :2] row-major: a1 = (x0,x2,x4), a2 = (x1,x3,x5);
// b1 = a1/max(|a1|,1e-12) (F.normalize), b2 = normalize(a2 - (b1.a2) b1), b3 = b1 x b2, R = [b1 b2 b3] (columns).
struct Rot6 { float b1[3], b2[3], a2[3], n1, n2, d; };

__device__ __forceinline__ Rot6 rot6d_frame(const float* x) {
  Rot6 f;
  const float a1[3] = {x[0], x[2], x[4]};
  f.a2[0] = x[1]; f.a2[1] = x[3]; f.a2[2] = x[5];
  f.n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
  for (int c = 0; c < 3; ++c) f.b1[c] = a1[c] / f.n1;
  f.d = f.b1[0] * f.a2[0] + f.b1[1] * f.a2[1] + f.b1[2] * f.a2[2];
  float v[3];
  for (int c = 0; c < 3; ++c) v[c] = f.a2[c] - f.d * f.b1[c];
  f.n2 = fmaxf(sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), 1e-12f);
  for (int c = 0; c < 3; ++c) f.b2[c] = v[c] / f.n2;
  return f;
}

__device__ __forceinline__ void rot6d_to_mat(const float* x, float* R) {
  const Rot6 f = rot6d_frame(x);
  const float b3[3] = {f.b1[1] * f.b2[2] - f.b1[2] * f.b2[1], f.b1[2] * f.b2[0] - f.b1[0] * f.b2[2],
                       f.b1[0] * f.b2[1] - f.b1[1] * f.b2[0]};
  for (int r = 0; r < 3; ++r) { R[3 * r] = f.b1[r]; R[3 * r + 1] = f.b2[r]; R[3 * r + 2] = b3[r]; }
}

// d(loss)/dR -> d(loss)/d(6D parameters)
__device__ __forceinline__ void rot6d_backward(const float* x, const float* dR, float* dx) {
  const Rot6 f = rot6d_frame(x);
  float db1[3], db2[3];
  const float db3[3] = {dR[2], dR[5], dR[8]};
  for (int r = 0; r < 3; ++r) { db1[r] = dR[3 * r]; db2[r] = dR[3 * r + 1]; }
  // b3 = b1 x b2 :  db1 += b2 x db3,  db2 += db3 x b1
  db1[0] += f.b2[1] * db3[2] - f.b2[2] * db3[1];
  db1[1] += f.b2[2] * db3[0] - f.b2[0] * db3[2];
  db1[2] += f.b2[0] * db3[1] - f.b2[1] * db3[0];
  db2[0] += db3[1] * f.b1[2] - db3[2] * f.b1[1];
  db2[1] += db3[2] * f.b1[0] - db3[0] * f.b1[2];
  db2[2] += db3[0] * f.b1[1] - db3[1] * f.b1[0];
  // b2 = v / |v|
  const float p2 = f.b2[0] * db2[0] + f.b2[1] * db2[1] + f.b2[2] * db2[2];
  float dv[3], da2[3], da1[3];
  for (int c = 0; c < 3; ++c) dv[c] = (db2[c] - (f.n2 > 1e-12f ? f.b2[c] * p2 : 0.f)) / f.n2;
  // v = a2 - d b1,  d = b1 . a2
  const float dd = -(dv[0] * f.b1[0] + dv[1] * f.b1[1] + dv[2] * f.b1[2]);
  for (int c = 0; c < 3; ++c) {
    da2[c] = dv[c] + dd * f.b1[c];
    db1[c] += -f.d * dv[c] + dd * f.a2[c];
  }
  // b1 = a1 / |a1|
  const float p1 = f.b1[0] * db1[0] + f.b1[1] * db1[1] + f.b1[2] * db1[2];
  for (int c = 0; c < 3; ++c) da1[c] = (db1[c] - (f.n1 > 1e-12f ? f.b1[c] * p1 : 0.f)) / f.n1;
  for (int c = 0; c < 3; ++c) { dx[2 * c] = da1[c]; dx[2 * c + 1] = da2[c]; }
}

// SMPL tree depth of every joint (root = 0): the chain is evaluated level by level, the joints of a level in parallel
__device__ __constant__ int kDepth[NJ] = {0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 7, 7, 8, 8};
constexpr int NLEVEL = 9;
constexpr int FKT = 32;   // threads per pose (one per joint, 8 idle)

struct FkShared {
  float Rl[NJ][9], Rg[NJ][9], tg[NJ][3], dRg[NJ][9], dc[NJ][3];
};

// forward chain of one pose into shared memory: thread j = joint j
__device__ __forceinline__ void fk_chain(FkShared& S, const float* __restrict__ bones_u, int rd,
                                         const float* __restrict__ rp, int j) {
  if (j < NJ) {
    if (rd == 6) rot6d_to_mat(bones_u + 6 * j, S.Rl[j]);
    else quat_to_mat(aa_to_quat(bones_u + 3 * j), S.Rl[j]);
  }
  __syncthreads();
  for (int lvl = 0; lvl < NLEVEL; ++lvl) {
    if (j < NJ && kDepth[j] == lvl) {
      if (j == 0) {
        for (int e = 0; e < 9; ++e) S.Rg[0][e] = S.Rl[0][e];
        for (int c = 0; c < 3; ++c) S.tg[0][c] = rp[c];
      } else {
        const int p = kParent[j];
        const float o[3] = {rp[3 * j] - rp[3 * p], rp[3 * j + 1] - rp[3 * p + 1], rp[3 * j + 2] - rp[3 * p + 2]};
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            S.Rg[j][3 * r + c] = S.Rg[p][3 * r] * S.Rl[j][c] + S.Rg[p][3 * r + 1] * S.Rl[j][3 + c] + S.Rg[p][3 * r + 2] * S.Rl[j][6 + c];
          S.tg[j][r] = S.Rg[p][3 * r] * o[0] + S.Rg[p][3 * r + 1] * o[1] + S.Rg[p][3 * r + 2] * o[2] + S.tg[p][r];
        }
      }
    }
    __syncthreads();
  }
}

// one block (32 threads) per pose.  l2ws / skts [U,24,4,4] row-major, rots [U,24,3,3], kp [U,24,3]; any output may be NULL.
__global__ __launch_bounds__(FKT) void k_fk_fwd(const float* __restrict__ bones, int rd, const float* __restrict__ pelvis,
                                                const float* __restrict__ rest, long long rest_stride, int n,
                                                float* __restrict__ l2ws, float* __restrict__ skts,
                                                float* __restrict__ rots, float* __restrict__ kp) {
  __shared__ FkShared S;
  const int u = blockIdx.x, j = threadIdx.x;
  const float* rp = rest + (long long)u * rest_stride;
  fk_chain(S, bones + (long long)u * NJ * rd, rd, rp, j);
  if (j >= NJ) return;
  float pv[3] = {0.f, 0.f, 0.f};
  if (pelvis) { pv[0] = pelvis[3 * u]; pv[1] = pelvis[3 * u + 1]; pv[2] = pelvis[3 * u + 2]; }
  const float* Rg = S.Rg[j];
  const float c[3] = {S.tg[j][0] + pv[0], S.tg[j][1] + pv[1], S.tg[j][2] + pv[2]};
  const long long m = ((long long)u * NJ + j) * 16;
  if (rots)
    for (int e = 0; e < 9; ++e) rots[((long long)u * NJ + j) * 9 + e] = S.Rl[j][e];
  if (l2ws) {
    for (int r = 0; r < 3; ++r) {
      l2ws[m + 4 * r] = Rg[3 * r]; l2ws[m + 4 * r + 1] = Rg[3 * r + 1]; l2ws[m + 4 * r + 2] = Rg[3 * r + 2];
      l2ws[m + 4 * r + 3] = c[r];
    }
    l2ws[m + 12] = 0.f; l2ws[m + 13] = 0.f; l2ws[m + 14] = 0.f; l2ws[m + 15] = 1.f;
  }
  if (skts) {   // inverse of a rigid transform: [R^T | -R^T c]
    for (int r = 0; r < 3; ++r) {
      skts[m + 4 * r] = Rg[r]; skts[m + 4 * r + 1] = Rg[3 + r]; skts[m + 4 * r + 2] = Rg[6 + r];
      skts[m + 4 * r + 3] = -(Rg[r] * c[0] + Rg[3 + r] * c[1] + Rg[6 + r] * c[2]);
    }
    skts[m + 12] = 0.f; skts[m + 13] = 0.f; skts[m + 14] = 0.f; skts[m + 15] = 1.f;
  }
  if (kp) { kp[((long long)u * NJ + j) * 3] = c[0]; kp[((long long)u * NJ + j) * 3 + 1] = c[1]; kp[((long long)u * NJ + j) * 3 + 2] = c[2]; }
}

// backward: gradients w.r.t. skts / l2ws (rows 0..2 used) / kp / rots  ->  gradients w.r.t. bones and pelvis
__global__ __launch_bounds__(FKT) void k_fk_bwd(const float* __restrict__ bones, int rd, const float* __restrict__ pelvis,
                                                const float* __restrict__ rest, long long rest_stride, int n,
                                                const float* __restrict__ g_skts, const float* __restrict__ g_l2ws,
                                                const float* __restrict__ g_kp, const float* __restrict__ g_rots,
                                                float* __restrict__ g_bones, float* __restrict__ g_pelvis) {
  __shared__ FkShared S;
  __shared__ float dpel[NJ][3];
  const int u = blockIdx.x, j = threadIdx.x;
  const float* rp = rest + (long long)u * rest_stride;
  fk_chain(S, bones + (long long)u * NJ * rd, rd, rp, j);     // recompute the chain
  float pv[3] = {0.f, 0.f, 0.f};
  if (pelvis) { pv[0] = pelvis[3 * u]; pv[1] = pelvis[3 * u + 1]; pv[2] = pelvis[3 * u + 2]; }
  // gradients w.r.t. the global rotation and the joint centre c = tg + pelvis of every joint
  if (j < NJ) {
    float dRg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dc[3] = {0.f, 0.f, 0.f};
    const long long m = ((long long)u * NJ + j) * 16;
    const float* Rg = S.Rg[j];
    const float c[3] = {S.tg[j][0] + pv[0], S.tg[j][1] + pv[1], S.tg[j][2] + pv[2]};
    if (g_l2ws)
      for (int r = 0; r < 3; ++r) {
        dRg[3 * r] += g_l2ws[m + 4 * r]; dRg[3 * r + 1] += g_l2ws[m + 4 * r + 1]; dRg[3 * r + 2] += g_l2ws[m + 4 * r + 2];
        dc[r] += g_l2ws[m + 4 * r + 3];
      }
    if (g_skts) {   // S = Rg^T, s = -Rg^T c
      float ds[3];
      for (int r = 0; r < 3; ++r) {
        ds[r] = g_skts[m + 4 * r + 3];
        for (int k = 0; k < 3; ++k) dRg[3 * k + r] += g_skts[m + 4 * r + k];     // dS[r][k] -> dRg[k][r]
      }
      for (int k = 0; k < 3; ++k) {
        dc[k] -= Rg[3 * k] * ds[0] + Rg[3 * k + 1] * ds[1] + Rg[3 * k + 2] * ds[2];
        for (int r = 0; r < 3; ++r) dRg[3 * k + r] -= c[k] * ds[r];
      }
    }
    if (g_kp)
      for (int r = 0; r < 3; ++r) dc[r] += g_kp[((long long)u * NJ + j) * 3 + r];
    for (int e = 0; e < 9; ++e) S.dRg[j][e] = dRg[e];
    for (int r = 0; r < 3; ++r) { S.dc[j][r] = dc[r]; dpel[j][r] = dc[r]; }
  }
  __syncthreads();
  if (g_pelvis && j < 3) {     // every joint centre contains the pelvis shift; fixed summation order
    float s = 0.f;
    for (int q = 0; q < NJ; ++q) s += dpel[q][j];
    g_pelvis[3 * u + j] = s;
  }
  // reverse chain, level by level: a parent gathers from its (already final) children
  for (int lvl = NLEVEL - 2; lvl >= 0; --lvl) {
    if (j < NJ && kDepth[j] == lvl) {
      for (int ch = j + 1; ch < NJ; ++ch) {
        if (kParent[ch] != j) continue;
        const float o[3] = {rp[3 * ch] - rp[3 * j], rp[3 * ch + 1] - rp[3 * j + 1], rp[3 * ch + 2] - rp[3 * j + 2]};
        // Rg_c = Rg_p Rl_c : dRg_p += dRg_c Rl_c^T ;  tg_c = Rg_p o + tg_p : dRg_p += dtg_c o^T, dtg_p += dtg_c
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            S.dRg[j][3 * r + c] += S.dRg[ch][3 * r] * S.Rl[ch][3 * c] + S.dRg[ch][3 * r + 1] * S.Rl[ch][3 * c + 1] +
                                   S.dRg[ch][3 * r + 2] * S.Rl[ch][3 * c + 2] + S.dc[ch][r] * o[c];
          S.dc[j][r] += S.dc[ch][r];
        }
      }
    }
    __syncthreads();
  }
  if (j >= NJ) return;
  float dRl[9];
  if (j == 0) {
    for (int e = 0; e < 9; ++e) dRl[e] = S.dRg[0][e];
  } else {
    const int p = kParent[j];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)   // dRl = Rg_p^T dRg_j
        dRl[3 * r + c] = S.Rg[p][r] * S.dRg[j][c] + S.Rg[p][3 + r] * S.dRg[j][3 + c] + S.Rg[p][6 + r] * S.dRg[j][6 + c];
  }
  if (g_rots)
    for (int e = 0; e < 9; ++e) dRl[e] += g_rots[((long long)u * NJ + j) * 9 + e];
  float da[6];
  if (rd == 6) rot6d_backward(bones + ((long long)u * NJ + j) * 6, dRl, da);
  else aa_backward(bones + ((long long)u * NJ + j) * 3, dRl, da);
  for (int e = 0; e < rd; ++e) g_bones[((long long)u * NJ + j) * rd + e] = da[e];
}


// ------------------------------------------------------------------------------------------------
// The pose layer's per-iteration work in three launches (round 4; the 384-rays-per-rank Mixamo step spent ~25 small torch
// launches here: index_select x 2, five per-ray expansions, their reshaped sums, two index_add_, zero fills, gradient
// accumulation adds -- PoseOptLayer.forward of core/pose_opt.py:318-331,372-445 and its autograd):
//   k_pose_batch_fwd   block (c, u) = ray chunk c (64 rays) x distinct pose u: if the chunk holds rays of pose u -- one ballot --
//                      the parameters are gathered through pose_idx[u], the chain is evaluated and the PER-RAY rows the
//                      reference's layer returns (kp / bones / skts / l2ws / rots [N, ...]) are written for those rays, a wave
//                      per ray, coalesced; chunk 0's blocks also write the unique-level outputs the regulariser reads.
//   k_pose_partial     block (c, u): the chunk's rays of pose u compacted in ray order (ballot prefix), their per-ray gradients
//                      summed in that order -> partial[u][c][1200] (+ the count, so that empty chunks are skipped later);
//   k_pose_batch_bwd   block u: partials summed in chunk order (fixed order: bit-reproducible), the unique-level gradients
//                      added, the reverse chain, and the result written or ADDED to rows pose_idx[u] of the full-size parameter
//                      gradients (distinct poses: no two blocks touch a row).
// (A first version with one block per pose for everything was SLOWER than the launches it replaced: 8 blocks writing 15 MB of
// rows and summing 384 rays each serially cost 0.75 ms at 3072 rays.)
// ------------------------------------------------------------------------------------------------
constexpr int PBT = 256;
constexpr int RPB = 64;           // rays per chunk = one ballot
constexpr int PW = NJ * 16 + NJ * 16 + NJ * 9 + NJ * 3 + NJ * 6;     // 1200: [skts | l2ws | rots | kp | bones(6)] of one pose
struct PoseRows {     // one pose's outputs (or summed output gradients) staged in LDS
  float skts[NJ * 16], l2ws[NJ * 16], rots[NJ * 9], kp[NJ * 3], bones[NJ * 6];
};
static_assert(sizeof(PoseRows) == PW * 4, "PoseRows is the 1200-float row");

__global__ __launch_bounds__(PBT) void k_pose_batch_fwd(const float* __restrict__ bones_all, int rd, const float* __restrict__ pelvis_all,
                                                        const float* __restrict__ rest, const long long* __restrict__ pose_idx,
                                                        const int* __restrict__ inverse, int n_rays,
                                                        float* __restrict__ u_kp, float* __restrict__ u_bones, float* __restrict__ u_rots,
                                                        float* __restrict__ r_kp, float* __restrict__ r_bones, float* __restrict__ r_skts,
                                                        float* __restrict__ r_l2ws, float* __restrict__ r_rots) {
  __shared__ FkShared S;
  __shared__ PoseRows R;
  __shared__ unsigned long long hits;
  const int c = blockIdx.x, u = blockIdx.y, t = threadIdx.x;
  const int n0 = c * RPB;
  if (t < 64) {
    const int n = n0 + t;
    const unsigned long long b = __ballot(n < n_rays && inverse[n] == u);
    if (t == 0) hits = b;
  }
  __syncthreads();
  const unsigned long long mine = hits;
  if (mine == 0ull && c != 0) return;          // no ray of this pose in the chunk (chunk 0 still owes the unique outputs)
  const long long pi = pose_idx[u];
  __shared__ float rest_s[NJ * 3], bones_s[NJ * 6];      // the chain reads these level by level: one global round trip, not nine
  if (t < NJ * 3) rest_s[t] = rest[t];
  for (int e = t; e < NJ * rd; e += PBT) bones_s[e] = bones_all[pi * NJ * rd + e];
  __syncthreads();
  const float* bones_u = bones_s;
  fk_chain(S, bones_u, rd, rest_s, t);
  if (t < NJ) {
    const int j = t;
    float pv[3] = {0.f, 0.f, 0.f};
    if (pelvis_all) { pv[0] = pelvis_all[3 * pi]; pv[1] = pelvis_all[3 * pi + 1]; pv[2] = pelvis_all[3 * pi + 2]; }
    const float* Rg = S.Rg[j];
    const float cc[3] = {S.tg[j][0] + pv[0], S.tg[j][1] + pv[1], S.tg[j][2] + pv[2]};
    for (int e = 0; e < 9; ++e) R.rots[9 * j + e] = S.Rl[j][e];
    for (int r = 0; r < 3; ++r) {
      R.l2ws[16 * j + 4 * r] = Rg[3 * r]; R.l2ws[16 * j + 4 * r + 1] = Rg[3 * r + 1]; R.l2ws[16 * j + 4 * r + 2] = Rg[3 * r + 2];
      R.l2ws[16 * j + 4 * r + 3] = cc[r];
      R.skts[16 * j + 4 * r] = Rg[r]; R.skts[16 * j + 4 * r + 1] = Rg[3 + r]; R.skts[16 * j + 4 * r + 2] = Rg[6 + r];
      R.skts[16 * j + 4 * r + 3] = -(Rg[r] * cc[0] + Rg[3 + r] * cc[1] + Rg[6 + r] * cc[2]);
      R.kp[3 * j + r] = cc[r];
    }
    for (int e = 0; e < 3; ++e) { R.l2ws[16 * j + 12 + e] = 0.f; R.skts[16 * j + 12 + e] = 0.f; }
    R.l2ws[16 * j + 15] = 1.f; R.skts[16 * j + 15] = 1.f;
    for (int e = 0; e < rd; ++e) R.bones[rd * j + e] = bones_u[rd * j + e];
  }
  __syncthreads();
  if (c == 0) {     // unique-level outputs
    for (int e = t; e < NJ * 3; e += PBT) if (u_kp) u_kp[(long long)u * NJ * 3 + e] = R.kp[e];
    for (int e = t; e < NJ * rd; e += PBT) if (u_bones) u_bones[(long long)u * NJ * rd + e] = R.bones[e];
    for (int e = t; e < NJ * 9; e += PBT) if (u_rots) u_rots[(long long)u * NJ * 9 + e] = R.rots[e];
  }
  // per-ray rows: a wave per ray of this pose in the chunk
  const int lane = t & 63, wave = t >> 6;
  for (int k = wave; k < RPB; k += PBT / 64) {
    if (!((mine >> k) & 1ull)) continue;       // wave-uniform
    const long long n = n0 + k;
    if (r_skts) for (int e = lane; e < NJ * 16; e += 64) r_skts[n * NJ * 16 + e] = R.skts[e];
    if (r_l2ws) for (int e = lane; e < NJ * 16; e += 64) r_l2ws[n * NJ * 16 + e] = R.l2ws[e];
    if (r_rots) for (int e = lane; e < NJ * 9; e += 64) r_rots[n * NJ * 9 + e] = R.rots[e];
    if (r_kp) for (int e = lane; e < NJ * 3; e += 64) r_kp[n * NJ * 3 + e] = R.kp[e];
    if (r_bones) for (int e = lane; e < NJ * rd; e += 64) r_bones[n * NJ * rd + e] = R.bones[e];
  }
}

// per-ray gradient sums of one (chunk, pose): partial [U][C][PW], count [U][C]
__global__ __launch_bounds__(PBT) void k_pose_partial(const int* __restrict__ inverse, int n_rays, int rd, const float* __restrict__ gr_kp,
                                                      const float* __restrict__ gr_bones, const float* __restrict__ gr_skts,
                                                      const float* __restrict__ gr_l2ws, const float* __restrict__ gr_rots,
                                                      float* __restrict__ partial, int* __restrict__ count) {
  __shared__ int list[RPB];
  __shared__ int n_list;
  const int c = blockIdx.x, u = blockIdx.y, C = gridDim.x, t = threadIdx.x;
  const int n0 = c * RPB;
  if (t < 64) {
    const int n = n0 + t;
    const bool hit = n < n_rays && inverse[n] == u;
    const unsigned long long b = __ballot(hit);
    if (hit) list[__popcll(b & ((1ull << t) - 1ull))] = n;
    if (t == 0) { n_list = __popcll(b); count[u * C + c] = __popcll(b); }
  }
  __syncthreads();
  const int nl = n_list;
  if (nl == 0) return;
  float* out = partial + ((long long)u * C + c) * PW;
  // thread = element of the 1200-float row, loop = the chunk's rays of this pose in ray order (independent loads: unrolled)
  for (int e = t; e < PW; e += PBT) {
    const float* src; int w, o;
    if (e < NJ * 16) { src = gr_skts; w = NJ * 16; o = e; }
    else if (e < 2 * NJ * 16) { src = gr_l2ws; w = NJ * 16; o = e - NJ * 16; }
    else if (e < 2 * NJ * 16 + NJ * 9) { src = gr_rots; w = NJ * 9; o = e - 2 * NJ * 16; }
    else if (e < 2 * NJ * 16 + NJ * 12) { src = gr_kp; w = NJ * 3; o = e - 2 * NJ * 16 - NJ * 9; }
    else { src = gr_bones; w = NJ * rd; o = e - 2 * NJ * 16 - NJ * 12; if (o >= w) src = nullptr; }
    float a = 0.f;
    if (src) {
#pragma unroll 8
      for (int k = 0; k < nl; ++k) a += src[(long long)list[k] * w + o];
    }
    out[e] = a;
  }
}

__global__ __launch_bounds__(PBT) void k_pose_batch_bwd(const float* __restrict__ bones_all, int rd, const float* __restrict__ pelvis_all,
                                                        const float* __restrict__ rest_g, const long long* __restrict__ pose_idx, int C,
                                                        const float* __restrict__ partial, const int* __restrict__ count,
                                                        const float* __restrict__ gu_kp, const float* __restrict__ gu_bones,
                                                        const float* __restrict__ gu_rots, float* __restrict__ g_bones_all,
                                                        float* __restrict__ g_pelvis_all, int accumulate) {
  __shared__ FkShared S;
  __shared__ PoseRows G;           // summed gradients w.r.t. this pose's outputs
  __shared__ float dpel[NJ][3];
  const int u = blockIdx.x, t = threadIdx.x;
  const long long pi = pose_idx[u];
  __shared__ float rest_s[NJ * 3], bones_s[NJ * 6];      // read level by level by the chains below: staged once
  if (t < NJ * 3) rest_s[t] = rest_g[t];
  for (int e = t; e < NJ * rd; e += PBT) bones_s[e] = bones_all[pi * NJ * rd + e];
  const float* rest = rest_s;
  const float* bones_u = bones_s;
  float* Gf = reinterpret_cast<float*>(&G);
  // the non-empty chunks of this pose, in chunk order (wave 0: ballot + prefix count), then thread = element, loop = those chunks:
  // independent loads (a loop over ALL chunks with a branch on the count was 48 serial round trips per element: 0.17 ms)
  constexpr int CL = 2048, EPT = (PW + PBT - 1) / PBT;
  __shared__ int clist[CL];
  __shared__ int n_clist;
  float a[EPT];
#pragma unroll
  for (int i = 0; i < EPT; ++i) a[i] = 0.f;
  for (int cb = 0; partial && cb < C; cb += CL) {
    if (t < 64) {
      int cnt = 0;
      for (int c0 = cb; c0 < C && c0 < cb + CL; c0 += 64) {
        const int c = c0 + t;
        const bool hit = c < C && c < cb + CL && count[u * C + c] > 0;
        const unsigned long long b = __ballot(hit);
        if (hit) clist[cnt + __popcll(b & ((1ull << t) - 1ull))] = c;
        cnt += __popcll(b);
      }
      if (t == 0) n_clist = cnt;
    }
    __syncthreads();
    const int nc = n_clist;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
      const int e = t + i * PBT;
      if (e < PW) {
        float acc = a[i];
#pragma unroll 8
        for (int k = 0; k < nc; ++k) acc += partial[((long long)u * C + clist[k]) * PW + e];
        a[i] = acc;
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < EPT; ++i) if (t + i * PBT < PW) Gf[t + i * PBT] = a[i];
  __syncthreads();
  for (int e = t; e < NJ * 9; e += PBT) if (gu_rots) G.rots[e] += gu_rots[(long long)u * NJ * 9 + e];
  for (int e = t; e < NJ * 3; e += PBT) if (gu_kp) G.kp[e] += gu_kp[(long long)u * NJ * 3 + e];
  for (int e = t; e < NJ * rd; e += PBT) if (gu_bones) G.bones[e] += gu_bones[(long long)u * NJ * rd + e];
  // ---- reverse chain on the summed gradients (k_fk_bwd's arithmetic)
  fk_chain(S, bones_u, rd, rest, t);
  float pv[3] = {0.f, 0.f, 0.f};
  if (pelvis_all) { pv[0] = pelvis_all[3 * pi]; pv[1] = pelvis_all[3 * pi + 1]; pv[2] = pelvis_all[3 * pi + 2]; }
  const int j = t;
  if (j < NJ) {
    float dRg[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dc[3] = {0.f, 0.f, 0.f};
    const float* Rg = S.Rg[j];
    const float c[3] = {S.tg[j][0] + pv[0], S.tg[j][1] + pv[1], S.tg[j][2] + pv[2]};
    const float* gl = G.l2ws + 16 * j;
    const float* gs = G.skts + 16 * j;
    for (int r = 0; r < 3; ++r) {
      dRg[3 * r] += gl[4 * r]; dRg[3 * r + 1] += gl[4 * r + 1]; dRg[3 * r + 2] += gl[4 * r + 2];
      dc[r] += gl[4 * r + 3];
    }
    float ds[3];
    for (int r = 0; r < 3; ++r) {
      ds[r] = gs[4 * r + 3];
      for (int k = 0; k < 3; ++k) dRg[3 * k + r] += gs[4 * r + k];
    }
    for (int k = 0; k < 3; ++k) {
      dc[k] -= Rg[3 * k] * ds[0] + Rg[3 * k + 1] * ds[1] + Rg[3 * k + 2] * ds[2];
      for (int r = 0; r < 3; ++r) dRg[3 * k + r] -= c[k] * ds[r];
    }
    for (int r = 0; r < 3; ++r) dc[r] += G.kp[3 * j + r];
    for (int e = 0; e < 9; ++e) S.dRg[j][e] = dRg[e];
    for (int r = 0; r < 3; ++r) { S.dc[j][r] = dc[r]; dpel[j][r] = dc[r]; }
  }
  __syncthreads();
  if (g_pelvis_all && j < 3) {
    float s = 0.f;
    for (int q = 0; q < NJ; ++q) s += dpel[q][j];
    float* o = g_pelvis_all + 3 * pi + j;
    *o = accumulate ? *o + s : s;
  }
  for (int lvl = NLEVEL - 2; lvl >= 0; --lvl) {
    if (j < NJ && kDepth[j] == lvl) {
      for (int ch = j + 1; ch < NJ; ++ch) {
        if (kParent[ch] != j) continue;
        const float o[3] = {rest[3 * ch] - rest[3 * j], rest[3 * ch + 1] - rest[3 * j + 1], rest[3 * ch + 2] - rest[3 * j + 2]};
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            S.dRg[j][3 * r + c] += S.dRg[ch][3 * r] * S.Rl[ch][3 * c] + S.dRg[ch][3 * r + 1] * S.Rl[ch][3 * c + 1] +
                                   S.dRg[ch][3 * r + 2] * S.Rl[ch][3 * c + 2] + S.dc[ch][r] * o[c];
          S.dc[j][r] += S.dc[ch][r];
        }
      }
    }
    __syncthreads();
  }
  if (j >= NJ) return;
  float dRl[9];
  if (j == 0) {
    for (int e = 0; e < 9; ++e) dRl[e] = S.dRg[0][e];
  } else {
    const int p = kParent[j];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        dRl[3 * r + c] = S.Rg[p][r] * S.dRg[j][c] + S.Rg[p][3 + r] * S.dRg[j][3 + c] + S.Rg[p][6 + r] * S.dRg[j][6 + c];
  }
  for (int e = 0; e < 9; ++e) dRl[e] += G.rots[9 * j + e];
  float da[6];
  if (rd == 6) rot6d_backward(bones_u + 6 * j, dRl, da);
  else aa_backward(bones_u + 3 * j, dRl, da);
  for (int e = 0; e < rd; ++e) {
    float* o = g_bones_all + (pi * NJ + j) * rd + e;
    const float v = da[e] + G.bones[rd * j + e];
    *o = accumulate ? *o + v : v;
  }
}

}  // namespace anerf

using namespace anerf;

// ------------------------------------------------------------------------------------------------
// Pose regulariser (_compute_kp_loss, core/trainer.py:382-403): per joint j >= 1 of a pose, d = (anchor - value)^2 per
// component, thresholded "d > tol ? d - tol : 0", summed over the components, averaged over rays x 23 joints, x coef.
// The reference evaluates it on the per-ray replicated batch; with w_u = (rays of pose u) / N it is the same number over the
// U distinct poses.  One block, thread = (pose, joint); loss and its gradient w.r.t. the values in one pass (the gradient is
// linear in the upstream scalar, which autograd applies).  value layout: vals [U,24,stride] with the first `dim` entries of
// the 6 (rot6d: rots[..., :3, :2] row-major = entries (r, c) at 3 r + c of a 3x3 row-major matrix) or 3 (axis-angle bones).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kp_loss(const float* __restrict__ vals, int rot6d, const float* __restrict__ anchors,
                                                 const float* __restrict__ w, int U, float tol, float coef,
                                                 float* __restrict__ loss, float* __restrict__ g_vals,
                                                 const float* __restrict__ base, float* __restrict__ total) {
  __shared__ float sh[4];
  float acc = 0.f;
  const int dim = rot6d ? 6 : 3, stride = rot6d ? 9 : 3;
  for (int i = threadIdx.x; i < U * 24; i += blockDim.x) {
    const int u = i / 24, j = i - 24 * u;
    const float wu = w[u] * coef / 23.0f;
    for (int c = 0; c < dim; ++c) {
      const int e = rot6d ? 3 * (c >> 1) + (c & 1) : c;        // rot6d component c = (row c / 2, column c % 2)
      float g = 0.f;
      if (j > 0) {
        const float d = anchors[(long long)i * dim + c] - vals[(long long)i * stride + e];
        const float d2 = d * d;
        if (d2 > tol) {
          acc += wu * (d2 - tol);
          g = -2.0f * wu * d;
        }
      }
      if (g_vals) g_vals[(long long)i * stride + e] = g;
    }
    if (g_vals && rot6d) {
      g_vals[(long long)i * 9 + 2] = 0.f; g_vals[(long long)i * 9 + 5] = 0.f; g_vals[(long long)i * 9 + 8] = 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float kp = sh[0] + sh[1] + sh[2] + sh[3];     // fixed order: bit-reproducible
    *loss = kp;
    if (total) *total = *base + kp;                      // anerf_kp_loss_add: the trainer's `total_loss + kp_loss` (one fp32 add, as torch's)
  }
}

extern "C" {

int anerf_kp_loss(const float* values, int32_t rot6d, const float* anchors, const float* pose_weights, int32_t n_poses, float tol,
                  float coef, float* loss, float* g_values, void* stream) {
  if (n_poses < 0) return set_error(ANERF_E_SHAPE, "kp_loss: n_poses >= 0");
  if (!loss) return set_error(ANERF_E_NULL, "kp_loss: loss is NULL");
  if (n_poses > 0 && (!values || !anchors || !pose_weights)) return set_error(ANERF_E_NULL, "kp_loss: NULL pointer");
  hipLaunchKernelGGL(k_kp_loss, dim3(1), dim3(256), 0, (hipStream_t)stream, values, (int)(rot6d != 0), anchors, pose_weights,
                     (int)n_poses, tol, coef, loss, g_values, (const float*)nullptr, (float*)nullptr);
  return check_launch("k_kp_loss");
}

int anerf_kp_loss_add(const float* values, int32_t rot6d, const float* anchors, const float* pose_weights, int32_t n_poses, float tol,
                      float coef, const float* base, float* loss, float* total, float* g_values, void* stream) {
  if (n_poses < 0) return set_error(ANERF_E_SHAPE, "kp_loss_add: n_poses >= 0");
  if (!loss || !base || !total) return set_error(ANERF_E_NULL, "kp_loss_add: loss / base / total is NULL");
  if (n_poses > 0 && (!values || !anchors || !pose_weights)) return set_error(ANERF_E_NULL, "kp_loss_add: NULL pointer");
  hipLaunchKernelGGL(k_kp_loss, dim3(1), dim3(256), 0, (hipStream_t)stream, values, (int)(rot6d != 0), anchors, pose_weights,
                     (int)n_poses, tol, coef, loss, g_values, base, total);
  return check_launch("k_kp_loss");
}

int anerf_fk_forward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose,
                     int64_t rest_pose_stride, int32_t n_poses, float* l2ws, float* skts, float* rots, float* kp,
                     void* stream) {
  if (rot_dim != 3 && rot_dim != 6) return set_error(ANERF_E_CONFIG, "fk_forward: rot_dim must be 3 (axis-angle) or 6 (rot6d)");
  if (n_poses < 0 || (rest_pose_stride != 0 && rest_pose_stride != 72)) return set_error(ANERF_E_SHAPE, "fk_forward: n_poses >= 0, rest_pose_stride 0 or 72");
  if (n_poses == 0) return ANERF_OK;
  if (!bones || !rest_pose) return set_error(ANERF_E_NULL, "fk_forward: NULL pointer");
  hipLaunchKernelGGL(k_fk_fwd, dim3(n_poses), dim3(FKT), 0, (hipStream_t)stream, bones, (int)rot_dim, pelvis,
                     rest_pose, (long long)rest_pose_stride, (int)n_poses, l2ws, skts, rots, kp);
  return check_launch("k_fk_fwd");
}

int anerf_fk_backward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose,
                      int64_t rest_pose_stride, int32_t n_poses, const float* g_skts, const float* g_l2ws,
                      const float* g_kp, const float* g_rots, float* g_bones, float* g_pelvis, void* stream) {
  if (rot_dim != 3 && rot_dim != 6) return set_error(ANERF_E_CONFIG, "fk_backward: rot_dim must be 3 (axis-angle) or 6 (rot6d)");
  if (n_poses < 0 || (rest_pose_stride != 0 && rest_pose_stride != 72)) return set_error(ANERF_E_SHAPE, "fk_backward: n_poses >= 0, rest_pose_stride 0 or 72");
  if (n_poses == 0) return ANERF_OK;
  if (!bones || !rest_pose || !g_bones) return set_error(ANERF_E_NULL, "fk_backward: NULL pointer");
  hipLaunchKernelGGL(k_fk_bwd, dim3(n_poses), dim3(FKT), 0, (hipStream_t)stream, bones, (int)rot_dim, pelvis,
                     rest_pose, (long long)rest_pose_stride, (int)n_poses, g_skts, g_l2ws, g_kp, g_rots, g_bones, g_pelvis);
  return check_launch("k_fk_bwd");
}


int anerf_pose_batch_forward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose, const int64_t* pose_idx,
                             int32_t n_unique, const int32_t* inverse, int32_t n_rays, float* u_kp, float* u_bones, float* u_rots,
                             float* r_kp, float* r_bones, float* r_skts, float* r_l2ws, float* r_rots, void* stream) {
  if (rot_dim != 3 && rot_dim != 6) return set_error(ANERF_E_CONFIG, "pose_batch_forward: rot_dim must be 3 (axis-angle) or 6 (rot6d)");
  if (n_unique < 0 || n_rays < 0 || n_unique > 65535) return set_error(ANERF_E_SHAPE, "pose_batch_forward: 0 <= n_unique <= 65535, n_rays >= 0");
  if (n_unique == 0) return ANERF_OK;
  if (!bones || !rest_pose || !pose_idx || (n_rays > 0 && !inverse)) return set_error(ANERF_E_NULL, "pose_batch_forward: NULL pointer");
  const int C = n_rays > 0 ? (n_rays + RPB - 1) / RPB : 1;
  hipLaunchKernelGGL(k_pose_batch_fwd, dim3(C, n_unique), dim3(PBT), 0, (hipStream_t)stream, bones, (int)rot_dim, pelvis, rest_pose,
                     reinterpret_cast<const long long*>(pose_idx), reinterpret_cast<const int*>(inverse), (int)n_rays, u_kp, u_bones, u_rots,
                     r_kp, r_bones, r_skts, r_l2ws, r_rots);
  return check_launch("k_pose_batch_fwd");
}

int64_t anerf_pose_batch_scratch_size(int32_t n_unique, int32_t n_rays) {
  if (n_unique < 0 || n_rays < 0) return set_error(ANERF_E_SHAPE, "pose_batch_scratch_size: negative sizes");
  const int64_t C = n_rays > 0 ? (n_rays + RPB - 1) / RPB : 1;
  return (int64_t)n_unique * C * (PW + 1) * 4;
}

int anerf_pose_batch_backward(const float* bones, int32_t rot_dim, const float* pelvis, const float* rest_pose, const int64_t* pose_idx,
                              int32_t n_unique, const int32_t* inverse, int32_t n_rays, const float* g_r_kp, const float* g_r_bones,
                              const float* g_r_skts, const float* g_r_l2ws, const float* g_r_rots, const float* g_u_kp,
                              const float* g_u_bones, const float* g_u_rots, float* g_bones, float* g_pelvis, int32_t accumulate,
                              void* scratch, int64_t scratch_bytes, void* stream) {
  if (rot_dim != 3 && rot_dim != 6) return set_error(ANERF_E_CONFIG, "pose_batch_backward: rot_dim must be 3 (axis-angle) or 6 (rot6d)");
  if (n_unique < 0 || n_rays < 0 || n_unique > 65535) return set_error(ANERF_E_SHAPE, "pose_batch_backward: 0 <= n_unique <= 65535, n_rays >= 0");
  if (n_unique == 0) return ANERF_OK;
  if (!bones || !rest_pose || !pose_idx || !g_bones || (n_rays > 0 && !inverse)) return set_error(ANERF_E_NULL, "pose_batch_backward: NULL pointer");
  const bool any_ray = n_rays > 0 && (g_r_kp || g_r_bones || g_r_skts || g_r_l2ws || g_r_rots);
  const int C = n_rays > 0 ? (n_rays + RPB - 1) / RPB : 1;
  float* partial = nullptr;
  int* count = nullptr;
  if (any_ray) {
    if (!scratch || scratch_bytes < anerf_pose_batch_scratch_size(n_unique, n_rays) || ((uintptr_t)scratch & 15))
      return set_error(ANERF_E_WORKSPACE, "pose_batch_backward: scratch of anerf_pose_batch_scratch_size bytes, 16-byte aligned");
    partial = reinterpret_cast<float*>(scratch);
    count = reinterpret_cast<int*>(partial + (int64_t)n_unique * C * PW);
    hipLaunchKernelGGL(k_pose_partial, dim3(C, n_unique), dim3(PBT), 0, (hipStream_t)stream, reinterpret_cast<const int*>(inverse),
                       (int)n_rays, (int)rot_dim, g_r_kp, g_r_bones, g_r_skts, g_r_l2ws, g_r_rots, partial, count);
    const int rc = check_launch("k_pose_partial");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_pose_batch_bwd, dim3(n_unique), dim3(PBT), 0, (hipStream_t)stream, bones, (int)rot_dim, pelvis, rest_pose,
                     reinterpret_cast<const long long*>(pose_idx), C, partial, count, g_u_kp, g_u_bones, g_u_rots, g_bones, g_pelvis,
                     (int)(accumulate != 0));
  return check_launch("k_pose_batch_bwd");
}

}  // extern "C"
