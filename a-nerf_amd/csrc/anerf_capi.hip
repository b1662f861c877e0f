// anerf_capi.hip -- the extern "C" boundary of libanerf_hip.so (declared in include/anerf.h).
// Host-side only: argument checking, weight-image layout / pack tables, kernel dispatch.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "anerf_dev.h"
#include "anerf_gemm.h"

namespace anerf {

static thread_local char g_err[256] = "";

int set_error(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return ANERF_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return ANERF_E_LAUNCH;
}

// launchers defined in the kernel translation units
struct MlpArgs;
int launch_pack(const AnerfNetParams*, const int32_t*, long long, float*, hipStream_t);
int launch_ray_bounds(const float*, int, const float*, int, int, float*, float*, hipStream_t);
int bounds_z_max_rays();
int launch_bounds_z(const float*, int, const float*, int, int, int, const float*, int, float*, hipStream_t);
int launch_coarse_z(const float*, const float*, const float*, int, int, int, const float*, int, float*, float*,
                    hipStream_t);
int launch_composite(const AnerfConfig*, const float*, const float*, const float*, int, const float*, int, int, float*,
                     float*, float*, float*, float*, float*, hipStream_t);
int launch_importance(const float*, const float*, int, int, int, const float*, int, float*, float*, long long*,
                      hipStream_t);
int launch_composite_bwd(const AnerfConfig*, const float*, const float*, const float*, int, const float*, int, int,
                         const float*, const float*, const float*, const float*, const float*, float*, hipStream_t);
int mlp_raw_entry(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays, int ray_stride,
                  const float* z, const float* skts, long long skt_stride, const float* cam, const float* codes,
                  int n_codes, float tau_v, float tau_d, const float* cut_v, const float* cut_d, const float* x,
                  int x_width, long long P, int N, int S, int nstages, float* raw, bool pre, const AnerfSaved* sv,
                  hipStream_t st, const float* pnoise = nullptr, const float* tau_dev = nullptr);
int mlp_bwd_entry(const float* packed_t, const float* aux, const float* draw, const AnerfSaved* sv, float* dz, float* df,
                  float* dzv, long long P, int nstages, hipStream_t st);
int launch_gather_raw(const float* raw_c, const float* raw_is, const long long* idx, int n, int S, int Ni, float* out, hipStream_t st);
int mlp_bwd_in_b3_entry(const float* packed_i, const float* dz, const float* dzv, float* dx, float* du, long long P,
                        long long Ppad, int nstages, int uw, hipStream_t st);
int mlp_bwd_b3_entry(const float* packed_t, const float* aux, const float* draw, const AnerfSaved* sv, float* dz, float* df,
                     float* dzv, long long P, int nstages, hipStream_t st);
int mlp_b3_entry(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays, int ray_stride,
                 const float* z, const float* skts, long long skt_stride, const float* cam, const float* codes, int n_codes,
                 float tau_v, float tau_d, const float* cut_v, const float* cut_d, long long P, int N, int S, int nstages,
                 float* raw, const AnerfSaved* sv, hipStream_t st, const float* pnoise = nullptr, const float* tau_dev = nullptr);
int launch_pack_b3(const AnerfNetParams* P, const int32_t* table, long long n, void* out, hipStream_t st);
int mlp_density_entry(const float* packed, const float* aux, const float* pts, const float* skts, float tau_v,
                      const float* cut_v, long long P, int nstages_trunk, float* sigma, hipStream_t st, int gate_bones);
int launch_gen_rays(int W, int x0, int y0, int bw, int bh, float fx, float fy, float cx, float cy, const float* c2w,
                    float near, float far, float* ray_batch, long long* valid_idx, hipStream_t st);
int launch_assemble(const float* rgb, const float* acc, const float* disp, const long long* valid_idx, int n, float* rgb_img,
                    float* disp_img, float* acc_img, hipStream_t st);
int mlp_bwd_in_entry(const float* packed_i, const float* dz, const float* dzv, float* dx, float* du, long long P,
                     long long Ppad, int nstages, int uw, hipStream_t st);
int launch_encode_bwd(int ld, const float* dx, const float* du, int uw, const float* rays, int ray_stride, const float* z,
                      const float* skts, long long skt_stride, float tau_v, float tau_d, const float* cut_v,
                      const float* cut_d, int n, int S, float* dY, float* dQ, float* dskts, bool accumulate, hipStream_t st,
                      const float* pnoise = nullptr, int gate_bones = 0, const float* tau_dev = nullptr);
int launch_pose_reduce(const float* dY, const float* dQ, const float* rays, int ray_stride, const float* z, int n, int S, float* dskts,
                       bool accumulate, hipStream_t st, const float* pnoise);
int mlp_bwd_in_enc_entry(int ld, int code, const float* packed_i, const float* dz, const float* dzv, float* du, long long P, long long Ppad,
                         int nstages, const float* rays, int ray_stride, const float* z, const float* skts, long long skt_stride,
                         float tau_v, float tau_d, const float* cut_v, const float* cut_d, int S, float* dY, float* dQ, const float* pnoise,
                         int gate_bones, const float* tau_dev, int n_rays, hipStream_t st);
int launch_gather_rows3(const float* a, const float* b, const long long* idx, int n, int S, int Ni, float* out, hipStream_t st);
int launch_code_reduce(const float* du, int uw, const float* cam, int n, int S, int n_codes, float* rowsum, float* dcodes,
                       hipStream_t st);

// ------------------------------------------------------------------------------------------------
// weight-stream layout (fp32 images, which = 0 / 1 / 2).  A segment = one Linear layer, padded to whole 32-fragment
// stages; a k-group = 8 contraction indices (4 per lane half = 4 MFMA k-steps) against NB 32-row output blocks = NB
// fragments of 64 lanes x float4, laid out QUARTER-major: fragment j = q * (NB / 4) + g of a k-group holds, for lane l,
// the four A operands of k-step q for the output blocks nb = 4g .. 4g+3:
//     { W[32(4g+e) + (l & 31)][col(kg, l >> 5, q)] : e = 0..3 }
// so the registers a `ds_read_b128` fills are all consumed by ONE quarter of the k-group's MFMAs (k-step q, blocks
// 4g..4g+3) and die together: the next k-group's fragments can be fetched a quarter at a time, three quarters (1536 MFMA
// cycles) ahead of their use, without a second fragment buffer.  (Round 1 stored [kg][nb][lane][k-step]: a fragment's four
// registers stayed live until the LAST quarter, every read for the next k-group had to wait for it, and the reads'
// latency -- four lock-step waves x 8 KiB -- was exposed at each of the 420 k-groups of a tile.)
// ------------------------------------------------------------------------------------------------
// element offset of (block nb, lane, k-step q) inside a k-group of NB blocks
static inline int64_t frag_pos(int NB, int nb, int lane, int q) {
  return ((int64_t)(q * (NB / 4) + (nb >> 2)) * 64 + lane) * 4 + (nb & 3);
}
struct Seg {
  int tensor;  // index into AnerfNetParams.w
  int K;       // torch in_features (row length)
  int NB;      // out_features / 32
  int nkg;     // k-groups
  int kind;    // 0 hidden (natural), 1 pts0, 2 pts5, 3 views
};

static bool config_ok(const AnerfConfig* c) {
  return c && c->n_joints == 24 && c->multires == 7 && (c->multires_views == 4 || c->multires_views == 0) &&
         (c->framecode_ch == 0 || c->framecode_ch == 16) && c->netdepth == 8 && c->netwidth == 256 && c->skip == 4 &&
         !(c->multires_views == 0 && c->framecode_ch != 0) && c->density_scale > 0.f;
}

static int dim_v(const AnerfConfig* c) { return 24 * (1 + 2 * c->multires); }
static int dim_x(const AnerfConfig* c) { return dim_v(c) + 72; }
static int dim_d(const AnerfConfig* c) { return 72 * (1 + 2 * c->multires_views); }

static std::vector<Seg> fwd_segments(const AnerfConfig* c) {
  std::vector<Seg> s;
  const int kx = dim_x(c) / 8;
  s.push_back({0, dim_x(c), 8, kx, 1});
  for (int i = 1; i <= 4; ++i) s.push_back({i, 256, 8, 32, 0});
  s.push_back({5, dim_x(c) + 256, 8, kx + 32, 2});
  s.push_back({6, 256, 8, 32, 0});
  s.push_back({7, 256, 8, 32, 0});
  s.push_back({9, 256, 8, 32, 0});
  const int kv = 256 + dim_d(c) + c->framecode_ch;
  s.push_back({10, kv, 4, kv / 8, 3});
  return s;
}

static int seg_stages(const Seg& s) { return (s.nkg * s.NB + STAGE_FRAGS - 1) / STAGE_FRAGS; }

// input column read by lane half h, slot t of k-group kg
static int seg_col(const AnerfConfig* c, const Seg& s, int kg, int h, int t) {
  auto owned = [&](int base, int g) {  // permuted (joint-owned) 3-vector channels: flat = 4g+t -> (a, comp)
    const int flat = 4 * g + t, a = flat / 3, comp = flat % 3;
    return base + 3 * (8 * (a >> 2) + 4 * h + (a & 3)) + comp;
  };
  const int kv = dim_v(c) / 8, kx = dim_x(c) / 8;
  switch (s.kind) {
    case 0: return 8 * kg + 4 * h + t;
    case 1:
    case 2:
      if (kg < kv) return 8 * kg + 4 * h + t;
      if (kg < kx) return owned(dim_v(c), kg - kv);
      return dim_x(c) + 8 * (kg - kx) + 4 * h + t;
    case 3: {
      if (kg < 32) return 8 * kg + 4 * h + t;
      const int nd = dim_d(c) / 8;
      if (kg < 32 + nd) {
        const int b = (kg - 32) / 9, g = (kg - 32) % 9;
        return owned(256 + 72 * b, g);
      }
      return 256 + dim_d(c) + 8 * (kg - 32 - nd) + 4 * h + t;
    }
  }
  return -1;
}

// backward-data image: W^T of (views feature columns, feature, pts 7,6,5[hidden cols],4,3,2,1), all 8 output blocks
struct BSeg { int tensor, K, c0, ncontract; };
static std::vector<BSeg> bwd_segments(const AnerfConfig* c) {
  const int kv = 256 + dim_d(c) + c->framecode_ch;
  std::vector<BSeg> s;
  s.push_back({10, kv, 0, 128});
  s.push_back({9, 256, 0, 256});
  s.push_back({7, 256, 0, 256});
  s.push_back({6, 256, 0, 256});
  s.push_back({5, dim_x(c) + 256, dim_x(c), 256});
  for (int l = 4; l >= 1; --l) s.push_back({l, 256, 0, 256});
  return s;
}
// ---- bf16x3 forward image (which = 3): k-steps of 16 input columns, (hi, lo) fragment pair per 32-row block
static int nu_pad(const AnerfConfig* c) { return (36 * (1 + 2 * c->multires_views) + c->framecode_ch / 2 + 7) / 8 * 8; }
static int b3_ksteps(const AnerfConfig* c, const Seg& s) {
  switch (s.kind) {
    case 0: return 16;
    case 1: return 27;
    case 2: return 27 + 16;
    default: return 16 + nu_pad(c) / 8;
  }
}
static int b3_stages(const AnerfConfig* c, const Seg& s) { return (b3_ksteps(c, s) * s.NB * 2 + STAGE_FRAGS - 1) / STAGE_FRAGS; }
// input column supplied by lane half hh, element e of k-step ks (-1: zero padding)
static int b3_col(const AnerfConfig* c, const Seg& s, int ks, int hh, int e) {
  auto own = [&](int a) { return 8 * (a >> 2) + 4 * hh + (a & 3); };
  auto hid = [&](int k) { return 16 * k + (e & 3) + 8 * (e >> 2) + 4 * hh; };
  auto xcol = [&](int k) {
    const int idx = 8 * k + e;
    if (idx < 180) return 24 * (idx / 12) + own(idx % 12);
    const int i = idx - 180;
    return dim_v(c) + 3 * own(i / 3) + i % 3;
  };
  switch (s.kind) {
    case 0: return hid(ks);
    case 1: return xcol(ks);
    case 2: return ks < 27 ? xcol(ks) : dim_x(c) + hid(ks - 27);
    default: {
      if (ks < 16) return hid(ks);
      const int idx = 8 * (ks - 16) + e, nband = 1 + 2 * c->multires_views;
      if (idx < 36 * nband) {
        const int b = idx / 36, i = idx % 36;
        return 256 + 72 * b + 3 * own(i / 3) + i % 3;
      }
      const int j = idx - 36 * nband;
      if (j < c->framecode_ch / 2) return 256 + dim_d(c) + 8 * hh + j;
      return -1;
    }
  }
}

static int bseg_stages(const BSeg& s) { return (s.ncontract / 8) * 8 / STAGE_FRAGS; }
static int u_width(const AnerfConfig* c) { return dim_d(c) + c->framecode_ch; }

}  // namespace anerf

using namespace anerf;

extern "C" {

const char* anerf_last_error(void) { return g_err; }
int anerf_version(void) { return 7; }

int anerf_layout(const AnerfConfig* cfg, int which, AnerfLayout* out) {
  if (!out) return set_error(ANERF_E_NULL, "out is NULL");
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (which < 0 || which > 5)
    return set_error(ANERF_E_CONFIG, "which must be 0 (W), 1 (W^T), 2 (input-gradient image), 3 (bf16x3 W), 4 (bf16x3 W^T) or "
                                     "5 (bf16x3 input-gradient image)");
  int stages = 0;
  if (which == 0)
    for (const Seg& s : fwd_segments(cfg)) stages += seg_stages(s);
  else if (which == 1)
    for (const BSeg& s : bwd_segments(cfg)) stages += bseg_stages(s);
  else if (which == 2 || which == 5)
    stages = 2 * (8 + 8) + ((u_width(cfg) + 255) / 256) * 4;
  else if (which == 4)
    for (const BSeg& s : bwd_segments(cfg)) stages += bseg_stages(s);   // same bytes per k as fp32: (hi, lo) bf16 = 4 B
  else
    for (const Seg& s : fwd_segments(cfg)) stages += b3_stages(cfg, s);
  out->n_stages = stages;
  out->stream_floats = (int64_t)stages * STAGE_FLOATS;
  out->aux_floats = AUX_FLOATS;
  out->x_width = dim_x(cfg) + dim_d(cfg) + (cfg->framecode_ch ? 1 : 0);
  return ANERF_OK;
}

int anerf_build_pack_table(const AnerfConfig* cfg, int which, int32_t* table) {
  AnerfLayout L;
  const int rc = anerf_layout(cfg, which, &L);
  if (rc) return rc;
  if (!table) return set_error(ANERF_E_NULL, "table is NULL");
  const int64_t n_stream_entries = which >= 3 ? 2 * L.stream_floats : L.stream_floats;   // bf16x3 images: one per bf16 element
  for (int64_t i = 0; i < n_stream_entries + L.aux_floats; ++i) table[i] = -1;
  int64_t pos = 0;
  if (which == 3) {
    for (const Seg& s : fwd_segments(cfg)) {
      for (int ks = 0; ks < b3_ksteps(cfg, s); ++ks)
        for (int nb = 0; nb < s.NB; ++nb)
          for (int part = 0; part < 2; ++part)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const int col = b3_col(cfg, s, ks, lane >> 5, e);
                if (col < 0) continue;
                const int n = 32 * nb + (lane & 31);
                table[pos + ((int64_t)((ks * s.NB + nb) * 2 + part) * 64 + lane) * 8 + e] =
                    (part << 29) | (s.tensor << 24) | (n * s.K + col);
              }
      pos += (int64_t)b3_stages(cfg, s) * STAGE_FRAGS * 512;
    }
  }
  if (which == 4) {
    // W^T in bf16x3 fragments: k-step ks contracts 16 rows n of W, supplied in the accumulator order of the split-bf16
    // kernels (lane half hh, element e -> n = 16 ks + (e & 3) + 8 (e >> 2) + 4 hh); block nb produces columns
    // c0 + 32 nb + (lane & 31) of W
    for (const BSeg& s : bwd_segments(cfg)) {
      for (int ks = 0; ks < s.ncontract / 16; ++ks)
        for (int nb = 0; nb < 8; ++nb)
          for (int part = 0; part < 2; ++part)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const int n = 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int col = s.c0 + 32 * nb + (lane & 31);
                table[pos + ((int64_t)((ks * 8 + nb) * 2 + part) * 64 + lane) * 8 + e] =
                    (part << 29) | (s.tensor << 24) | (n * s.K + col);
              }
      pos += (int64_t)bseg_stages(s) * STAGE_FRAGS * 512;
    }
  }
  if (which == 1) {
    for (const BSeg& s : bwd_segments(cfg)) {
      for (int kg = 0; kg < s.ncontract / 8; ++kg)
        for (int nb = 0; nb < 8; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t) {
              const int n = 8 * kg + 4 * (lane >> 5) + t;          // contraction index = row of W
              const int col = s.c0 + 32 * nb + (lane & 31);        // produced index   = column of W
              table[pos + (int64_t)kg * 8 * FRAG_FLOATS + frag_pos(8, nb, lane, t)] = (s.tensor << 24) | (n * s.K + col);
            }
      pos += (int64_t)bseg_stages(s) * STAGE_FLOATS;
    }
  }
  if (which == 5) {
    // the which=2 image as bf16x3 fragments: k-steps of 16 contraction rows in the accumulator order of the split-bf16
    // kernels; same stream-column (perm) semantics of the produced columns, so k_encode_bwd is unchanged
    std::vector<int32_t> px(dim_x(cfg)), pu(u_width(cfg));
    anerf_build_perm_tables(cfg, px.data(), pu.data());
    auto emit3 = [&](int tensor, int K, int colbase, const std::vector<int32_t>& perm, int gi, int nks) {
      for (int ks = 0; ks < nks; ++ks)
        for (int nb = 0; nb < 8; ++nb)
          for (int part = 0; part < 2; ++part)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 8; ++e) {
                const int n = 16 * ks + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int cs = 256 * gi + 32 * nb + (lane & 31);     // stream column
                if (cs < (int)perm.size())
                  table[pos + ((int64_t)((ks * 8 + nb) * 2 + part) * 64 + lane) * 8 + e] =
                      (part << 29) | (tensor << 24) | (n * K + colbase + perm[cs]);
              }
      pos += (int64_t)(nks * 16 / STAGE_FRAGS) * STAGE_FRAGS * 512;
    };
    for (int gi = 0; gi < 2; ++gi) {
      emit3(0, dim_x(cfg), 0, px, gi, 16);
      emit3(5, dim_x(cfg) + 256, 0, px, gi, 16);
    }
    for (int gi = 0; gi < (u_width(cfg) + 255) / 256; ++gi) emit3(10, 256 + u_width(cfg), 256, pu, gi, 8);
  }
  if (which == 2) {
    // [x columns 256*gi .. +255 of W0^T (32 kg) then of W5^T (32 kg)] for gi = 0,1; then [u columns of Wv^T (16 kg)]
    std::vector<int32_t> px(dim_x(cfg)), pu(u_width(cfg));
    anerf_build_perm_tables(cfg, px.data(), pu.data());
    auto emit = [&](int tensor, int K, int colbase, const std::vector<int32_t>& perm, int gi, int nkg) {
      for (int kg = 0; kg < nkg; ++kg)
        for (int nb = 0; nb < 8; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int t = 0; t < 4; ++t) {
              const int n = 8 * kg + 4 * (lane >> 5) + t;
              const int cs = 256 * gi + 32 * nb + (lane & 31);     // stream column
              if (cs < (int)perm.size())
                table[pos + (int64_t)kg * 8 * FRAG_FLOATS + frag_pos(8, nb, lane, t)] = (tensor << 24) | (n * K + colbase + perm[cs]);
            }
      pos += (int64_t)(nkg * 8 / STAGE_FRAGS) * STAGE_FLOATS;
    };
    for (int gi = 0; gi < 2; ++gi) {
      emit(0, dim_x(cfg), 0, px, gi, 32);
      emit(5, dim_x(cfg) + 256, 0, px, gi, 32);
    }
    for (int gi = 0; gi < (u_width(cfg) + 255) / 256; ++gi) emit(10, 256 + u_width(cfg), 256, pu, gi, 16);
  }
  for (const Seg& s : fwd_segments(cfg)) {
    if (which != 0) break;
    for (int kg = 0; kg < s.nkg; ++kg)
      for (int nb = 0; nb < s.NB; ++nb)
        for (int lane = 0; lane < 64; ++lane)
          for (int t = 0; t < 4; ++t) {
            const int n = 32 * nb + (lane & 31), h = lane >> 5;
            const int col = seg_col(cfg, s, kg, h, t);
            table[pos + (int64_t)kg * s.NB * FRAG_FLOATS + frag_pos(s.NB, nb, lane, t)] = (s.tensor << 24) | (n * s.K + col);
          }
    pos += (int64_t)seg_stages(s) * STAGE_FLOATS;
  }
  int32_t* aux = table + n_stream_entries;
  for (int i = 0; i < 8; ++i)
    for (int n = 0; n < 256; ++n) aux[AUX_B0 + 256 * i + n] = ((12 + i) << 24) | n;
  for (int n = 0; n < 256; ++n) aux[AUX_BF + n] = ((12 + 9) << 24) | n;
  for (int n = 0; n < 128; ++n) aux[AUX_BV + n] = ((12 + 10) << 24) | n;
  for (int n = 0; n < 256; ++n) aux[AUX_WA + n] = (8 << 24) | n;
  aux[AUX_BA] = ((12 + 8) << 24) | 0;
  for (int n = 0; n < 384; ++n) aux[AUX_WC + n] = (11 << 24) | n;
  for (int n = 0; n < 3; ++n) aux[AUX_BC + n] = ((12 + 11) << 24) | n;
  return ANERF_OK;
}

int anerf_pack_params(const AnerfNetParams* params, const int32_t* table, int64_t n, float* out, void* stream) {
  if (!params || !table || !out) return set_error(ANERF_E_NULL, "pack: NULL pointer");
  for (int i = 0; i < 12; ++i)
    if (!params->w[i] || !params->b[i]) return set_error(ANERF_E_NULL, "pack: NULL tensor");
  if (!sched_ok(*params)) return set_error(ANERF_E_SHAPE, "pack: sched_dim_x must be 432 and sched_dim_u 72 / 648 / 664 (the encoded widths of the supported layouts; trunk width 256)");
  return launch_pack(params, table, n, out, (hipStream_t)stream);
}

int anerf_pack_params_b3(const AnerfNetParams* params, const int32_t* table, int64_t stream_floats, int64_t aux_floats,
                         float* out, void* stream) {
  if (!params || !table || !out) return set_error(ANERF_E_NULL, "pack_b3: NULL pointer");
  for (int i = 0; i < 12; ++i)
    if (!params->w[i] || !params->b[i]) return set_error(ANERF_E_NULL, "pack_b3: NULL tensor");
  if (!sched_ok(*params)) return set_error(ANERF_E_SHAPE, "pack_b3: sched_dim_x must be 432 and sched_dim_u 72 / 648 / 664 (the encoded widths of the supported layouts; trunk width 256)");
  int rc = launch_pack_b3(params, table, 2 * stream_floats, out, (hipStream_t)stream);
  if (rc) return rc;
  return launch_pack(params, table + 2 * stream_floats, aux_floats, out + stream_floats, (hipStream_t)stream);
}

int anerf_mlp_raw_b3(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays, int32_t ray_stride,
                     const float* z_vals, const float* skts, int64_t skt_ray_stride, const float* cam_idx,
                     const float* codes, int32_t n_codes, float tau_v, float tau_d, const float* cutoff_v,
                     const float* cutoff_d, int32_t n_rays, int32_t n_samples, float* raw, void* stream) {
  if (n_rays == 0) return ANERF_OK;
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 3, &L);
  if (rc) return rc;
  if (!packed || !aux || !rays || !z_vals || !skts || !cutoff_v || !cutoff_d || !raw)
    return set_error(ANERF_E_NULL, "mlp_raw_b3: NULL pointer");
  if (cfg->framecode_ch && (!cam_idx || !codes || n_codes < 1)) return set_error(ANERF_E_NULL, "mlp_raw_b3: frame codes");
  if (n_samples < MIN_SAMPLES || n_samples > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "mlp_raw_b3: 8 <= samples <= 512");
  if (skt_ray_stride != 0 && skt_ray_stride != 384) return set_error(ANERF_E_SHAPE, "mlp_raw_b3: skt_ray_stride 0|384");
  if (ray_stride < 6) return set_error(ANERF_E_SHAPE, "mlp_raw_b3: ray_stride >= 6");
  return mlp_b3_entry(cfg, packed, aux, rays, ray_stride, z_vals, skts, skt_ray_stride, cam_idx, codes, n_codes, tau_v, tau_d,
                      cutoff_v, cutoff_d, (long long)n_rays * n_samples, n_rays, n_samples, L.n_stages, raw, nullptr,
                      (hipStream_t)stream);
}

static int ray_bounds_checked(const float* rays, int32_t ray_stride, const float* cyls, int cyl_stride, int32_t n_rays, float* near_far,
                              float* stats_ws, void* stream) {
  if (n_rays == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  if (!rays || !cyls || !near_far || !stats_ws) return set_error(ANERF_E_NULL, "ray_bounds: NULL pointer");
  if ((uintptr_t)stats_ws & 7) return set_error(ANERF_E_WORKSPACE, "ray_bounds: stats_ws must be 8-byte aligned");
  if (ray_stride < 8 || n_rays < 0) return set_error(ANERF_E_SHAPE, "ray_bounds: ray_stride >= 8 required");
  return launch_ray_bounds(rays, ray_stride, cyls, cyl_stride, n_rays, near_far, stats_ws, (hipStream_t)stream);
}

int anerf_ray_bounds(const float* rays, int32_t ray_stride, const float* cyls, int32_t n_rays, float* near_far,
                     float* stats_ws, void* stream) {
  return ray_bounds_checked(rays, ray_stride, cyls, 5, n_rays, near_far, stats_ws, stream);
}

int anerf_coarse_z(const float* near_far, const float* stats_ws, const float* rays, int32_t ray_stride,
                    int32_t n_rays, int32_t n_samples, const float* t_rand, int32_t lindisp, float* z_vals,
                    float* near_far_fixed, void* stream) {
  if (n_rays == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  if (!near_far || !stats_ws || !rays || !z_vals) return set_error(ANERF_E_NULL, "coarse_z: NULL pointer");
  if (n_samples < 2 || n_samples > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "coarse_z: 2 <= N_samples <= 512");
  if (n_rays == 0) return ANERF_OK;
  return launch_coarse_z(near_far, stats_ws, rays, ray_stride, n_rays, n_samples, t_rand, lindisp, z_vals,
                         near_far_fixed, (hipStream_t)stream);
}

int anerf_mlp_raw(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays,
                  int32_t ray_stride, const float* z_vals, const float* skts, int64_t skt_ray_stride,
                  const float* cam_idx, const float* codes, int32_t n_codes, float tau_v, float tau_d,
                  const float* cutoff_v, const float* cutoff_d, int32_t n_rays, int32_t n_samples, float* raw,
                  void* stream) {
  if (n_rays == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 0, &L);
  if (rc) return rc;
  if (!packed || !aux || !rays || !z_vals || !skts || !cutoff_v || !cutoff_d || !raw)
    return set_error(ANERF_E_NULL, "mlp_raw: NULL pointer");
  if (cfg->framecode_ch && (!cam_idx || !codes || n_codes < 1)) return set_error(ANERF_E_NULL, "mlp_raw: frame codes");
  if (n_samples < MIN_SAMPLES || n_samples > MAX_SAMPLES)
    return set_error(ANERF_E_SHAPE, "mlp_raw: 8 <= samples per ray <= 512");
  if (skt_ray_stride != 0 && skt_ray_stride != 384) return set_error(ANERF_E_SHAPE, "mlp_raw: skt_ray_stride 0|384");
  if (ray_stride < 6) return set_error(ANERF_E_SHAPE, "mlp_raw: ray_stride >= 6");
  return mlp_raw_entry(cfg, packed, aux, rays, ray_stride, z_vals, skts, skt_ray_stride, cam_idx, codes, n_codes,
                       tau_v, tau_d, cutoff_v, cutoff_d, nullptr, 0, (long long)n_rays * n_samples, n_rays, n_samples,
                       L.n_stages, raw, false, nullptr, (hipStream_t)stream);
}

int anerf_mlp_forward(const AnerfConfig* cfg, const float* packed, const float* aux, const float* x,
                      int64_t n_points, const float* codes, int32_t n_codes, float* raw, void* stream) {
  if (n_points == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 0, &L);
  if (rc) return rc;
  if (!packed || !aux || !x || !raw) return set_error(ANERF_E_NULL, "mlp_forward: NULL pointer");
  if (cfg->framecode_ch && (!codes || n_codes < 1)) return set_error(ANERF_E_NULL, "mlp_forward: frame codes");
  if (n_points < 0) return set_error(ANERF_E_SHAPE, "mlp_forward: n_points < 0");
  return mlp_raw_entry(cfg, packed, aux, nullptr, 0, nullptr, nullptr, 0, nullptr, codes, n_codes, 0.f, 0.f, nullptr,
                       nullptr, x, L.x_width, n_points, 0, 1, L.n_stages, raw, true, nullptr, (hipStream_t)stream);
}

int anerf_composite(const AnerfConfig* cfg, const float* raw, const float* z_vals, const float* rays,
                    int32_t ray_stride, const float* noise, int32_t n_rays, int32_t n_samples, float* rgb_map,
                    float* disp_map, float* acc_map, float* weights, float* alpha, float* depth_map, void* stream) {
  if (n_rays == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  if (!cfg || !raw || !z_vals || !rays || !rgb_map || !disp_map || !acc_map || !weights || !alpha)
    return set_error(ANERF_E_NULL, "composite: NULL pointer");
  if (n_samples < 1 || n_samples > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "composite: 1 <= samples <= 512");
  if (cfg->density_act != 0 && cfg->density_act != 1) return set_error(ANERF_E_CONFIG, "composite: density_act");
  if (n_rays == 0) return ANERF_OK;
  return launch_composite(cfg, raw, z_vals, rays, ray_stride, noise, n_rays, n_samples, rgb_map, disp_map, acc_map,
                          weights, alpha, depth_map, (hipStream_t)stream);
}

int anerf_importance(const float* z_vals, const float* weights, int32_t n_rays, int32_t n_samples,
                     int32_t n_importance, const float* u, int32_t single_net, float* z_samples, float* z_merged,
                     int64_t* sorted_idx, void* stream) {
  if (n_rays == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  if (!z_vals || !weights || !z_samples || !z_merged) return set_error(ANERF_E_NULL, "importance: NULL pointer");
  if (n_samples < 3 || n_importance < 1 || n_samples + n_importance > MAX_SAMPLES)
    return set_error(ANERF_E_SHAPE, "importance: need S >= 3, Ni >= 1, S + Ni <= 512");
  if (n_rays == 0) return ANERF_OK;
  return launch_importance(z_vals, weights, n_rays, n_samples, n_importance, u, single_net, z_samples, z_merged,
                           (long long*)sorted_idx, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// training path
// ---------------------------------------------------------------------------------------------------------------
// Number of blocks (4 x 128x128 wave tiles each) of the weight-gradient GEMM; the alpha / rgb head problems ride in two of them.
static void gemm_block_counts(const AnerfConfig* cfg, int* nheavy) {
  const int tx = (dim_x(cfg) + 127) / 128;                       // column tiles of X'
  const int tv = 2 + (u_width(cfg) + 127) / 128;                 // views layer: feature (2 tiles) + U' tiles, M = 128
  // M = 256 problems pair their column tiles 2 x 2:  L0 (X'), L1-4, L5 (X' and h4), L6, L7, feature
  *nheavy = (tx + 1) / 2 + 4 + (tx + 1) / 2 + 1 + 2 + 1 + (tv + 3) / 4;
}

int anerf_train_layout(const AnerfConfig* cfg, int64_t n_points, AnerfTrainLayout* out) {
  if (!out) return set_error(ANERF_E_NULL, "out is NULL");
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (n_points < 0) return set_error(ANERF_E_SHAPE, "n_points < 0");
  out->p_pad = (n_points + 127) / 128 * 128;
  if (out->p_pad == 0) out->p_pad = 128;
  out->x_width = dim_x(cfg);
  out->u_width = u_width(cfg);
  int nh, rh, ch;
  gemm_block_counts(cfg, &nh);
  gemm_plan_rows(out->p_pad, nh, &rh, &ch);
  out->gemm_chunks = ch;
  // partials: heavy problems (M*N + bias M) x chunks; head problems (alpha 1 x 256 in 2 row slots, rgb 3 x 128 in 4) likewise
  const int64_t heavy = 256LL * dim_x(cfg) + 256 + 6 * (256LL * 256 + 256) + 256LL * dim_x(cfg) + 256 + 256LL * 256 +
                        (256LL * 256 + 256) + (128LL * 256 + 128) + 128LL * u_width(cfg);
  const int64_t heads = 4 * (3LL * 128 + 3) + 2 * (1LL * 256 + 1);
  out->gemm_ws_floats = (heavy + heads) * ch;
  return ANERF_OK;
}

int anerf_build_perm_tables(const AnerfConfig* cfg, int32_t* perm_x, int32_t* perm_u) {
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (!perm_x || !perm_u) return set_error(ANERF_E_NULL, "perm tables NULL");
  const std::vector<Seg> segs = fwd_segments(cfg);
  const Seg& s0 = segs.front();
  const Seg& sv = segs.back();
  for (int kg = 0; kg < dim_x(cfg) / 8; ++kg)
    for (int h = 0; h < 2; ++h)
      for (int t = 0; t < 4; ++t) perm_x[8 * kg + 4 * h + t] = seg_col(cfg, s0, kg, h, t);
  for (int kg = 0; kg < u_width(cfg) / 8; ++kg)
    for (int h = 0; h < 2; ++h)
      for (int t = 0; t < 4; ++t) perm_u[8 * kg + 4 * h + t] = seg_col(cfg, sv, 32 + kg, h, t) - 256;
  return ANERF_OK;
}

static int saved_ok(const AnerfSaved* s) {
  return s && s->h && s->f && s->g && s->x && s->u && s->p_pad > 0 && s->p_pad % 128 == 0;
}

int anerf_mlp_raw_train(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays,
                        int32_t ray_stride, const float* z_vals, const float* skts, int64_t skt_ray_stride,
                        const float* cam_idx, const float* codes, int32_t n_codes, float tau_v, float tau_d,
                        const float* cutoff_v, const float* cutoff_d, int32_t n_rays, int32_t n_samples, float* raw,
                        const AnerfSaved* saved, void* stream) {
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 0, &L);
  if (rc) return rc;
  if (!packed || !aux || !rays || !z_vals || !skts || !cutoff_v || !cutoff_d || !raw)
    return set_error(ANERF_E_NULL, "mlp_raw_train: NULL pointer");
  if (!saved_ok(saved) || saved->p_pad < (int64_t)n_rays * n_samples)
    return set_error(ANERF_E_WORKSPACE, "mlp_raw_train: AnerfSaved planes missing or too small");
  if (cfg->framecode_ch && (!cam_idx || !codes || n_codes < 1)) return set_error(ANERF_E_NULL, "mlp_raw_train: frame codes");
  if (n_samples < MIN_SAMPLES || n_samples > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "mlp_raw_train: 8 <= samples <= 512");
  if (skt_ray_stride != 0 && skt_ray_stride != 384) return set_error(ANERF_E_SHAPE, "mlp_raw_train: skt_ray_stride 0|384");
  return mlp_raw_entry(cfg, packed, aux, rays, ray_stride, z_vals, skts, skt_ray_stride, cam_idx, codes, n_codes, tau_v,
                       tau_d, cutoff_v, cutoff_d, nullptr, 0, (long long)n_rays * n_samples, n_rays, n_samples,
                       L.n_stages, raw, false, saved, (hipStream_t)stream);
}

int anerf_build_perm_tables_b3(const AnerfConfig* cfg, int32_t* perm_x, int32_t* perm_u) {
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (!perm_x || !perm_u) return set_error(ANERF_E_NULL, "perm tables NULL");
  const std::vector<Seg> segs = fwd_segments(cfg);
  const Seg& s0 = segs.front();
  const Seg& sv = segs.back();
  const int nx = dim_x(cfg) / 2, nu = u_width(cfg) / 2;            // values per lane half
  for (int h = 0; h < 2; ++h) {
    for (int i = 0; i < nx; ++i) perm_x[h * nx + i] = b3_col(cfg, s0, i / 8, h, i % 8);
    for (int i = 0; i < nu; ++i) perm_u[h * nu + i] = b3_col(cfg, sv, 16 + i / 8, h, i % 8) - 256;
  }
  return ANERF_OK;
}

int anerf_mlp_raw_train_b3(const AnerfConfig* cfg, const float* packed, const float* aux, const float* rays,
                           int32_t ray_stride, const float* z_vals, const float* skts, int64_t skt_ray_stride,
                           const float* cam_idx, const float* codes, int32_t n_codes, float tau_v, float tau_d,
                           const float* cutoff_v, const float* cutoff_d, int32_t n_rays, int32_t n_samples, float* raw,
                           const AnerfSaved* saved, void* stream) {
  if (n_rays == 0) return ANERF_OK;
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 3, &L);
  if (rc) return rc;
  if (!packed || !aux || !rays || !z_vals || !skts || !cutoff_v || !cutoff_d || !raw)
    return set_error(ANERF_E_NULL, "mlp_raw_train_b3: NULL pointer");
  if (cfg->framecode_ch && (!cam_idx || !codes || n_codes < 1)) return set_error(ANERF_E_NULL, "mlp_raw_train_b3: frame codes");
  if (n_samples < MIN_SAMPLES || n_samples > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "mlp_raw_train_b3: 8 <= samples <= 512");
  if (skt_ray_stride != 0 && skt_ray_stride != 384) return set_error(ANERF_E_SHAPE, "mlp_raw_train_b3: skt_ray_stride 0|384");
  if (ray_stride < 6) return set_error(ANERF_E_SHAPE, "mlp_raw_train_b3: ray_stride >= 6");
  if (!saved_ok(saved) || saved->p_pad < (int64_t)n_rays * n_samples) return set_error(ANERF_E_WORKSPACE, "mlp_raw_train_b3: AnerfSaved");
  return mlp_b3_entry(cfg, packed, aux, rays, ray_stride, z_vals, skts, skt_ray_stride, cam_idx, codes, n_codes, tau_v, tau_d,
                      cutoff_v, cutoff_d, (long long)n_rays * n_samples, n_rays, n_samples, L.n_stages, raw, saved,
                      (hipStream_t)stream);
}

int anerf_composite_backward(const AnerfConfig* cfg, const float* raw, const float* z_vals, const float* rays,
                             int32_t ray_stride, const float* noise, int32_t n_rays, int32_t n_samples,
                             const float* g_rgb, const float* g_acc, const float* g_disp, const float* g_alpha,
                             const float* g_weights, float* draw, void* stream) {
  if (!cfg || !raw || !z_vals || !rays || !g_rgb || !draw) return set_error(ANERF_E_NULL, "composite_backward: NULL pointer");
  if (n_samples < 1 || n_samples > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "composite_backward: 1 <= samples <= 512");
  if (n_rays == 0) return ANERF_OK;
  return launch_composite_bwd(cfg, raw, z_vals, rays, ray_stride, noise, n_rays, n_samples, g_rgb, g_acc, g_disp, g_alpha,
                              g_weights, draw, (hipStream_t)stream);
}

int anerf_mlp_backward(const AnerfConfig* cfg, const float* packed_t, const float* aux, const float* draw,
                       const AnerfSaved* saved, float* dz, float* df, float* dzv, int64_t n_points, void* stream) {
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 1, &L);
  if (rc) return rc;
  if (!packed_t || !aux || !draw || !dz || !df || !dzv) return set_error(ANERF_E_NULL, "mlp_backward: NULL pointer");
  if (!saved_ok(saved) || saved->p_pad < n_points) return set_error(ANERF_E_WORKSPACE, "mlp_backward: AnerfSaved");
  return mlp_bwd_entry(packed_t, aux, draw, saved, dz, df, dzv, n_points, L.n_stages, (hipStream_t)stream);
}

static int weight_grads_impl(const AnerfConfig* cfg, const AnerfSaved* sv, const float* dz, const float* df, const float* dzv,
                             const float* draw, int64_t n_points, const int32_t* perm_x, const int32_t* perm_u,
                             const AnerfNetGrads* gr, float* workspace, int64_t ws_floats, bool b3, bool accumulate,
                             void* stream, GemmBatch* defer = nullptr) {
  // defer != NULL: only the GEMM is enqueued and its reduction descriptor is handed back (anerf_backward reduces both passes of a
  // step with ONE k_reduce_dw2 launch)
  AnerfTrainLayout T;
  const int rc = anerf_train_layout(cfg, n_points, &T);
  if (rc) return rc;
  if (!saved_ok(sv) || sv->p_pad != T.p_pad) return set_error(ANERF_E_WORKSPACE, "weight_grads: AnerfSaved.p_pad mismatch");
  if (!dz || !df || !dzv || !draw || !perm_x || !perm_u || !gr || !workspace) return set_error(ANERF_E_NULL, "weight_grads: NULL");
  if (ws_floats < T.gemm_ws_floats) return set_error(ANERF_E_WORKSPACE, "weight_grads: workspace too small");
  for (int i = 0; i < 12; ++i)
    if (!gr->w[i] || !gr->b[i]) return set_error(ANERF_E_NULL, "weight_grads: NULL gradient tensor");
  const long long pp = T.p_pad;
  const int DX = dim_x(cfg), UW = u_width(cfg), KV = 256 + UW;
  GemmBatch G;
  GemmPlan P;
  memset(&G, 0, sizeof(G));
  memset(&P, 0, sizeof(P));
  P.p_pad = pp;
  gemm_block_counts(cfg, &P.nheavy);
  gemm_plan_rows(pp, P.nheavy, &P.rows_h, &P.chunks_h);
  long long ws_pos = 0, out_pos = 0;
  int np = 0, nmat = 0;
  bool plan_overflow = false;          // an operand / tile table would have been written past its end: reported after planning
  constexpr int MAX_MAT = (int)(sizeof(P.mat) / sizeof(P.mat[0]));
  auto mat = [&](const float* ptr, int ld) {
    for (int i = 0; i < nmat; ++i)
      if (P.mat[i].ptr == ptr) return i;
    if (nmat >= MAX_MAT) { plan_overflow = true; return 0; }     // bounds-checked HERE: the table lives on this stack frame
    P.mat[nmat].ptr = ptr; P.mat[nmat].ld = ld; P.mat[nmat].ncols = ld;
    return nmat++;
  };
  struct WT { int prob, a_mat, b_mat, m0, n0; };                 // one wave tile
  std::vector<WT> sq, wide;                                      // M = 256 (2 x 2 blocks), M = 128 (1 x 4)
  // `slots` > 0: a head problem (no wave tiles; hosted, see GemmHead) whose partials come in chunks x slots pieces
  auto add = [&](const float* A, int lda, int M, const float* B, int ldb, int N, float* dst, int dst_ld, int col0,
                 const int* cmap, int m_first, int m_count, float* bias, int bm_first, int bm_count, int slots = 0) {
    const int chunks = P.chunks_h * (slots ? slots : 1);
    GemmProb& p = G.p[np];
    p.dst = dst; p.bias_dst = bias; p.colmap = cmap; p.M = M; p.N = N; p.chunks = chunks;
    p.dst_ld = dst_ld; p.dst_col0 = col0; p.m_first = m_first; p.m_count = m_count;
    p.bm_first = bm_first; p.bm_count = bm_count;
    p.part_off = ws_pos; ws_pos += (long long)chunks * M * N;
    p.bias_off = -1;
    if (bias) { p.bias_off = ws_pos; ws_pos += (long long)chunks * M; }
    p.out_base = out_pos; out_pos += (long long)M * N + (bias ? M : 0);
    const int am = mat(A, lda), bm = mat(B, ldb);
    if (!slots) {
      std::vector<WT>& list = M == 256 ? sq : wide;
      for (int n0 = 0; n0 < N; n0 += 128)
        for (int m0 = 0; m0 < M; m0 += 128) list.push_back({np, am, bm, m0, n0});
    }
    return np++;
  };
  auto DZ = [&](int l) { return dz + (long long)l * pp * 256; };
  auto H = [&](int l) { return sv->h + (long long)l * pp * 256; };
  const int p_x0 = add(DZ(0), 256, 256, sv->x, DX, DX, gr->w[0], DX, 0, perm_x, 0, 256, gr->b[0], 0, 256);
  for (int l = 1; l <= 4; ++l) add(DZ(l), 256, 256, H(l - 1), 256, 256, gr->w[l], 256, 0, nullptr, 0, 256, gr->b[l], 0, 256);
  const int p_x5 = add(DZ(5), 256, 256, sv->x, DX, DX, gr->w[5], DX + 256, 0, perm_x, 0, 256, gr->b[5], 0, 256);
  add(DZ(5), 256, 256, H(4), 256, 256, gr->w[5], DX + 256, DX, nullptr, 0, 256, nullptr, 0, 0);
  add(DZ(6), 256, 256, H(5), 256, 256, gr->w[6], 256, 0, nullptr, 0, 256, gr->b[6], 0, 256);
  const int p_rgb_host = add(DZ(7), 256, 256, H(6), 256, 256, gr->w[7], 256, 0, nullptr, 0, 256, gr->b[7], 0, 256);
  const int p_alpha_host = add(df, 256, 256, H(7), 256, 256, gr->w[9], 256, 0, nullptr, 0, 256, gr->b[9], 0, 256);
  add(dzv, 128, 128, sv->f, 256, 256, gr->w[10], KV, 0, nullptr, 0, 128, gr->b[10], 0, 128);
  const int p_u = add(dzv, 128, 128, sv->u, UW, UW, gr->w[10], KV, 256, perm_u, 0, 128, nullptr, 0, 0);
  // frequency schedule (ABI 5): the images were packed as W diag(s); the gradient of W is the image's gradient times diag(s)
  G.p[p_x0].colscale = G.p[p_x5].colscale = gr->sched_x;
  G.p[p_u].colscale = gr->sched_u;
  // heads: rgb_linear <- draw[:, 0:3]^T g (4 row slots), alpha_linear <- draw[:, 3]^T h7 (2 row slots)
  const int p_rgb = add(draw, 4, 3, sv->g, 128, 128, gr->w[11], 128, 0, nullptr, 0, 3, gr->b[11], 0, 3, 4);
  const int p_alpha = add(draw, 4, 1, H(7), 256, 256, gr->w[8], 256, 0, nullptr, 0, 1, gr->b[8], 0, 1, 2);
  P.draw_mat = mat(draw, 4);
  G.nprob = np;
  G.accumulate = accumulate ? 1 : 0;
  G.total_out = out_pos;
  if (ws_pos > ws_floats || ws_pos >= (1LL << 31)) return set_error(ANERF_E_WORKSPACE, "weight_grads: workspace accounting");
  // ---- group the wave tiles into blocks of 4 that share LDS operand tiles
  int nb = 0;
  auto tile_of = [&](GemmBlock& B, int m, int col0) {
    for (int i = 0; i < B.ntiles; ++i)
      if (B.t[i].mat == m && B.t[i].col0 == col0) return i;
    if (B.ntiles >= (int)(sizeof(B.t) / sizeof(B.t[0]))) { plan_overflow = true; return 0; }
    B.t[B.ntiles].mat = m; B.t[B.ntiles].col0 = col0;
    return B.ntiles++;
  };
  P.head[0].blk = P.head[1].blk = -1;
  auto emit = [&](const std::vector<WT>& list, size_t per_block) {
    for (size_t i0 = 0; i0 < list.size(); i0 += per_block) {
      if (!list.empty() && list[i0].prob == p_alpha_host) P.head[0].blk = nb;
      if (!list.empty() && list[i0].prob == p_rgb_host) P.head[1].blk = nb;
      if (nb >= (int)(sizeof(P.blk) / sizeof(P.blk[0]))) { plan_overflow = true; return; }
      GemmBlock& B = P.blk[nb++];
      B.ntiles = 0; B.pad_ = 0;
      for (int w = 0; w < 4; ++w) B.w[w].a_tile = -1;
      for (size_t k = 0; k < per_block && i0 + k < list.size(); ++k) {
        const WT& t = list[i0 + k];
        const GemmProb& p = G.p[t.prob];
        GemmWave& W = B.w[k];
        W.a_tile = tile_of(B, t.a_mat, t.m0);
        W.b_tile = tile_of(B, t.b_mat, t.n0);
        W.part_off = (int)p.part_off;
        W.bias_off = (p.bias_off >= 0 && t.n0 == 0) ? (int)p.bias_off : -1;
        W.M = p.M; W.N = p.N; W.m0 = t.m0; W.n0 = t.n0;
      }
    }
  };
  // M = 256: wave tiles were pushed n-major with both m tiles adjacent -> 4 consecutive = {n, n+128} x {m 0, 128}.
  // A problem with an odd number of column tiles (X': 4 tiles -> even; defensive) must not share a block with the next.
  {
    std::vector<WT> run;
    for (size_t i = 0; i <= sq.size(); ++i) {
      if (i == sq.size() || (!run.empty() && sq[i].prob != run.back().prob)) {
        emit(run, 4);
        run.clear();
      }
      if (i < sq.size()) run.push_back(sq[i]);
    }
  }
  emit(wide, 4);          // views layer: one A tile (dzv) x up to 4 column tiles of f / U'
  if (plan_overflow || nb != P.nheavy) return set_error(ANERF_E_CONFIG, "weight_grads: block plan mismatch (operand / tile / block table full)");
  // head hosts: the feature layer's block (its two B tiles ARE the h7 rows) and layer 7's block + the g rows as a fifth tile
  if (P.head[0].blk < 0 || P.head[1].blk < 0 || P.blk[P.head[0].blk].ntiles != 4 || P.blk[P.head[1].blk].ntiles != 4)
    return set_error(ANERF_E_CONFIG, "weight_grads: head host blocks");
  P.head[0].tile = -1;    // (each wave's own b_tile)
  P.head[0].part_off = (int)G.p[p_alpha].part_off; P.head[0].bias_off = (int)G.p[p_alpha].bias_off;
  P.head[1].tile = tile_of(P.blk[P.head[1].blk], mat(sv->g, 128), 0);
  if (plan_overflow) return set_error(ANERF_E_CONFIG, "weight_grads: no room for the rgb head's fifth tile");
  // the hosted head FMAs (HeadWork, anerf_gemm.hip) take wave & 1 as the row tile and wave >> 1 as the column tile of a 2 x 2 block:
  // that is the order `emit` filled the host blocks' waves in (n-major, both m tiles adjacent) -- checked, not assumed
  for (int hb = 0; hb < 2; ++hb) {
    const GemmBlock& B = P.blk[P.head[hb].blk];
    for (int w = 0; w < 4; ++w)
      if (B.w[w].a_tile < 0 || B.w[w].m0 != 128 * (w & 1) || B.w[w].n0 != B.w[w & 2].n0)
        return set_error(ANERF_E_CONFIG, "weight_grads: head host block is not in (row tile = wave & 1, column tile = wave >> 1) order");
  }
  P.head[1].part_off = (int)G.p[p_rgb].part_off; P.head[1].bias_off = (int)G.p[p_rgb].bias_off;
  if (defer) *defer = G;
  return launch_weight_grads(P, G, workspace, b3, (hipStream_t)stream, defer == nullptr);
}

int anerf_mlp_backward_b3(const AnerfConfig* cfg, const float* packed_t, const float* aux, const float* draw,
                          const AnerfSaved* saved, float* dz, float* df, float* dzv, int64_t n_points, void* stream) {
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 4, &L);
  if (rc) return rc;
  if (!packed_t || !aux || !draw || !dz || !df || !dzv) return set_error(ANERF_E_NULL, "mlp_backward_b3: NULL pointer");
  if (!saved_ok(saved) || saved->p_pad < n_points) return set_error(ANERF_E_WORKSPACE, "mlp_backward_b3: AnerfSaved");
  return mlp_bwd_b3_entry(packed_t, aux, draw, saved, dz, df, dzv, n_points, L.n_stages, (hipStream_t)stream);
}

int anerf_weight_grads(const AnerfConfig* cfg, const AnerfSaved* sv, const float* dz, const float* df, const float* dzv,
                       const float* draw, int64_t n_points, const int32_t* perm_x, const int32_t* perm_u,
                       const AnerfNetGrads* gr, float* workspace, int64_t ws_floats, void* stream) {
  return weight_grads_impl(cfg, sv, dz, df, dzv, draw, n_points, perm_x, perm_u, gr, workspace, ws_floats, false, false, stream);
}

int anerf_weight_grads_b3(const AnerfConfig* cfg, const AnerfSaved* sv, const float* dz, const float* df, const float* dzv,
                          const float* draw, int64_t n_points, const int32_t* perm_x, const int32_t* perm_u,
                          const AnerfNetGrads* gr, float* workspace, int64_t ws_floats, void* stream) {
  return weight_grads_impl(cfg, sv, dz, df, dzv, draw, n_points, perm_x, perm_u, gr, workspace, ws_floats, true, false, stream);
}

int anerf_input_grads(const AnerfConfig* cfg, const float* packed_i, const float* dz, const float* dzv, int64_t p_pad,
                      int64_t n_points, float* dx, float* du, void* stream) {
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 2, &L);
  if (rc) return rc;
  if (!packed_i || !dz || !dzv || !dx || !du) return set_error(ANERF_E_NULL, "input_grads: NULL pointer");
  if (p_pad < n_points || p_pad % 128) return set_error(ANERF_E_WORKSPACE, "input_grads: p_pad");
  return mlp_bwd_in_entry(packed_i, dz, dzv, dx, du, n_points, p_pad, L.n_stages, u_width(cfg), (hipStream_t)stream);
}

int anerf_input_grads_b3(const AnerfConfig* cfg, const float* packed_i, const float* dz, const float* dzv, int64_t p_pad,
                         int64_t n_points, float* dx, float* du, void* stream) {
  AnerfLayout L;
  const int rc = anerf_layout(cfg, 5, &L);
  if (rc) return rc;
  if (!packed_i || !dz || !dzv || !dx || !du) return set_error(ANERF_E_NULL, "input_grads_b3: NULL pointer");
  if (p_pad < n_points || p_pad % 128) return set_error(ANERF_E_WORKSPACE, "input_grads_b3: p_pad");
  return mlp_bwd_in_b3_entry(packed_i, dz, dzv, dx, du, n_points, p_pad, L.n_stages, u_width(cfg), (hipStream_t)stream);
}

int anerf_encode_backward(const AnerfConfig* cfg, const float* dx, const float* du, const float* rays,
                          int32_t ray_stride, const float* z_vals, const float* skts, int64_t skt_ray_stride, float tau_v,
                          float tau_d, const float* cutoff_v, const float* cutoff_d, int32_t n_rays, int32_t n_samples,
                          float* dy_ws, float* dq_ws, float* dskts, void* stream) {
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (!dx || !du || !rays || !z_vals || !skts || !cutoff_v || !cutoff_d || !dy_ws || !dq_ws || !dskts)
    return set_error(ANERF_E_NULL, "encode_backward: NULL pointer");
  if (skt_ray_stride != 384) return set_error(ANERF_E_SHAPE, "encode_backward: skts must be per ray (stride 384)");
  if (n_rays == 0) return ANERF_OK;
  return launch_encode_bwd(cfg->multires_views, dx, du, u_width(cfg), rays, ray_stride, z_vals, skts, skt_ray_stride, tau_v,
                           tau_d, cutoff_v, cutoff_d, n_rays, n_samples, dy_ws, dq_ws, dskts, false, (hipStream_t)stream, nullptr,
                           cfg->cutoff_bones);
}

int anerf_code_grads(const AnerfConfig* cfg, const float* du, const float* cam_idx, int32_t n_rays, int32_t n_samples,
                     float* dcodes, int32_t n_codes, float* rowsum_ws, void* stream) {
  if (!config_ok(cfg) || cfg->framecode_ch != 16) return set_error(ANERF_E_CONFIG, "code_grads: framecode_ch must be 16");
  if (!du || !cam_idx || !dcodes || !rowsum_ws || n_codes < 1) return set_error(ANERF_E_NULL, "code_grads: NULL pointer");
  if (n_rays == 0) return ANERF_OK;
  return launch_code_reduce(du, u_width(cfg), cam_idx, n_rays, n_samples, n_codes, rowsum_ws, dcodes, (hipStream_t)stream);
}

int anerf_density(const AnerfConfig* cfg, const float* packed, const float* aux, const float* pts, const float* skts,
                  float tau_v, const float* cutoff_v, int64_t n_points, float* sigma_raw, void* stream) {
  if (n_points == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (!packed || !aux || !pts || !skts || !cutoff_v || !sigma_raw) return set_error(ANERF_E_NULL, "density: NULL pointer");
  if (n_points < 0) return set_error(ANERF_E_SHAPE, "density: n_points < 0");
  int stages = 0, k = 0;
  for (const Seg& s : fwd_segments(cfg)) {
    if (k++ == 8) break;                 // segments 0..7 = pts_linears.0..7
    stages += seg_stages(s);
  }
  return mlp_density_entry(packed, aux, pts, skts, tau_v, cutoff_v, n_points, stages, sigma_raw, (hipStream_t)stream,
                           cfg->cutoff_bones);
}

int anerf_gen_rays(int32_t H, int32_t W, float focal_x, float focal_y, float center_x, float center_y, const float* c2w,
                   int32_t x0, int32_t y0, int32_t x1, int32_t y1, float near, float far, float* ray_batch,
                   int64_t* valid_idx, void* stream) {
  if (!c2w || !ray_batch || !valid_idx) return set_error(ANERF_E_NULL, "gen_rays: NULL pointer");
  if (x0 < 0 || y0 < 0 || x1 > W || y1 > H || x1 < x0 || y1 < y0) return set_error(ANERF_E_SHAPE, "gen_rays: bbox");
  if (x1 == x0 || y1 == y0) return ANERF_OK;
  return launch_gen_rays(W, x0, y0, x1 - x0, y1 - y0, focal_x, focal_y, center_x, center_y, c2w, near, far, ray_batch,
                         (long long*)valid_idx, (hipStream_t)stream);
}

int anerf_assemble_frame(const float* rgb_map, const float* acc_map, const float* disp_map, const int64_t* valid_idx,
                         int32_t n_rays, float* rgb_img, float* disp_img, float* acc_img, void* stream) {
  if (n_rays == 0) return ANERF_OK;   /* empty batch: nothing to enqueue, pointers may be NULL */
  if (!rgb_map || !acc_map || !valid_idx || !rgb_img) return set_error(ANERF_E_NULL, "assemble_frame: NULL pointer");
  if (disp_img && !disp_map) return set_error(ANERF_E_NULL, "assemble_frame: disp_map");
  if (n_rays == 0) return ANERF_OK;
  return launch_assemble(rgb_map, acc_map, disp_map, (const long long*)valid_idx, n_rays, rgb_img, disp_img, acc_img,
                         (hipStream_t)stream);
}

// ---- one-call forward ---------------------------------------------------------------------------------------------
namespace {
struct FwdWs {
  int64_t near_far, stats, z, raw, weights, zs, zm, idx, raw_is, raw_f, weights_f, pn_f, total;
};
FwdWs fwd_ws(int64_t n, int64_t S, int64_t Ni) {
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  FwdWs w;
  int64_t o = 0;
  w.near_far = o; o += up(n * 2 * 4);
  w.stats = o; o += up(32);
  w.z = o; o += up(n * S * 4);
  w.raw = o; o += up(n * S * 16);
  w.weights = o; o += up(n * S * 4);
  w.zs = o; o += up(n * Ni * 4);
  w.zm = o; o += up(n * (S + Ni) * 4);
  w.idx = o; o += up(n * (S + Ni) * 8);
  w.raw_is = o; o += up(n * Ni * 16);
  w.raw_f = o; o += up(n * (S + Ni) * 16);
  w.weights_f = o; o += up(n * (S + Ni) * 4);
  w.pn_f = o; o += up(n * (S + Ni) * 12);      // point noise of the merged samples (ray_noise_std > 0 only)
  w.total = o;
  return w;
}
}  // namespace

int64_t anerf_workspace_size(const AnerfConfig* cfg, int32_t n_rays, int32_t n_samples, int32_t n_importance) {
  if (!config_ok(cfg) || n_rays < 0 || n_samples < 1 || n_importance < 0) return set_error(ANERF_E_SHAPE, "workspace_size: bad sizes");
  return fwd_ws(n_rays, n_samples, n_importance).total;
}

namespace {
// saved-activation planes of one network pass inside the training workspace
struct SavedWs { int64_t h, f, g, x, u, bytes, p_pad; };
SavedWs saved_ws(const AnerfConfig* cfg, int64_t n_points) {
  AnerfTrainLayout T;
  anerf_train_layout(cfg, n_points > 0 ? n_points : 1, &T);
  SavedWs s;
  int64_t o = 0;
  const int64_t pp = T.p_pad;
  s.p_pad = pp;
  s.h = o; o += 8 * pp * 256 * 4;
  s.f = o; o += pp * 256 * 4;
  s.g = o; o += pp * 128 * 4;
  s.x = o; o += pp * T.x_width * 4;
  s.u = o; o += pp * T.u_width * 4;
  s.bytes = o;
  return s;
}
AnerfSaved saved_at(char* base, const SavedWs& s) {
  AnerfSaved a;
  a.h = reinterpret_cast<float*>(base + s.h);
  a.f = reinterpret_cast<float*>(base + s.f);
  a.g = reinterpret_cast<float*>(base + s.g);
  a.x = reinterpret_cast<float*>(base + s.x);
  a.u = reinterpret_cast<float*>(base + s.u);
  a.p_pad = s.p_pad;
  return a;
}
// rows [P, p_pad) of `planes` row-major [p_pad][width] planes := 0 (the weight-gradient GEMM reads them)
int zero_pad_rows(float* base, int planes, int64_t p_pad, int64_t P, int width, hipStream_t st) {
  if (p_pad <= P) return ANERF_OK;
  const hipError_t e = hipMemset2DAsync(base + P * width, (size_t)p_pad * width * 4, 0, (size_t)(p_pad - P) * width * 4, planes, st);
  return e == hipSuccess ? ANERF_OK : set_error(ANERF_E_LAUNCH, "zero_pad_rows: hipMemset2DAsync failed");
}
int zero_saved_pads(const AnerfConfig* cfg, const AnerfSaved& a, int64_t P, hipStream_t st) {
  int rc = zero_pad_rows(a.h, 8, a.p_pad, P, 256, st);
  if (!rc) rc = zero_pad_rows(a.f, 1, a.p_pad, P, 256, st);
  if (!rc) rc = zero_pad_rows(a.g, 1, a.p_pad, P, 128, st);
  if (!rc) rc = zero_pad_rows(a.x, 1, a.p_pad, P, dim_x(cfg), st);
  if (!rc) rc = zero_pad_rows(a.u, 1, a.p_pad, P, u_width(cfg), st);
  return rc;
}

// ABI revision 3: optional caller-owned timing events around the MFMA kernels (AnerfProfile)
inline void prof_rec(const AnerfProfile* p, int slot, void* stream) {
  if (p && p->ev[slot]) (void)hipEventRecord((hipEvent_t)p->ev[slot], (hipStream_t)stream);
}

int forward_check(const AnerfConfig* cfg, const AnerfForwardIO* io, const char* who) {
  if (!config_ok(cfg)) return set_error(ANERF_E_CONFIG, "unsupported AnerfConfig");
  if (!io) return set_error(ANERF_E_NULL, "forward: io is NULL");
  if (io->n_rays < 0 || io->n_importance < 0) return set_error(ANERF_E_SHAPE, "forward: negative sizes");
  if (io->precision != 0 && io->precision != 1) return set_error(ANERF_E_CONFIG, "forward: precision must be 0 (fp32) or 1 (bf16x3)");
  (void)who;
  return ANERF_OK;
}

// RayCaster.render_rays for one caster call; sv_c / sv_f != NULL selects the training kernels (activations saved)
int forward_impl(const AnerfConfig* cfg, const AnerfForwardIO* io, char* ws, const FwdWs& w, const AnerfSaved* sv_c,
                 const AnerfSaved* sv_f, void* stream) {
  const int n = io->n_rays, S = io->n_samples, Ni = io->n_importance;
  if (!io->rgb_map || !io->disp_map || !io->acc_map || !io->alpha) return set_error(ANERF_E_NULL, "forward: output maps");
  if (Ni > 0 && !io->single_net && (!io->packed_f || !io->aux_f)) return set_error(ANERF_E_NULL, "forward: fine network image");
  auto F = [&](int64_t off) { return reinterpret_cast<float*>(ws + off); };
  if ((io->pts_noise != nullptr) != (Ni > 0 ? io->pts_noise_is != nullptr : io->pts_noise != nullptr))
    return set_error(ANERF_E_NULL, "forward: pts_noise and pts_noise_is go together when n_importance > 0");
  auto mlp = [&](const float* packed, const float* aux, const float* codes, const float* zz, int ns, float* raw,
                 const AnerfSaved* sv, const float* pn) {
    // ABI revision 6: io->step = the device-resident step block; the TRAINING kernels then read {tau_v, tau_d} from it
    const float* tau_dev = (sv && io->step) ? &io->step->tau_v : nullptr;
    if (pn || tau_dev) {   // ray_noise_std > 0 / device-resident tau: same kernels through the internal entries, which take both
      AnerfLayout L;
      int r = anerf_layout(cfg, io->precision == 1 ? 3 : 0, &L);
      if (r) return r;
      if (!packed || !aux || !io->rays || !zz || !io->skts || !io->cutoff_v || !io->cutoff_d || !raw)
        return set_error(ANERF_E_NULL, "forward: NULL pointer");
      if (cfg->framecode_ch && (!io->cam_idx || !codes || io->n_codes < 1)) return set_error(ANERF_E_NULL, "forward: frame codes");
      if (ns < MIN_SAMPLES || ns > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "forward: 8 <= samples per ray <= 512");
      if ((io->skt_ray_stride != 0 && io->skt_ray_stride != 384) || io->ray_stride < 6) return set_error(ANERF_E_SHAPE, "forward: strides");
      if (io->precision == 1)
        return mlp_b3_entry(cfg, packed, aux, io->rays, io->ray_stride, zz, io->skts, io->skt_ray_stride, io->cam_idx, codes, io->n_codes,
                            io->tau_v, io->tau_d, io->cutoff_v, io->cutoff_d, (long long)n * ns, n, ns, L.n_stages, raw, sv,
                            (hipStream_t)stream, pn, tau_dev);
      return mlp_raw_entry(cfg, packed, aux, io->rays, io->ray_stride, zz, io->skts, io->skt_ray_stride, io->cam_idx, codes, io->n_codes,
                           io->tau_v, io->tau_d, io->cutoff_v, io->cutoff_d, nullptr, 0, (long long)n * ns, n, ns, L.n_stages, raw, false,
                           sv, (hipStream_t)stream, pn, tau_dev);
    }
    if (sv)
      return (io->precision == 1 ? anerf_mlp_raw_train_b3 : anerf_mlp_raw_train)(
          cfg, packed, aux, io->rays, io->ray_stride, zz, io->skts, io->skt_ray_stride, io->cam_idx, codes, io->n_codes,
          io->tau_v, io->tau_d, io->cutoff_v, io->cutoff_d, n, ns, raw, sv, stream);
    return (io->precision == 1 ? anerf_mlp_raw_b3 : anerf_mlp_raw)(
        cfg, packed, aux, io->rays, io->ray_stride, zz, io->skts, io->skt_ray_stride, io->cam_idx, codes, io->n_codes,
        io->tau_v, io->tau_d, io->cutoff_v, io->cutoff_d, n, ns, raw, stream);
  };
  // ABI revision 4: io->cyl_shared = one cylinder [5] for every ray of the call (a frame's rays under render_path)
  int rc;
  if (n <= bounds_z_max_rays()) {
    // A2 + A3 as ONE launch (k_bounds_z: bit-identical to the staged pair, tests/test_hip_edge_cases.py); same argument checks
    if (!io->rays || !io->cyls) return set_error(ANERF_E_NULL, "ray_bounds: NULL pointer");
    if (io->ray_stride < 8) return set_error(ANERF_E_SHAPE, "ray_bounds: ray_stride >= 8 required");
    if (S < 2 || S > MAX_SAMPLES) return set_error(ANERF_E_SHAPE, "coarse_z: 2 <= N_samples <= 512");
    rc = launch_bounds_z(io->rays, io->ray_stride, io->cyls, io->cyl_shared ? 0 : 5, n, S, io->t_rand, io->lindisp, F(w.z), (hipStream_t)stream);
  } else {
    rc = ray_bounds_checked(io->rays, io->ray_stride, io->cyls, io->cyl_shared ? 0 : 5, n, F(w.near_far), F(w.stats), stream);
    if (rc) return rc;
    rc = anerf_coarse_z(F(w.near_far), F(w.stats), io->rays, io->ray_stride, n, S, io->t_rand, io->lindisp, F(w.z), nullptr, stream);
  }
  if (rc) return rc;
  prof_rec(io->profile, ANERF_PROF_FWD(0), stream);
  rc = mlp(io->packed_c, io->aux_c, io->codes_c, F(w.z), S, F(w.raw), sv_c, io->pts_noise);
  if (rc) return rc;
  prof_rec(io->profile, ANERF_PROF_FWD(0) + 1, stream);
  const bool hier = Ni > 0;
  float* alpha_c = hier ? io->alpha0 : io->alpha;
  float* scratch_alpha = F(w.weights_f);            // coarse alpha lands here when the caller does not want alpha0
  rc = anerf_composite(cfg, F(w.raw), F(w.z), io->rays, io->ray_stride, io->noise, n, S,
                       hier ? (io->rgb0 ? io->rgb0 : io->rgb_map) : io->rgb_map, hier ? (io->disp0 ? io->disp0 : io->disp_map) : io->disp_map,
                       hier ? (io->acc0 ? io->acc0 : io->acc_map) : io->acc_map, F(w.weights), alpha_c ? alpha_c : scratch_alpha, nullptr, stream);
  if (rc || !hier) return rc;
  rc = anerf_importance(F(w.z), F(w.weights), n, S, Ni, io->u_imp, io->single_net, F(w.zs), F(w.zm),
                        reinterpret_cast<int64_t*>(ws + w.idx), stream);
  if (rc) return rc;
  if (io->single_net) {
    rc = mlp(io->packed_c, io->aux_c, io->codes_c, F(w.zs), Ni, F(w.raw_is), nullptr, io->pts_noise_is);
    if (rc) return rc;
    rc = launch_gather_raw(F(w.raw), F(w.raw_is), reinterpret_cast<const long long*>(ws + w.idx), n, S, Ni, F(w.raw_f), (hipStream_t)stream);
  } else {
    const float* pn_f = nullptr;
    if (io->pts_noise) {   // the merged samples keep their own offsets: gather [coarse | importance] by the sort order (raycasters.py:679-709)
      rc = launch_gather_rows3(io->pts_noise, io->pts_noise_is, reinterpret_cast<const long long*>(ws + w.idx), n, S, Ni, F(w.pn_f),
                               (hipStream_t)stream);
      if (rc) return rc;
      pn_f = F(w.pn_f);
    }
    prof_rec(io->profile, ANERF_PROF_FWD(1), stream);
    rc = mlp(io->packed_f, io->aux_f, io->codes_f, F(w.zm), S + Ni, F(w.raw_f), sv_f, pn_f);
    prof_rec(io->profile, ANERF_PROF_FWD(1) + 1, stream);
  }
  if (rc) return rc;
  return anerf_composite(cfg, F(w.raw_f), F(w.zm), io->rays, io->ray_stride, io->noise_fine, n, S + Ni, io->rgb_map, io->disp_map,
                         io->acc_map, F(w.weights_f), io->alpha, nullptr, stream);
}

// training workspace = forward workspace + saved planes of the coarse pass (+ of the fine pass)
struct TrainWs { FwdWs fwd; SavedWs sc, sf; int64_t off_c, off_f, total; };
TrainWs train_ws(const AnerfConfig* cfg, int64_t n, int64_t S, int64_t Ni) {
  TrainWs t;
  t.fwd = fwd_ws(n, S, Ni);
  t.sc = saved_ws(cfg, n * S);
  t.off_c = t.fwd.total;
  t.off_f = t.off_c + t.sc.bytes;
  t.sf = saved_ws(cfg, Ni > 0 ? n * (S + Ni) : 0);
  t.total = t.off_f + (Ni > 0 ? t.sf.bytes : 0);
  return t;
}

// backward scratch of ONE network pass over P points (reused by the second pass)
struct BwdWs { int64_t draw, dz, df, dzv, gemm, dx, du, dy, dq, rowsum, total; };
BwdWs bwd_ws(const AnerfConfig* cfg, int64_t P, bool input_grads) {
  auto up = [](int64_t b) { return (b + 255) / 256 * 256; };
  AnerfTrainLayout T;
  anerf_train_layout(cfg, P > 0 ? P : 1, &T);
  const int64_t pp = T.p_pad;
  BwdWs b;
  int64_t o = 0;
  b.draw = o; o += up(pp * 16);
  b.dz = o; o += up(8 * pp * 256 * 4);
  b.df = o; o += up(pp * 256 * 4);
  b.dzv = o; o += up(pp * 128 * 4);
  b.gemm = o; o += up(T.gemm_ws_floats * 4);
  b.dx = b.du = b.dy = b.dq = b.rowsum = o;
  if (input_grads) {
    b.rowsum = o; o += up((P / 8 + 1) * 16 * 4);     // [N][16], N <= P / 8 (MIN_SAMPLES)
    b.dx = o; o += up(pp * T.x_width * 4);
    b.du = o; o += up(pp * T.u_width * 4);
    b.dy = o; o += up(P * 72 * 4);
    b.dq = o; o += up(P * 72 * 4);
  }
  b.total = o;
  return b;
}
}  // namespace

int anerf_forward(const AnerfConfig* cfg, const AnerfForwardIO* io, void* workspace, int64_t ws_bytes, void* stream) {
  const int rc = forward_check(cfg, io, "forward");
  if (rc) return rc;
  if (io->n_rays == 0) return ANERF_OK;
  const FwdWs w = fwd_ws(io->n_rays, io->n_samples, io->n_importance);
  if (!workspace || ws_bytes < w.total || ((uintptr_t)workspace & 15)) return set_error(ANERF_E_WORKSPACE, "forward: workspace");
  return forward_impl(cfg, io, static_cast<char*>(workspace), w, nullptr, nullptr, stream);
}

int64_t anerf_train_workspace_size(const AnerfConfig* cfg, int32_t n_rays, int32_t n_samples, int32_t n_importance) {
  if (!config_ok(cfg) || n_rays < 0 || n_samples < 1 || n_importance < 0) return set_error(ANERF_E_SHAPE, "train_workspace_size: bad sizes");
  return train_ws(cfg, n_rays, n_samples, n_importance).total;
}

int64_t anerf_backward_scratch_size(const AnerfConfig* cfg, int32_t n_rays, int32_t n_samples, int32_t n_importance,
                                    int32_t input_grads) {
  if (!config_ok(cfg) || n_rays < 0 || n_samples < 1 || n_importance < 0) return set_error(ANERF_E_SHAPE, "backward_scratch_size: bad sizes");
  // + a second home for the fine pass's GEMM partials: when both passes run in one call they wait there for the merged reduction
  const BwdWs f = bwd_ws(cfg, (int64_t)n_rays * (n_samples + n_importance), input_grads != 0);
  AnerfTrainLayout T;
  anerf_train_layout(cfg, (int64_t)n_rays * (n_samples + n_importance) > 0 ? (int64_t)n_rays * (n_samples + n_importance) : 1, &T);
  return f.total + (n_importance > 0 ? (T.gemm_ws_floats * 4 + 255) / 256 * 256 : 0);
}

int anerf_train_forward(const AnerfConfig* cfg, const AnerfForwardIO* io, void* workspace, int64_t ws_bytes, void* stream) {
  int rc = forward_check(cfg, io, "train_forward");
  if (rc) return rc;
  if (io->single_net) return set_error(ANERF_E_CONFIG, "train_forward: single_net trains through the staged entry points");
  if (io->n_rays == 0) return ANERF_OK;
  const int64_t n = io->n_rays, S = io->n_samples, Ni = io->n_importance;
  const TrainWs t = train_ws(cfg, n, S, Ni);
  if (!workspace || ws_bytes < t.total || ((uintptr_t)workspace & 15)) return set_error(ANERF_E_WORKSPACE, "train_forward: workspace");
  char* ws = static_cast<char*>(workspace);
  const AnerfSaved sc = saved_at(ws + t.off_c, t.sc), sf = saved_at(ws + t.off_f, t.sf);
  rc = zero_saved_pads(cfg, sc, n * S, (hipStream_t)stream);
  if (!rc && Ni > 0) rc = zero_saved_pads(cfg, sf, n * (S + Ni), (hipStream_t)stream);
  if (rc) return rc;
  return forward_impl(cfg, io, ws, t.fwd, &sc, Ni > 0 ? &sf : nullptr, stream);
}

int anerf_backward(const AnerfConfig* cfg, const AnerfForwardIO* io, const AnerfBackwardIO* b, void* workspace,
                   int64_t ws_bytes, void* scratch, int64_t scratch_bytes, void* stream) {
  int rc = forward_check(cfg, io, "backward");
  if (rc) return rc;
  if (!b) return set_error(ANERF_E_NULL, "backward: AnerfBackwardIO is NULL");
  if (io->single_net) return set_error(ANERF_E_CONFIG, "backward: single_net trains through the staged entry points");
  if (io->n_rays == 0) return ANERF_OK;
  const int64_t n = io->n_rays, S = io->n_samples, Ni = io->n_importance;
  const bool hier = Ni > 0, b3 = io->precision == 1;
  const bool want_in = b->g_skts || b->g_codes_c || b->g_codes_f;
  const TrainWs t = train_ws(cfg, n, S, Ni);
  if (!workspace || ws_bytes < t.total || ((uintptr_t)workspace & 15)) return set_error(ANERF_E_WORKSPACE, "backward: workspace");
  const int64_t fine_total = bwd_ws(cfg, n * (S + Ni), want_in).total;
  if (!scratch || ((uintptr_t)scratch & 15) || scratch_bytes < fine_total) return set_error(ANERF_E_WORKSPACE, "backward: scratch");
  // one k_reduce_dw2 for both passes when they run in this call, no per-kernel timing is asked for and the scratch has the extra
  // region (anerf_backward_scratch_size): the fine pass's GEMM partials are written THERE, out of the coarse pass's way
  AnerfTrainLayout Tf;
  anerf_train_layout(cfg, n * (S + Ni), &Tf);
  const int64_t gemm_f_bytes = (Tf.gemm_ws_floats * 4 + 255) / 256 * 256;
  const bool merge_reduce = hier && b->passes == 0 && !b->profile && scratch_bytes >= fine_total + gemm_f_bytes;
  GemmBatch G_fine;
  if (!b->g_rgb || !b->perm_x || !b->perm_u || !b->packed_t_c || (hier && (!b->packed_t_f || !b->g_rgb0)))
    return set_error(ANERF_E_NULL, "backward: NULL pointer");
  if (want_in && (!b->packed_i_c || (hier && !b->packed_i_f))) return set_error(ANERF_E_NULL, "backward: input-gradient weight image");
  if (b->g_skts && io->skt_ray_stride != 384) return set_error(ANERF_E_SHAPE, "backward: g_skts needs per-ray skts (stride 384)");
  char* ws = static_cast<char*>(workspace);
  char* sb = static_cast<char*>(scratch);
  hipStream_t st = (hipStream_t)stream;
  auto F = [&](int64_t off) { return reinterpret_cast<float*>(ws + off); };
  const int uw = u_width(cfg);
  if (b->passes < 0 || (b->passes > 3 && b->passes != 4 && b->passes != 8 && b->passes != 16 && b->passes != 32))
    return set_error(ANERF_E_CONFIG, "backward: passes must be 0..3, 4, 8, 16 or 32");
  // ABI revision 6: passes = 4 / 8 split the COARSE pass once more -- 4 = everything that ends in PARAMETER gradients (weights, biases,
  // frame codes: complete when this call is enqueued, so their all-reduce can start), 8 = its pose-gradient tail (k_encode_bwd +
  // k_pose_reduce into g_skts), which reads dx / du of the preceding passes = 4 call from the SAME scratch
  // Round 6 (still ABI revision 7: new VALUES of an existing field): passes = 16 / 32 split the passes = 4 part once more -- 16 = the
  // coarse pass up to its WEIGHT gradients (composite backward, k_mlp_bwd, GEMM + reduction: 99.9 % of the coarse network's bucket
  // bytes are final here, and the 3.46 MB all-reduce can start under the 180 us input-gradient kernel), 32 = the input-gradient
  // part (k_mlp_bwd_in[_enc] + frame-code gradients), which reads dz / dzv of the passes = 16 call from the SAME scratch
  const int coarse_part = b->passes == 4 ? 1 : (b->passes == 8 ? 2 : (b->passes == 16 ? 3 : (b->passes == 32 ? 4 : 0)));
  if ((coarse_part == 1 || coarse_part == 2) && (!hier || !b->g_skts))
    return set_error(ANERF_E_CONFIG, "backward: passes = 4 / 8 need n_importance > 0 and g_skts");
  if (coarse_part >= 3 && (!hier || !want_in)) return set_error(ANERF_E_CONFIG, "backward: passes = 16 / 32 need n_importance > 0 and input gradients (g_skts or frame codes)");
  const bool do_fine = hier && (b->passes == 0 || (b->passes <= 3 && (b->passes & 1)));
  const bool do_coarse = !hier || b->passes == 0 || (b->passes & 2) || coarse_part;
  const bool coarse_only = hier && !do_fine;              // second half of a split backward: g_skts already holds the fine pass
  // (no zero fill of g_skts: the first pose-gradient pass WRITES all four rows of every 4 x 4 block, row 3 as zeros -- k_pose_reduce)
  bool skts_written = coarse_only;
  // one network pass: composite backward -> dz chain -> weight gradients (-> input gradients -> pose / code gradients)
  auto pass = [&](int which_pass, const AnerfSaved& sv, const float* raw, const float* zz, int ns, const float* noise, const float* g_rgb,
                  const float* g_acc, const float* g_disp, const float* g_alpha, const float* packed_t, const float* aux,
                  const float* packed_i, const AnerfNetGrads* gr, float* g_codes, const float* pn, int part) {
    const int64_t P = n * ns;
    const BwdWs w = bwd_ws(cfg, P, want_in);
    auto B = [&](int64_t off) { return reinterpret_cast<float*>(sb + off); };
    const int64_t pp = sv.p_pad;
    // round 6: the fp32 one-call backward applies the encoding's backward INSIDE k_mlp_bwd_in (k_mlp_bwd_in_enc: dX' / dU' never
    // stored, dY / dQ written by the tile kernel); the pose tail is then k_pose_reduce alone.  ANERF_NO_FUSED_ENCODE_BWD=1 keeps the
    // separate k_mlp_bwd_in -> k_encode_bwd pair (A/B measurements; the split-bf16 path and the staged entry points always use it)
    static const bool no_fuse = getenv("ANERF_NO_FUSED_ENCODE_BWD") != nullptr && getenv("ANERF_NO_FUSED_ENCODE_BWD")[0] == '1';
    const bool fuse_enc = b->g_skts && !b3 && !no_fuse &&
                          (cfg->multires_views == 4 || (cfg->multires_views == 0 && cfg->framecode_ch == 0)) &&
                          (cfg->framecode_ch == 0 || cfg->framecode_ch == 16) && cfg->multires == 7;
    auto pose_tail = [&]() {
      if (fuse_enc) {
        int r3 = launch_pose_reduce(B(w.dy), B(w.dq), io->rays, io->ray_stride, zz, (int)n, ns, b->g_skts, skts_written, st, pn);
        if (!r3) skts_written = true;
        return r3;
      }
      int r2 = launch_encode_bwd(cfg->multires_views, B(w.dx), B(w.du), uw, io->rays, io->ray_stride, zz, io->skts, io->skt_ray_stride,
                                 io->tau_v, io->tau_d, io->cutoff_v, io->cutoff_d, (int)n, ns, B(w.dy), B(w.dq), b->g_skts, skts_written, st, pn,
                                 cfg->cutoff_bones, io->step ? &io->step->tau_v : nullptr);
      if (!r2) skts_written = true;
      return r2;
    };
    if (part == 2) return pose_tail();       // passes = 8: dx / du are where the passes = 4 call left them
    int r = ANERF_OK;
    if (part != 4) {                         // (passes = 32: dz / dzv are where the passes = 16 call left them)
    r = zero_pad_rows(B(w.draw), 1, pp, P, 4, st);
    if (!r) r = zero_pad_rows(B(w.dz), 8, pp, P, 256, st);
    if (!r) r = zero_pad_rows(B(w.df), 1, pp, P, 256, st);
    if (!r) r = zero_pad_rows(B(w.dzv), 1, pp, P, 128, st);
    if (r) return r;
    r = anerf_composite_backward(cfg, raw, zz, io->rays, io->ray_stride, noise, (int)n, ns, g_rgb, g_acc, g_disp, g_alpha, nullptr,
                                 B(w.draw), stream);
    if (r) return r;
    prof_rec(b->profile, ANERF_PROF_BWD(which_pass), stream);
    r = (b3 ? anerf_mlp_backward_b3 : anerf_mlp_backward)(cfg, packed_t, aux, B(w.draw), &sv, B(w.dz), B(w.df), B(w.dzv), P, stream);
    if (r) return r;
    prof_rec(b->profile, ANERF_PROF_BWD(which_pass) + 1, stream);
    AnerfTrainLayout T;
    anerf_train_layout(cfg, P, &T);
    prof_rec(b->profile, ANERF_PROF_GEMM(which_pass), stream);
    if (merge_reduce && which_pass == 1) {
      r = weight_grads_impl(cfg, &sv, B(w.dz), B(w.df), B(w.dzv), B(w.draw), P, b->perm_x, b->perm_u, gr, B(fine_total),
                            T.gemm_ws_floats, b3, b->accumulate != 0, stream, &G_fine);
    } else if (merge_reduce) {
      GemmBatch G_coarse;
      r = weight_grads_impl(cfg, &sv, B(w.dz), B(w.df), B(w.dzv), B(w.draw), P, b->perm_x, b->perm_u, gr, B(w.gemm),
                            T.gemm_ws_floats, b3, b->accumulate != 0, stream, &G_coarse);
      if (!r) r = launch_reduce_dw2(G_fine, B(fine_total), G_coarse, B(w.gemm), st);
    } else {
      r = weight_grads_impl(cfg, &sv, B(w.dz), B(w.df), B(w.dzv), B(w.draw), P, b->perm_x, b->perm_u, gr, B(w.gemm),
                            T.gemm_ws_floats, b3, b->accumulate != 0, stream);
    }
    prof_rec(b->profile, ANERF_PROF_GEMM(which_pass) + 1, stream);
    }   // part != 4
    if (r || !want_in || part == 3) return r;      // (passes = 16 ends behind the weight gradients)
    prof_rec(b->profile, ANERF_PROF_BWD_IN(which_pass), stream);
    if (fuse_enc) {
      AnerfLayout Li;
      r = anerf_layout(cfg, 2, &Li);
      if (r) return r;
      if (!packed_i) return set_error(ANERF_E_NULL, "backward: input-gradient weight image");
      r = mlp_bwd_in_enc_entry(cfg->multires_views, cfg->framecode_ch, packed_i, B(w.dz), B(w.dzv), B(w.du), P, pp, Li.n_stages, io->rays,
                               io->ray_stride, zz, io->skts, io->skt_ray_stride, io->tau_v, io->tau_d, io->cutoff_v, io->cutoff_d, ns,
                               B(w.dy), B(w.dq), pn, cfg->cutoff_bones, io->step ? &io->step->tau_v : nullptr, (int)n, st);
    } else {
      r = (b3 ? anerf_input_grads_b3 : anerf_input_grads)(cfg, packed_i, B(w.dz), B(w.dzv), pp, P, B(w.dx), B(w.du), stream);
    }
    if (r) return r;
    prof_rec(b->profile, ANERF_PROF_BWD_IN(which_pass) + 1, stream);
    // frame-code gradients first: they are PARAMETER gradients (all-reduced), the pose gradients behind them are not
    // (independent kernels: both only read dx / du)
    if (g_codes) {
      // accumulate = 1: the frame-code gradients are ADDED to the caller's tensor too (k_code_reduce adds; no zero fill)
      if (!b->accumulate && hipMemsetAsync(g_codes, 0, (size_t)io->n_codes * 16 * 4, st) != hipSuccess) return set_error(ANERF_E_LAUNCH, "backward: hipMemsetAsync");
      r = anerf_code_grads(cfg, B(w.du), io->cam_idx, (int)n, ns, g_codes, io->n_codes, B(w.rowsum), stream);
      if (r) return r;
    }
    if (b->g_skts && part != 1 && part != 4) r = pose_tail();
    return r;
  };
  if (do_fine) {   // the fine pass first, as autograd runs it
    const AnerfSaved sf = saved_at(ws + t.off_f, t.sf);
    rc = pass(1, sf, F(t.fwd.raw_f), F(t.fwd.zm), (int)(S + Ni), io->noise_fine, b->g_rgb, b->g_acc, b->g_disp, b->g_alpha,
              b->packed_t_f, io->aux_f, b->packed_i_f, &b->grads_f, b->g_codes_f, io->pts_noise ? F(t.fwd.pn_f) : nullptr, 0);
    if (rc) return rc;
  }
  if (!do_coarse) return ANERF_OK;
  const AnerfSaved sc = saved_at(ws + t.off_c, t.sc);
  return pass(0, sc, F(t.fwd.raw), F(t.fwd.z), (int)S, io->noise, hier ? b->g_rgb0 : b->g_rgb, hier ? b->g_acc0 : b->g_acc,
              hier ? b->g_disp0 : b->g_disp, hier ? b->g_alpha0 : b->g_alpha, b->packed_t_c, io->aux_c, b->packed_i_c, &b->grads_c,
              b->g_codes_c, io->pts_noise, coarse_part);
}

}  // extern "C"
