// train_step_demo -- ONE WHOLE TRAINING ITERATION of the hot path through the C ABI, from plain C++ (no Python, no torch): random
// inputs -> weight-image gather -> anerf_train_forward (ray bounds, depths, fused encode + 8x256 MLP x 2 networks, importance
// resampling, compositing) -> anerf_loss -> anerf_backward (all 48 parameter gradients) -> Adam.  Run twice from the same start:
//   (a) eagerly, the per-iteration scalars as ARGUMENTS (anerf_rand_fill / anerf_adam_step),
//   (b) captured ONCE into a hipGraph with the plain HIP runtime API and replayed, the scalars in the device-resident step block
//       (anerf_rand_fill_dev / anerf_adam_step_dev / AnerfForwardIO.step + anerf_step_block_write) -- no node update.
// Must end on bit-identical parameters and losses, with the loss going down.  This is the drop-in boundary of SURVEY 8(b) used the
// way a non-Python host would use it: device pointers, sizes, a stream; the library never allocates and never synchronises.
// Built by __graft_entry__.build(), run by tests/test_graph_step.py on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "anerf.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #x, r_, anerf_last_error()); exit(3); } } while (0)

template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }

int main(int argc, char** argv) {
  // train_step_demo [n_rays [timed_iterations]]: with a second argument the eager and the captured iteration are also TIMED
  // (HIP events over `timed_iterations` back-to-back iterations each): what a non-Python host pays per step
  const AnerfConfig cfg = {24, 7, 4, 0, 8, 256, 4, 0, 1.0f, 0.0f, 0};   // configs/surreal/surreal.txt
  const int N = argc > 1 ? atoi(argv[1]) : 192, S = 64, NI = 16, ITERS = 6;
  const int TIMED = argc > 2 ? atoi(argv[2]) : 0;
  if (N < 16 || N > 65536) { fprintf(stderr, "n_rays in 16..65536\n"); return 2; }
  const uint64_t SEED = 0xA5EEDull;
  hipStream_t st;
  CK(hipStreamCreate(&st));

  // ---- parameters of both networks in ONE flat buffer (torch Linear [out][in] row-major; order w0, b0, w1, b1, ...)
  const int OUT[12] = {256, 256, 256, 256, 256, 256, 256, 256, 1, 256, 128, 3};
  const int IN[12] = {432, 256, 256, 256, 256, 688, 256, 256, 256, 256, 904, 128};
  int64_t per_net = 0;
  for (int i = 0; i < 12; ++i) per_net += (int64_t)OUT[i] * IN[i] + OUT[i];
  const int64_t NP = 2 * per_net;                                        // 1 728 520
  float *P = dalloc<float>(NP + 4), *G = dalloc<float>(NP + 4), *M = dalloc<float>(NP + 4), *V = dalloc<float>(NP + 4);
  AnerfNetParams np[2];
  AnerfNetGrads ng[2];
  memset(np, 0, sizeof(np));
  memset(ng, 0, sizeof(ng));
  std::vector<AnerfRandJob> init;
  for (int n = 0; n < 2; ++n) {
    int64_t o = n * per_net;
    for (int i = 0; i < 12; ++i) {
      np[n].w[i] = P + o; ng[n].w[i] = G + o;
      init.push_back({P + o, (int64_t)OUT[i] * IN[i], 1, 1.0f / sqrtf((float)IN[i])});
      o += (int64_t)OUT[i] * IN[i];
      np[n].b[i] = P + o; ng[n].b[i] = G + o;
      init.push_back({P + o, OUT[i], 1, 0.05f});
      o += OUT[i];
    }
  }
  auto init_params = [&]() {                                             // the same draws every time: both runs start equal
    for (size_t k = 0; k < init.size(); k += 6) AK(anerf_rand_fill(&init[k], (int)(init.size() - k < 6 ? init.size() - k : 6), SEED, 1000 + k, st));
    const float one = 1.0f;                                              // alpha_linear.bias = +1 (a default init renders zero density)
    for (int n = 0; n < 2; ++n) CK(hipMemcpyAsync((void*)np[n].b[8], &one, 4, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(G, 0, (NP + 4) * 4, st));
    CK(hipMemsetAsync(M, 0, (NP + 4) * 4, st));
    CK(hipMemsetAsync(V, 0, (NP + 4) * 4, st));
    CK(hipStreamSynchronize(st));
  };

  // ---- weight images (which = 0: forward, 1: backward-data) of both networks: one gather launch per iteration
  AnerfPackJob jobs[4];
  float* img[2][2];
  AnerfLayout L[2];
  for (int w = 0; w < 2; ++w) {
    AK(anerf_layout(&cfg, w, &L[w]));
    const int64_t n = L[w].stream_floats + L[w].aux_floats;
    std::vector<int32_t> tab(n);
    AK(anerf_build_pack_table(&cfg, w, tab.data()));
    int32_t* dtab = dalloc<int32_t>(n);
    CK(hipMemcpy(dtab, tab.data(), n * 4, hipMemcpyHostToDevice));
    for (int net = 0; net < 2; ++net) {
      img[net][w] = dalloc<float>(n);
      jobs[2 * net + w] = {np[net], dtab, n, img[net][w], 0};
    }
  }
  int32_t hx[432], hu[648];
  AK(anerf_build_perm_tables(&cfg, hx, hu));
  int32_t *perm_x = dalloc<int32_t>(432), *perm_u = dalloc<int32_t>(648);
  CK(hipMemcpy(perm_x, hx, sizeof(hx), hipMemcpyHostToDevice));
  CK(hipMemcpy(perm_u, hu, sizeof(hu), hipMemcpyHostToDevice));

  // ---- a synthetic batch: rays from z = +3 towards a unit cylinder around the origin, one pose (24 bone frames), white target
  std::vector<float> ro(3 * N), rd(3 * N), tgt(3 * N), skt(24 * 16, 0.f), cut(24, 0.5f);
  for (int r = 0; r < N; ++r) {
    ro[3 * r] = 0.f; ro[3 * r + 1] = 0.f; ro[3 * r + 2] = 3.f;
    rd[3 * r] = -0.25f + 0.5f * (float)(r % 16) / 15.f; rd[3 * r + 1] = -0.3f + 0.6f * (float)((r / 16) % 12) / 11.f; rd[3 * r + 2] = -1.f;
    tgt[3 * r] = 0.9f; tgt[3 * r + 1] = 0.2f + 0.5f * (float)(r % 7) / 6.f; tgt[3 * r + 2] = 0.1f;
  }
  for (int j = 0; j < 24; ++j) {                                         // world -> bone: identity rotation, joints on a helix
    float* m = &skt[16 * j];
    m[0] = m[5] = m[10] = m[15] = 1.f;
    m[3] = -0.3f * cosf(0.7f * j); m[7] = -(-0.8f + 1.6f * j / 23.f); m[11] = -0.3f * sinf(0.7f * j);
  }
  const float cyl[5] = {0.f, 0.f, 1.0f, -1.0f, 1.0f};                    // centre (x, z), radius, y range
  const float bg[3] = {1.f, 1.f, 1.f};
  float *d_ro = dalloc<float>(3 * N), *d_rd = dalloc<float>(3 * N), *d_tgt = dalloc<float>(3 * N), *d_skt = dalloc<float>(24 * 16),
        *d_cut = dalloc<float>(24), *d_cyl = dalloc<float>(5), *d_bg = dalloc<float>(3), *rays = dalloc<float>(11 * N);
  CK(hipMemcpy(d_ro, ro.data(), 12 * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_rd, rd.data(), 12 * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tgt, tgt.data(), 12 * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_skt, skt.data(), 24 * 16 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cut, cut.data(), 24 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_cyl, cyl, 20, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_bg, bg, 12, hipMemcpyHostToDevice));
  AK(anerf_make_ray_batch(d_ro, d_rd, N, 0.f, 1.f, 11, rays, st));

  // ---- outputs, random inputs, workspaces (the caller owns every byte)
  float *rgb = dalloc<float>(3 * N), *disp = dalloc<float>(N), *acc = dalloc<float>(N), *alpha = dalloc<float>((size_t)N * (S + NI)),
        *rgb0 = dalloc<float>(3 * N), *disp0 = dalloc<float>(N), *acc0 = dalloc<float>(N), *alpha0 = dalloc<float>((size_t)N * S);
  float *t_rand = dalloc<float>((size_t)N * S), *u_imp = dalloc<float>((size_t)N * NI), *noise = dalloc<float>((size_t)N * S),
        *noise_f = dalloc<float>((size_t)N * (S + NI));
  float *g_rgb = dalloc<float>(3 * N), *g_acc = dalloc<float>(N), *g_rgb0 = dalloc<float>(3 * N), *g_acc0 = dalloc<float>(N),
        *loss4 = dalloc<float>(4 * (ITERS + 1)), *lpart = dalloc<float>(4 * anerf_loss_blocks(N)), *apart = dalloc<float>(anerf_adam_blocks(NP)),
        *norms = dalloc<float>(2);
  const int64_t ws_bytes = anerf_train_workspace_size(&cfg, N, S, NI), sc_bytes = anerf_backward_scratch_size(&cfg, N, S, NI, 0);
  if (ws_bytes < 0 || sc_bytes < 0) { fprintf(stderr, "workspace size: %s\n", anerf_last_error()); return 3; }
  char *ws = dalloc<char>(ws_bytes), *scratch = dalloc<char>(sc_bytes);
  AnerfStepBlock* block = dalloc<AnerfStepBlock>(1);

  AnerfForwardIO io;
  memset(&io, 0, sizeof(io));
  io.packed_c = img[0][0]; io.aux_c = img[0][0] + L[0].stream_floats; io.packed_f = img[1][0]; io.aux_f = img[1][0] + L[0].stream_floats;
  io.rays = rays; io.ray_stride = 11; io.skts = d_skt; io.skt_ray_stride = 0; io.cyls = d_cyl; io.cyl_shared = 1;
  io.t_rand = t_rand; io.u_imp = u_imp; io.noise = noise; io.noise_fine = noise_f;
  io.cutoff_v = d_cut; io.cutoff_d = d_cut; io.tau_v = io.tau_d = 20.f;
  io.n_rays = N; io.n_samples = S; io.n_importance = NI;
  io.rgb_map = rgb; io.disp_map = disp; io.acc_map = acc; io.alpha = alpha; io.rgb0 = rgb0; io.disp0 = disp0; io.acc0 = acc0; io.alpha0 = alpha0;
  AnerfBackwardIO bw;
  memset(&bw, 0, sizeof(bw));
  bw.g_rgb = g_rgb; bw.g_acc = g_acc; bw.g_rgb0 = g_rgb0; bw.g_acc0 = g_acc0;
  bw.packed_t_c = img[0][1]; bw.packed_t_f = img[1][1]; bw.perm_x = perm_x; bw.perm_u = perm_u;
  bw.grads_c = ng[0]; bw.grads_f = ng[1]; bw.accumulate = 1;            // added to the (zeroed) flat gradient buffer, as FusedAdam does

  AnerfRandJob rj[4] = {{t_rand, (int64_t)N * S, 0, 1.f}, {u_imp, (int64_t)N * NI, 0, 1.f}, {noise, (int64_t)N * S, 1, 1.f},
                        {noise_f, (int64_t)N * (S + NI), 1, 1.f}};
  auto lr_of = [](int it) { return 5e-4f * powf(0.98f, (float)it); };
  auto tau_of = [](int it) { return 20.f * powf(1.01f, (float)it); };

  // one iteration; dev = false: scalars as arguments, true: from the step block (what gets captured)
  auto iteration = [&](int it, bool dev) {
    if (dev) AK(anerf_rand_fill_dev(rj, 4, block, 0, st)); else AK(anerf_rand_fill(rj, 4, SEED, (uint64_t)it, st));
    AK(anerf_pack_params_multi(jobs, 4, st));
    AnerfForwardIO f = io;
    if (dev) f.step = block; else f.tau_v = f.tau_d = tau_of(it);
    AK(anerf_train_forward(&cfg, &f, ws, ws_bytes, st));
    AK(anerf_loss(rgb, acc, rgb0, acc0, d_tgt, d_bg, 0, N, 0, 0.1f, 1.0f, loss4 + (dev || it > ITERS ? 0 : 4 * it), g_rgb, g_acc, g_rgb0, g_acc0, lpart, st));
    AK(anerf_backward(&cfg, &f, &bw, ws, ws_bytes, scratch, sc_bytes, st));
    if (dev) AK(anerf_adam_step_dev(P, G, M, V, NP, 0.9f, 0.999f, 1e-8f, block, 0, 1, 48, apart, norms, st));
    else AK(anerf_adam_step(P, G, M, V, NP, lr_of(it), 0.9f, 0.999f, 1e-8f, it, 1.0f, 1, 48, apart, norms, st));
  };

  // ---- (a) eager
  init_params();
  for (int it = 1; it <= ITERS; ++it) iteration(it, false);
  CK(hipStreamSynchronize(st));
  std::vector<float> pa(NP), la(4 * (ITERS + 1));
  CK(hipMemcpy(pa.data(), P, NP * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(la.data(), loss4, la.size() * 4, hipMemcpyDeviceToHost));

  // ---- (b) captured once, replayed
  init_params();
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  iteration(0, true);
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  size_t n_nodes = 0;
  CK(hipGraphGetNodes(graph, nullptr, &n_nodes));
  std::vector<float> lb(4 * (ITERS + 1), 0.f);
  for (int it = 1; it <= ITERS; ++it) {
    AnerfStepValues v;
    memset(&v, 0, sizeof(v));
    v.rng_seed = SEED; v.rng_offset = (uint64_t)it; v.tau_v = v.tau_d = tau_of(it);
    v.n_groups = 1; v.lr[0] = lr_of(it); v.beta1[0] = 0.9f; v.beta2[0] = 0.999f; v.adam_step[0] = it; v.grad_scale[0] = 1.0f;
    AK(anerf_step_block_write(block, &v, st));
    CK(hipGraphLaunch(exec, st));
    CK(hipMemcpyAsync(&lb[4 * it], loss4, 16, hipMemcpyDeviceToHost, st));    // the captured step writes its loss to slot 0
    CK(hipStreamSynchronize(st));
  }
  std::vector<float> pb(NP);
  CK(hipMemcpy(pb.data(), P, NP * 4, hipMemcpyDeviceToHost));

  // ---- verdict
  if (memcmp(pa.data(), pb.data(), NP * 4) != 0) { printf("MISMATCH: parameters differ between the eager and the captured run\n"); return 1; }
  for (int it = 1; it <= ITERS; ++it)
    if (memcmp(&la[4 * it], &lb[4 * it], 16) != 0) { printf("MISMATCH: loss of iteration %d: %.9g vs %.9g\n", it, la[4 * it], lb[4 * it]); return 1; }
  if (!(la[4] > 0.f) || !isfinite(la[4 * ITERS]) || !(la[4 * ITERS] < la[4])) { printf("loss did not go down: %.6f -> %.6f\n", la[4], la[4 * ITERS]); return 1; }
  if (TIMED > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms_eager = 0.f, ms_graph = 0.f;
    for (int w = 0; w < 3; ++w) iteration(ITERS + 1 + w, false);
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < TIMED; ++k) iteration(ITERS + 4 + k, false);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_eager, e0, e1));
    AnerfStepValues v;
    memset(&v, 0, sizeof(v));
    v.rng_seed = SEED; v.tau_v = v.tau_d = 20.f; v.n_groups = 1; v.lr[0] = 1e-4f; v.beta1[0] = 0.9f; v.beta2[0] = 0.999f; v.grad_scale[0] = 1.f;
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < TIMED; ++k) {
      v.rng_offset = 100 + k; v.adam_step[0] = ITERS + 1 + k;
      AK(anerf_step_block_write(block, &v, st));
      CK(hipGraphLaunch(exec, st));
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_graph, e0, e1));
    printf("train_step_demo timing: %d rays, %d iterations each: eager %.4f ms / iteration, captured graph %.4f ms / iteration (C++ host, HIP events)\n",
           N, TIMED, ms_eager / TIMED, ms_graph / TIMED);
  }
  printf("train_step_demo: %d iterations (%d rays, %d+%d samples) through the C ABI alone; ONE captured graph of %zu nodes replayed %d times, "
         "no node update; parameters and losses bit-identical to the eager run; loss %.6f -> %.6f\n", ITERS, N, S, NI, n_nodes, ITERS, la[4], la[4 * ITERS]);
  return 0;
}
