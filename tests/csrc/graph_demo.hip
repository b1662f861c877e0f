// graph_demo -- the C ABI's "stream-only" contract exercised WITHOUT torch: a piece of the training iteration (fresh random
// gradients from anerf_rand_fill_dev, then anerf_adam_step_dev) is captured into a hipGraph with the plain HIP runtime API and
// replayed; per iteration only anerf_step_block_write changes (Philox offset, learning rate, Adam step count) -- no node update.
// The result must be bit-identical to the same iterations issued eagerly through the by-value entry points
// (anerf_rand_fill / anerf_adam_step).  Built by __graft_entry__.build(), run by tests/test_graph_step.py on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "anerf.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, anerf_last_error()); return 3; } } while (0)

int main() {
  const int64_t n = (1 << 20) + 3;                 // a tail that is not a multiple of 4
  const size_t bytes = ((n + 3) / 4 * 4) * sizeof(float);
  const uint64_t seed = 0x1234ABCD5678ull;
  const int iters = 6;
  float *buf[2][4];                                // [eager | graph][p, g, m, v]
  for (int s = 0; s < 2; ++s)
    for (int k = 0; k < 4; ++k) { CK(hipMalloc(&buf[s][k], bytes)); CK(hipMemset(buf[s][k], 0, bytes)); }
  float *partials[2], *norms[2];
  const int nblk = anerf_adam_blocks(n);
  for (int s = 0; s < 2; ++s) { CK(hipMalloc(&partials[s], nblk * sizeof(float))); CK(hipMalloc(&norms[s], 2 * sizeof(float))); }
  AnerfStepBlock* block;
  CK(hipMalloc(&block, sizeof(AnerfStepBlock)));
  CK(hipMemset(block, 0, sizeof(AnerfStepBlock)));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // identical starting parameters on both sides: one eager draw each with the same (seed, offset)
  for (int s = 0; s < 2; ++s) {
    AnerfRandJob j0 = {buf[s][0], n, 1, 0.5f};
    AK(anerf_rand_fill(&j0, 1, seed, 999, st));
  }
  auto lr_of = [](int it) { return 5e-4f * (1.0f - 0.05f * (float)it); };

  // ---- eager: by-value entry points
  for (int it = 1; it <= iters; ++it) {
    AnerfRandJob jg = {buf[0][1], n, 1, 0.1f};
    AK(anerf_rand_fill(&jg, 1, seed, (uint64_t)it, st));
    AK(anerf_adam_step(buf[0][0], buf[0][1], buf[0][2], buf[0][3], n, lr_of(it), 0.9f, 0.999f, 1e-8f, it, 0.125f, 1, 24, partials[0], norms[0], st));
  }
  CK(hipStreamSynchronize(st));

  // ---- captured: one graph, replayed; the per-iteration scalars travel through the device-resident step block
  hipGraph_t graph;
  hipGraphExec_t exec;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  {
    AnerfRandJob jg = {buf[1][1], n, 1, 0.1f};
    AK(anerf_rand_fill_dev(&jg, 1, block, 0, st));
    AK(anerf_adam_step_dev(buf[1][0], buf[1][1], buf[1][2], buf[1][3], n, 0.9f, 0.999f, 1e-8f, block, 1, 1, 24, partials[1], norms[1], st));
  }
  CK(hipStreamEndCapture(st, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  for (int it = 1; it <= iters; ++it) {
    AnerfStepValues v;
    memset(&v, 0, sizeof(v));
    v.rng_seed = seed; v.rng_offset = (uint64_t)it; v.tau_v = v.tau_d = 20.f;
    v.n_groups = 2;                                // group 0 idle (step 0: its entries stay), group 1 = the one the graph steps
    v.lr[1] = lr_of(it); v.beta1[1] = 0.9f; v.beta2[1] = 0.999f; v.adam_step[1] = it; v.grad_scale[1] = 0.125f;
    AK(anerf_step_block_write(block, &v, st));
    CK(hipGraphLaunch(exec, st));
  }
  CK(hipStreamSynchronize(st));

  // ---- compare every byte of (p, m, v) and the gradient norms
  std::vector<char> a(bytes), b(bytes);
  for (int k : {0, 2, 3}) {
    CK(hipMemcpy(a.data(), buf[0][k], bytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), buf[1][k], bytes, hipMemcpyDeviceToHost));
    if (memcmp(a.data(), b.data(), (size_t)n * 4) != 0) { printf("MISMATCH in buffer %d\n", k); return 1; }
  }
  float na[2], nb[2];
  CK(hipMemcpy(na, norms[0], 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(nb, norms[1], 8, hipMemcpyDeviceToHost));
  if (memcmp(na, nb, 8) != 0 || !(na[0] > 0.f)) { printf("MISMATCH in norms %g %g vs %g %g\n", na[0], na[1], nb[0], nb[1]); return 1; }
  float p0;
  memcpy(&p0, a.data(), 4);
  printf("graph_demo: %d replays of ONE captured graph (no node update) bit-identical to the eager by-value calls; n = %lld, total_norm = %.6f\n",
         iters, (long long)n, na[0]);
  CK(hipGraphExecDestroy(exec));
  CK(hipGraphDestroy(graph));
  return 0;
}
