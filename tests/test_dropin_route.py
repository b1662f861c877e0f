"""GPU: the drop-in route with the inputs the reference's own caller builds.

run_nerf.render_path (run_nerf.py:62-88) hands every frame's pose to render() as `x.clone().expand(n_rays, ...)` -- stride-0
views -- and batchify_rays (trainer.py:64-79) slices them per 4096-ray chunk.  The caster must pass such inputs to the kernels as
their single row (zero ray stride: the shared-pose prologue of k_mlp_fwd, one cylinder for k_ray_bounds) instead of writing N
copies: same bits as the replicated form, and no allocation beyond the call's outputs and workspace.
"""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

from cases import build
from test_hip_backward import make_caster

pytestmark = pytest.mark.gpu

ops = importlib.import_module("a-nerf_amd.ops")
_lib = importlib.import_module("a-nerf_amd._lib")
synth = importlib.import_module("a-nerf_amd.synth")
render_mod = importlib.import_module("a-nerf_amd.render")


def dev(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32, device="cuda")


def frame_inputs(n, expand=True):
    """rays of the bench frame + the pose inputs as run_nerf.reuse_input builds them (expand) or replicated in memory"""
    sc = synth.make_scene(0, 512, 512, 600.0)
    rays = (dev(sc["rays_o"][:n]), dev(sc["rays_d"][:n]))
    reuse = lambda x, *sh: dev(x)[None].clone().expand(n, *sh)
    batch = dict(kp_batch=reuse(sc["pose"]["kp"], 24, 3), skts=reuse(sc["pose"]["skts"], 24, 4, 4), cyls=reuse(sc["cyl"], 5),
                 bones=reuse(sc["pose"]["bones"], 24, 3))
    if not expand:
        batch = {k: v.contiguous() for k, v in batch.items()}
    return rays, batch


RK = dict(perturb=False, N_importance=16, N_samples=64, use_viewdirs=True, raw_noise_std=0., ray_noise_std=0., ext_scale=0.001,
          preproc_kwargs={"density_scale": 1.0, "density_fn": torch.nn.functional.relu}, lindisp=False, nerf_type="nerf")


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_stride0_expanded_inputs_equal_replicated_inputs_bit_for_bit(precision):
    """render() -> batchify_rays(chunk 4096) -> RayCaster over 3 chunks (ragged last one), 64 + 16 samples"""
    caster = make_caster(build("eval_hier")).eval()
    caster.render_precision = precision
    n = 2 * 4096 + 777
    outs = []
    for expand in (True, False):
        rays, batch = frame_inputs(n, expand)
        assert ops.row_shared(batch["skts"]) == expand
        with torch.no_grad():
            outs.append(render_mod.render(512, 512, 600.0, rays=rays, chunk=4096, ray_caster=caster, cams=None, subject_idxs=None,
                                          **batch, **RK))
    assert set(outs[0]) == {"rgb_map", "disp_map", "acc_map", "alpha", "rgb0", "disp0", "acc0", "alpha0"}
    for k in outs[0]:
        assert outs[0][k].shape[0] == n
        assert torch.equal(outs[0][k], outs[1][k]), k
    assert torch.isfinite(outs[0]["rgb_map"]).all() and float(outs[0]["acc_map"].max()) > 0.1


def test_a_chunk_call_allocates_its_outputs_and_workspace_only():
    """One 4096-ray caster call with stride-0 expanded inputs: the bytes the caching allocator hands out during the call are the
    output maps + the library's workspace (+ small change); the replicated [4096,24,4,4] pose (6.3 MB) is never written.  Every
    single allocation of the call is listed from the allocator's own trace: none of 1 MB or more besides the workspace and the
    alpha map (4096 x 64 floats = exactly 1 MiB)."""
    caster = make_caster(build("eval_s32")).eval()
    n, S = 4096, 64
    rays, batch = frame_inputs(n, True)
    rb = ops.make_ray_batch(*rays)
    rk = dict(RK, N_importance=0, N_samples=S)
    rk.pop("use_viewdirs")                          # consumed by render(); the caster call itself does not take it
    with torch.no_grad():
        caster(rb, **batch, **rk)                   # warm: weight image packed, tables cached
        torch.cuda.synchronize()
        cc = caster.network.path_cfg.c()
        ws_bytes = int(_lib.load().anerf_workspace_size(C.byref(cc), n, S, 0))
        out_bytes = 4 * (n * 3 + n + n + n * S)
        torch.cuda.memory._record_memory_history(max_entries=10000)
        a0 = torch.cuda.memory_stats()["allocated_bytes.all.allocated"]
        ret = caster(rb, **batch, **rk)
        a1 = torch.cuda.memory_stats()["allocated_bytes.all.allocated"]
        snap = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
    allocs = [e["size"] for tr in snap["device_traces"] for e in tr if e["action"] == "alloc"]
    print(f"chunk call: {a1 - a0} B allocated in {len(allocs)} allocations; outputs {out_bytes} B + workspace {ws_bytes} B; sizes {sorted(allocs)[-6:]}")
    assert a1 - a0 <= out_bytes + ws_bytes + 64 * 1024
    big = sorted(s for s in allocs if s >= (1 << 20))
    assert len(big) <= 2 and all(s <= max(ws_bytes + 512, n * S * 4) for s in big), big
    assert ret["rgb_map"].shape == (n, 3)
    # and the replicated form of the same call does pay for the copy (the test can see what it guards against)
    with torch.no_grad():
        rep = {k: v.contiguous() for k, v in batch.items()}
        b0 = torch.cuda.memory_stats()["allocated_bytes.all.allocated"]
        ret2 = caster(rb, **rep, **rk)
        b1 = torch.cuda.memory_stats()["allocated_bytes.all.allocated"]
    assert torch.equal(ret2["rgb_map"], ret["rgb_map"])
    assert b1 - b0 <= out_bytes + ws_bytes + 64 * 1024         # already contiguous: no copy either


def test_shared_cylinder_matches_per_ray_cylinders_incl_nan_fallback():
    """cyl_shared (ABI revision 4) against the per-ray [N,5] form on the NaN-fallback case (rays that miss the cylinder take the
    call's mean near / far, ray_utils.py:327-339): bit-equal outputs"""
    c = build("nan_fallback")
    cfg = ops.PathConfig()
    net = ops.pack_params(cfg, {k: dev(v) for k, v in c["Pc"].items()})
    rb = ops.make_ray_batch(dev(c["rays_o"]), dev(c["rays_d"]))
    cyl = dev(c["cyls"])
    assert float((cyl - cyl[:1]).abs().max()) == 0.0          # one pose: all rows equal
    skt = dev(c["skts"])
    per_ray = ops.forward(cfg, net, None, rb, skt, cyl, c["S"], 0)
    shared = ops.forward(cfg, net, None, rb, skt, cyl[:1].expand(c["n"], -1), c["S"], 0)
    one_row = ops.forward(cfg, net, None, rb, skt, cyl[:1], c["S"], 0)
    for k in per_ray:
        assert torch.equal(per_ray[k], shared[k]) and torch.equal(per_ray[k], one_row[k]), k


def test_render_staticcam_and_tensor_near_far():
    """render(c2w_staticcam=...) (trainer.py:118-122: the second camera's origins / directions for the full H x W image, the given
    rays only as view directions -- which the path slices off, raycasters.py:415) equals rendering the second camera's rays
    directly; near / far given as tensors (the reference multiplies them into ones_like, trainer.py:131) equal the scalar form."""
    caster = make_caster(build("eval_s32")).eval()
    H = W = 24
    focal = 30.0
    sc = synth.make_scene(0, H, W, focal)
    c2w_a = dev(synth.default_c2w())
    c2w_b = c2w_a.clone()
    c2w_b[0, 3] += 0.3                      # a second camera, shifted sideways
    rb_a, _ = ops.gen_rays(H, W, focal, c2w_a, (0, 0, W, H))
    rb_b, _ = ops.gen_rays(H, W, focal, c2w_b, (0, 0, W, H))
    n = H * W
    reuse = lambda x, *sh: dev(x)[None].clone().expand(n, *sh)
    batch = dict(kp_batch=reuse(sc["pose"]["kp"], 24, 3), skts=reuse(sc["pose"]["skts"], 24, 4, 4), cyls=reuse(sc["cyl"], 5),
                 bones=reuse(sc["pose"]["bones"], 24, 3))
    rk = dict(RK, N_importance=0, N_samples=32)
    with torch.no_grad():
        direct = render_mod.render(H, W, focal, rays=(rb_b[:, 0:3], rb_b[:, 3:6]), chunk=256, ray_caster=caster, cams=None, subject_idxs=None,
                                   **batch, **rk)
        static = render_mod.render(H, W, focal, rays=(rb_a[:, 0:3].reshape(H, W, 3), rb_a[:, 3:6].reshape(H, W, 3)), chunk=256,
                                   c2w_staticcam=c2w_b[:3, :4], ray_caster=caster, cams=None, subject_idxs=None, **batch, **rk)
        tens = render_mod.render(H, W, focal, rays=(rb_b[:, 0:3], rb_b[:, 3:6]), chunk=256, near=torch.zeros(n, 1, device="cuda"),
                                 far=torch.ones(1, device="cuda"), ray_caster=caster, cams=None, subject_idxs=None, **batch, **rk)
    assert static["rgb_map"].shape == (H, W, 3) and static["alpha"].shape == (H, W, 32)
    for k in direct:
        assert torch.equal(static[k].reshape(direct[k].shape), direct[k]), k
        assert torch.equal(tens[k], direct[k]), k
    assert float(direct["acc_map"].max()) > 0.1
