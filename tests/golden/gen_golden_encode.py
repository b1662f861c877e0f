#!/usr/bin/env python3
"""Golden vectors for rows A4-A6 alone: the reference's ENCODED network input (world->bone transform, RelDist / VecNorm
features, cutoff positional encodings; core/raycasters.py:476-577, core/encoders.py:8-193, core/cutoff_embedder.py:111-174)
for a few hundred samples, so that the fused kernel's encode stage is compared directly instead of through 11 layers.

  (a) eval, S = 32, one shared pose (the eval_s32 case): full rows `X = cat(v, r, d)` [.,1080] of 4 rays (128 samples);
  (b) train mode, pytest randomness, per-ray poses, 64 + 16 samples (the train_pytest case): the rows of BOTH network passes
      (coarse: the 64 jittered depths; fine: the 80 merged depths = the reference's gather-merge of its coarse and
      importance encodings) of 2 rays (288 samples), captured by wrapping the reference's own `RayCaster.encode_inputs`
      during its `render()` call, together with the depths it sampled (captured at sample_pts / sample_pts_is);
The generator also evaluates the reference's encoders in float64 and prints how far its own fp32 rows are from them (3e-7).

Run (build container only):  python tests/golden/gen_golden_encode.py   -> tests/golden/encode_rows.npz
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
gg = importlib.import_module("gen_golden")

EVAL_RAYS = [0, 27, 55, 95]
TRAIN_RAYS = [1, 17]               # one ray of two different poses of scene_batch(48, [4, 5, 6], per_ray_pose=True); printed below


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cp = gg.import_reference()
    from core.utils.ray_utils import get_near_far_in_cylinder, sample_from_lineseg
    t = gg.t
    t64 = lambda x: torch.tensor(np.asarray(x), dtype=torch.float64)
    args, caster, rk_train, rk_test = gg.build_caster(cp, "configs/surreal/surreal.txt", 11, 12)
    out = {}

    def encode(pts, ro, rd, kp, skts, bones, dt):
        c = lambda x: torch.as_tensor(np.asarray(x), dtype=dt) if not torch.is_tensor(x) else x.to(dt)
        with torch.no_grad():
            e = caster.encode_inputs(c(pts), [c(ro)[:, None], c(rd)[:, None]], c(kp), c(skts), c(bones),
                                     joint_coords=caster.get_subject_joint_coords(None, "cpu").to(dt), network=caster.network,
                                     **rk_test["preproc_kwargs"])
        return torch.cat([e["v"], e["r"], e["d"]], -1)

    # (a) the eval_s32 inputs
    caster.eval()
    ro, rd, kp, skts, bones, cyls, _ = gg.scene_batch(96, [0], ray_seed=1)
    near, far = get_near_far_in_cylinder(t(ro), t(rd), t(cyls), near=torch.zeros(96, 1), far=torch.ones(96, 1))
    z = sample_from_lineseg(near, far, 96, 32, perturb=0.)
    pts = t(ro)[:, None] + t(rd)[:, None] * z[..., None]
    X = encode(pts, ro, rd, kp, skts, bones, torch.float32)
    eval_in = (ro, rd, kp, skts, bones, z.clone())
    out["eval_rays"] = np.array(EVAL_RAYS)
    out["eval_z"] = z.numpy()
    out["eval_X"] = X[EVAL_RAYS].numpy()

    # (b) the train_pytest call; the reference's own depths and sort order are captured at sample_pts / sample_pts_is, its
    # encodings at encode_inputs
    caster.train()
    ro, rd, kp, skts, bones, cyls, which = gg.scene_batch(48, [4, 5, 6], ray_seed=4, per_ray_pose=True)
    print("train rays", TRAIN_RAYS, "-> poses", [int(which[i]) for i in TRAIN_RAYS])
    enc_calls, cap = [], {}
    real_enc, real_sp, real_spi = caster.encode_inputs, caster.sample_pts, caster.sample_pts_is

    def spy_enc(pts, *a, **k):
        e = real_enc(pts, *a, **k)
        enc_calls.append(torch.cat([e["v"], e["r"], e["d"]], -1).detach().clone())
        return e

    def spy_sp(*a, **k):
        pts, zv = real_sp(*a, **k)
        cap["z_coarse"] = zv.detach().clone()
        return pts, zv

    def spy_spi(*a, **k):
        r = real_spi(*a, **k)
        cap["z_merged"], cap["z_is"], cap["idx"] = r[1].detach().clone(), r[2].detach().clone(), r[3].detach().clone()
        return r
    caster.encode_inputs, caster.sample_pts, caster.sample_pts_is = spy_enc, spy_sp, spy_spi
    try:
        gg.run_render(rk_train, ro, rd, kp, t(skts), bones, cyls, pytest=True)
    finally:
        caster.encode_inputs, caster.sample_pts, caster.sample_pts_is = real_enc, real_sp, real_spi
    # the reference encodes the 64 coarse samples, then ONLY the 16 importance samples, and merges the two sets of encodings by
    # the sort order of the depths (_merge_encodings, raycasters.py:679-709; merge_samples :796-812); the fused kernel
    # re-encodes all 80 merged depths.  Rows of the fine pass = that merge, with the reference's own sorted_idxs.
    assert len(enc_calls) == 2 and enc_calls[0].shape == (48, 64, 1080) and enc_calls[1].shape == (48, 16, 1080)
    X_c, X_i = enc_calls
    X_m = torch.gather(torch.cat([X_c, X_i], 1), 1, cap["idx"][..., None].expand(-1, -1, 1080))
    assert torch.equal(torch.gather(torch.cat([cap["z_coarse"], cap["z_is"]], -1), 1, cap["idx"]), cap["z_merged"])
    out["train_rays"] = np.array(TRAIN_RAYS)
    out["train_z_coarse"], out["train_z_fine"] = cap["z_coarse"].numpy(), cap["z_merged"].numpy()
    out["train_X_coarse"], out["train_X_fine"] = X_c[TRAIN_RAYS].numpy(), X_m[TRAIN_RAYS].numpy()

    # yardstick, printed only: the same rows from the reference's encoders evaluated in float64 on the exact sample points
    # o + d z of the fp32 inputs.  The fp32 reference sits within 3e-7 of them (the cutoff gates zero every joint whose
    # distance is large enough for 2^6 x its rounding error to matter), so a direct 2e-6 gate on the kernel's rows is meaningful.
    caster.double()
    e_ro, e_rd, e_kp, e_skts, e_bones, e_z = eval_in
    p64 = t64(e_ro)[:, None] + t64(e_rd)[:, None] * e_z.double()[..., None]
    x64 = {"eval_X": encode(p64, e_ro, e_rd, e_kp, e_skts, e_bones, torch.float64)[EVAL_RAYS].numpy()}
    for tag, zz in (("coarse", cap["z_coarse"]), ("fine", cap["z_merged"])):
        p64 = t64(ro)[:, None] + t64(rd)[:, None] * zz.double()[..., None]
        x64[f"train_X_{tag}"] = encode(p64, ro, rd, kp, skts, bones, torch.float64)[TRAIN_RAYS].numpy()
    for k, v in x64.items():
        print(k, "max |fp32 reference - fp64 reference| =", float(np.abs(out[k] - v).max()))
    np.savez_compressed(os.path.join(gg.OUT, "encode_rows.npz"), **out)
    print({k: v.shape for k, v in out.items()}, os.path.getsize(os.path.join(gg.OUT, "encode_rows.npz")), "bytes")


if __name__ == "__main__":
    main()
