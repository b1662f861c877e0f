"""Pose-refinement layer: mirror of the reference's PoseOptLayer (core/pose_opt.py:242-445) on the
fused forward-kinematics kernels (anerf_fk.hip).

Same constructor arguments, parameters (`pelvis`, `bones`), buffers (`rest_pose`), `forward(idxs)` 5-tuple
`(kp, bones, skts, l2ws, rots)`, `calculate_kinematic`, `update_cache`, so `poseopt_layer_state_dict` checkpoints
(core/trainer.py:498-505, pose_opt.py:211-240) load unchanged.  Supported set = what the shipped configs use: SMPL
skeleton, axis-angle bones, one rest pose shared by all poses or one per pose; `use_rot6d` and the multi-view `kp_map`
variant raise NotImplementedError at construction.
`skts` returned here feed RayCaster.render_rays(skts=...); their gradient (the hot path's dskts) flows back to
`bones` / `pelvis` through one backward kernel.  No CPU fallback.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops


class _FkFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, bones, pelvis, rest_pose):
        out = ops.fk_forward(bones, rest_pose, pelvis)
        ctx.save_for_backward(bones, pelvis, rest_pose)
        return out["kp"], out["skts"], out["l2ws"], out["rots"]

    @staticmethod
    def backward(ctx, g_kp, g_skts, g_l2ws, g_rots):
        bones, pelvis, rest_pose = ctx.saved_tensors
        gb, gp = ops.fk_backward(bones, rest_pose, pelvis, g_skts=g_skts, g_l2ws=g_l2ws, g_kp=g_kp, g_rots=g_rots)
        return gb, gp, None


class _RepeatPoses(torch.autograd.Function):
    """x_u -> each row repeated r times (the per-ray view of per-pose data when every pose owns r consecutive rays, which
    is how the reference's sampler lays a batch out: N_sample_images images x N_rand / N_sample_images rays).  Backward is
    one reshaped sum -- deterministic -- instead of torch's sort-based index backward (0.12 ms per call at 3072 rays)."""

    @staticmethod
    def forward(ctx, x_u, r, tiled):
        ctx.r, ctx.tiled = r, tiled
        if tiled:                                                   # rays cycle through the poses: 0 1 .. u-1 0 1 ..
            return x_u.repeat((r,) + (1,) * (x_u.dim() - 1))
        return x_u.repeat_interleave(r, dim=0)                      # r consecutive rays per pose

    @staticmethod
    def backward(ctx, g):
        u = g.shape[0] // ctx.r
        if ctx.tiled:
            return g.reshape((ctx.r, u) + tuple(g.shape[1:])).sum(0), None, None
        return g.reshape((u, ctx.r) + tuple(g.shape[1:])).sum(1), None, None


def expand_poses(x_u, inverse_idxs):
    """x_u[inverse_idxs] (pose_opt.py:433-437); inverse_idxs is a HOST array, so its structure is known without a sync."""
    inv = np.asarray(inverse_idxs)
    u = x_u.shape[0]
    if u > 0 and len(inv) % u == 0:
        r = len(inv) // u
        if np.array_equal(inv, np.repeat(np.arange(u), r)):
            return x_u if r == 1 else _RepeatPoses.apply(x_u, r, False)
        if np.array_equal(inv, np.tile(np.arange(u), r)):
            return _RepeatPoses.apply(x_u, r, True)
    return x_u[torch.as_tensor(inv, device=x_u.device)]


def calculate_kinematic(bones, pelvis, rest_pose):
    """(kp, skts, l2ws, rots) of axis-angle `bones` [U,24,3] (+ `pelvis` [U,3]); differentiable w.r.t. both."""
    return _FkFn.apply(bones, pelvis, rest_pose)


class PoseOptLayer(nn.Module):
    def __init__(self, kps, bones, rest_pose, skel_type=None, kp_map=None, kp_uidxs=None, use_cache=False, use_rot6d=False,
                 beta=None, rest_pose_idxs=None):
        super().__init__()
        if use_rot6d:
            raise NotImplementedError("rot6d bones: not in the fused FK set (axis-angle only)")
        if kp_map is not None or kp_uidxs is not None:
            raise NotImplementedError("multi-view kp_map: not in the fused FK set")
        if skel_type is not None and getattr(skel_type, "root_id", 0) != 0:
            raise NotImplementedError("only the SMPL skeleton (root_id 0) is supported")
        kps, bones = torch.as_tensor(kps, dtype=torch.float32), torch.as_tensor(bones, dtype=torch.float32)
        if bones.dim() != 3 or bones.shape[1:] != (24, 3):
            raise NotImplementedError(f"bones must be [N,24,3] axis-angle, got {tuple(bones.shape)}")
        self.use_cache = use_cache
        self.use_rot6d = False
        self.unroll_kinematic_chain = True
        self.root_id = 0
        self.kp_map = self.kp_uidxs = None
        self.rest_pose_idxs = rest_pose_idxs
        self.beta = torch.as_tensor(beta) if beta is not None else None
        self.register_buffer("rest_pose", torch.as_tensor(rest_pose, dtype=torch.float32).reshape(-1, 24, 3).clone())
        self.register_parameter("pelvis", nn.Parameter(kps[:, 0].clone()))
        self.register_parameter("bones", nn.Parameter(bones.clone()))
        self.N_kps = self.pelvis.shape[0]
        if use_cache:
            self.update_cache()

    def get_rest_pose(self, kp_idxs=None, rest_pose_idxs=None):
        if len(self.rest_pose) == 1:
            return self.rest_pose
        if rest_pose_idxs is not None:
            return self.rest_pose[rest_pose_idxs]
        return self.rest_pose[self.rest_pose_idxs[kp_idxs]]

    def get_pelvis(self):
        return self.pelvis

    def idx_to_params(self, idx):
        return self.pelvis[idx], self.bones[idx]

    def calculate_kinematic(self, idxs, rest_pose_idxs=None):
        if idxs is None:
            idxs = np.arange(len(self.pelvis))
        idxs = np.atleast_1d(np.asarray(idxs.cpu() if torch.is_tensor(idxs) else idxs))
        unique_idxs, inverse_idxs = np.unique(idxs, return_inverse=True)       # FK once per distinct pose
        rest = self.get_rest_pose(unique_idxs, rest_pose_idxs)
        pelvis, bone = self.idx_to_params(unique_idxs)
        kp, skts, l2ws, rots = calculate_kinematic(bone.contiguous(), pelvis.contiguous(), rest)
        return tuple(expand_poses(t, inverse_idxs) for t in (kp, bone, skts, l2ws, rots))

    @torch.no_grad()
    def update_cache(self):
        kps, bones, skts, l2ws, rots = self.calculate_kinematic(np.arange(len(self.pelvis)))
        self.cache_kps, self.cache_bones, self.cache_skts, self.cache_l2ws, self.cache_rots = kps, bones, skts, l2ws, rots

    def forward(self, idxs, rest_pose_idxs=None):
        if not self.use_cache:
            return self.calculate_kinematic(idxs, rest_pose_idxs)
        return self.cache_kps[idxs], self.cache_bones[idxs], self.cache_skts[idxs], self.cache_l2ws[idxs], self.cache_rots[idxs]


def load_poseopt_from_state_dict(state_dict):
    """pose_opt.py:211-240: rebuild the layer from a checkpoint's `poseopt_layer_state_dict`."""
    sd = state_dict["poseopt_layer_state_dict"]
    if "kp_map" in sd or sd["bones"].shape[-1] != 3:
        raise NotImplementedError("multi-view / rot6d pose checkpoints are outside the fused FK set")
    n = sd["pelvis"].shape[0]
    layer = PoseOptLayer(torch.zeros(n, 24, 3), torch.zeros(n, 24, 3), torch.zeros(sd["rest_pose"].shape))
    layer.load_state_dict(sd)
    return layer
