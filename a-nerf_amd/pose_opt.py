"""Pose-refinement layer: mirror of the reference's PoseOptLayer (core/pose_opt.py:242-445) on the
fused forward-kinematics kernels (anerf_fk.hip).

Same constructor arguments, parameters (`pelvis`, `bones`), buffers (`rest_pose`), `forward(idxs)` 5-tuple
`(kp, bones, skts, l2ws, rots)`, `calculate_kinematic`, `update_cache`, so `poseopt_layer_state_dict` checkpoints
(core/trainer.py:498-505, pose_opt.py:211-240) load unchanged.  Supported set = what the shipped configs use: SMPL
skeleton; axis-angle bones or `use_rot6d` 6D rotations (opt_rot6d = True in the mixamo / h36m / perfcap configs); one
rest pose shared by all poses or one per pose; the multi-view `kp_map` variant (h36m: per-view root rotation +
shared body bones, pose_opt.py:293-296,318-331).
`skts` returned here feed RayCaster.render_rays(skts=...); their gradient (the hot path's dskts) flows back to
`bones` / `pelvis` through one backward kernel.  No CPU fallback.
"""
import weakref

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


class _PoseBatchFn(torch.autograd.Function):
    """PoseOptLayer.forward for one batch (parameter lookup -> FK per distinct pose -> per-ray rows) as ONE launch each way
    (anerf_pose_batch_forward / _backward; core/pose_opt.py:318-331,372-445 and its autograd).  Outputs: the five per-ray
    tensors of the reference's layer, then the unique-level kp / bones / rots the pose regulariser reads.
    Gradients reach the parameters either as dense tensors through autograd (rows of the batch's poses written into zeros), or
    -- when the layer is attached to a FusedAdam whose flat bucket owns `pelvis.grad` / `bones.grad` -- added in place by the
    kernel (no zero fill, no AccumulateGrad add), and autograd is told there is nothing to accumulate."""

    @staticmethod
    def forward(ctx, pelvis, bones, rest, pose_idx, inverse, layer_ref):
        uniq, rays = ops.pose_batch_forward(bones, pelvis, rest, pose_idx, inverse)
        ctx.save_for_backward(pelvis, bones, rest, pose_idx, inverse)
        ctx.layer_ref = layer_ref
        ctx.set_materialize_grads(False)
        return rays["kp"], rays["bones"], rays["skts"], rays["l2ws"], rays["rots"], uniq["kp"], uniq["bones"], uniq["rots"]

    @staticmethod
    def backward(ctx, g_kp, g_bones, g_skts, g_l2ws, g_rots, gu_kp, gu_bones, gu_rots):
        pelvis, bones, rest, pose_idx, inverse = ctx.saved_tensors
        layer = ctx.layer_ref() if ctx.layer_ref is not None else None
        sink = getattr(layer, "_anerf_grad_sink", None) if layer is not None else None
        sink = sink() if sink is not None else None
        # (saved tensors come back as new Python objects: compare storage, not identity)
        direct = sink is not None and layer.pelvis.data_ptr() == pelvis.data_ptr() and layer.bones.data_ptr() == bones.data_ptr() and \
            sink.owns_grads([layer.pelvis, layer.bones], bones.device)
        if direct:
            gb, gp = layer.bones.grad, layer.pelvis.grad
        else:
            gb, gp = torch.zeros_like(bones), torch.zeros_like(pelvis)
        ops.pose_batch_backward(bones, pelvis, rest, pose_idx, inverse, dict(kp=g_kp, bones=g_bones, skts=g_skts, l2ws=g_l2ws, rots=g_rots),
                                dict(kp=gu_kp, bones=gu_bones, rots=gu_rots), gb, gp, accumulate=direct)
        return (None, None, None, None, None, None) if direct else (gp, gb, None, None, None, None)


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


class _KpLossFn(torch.autograd.Function):
    """(values, ..., base) -> kp_loss, or (kp_loss, base + kp_loss) when a base loss is given: the regulariser AND the trainer's
    `total = rgb losses + kp_loss` in the one launch (anerf_kp_loss_add, ABI revision 7)."""

    @staticmethod
    def forward(ctx, values, anchors, weights, rot6d, tol, coef, base):
        import ctypes as C
        from . import _lib
        values, anchors, weights = ops._f32c(values, "values"), ops._f32c(anchors, "anchors"), ops._f32c(weights, "weights")
        u = values.shape[0]
        out = torch.empty(2, dtype=torch.float32, device=values.device)          # [kp_loss, base + kp_loss]
        g = torch.empty_like(values) if ctx.needs_input_grad[0] else None
        ctx.g, ctx.with_base = g, base is not None
        ctx.set_materialize_grads(False)
        if base is None:
            _lib.check(_lib.load().anerf_kp_loss(ops._p(values), int(bool(rot6d)), ops._p(anchors), ops._p(weights), u, float(tol), float(coef),
                                                 ops._p(out), ops._p(g), ops._stream()), "anerf_kp_loss")
            return out[0]
        if base.numel() != 1 or base.dtype != torch.float32 or base.device != values.device:
            raise ValueError("kp_loss(add_to=...): a float32 scalar on the values' device")
        _lib.check(_lib.load().anerf_kp_loss_add(ops._p(values), int(bool(rot6d)), ops._p(anchors), ops._p(weights), u, float(tol), float(coef),
                                                 C.c_void_p(base.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 4),
                                                 ops._p(g), ops._stream()), "anerf_kp_loss_add")
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_kp, g_total=None):
        from .optim import is_unit_seed
        # upstream of `values`: through kp_loss itself and / or through the total; the unit seed of optim.backward() passes unscaled
        ups = [u for u in (g_kp, g_total) if u is not None]
        gv = None
        if ctx.g is not None and ups:
            up = ups[0] if len(ups) == 1 else ups[0] + ups[1]
            gv = ctx.g if is_unit_seed(up) else ctx.g * up
        return gv, None, None, None, None, None, (g_total if ctx.with_base else None)


def kp_loss(values, anchors, pose_weights, rot6d, tol, coef, add_to=None):
    """Trainer._compute_kp_loss (core/trainer.py:382-403) over the U distinct poses of a batch in one launch each way.
    values: rots [U,24,3,3] (rot6d = True; the reference compares rots[..., :3, :2]) or axis-angle bones [U,24,3];
    anchors [U,24,6] / [U,24,3] (popt_anchors of those poses); pose_weights [U] = rays of pose u / N.
    add_to: a scalar loss tensor (the trainer's total so far) -> returns (kp_loss, add_to + kp_loss), the sum formed by the same
    launch (no separate add kernel); differentiable w.r.t. both."""
    want = (24, 3, 3) if rot6d else (24, 3)
    if tuple(values.shape[1:]) != want or tuple(anchors.shape[1:]) != ((24, 6) if rot6d else (24, 3)) or anchors.shape[0] != values.shape[0]:
        raise ValueError(f"kp_loss: values {tuple(values.shape)} / anchors {tuple(anchors.shape)} do not match rot6d={rot6d}")
    return _KpLossFn.apply(values, anchors, pose_weights, rot6d, tol, coef, add_to)


def calculate_kinematic(bones, pelvis, rest_pose):
    """(kp, skts, l2ws, rots) of axis-angle `bones` [U,24,3] or rot6d `bones` [U,24,6] (+ `pelvis` [U,3]);
    differentiable w.r.t. both."""
    return _FkFn.apply(bones, pelvis, rest_pose)


# ---- host-side rotation conversions: parameter initialisation and pose export only (never on the per-step path, which
# is the FK kernel).  pytorch3d (absent here) is what the reference calls; restated from its published formulae.
def axisang_to_rot(a):
    """skeleton_utils.py:411 axisang_to_rot = pytorch3d axis_angle_to_matrix: [...,3] -> [...,3,3] (same map as anerf_fk.hip)."""
    th = torch.linalg.norm(a, dim=-1, keepdim=True)
    half = 0.5 * th
    small = th.abs() < 1e-6
    k = torch.where(small, 0.5 - th * th / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(th), th))
    r, (i, j, kk) = torch.cos(half)[..., 0], (a * k).unbind(-1)
    two_s = 2.0 / (r * r + i * i + j * j + kk * kk)
    o = torch.stack([1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
                     two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
                     two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)], -1)
    return o.reshape(a.shape[:-1] + (3, 3))


def rot6d_to_rotmat(x):
    """skeleton_utils.py:420-436 (Zhou et al. 2019): [...,6] -> [...,3,3]; x = the first two columns of R, row-major."""
    sh = x.shape[:-1]
    x = x.reshape(-1, 3, 2)
    b1 = torch.nn.functional.normalize(x[:, :, 0], dim=-1)
    a2 = x[:, :, 1]
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.linalg.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-1).reshape(*sh, 3, 3)


def rot_to_rot6d(rot):
    """skeleton_utils.py:408"""
    return rot[..., :3, :2].flatten(start_dim=-2)


def rot_to_axisang(rot):
    """skeleton_utils.py:405 rot_to_axisang = pytorch3d matrix_to_axis_angle: [...,3,3] -> [...,3], angle in [0, pi].
    Through the best-conditioned of the four quaternion candidates (Shepperd), as pytorch3d's matrix_to_quaternion."""
    m = rot[..., :3, :3]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[..., r, c] for r in range(3) for c in range(3)]
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22,
                                                1 - m00 - m11 + m22], -1), min=0.0))
    cand = torch.stack([torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
                        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
                        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4)))[..., 0, :]
    q = torch.where(q[..., :1] < 0, -q, q)                                  # real part >= 0: angle in [0, pi]
    n = torch.linalg.norm(q[..., 1:], dim=-1, keepdim=True)
    half = torch.atan2(n, q[..., :1])
    ang = 2.0 * half
    small = ang.abs() < 1e-6
    sin_half_over_ang = torch.where(small, 0.5 - ang * ang / 48.0, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return q[..., 1:] / sin_half_over_ang


def rot6d_to_axisang(x):
    """skeleton_utils.py:417"""
    return rot_to_axisang(rot6d_to_rotmat(x))


class PoseOptLayer(nn.Module):
    def __init__(self, kps, bones, rest_pose, skel_type=None, kp_map=None, kp_uidxs=None, use_cache=False, use_rot6d=False,
                 beta=None, rest_pose_idxs=None):
        super().__init__()
        if skel_type is not None and getattr(skel_type, "root_id", 0) != 0:
            raise NotImplementedError("only the SMPL skeleton (root_id 0) is supported")
        kps, bones = torch.as_tensor(kps, dtype=torch.float32), torch.as_tensor(bones, dtype=torch.float32)
        if bones.dim() != 3 or bones.shape[1:] != (24, 3):
            raise NotImplementedError(f"bones must be [N,24,3] axis-angle, got {tuple(bones.shape)}")
        self.use_cache = use_cache
        self.use_rot6d = bool(use_rot6d)
        self.unroll_kinematic_chain = True
        self.root_id = 0
        self.rest_pose_idxs = rest_pose_idxs
        if kp_map is not None:                                       # multi-view (pose_opt.py:259-263)
            self.register_buffer("kp_map", torch.as_tensor(np.asarray(kp_map)).long())
            self.register_buffer("kp_uidxs", torch.as_tensor(np.asarray(kp_uidxs)).long())
        else:
            self.kp_map = self.kp_uidxs = None
        self.beta = torch.as_tensor(beta) if beta is not None else None
        self.register_buffer("rest_pose", torch.as_tensor(rest_pose, dtype=torch.float32).reshape(-1, 24, 3).clone())
        self.register_parameter("pelvis", nn.Parameter(kps[:, 0].clone()))
        if self.use_rot6d:                                           # pose_opt.py:284-289: first two columns of R
            bones = rot_to_rot6d(axisang_to_rot(bones))
        if self.kp_map is None:
            self.register_parameter("bones", nn.Parameter(bones.clone()))
        else:                                                        # per-view root rotation, body bones shared by the views
            self.register_parameter("root_bones", nn.Parameter(bones[:, 0].clone()))
            self.register_parameter("bones", nn.Parameter(bones[self.kp_uidxs.to(bones.device), 1:].clone()))
        self.N_kps = self.pelvis.shape[0]
        if use_cache:
            self.update_cache()

    def get_rest_pose(self, kp_idxs=None, rest_pose_idxs=None):
        if len(self.rest_pose) == 1:
            return self.rest_pose
        if rest_pose_idxs is not None:
            return self.rest_pose[rest_pose_idxs]
        return self.rest_pose[self.rest_pose_idxs[kp_idxs]]

    def _device_index(self, idx):
        """pose indices as an int64 device tensor.  A host array is uploaded ONCE per distinct content and kept: the upload of
        pageable host memory is a blocking copy queued behind everything already on the stream -- one per training step made
        every step end in a host sync (the reference pays it too: `kp_idx.cpu().numpy()`, trainer.py:297-299)."""
        if torch.is_tensor(idx):
            return idx.to(self.pelvis.device).long().reshape(-1)
        arr = np.ascontiguousarray(np.asarray(idx).reshape(-1).astype(np.int64))
        cache = self.__dict__.setdefault("_idx_cache", {})
        key = (arr.tobytes(), str(self.pelvis.device))
        hit = cache.get(key)
        if hit is None:
            if len(cache) > 256:
                cache.clear()
            hit = cache[key] = torch.as_tensor(arr, device=self.pelvis.device)
        return hit

    def idx_to_params(self, idx):
        """pose_opt.py:318-331"""
        idx = self._device_index(idx)
        # index_select: its backward is one index_add_ (idx holds distinct poses on the training path); `tensor[idx]` goes
        # through index_put_'s sort-based backward (a radix sort + two kernels per parameter and step)
        sel = lambda t, i: torch.index_select(t, 0, i)
        if self.kp_map is None:
            return sel(self.pelvis, idx), sel(self.bones, idx)
        return sel(self.pelvis, idx), torch.cat([sel(self.root_bones, idx)[:, None, :], sel(self.bones, self.kp_map[idx])], dim=1)

    def get_pelvis(self, idx=None):
        return self.idx_to_params(np.arange(self.N_kps) if idx is None else idx)[0]

    def get_beta(self):
        return self.beta

    def get_bones(self, idx=None):
        """axis-angle bones of the (refined) poses, whatever the parametrisation (pose_opt.py:342-350)"""
        bones = self.idx_to_params(np.arange(self.N_kps) if idx is None else idx)[1]
        return rot6d_to_axisang(bones) if self.use_rot6d else bones

    def to_bones3d(self, bones):
        return bones if bones.shape[-1] == 3 else rot6d_to_axisang(bones)

    def stage_batch(self, idxs):
        """For a training step replayed from a captured hipGraph (graph_step.GraphedTrainStep): group the batch's rays by pose on the
        host and upload (distinct pose rows [U] int64, each ray's slot [N] int32, each pose's share of the rays [U] float32) into
        PERSISTENT device tensors -- one set per (U, N) -- from pinned staging, stream-ordered.  Call it OUTSIDE the captured region,
        before the step: the following `forward(idxs)` with the same indices then reads those tensors and uploads nothing, so the
        captured kernels see static addresses whose contents follow the batch.  Returns (U, N): batches with another number of
        distinct poses need their own graph."""
        arr = np.ascontiguousarray(np.asarray(idxs.cpu() if torch.is_tensor(idxs) else idxs).reshape(-1).astype(np.int64))
        uniq, inv = np.unique(arr, return_inverse=True)
        counts = np.bincount(inv, minlength=len(uniq))
        dev = self.pelvis.device
        slots = self.__dict__.setdefault("_static_idx", {})
        key = (len(uniq), len(arr))
        sl = slots.get((key, str(dev)))
        if sl is None:
            sl = slots[(key, str(dev))] = (torch.empty(len(uniq), dtype=torch.int64, device=dev), torch.empty(len(arr), dtype=torch.int32, device=dev),
                                           torch.empty(len(uniq), dtype=torch.float32, device=dev))
        # staging: a small PERSISTENT pinned ring per (U, N) -- three slots, an event per slot -- instead of three fresh pin_memory()
        # tensors per iteration (pinning costs ~0.2 ms per tensor, tools/leases/r05_probe_upload.py; and a pinned ALLOCATION from
        # another thread aborts a capture in "global" capture-error mode).  A slot is rewritten only after the copies that read it
        # have run (event.synchronize(): three iterations back, normally long done).
        rings = self.__dict__.setdefault("_static_ring", {})
        ring = rings.get((key, str(dev)))
        if ring is None:
            mk = lambda: (torch.empty(len(uniq), dtype=torch.int64).pin_memory(), torch.empty(len(arr), dtype=torch.int32).pin_memory(),
                          torch.empty(len(uniq), dtype=torch.float32).pin_memory(), torch.cuda.Event())
            ring = rings[(key, str(dev))] = {"slots": [mk() for _ in range(3)], "next": 0, "used": [False, False, False]}
        k = ring["next"]
        ring["next"] = (k + 1) % 3
        h_idx, h_inv, h_w, ev = ring["slots"][k]
        if ring["used"][k]:
            ev.synchronize()
        h_idx.numpy()[:] = uniq
        h_inv.numpy()[:] = inv
        h_w.numpy()[:] = counts / float(len(arr))
        sl[0].copy_(h_idx, non_blocking=True)
        sl[1].copy_(h_inv, non_blocking=True)
        sl[2].copy_(h_w, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        ring["used"][k] = True
        self.__dict__["_staged"] = (arr.tobytes(), sl, uniq, counts)
        return key

    def _batch_index(self, idxs):
        """(unique pose rows int64 [U], inverse int32 [N]) on the device + the host arrays; uploaded once per distinct batch layout
        (a blocking copy of pageable memory per step would be a host sync per step); or the persistent tensors stage_batch() just
        filled for exactly these indices"""
        arr = np.ascontiguousarray(np.asarray(idxs).reshape(-1).astype(np.int64))
        st = self.__dict__.get("_staged")
        if st is not None and st[1][0].device == self.pelvis.device and st[0] == arr.tobytes():
            return st[1][0], st[1][1], st[2], st[3]
        cache = self.__dict__.setdefault("_batch_cache", {})
        key = (arr.tobytes(), str(self.pelvis.device))
        hit = cache.get(key)
        if hit is None:
            if len(cache) > 256:
                cache.clear()
            uniq, inv = np.unique(arr, return_inverse=True)
            dev = self.pelvis.device
            hit = cache[key] = (torch.as_tensor(uniq.astype(np.int64), device=dev), torch.as_tensor(inv.astype(np.int32), device=dev), uniq,
                                np.bincount(inv, minlength=len(uniq)))
        return hit

    def calculate_kinematic(self, idxs, rest_pose_idxs=None):
        if idxs is None:
            idxs = np.arange(len(self.pelvis))
        idxs = np.atleast_1d(np.asarray(idxs.cpu() if torch.is_tensor(idxs) else idxs))
        if self.kp_map is None and len(self.rest_pose) == 1 and self.pelvis.is_cuda and getattr(self, "fused_batch", True):
            # single-view layer with one rest pose (surreal / mixamo / perfcap): the whole call is one launch each way
            pose_idx, inverse, uniq_host, counts = self._batch_index(idxs)
            kp, bone, skts, l2ws, rots, kp_u, bone_u, rots_u = _PoseBatchFn.apply(self.pelvis, self.bones, self.rest_pose, pose_idx, inverse,
                                                                                  weakref.ref(self))
            self.last_unique = {"idxs": uniq_host, "counts": counts, "rots": rots_u, "bones": bone_u, "kp": kp_u}
            st = self.__dict__.get("_staged")
            if st is not None and st[1][0].data_ptr() == pose_idx.data_ptr():      # staged batch: the device-side twins of idxs / counts
                self.last_unique.update(idx_dev=st[1][0], w_dev=st[1][2])
            return kp, bone, skts, l2ws, rots
        unique_idxs, inverse_idxs = np.unique(idxs, return_inverse=True)       # FK once per distinct pose
        rest = self.get_rest_pose(unique_idxs, rest_pose_idxs)
        pelvis, bone = self.idx_to_params(unique_idxs)
        kp, skts, l2ws, rots = calculate_kinematic(bone.contiguous(), pelvis.contiguous(), rest)
        # what the regulariser needs per DISTINCT pose (kp_loss): the FK outputs before the per-ray expansion
        self.last_unique = {"idxs": unique_idxs, "counts": np.bincount(inverse_idxs, minlength=len(unique_idxs)), "rots": rots,
                            "bones": bone, "kp": kp}
        return tuple(expand_poses(t, inverse_idxs) for t in (kp, bone, skts, l2ws, rots))

    @torch.no_grad()
    def update_cache(self):
        kps, bones, skts, l2ws, rots = self.calculate_kinematic(np.arange(len(self.pelvis)))
        self.cache_kps, self.cache_bones, self.cache_skts, self.cache_l2ws, self.cache_rots = kps, bones, skts, l2ws, rots

    def forward(self, idxs, rest_pose_idxs=None):
        if not self.use_cache:
            return self.calculate_kinematic(idxs, rest_pose_idxs)
        return self.cache_kps[idxs], self.cache_bones[idxs], self.cache_skts[idxs], self.cache_l2ws[idxs], self.cache_rots[idxs]


def create_popt(args, data_attrs, ckpt=None, device=None):
    """core/pose_opt.py:14-83, same arguments and the same `(pose_optimizer, popt_kwargs)` pair run_nerf.py:523 unpacks: the pose
    layer over the dataset's poses (`data_attrs` = dataset.get_meta() / H5PoseData.data_attrs()), a torch Adam over its parameters
    (lr = opt_pose_lrate), the regularisation anchors {kps, bones, rots, beta}; a checkpoint (or --init_poseopt) restores layer,
    optimiser and anchors unless --no_poseopt_reload, --use_ckpt_anchor re-derives the anchors from the restored layer.
    The reference's `--use_ckpt_anchor` branch unpacks four of forward's five return values (pose_opt.py:65 against :311-316) and
    raises; here it does what its comment describes (anchors = the restored layer's poses, bones back in axis-angle).
    To put the layer into the flat data-parallel bucket: `optim.FusedAdam.from_torch(optimizer, pose_optimizer,
    pose_step_every=args.opt_pose_step)` takes both torch Adams over (hyper-parameters and restored state) and
    `fused.group_optimizer(1)` is the Trainer's pose_optimizer."""
    skel_type = data_attrs["skel_type"]
    rest_pose = torch.as_tensor(np.asarray(data_attrs["rest_pose"])).reshape(-1, len(skel_type.joint_names), 3)
    beta = torch.tensor(data_attrs["betas"])
    init_kps, init_bones = torch.tensor(data_attrs["kp3d"]), torch.tensor(data_attrs["bones"])
    popt_layer = PoseOptLayer(init_kps.clone().to(device), init_bones.clone().to(device), rest_pose.to(device), beta=beta, skel_type=skel_type,
                              kp_map=data_attrs.get("kp_map", None), kp_uidxs=data_attrs.get("kp_uidxs", None),
                              rest_pose_idxs=data_attrs.get("rest_pose_idxs", None), use_cache=args.opt_pose_cache, use_rot6d=args.opt_rot6d)
    popt_layer = popt_layer.to(device)
    pose_optimizer = torch.optim.Adam(params=list(popt_layer.parameters()), lr=args.opt_pose_lrate, betas=(0.9, 0.999))
    anchor_kps, anchor_bones, anchor_beta = init_kps, init_bones, beta
    if (ckpt is not None or args.init_poseopt is not None) and not args.no_poseopt_reload:
        pose_ckpt = torch.load(args.init_poseopt, weights_only=False) if args.init_poseopt is not None else ckpt
        popt_layer.load_state_dict(pose_ckpt["poseopt_layer_state_dict"])
        pose_optimizer.load_state_dict(pose_ckpt["pose_optimizer_state_dict"])
        if "poseopt_anchors" in pose_ckpt:
            anchor_kps, anchor_bones, anchor_beta = (pose_ckpt["poseopt_anchors"][k] for k in ("kps", "bones", "beta"))
        if args.use_ckpt_anchor:      # the checkpoint's poses become the optimisation constraint
            with torch.no_grad():
                anchor_kps, anchor_bones, _, _, _ = popt_layer(np.arange(anchor_bones.shape[0]))
            anchor_kps, anchor_bones = anchor_kps.cpu().clone(), popt_layer.to_bones3d(anchor_bones).cpu().clone()
            anchor_beta = popt_layer.get_beta()
        print("load smpl state dict")
    # anchor rotations recomputed from the bones, so that the two are consistent
    anchor_rots = axisang_to_rot(anchor_bones.reshape(-1, 3)).reshape(*anchor_kps.shape[:2], 3, 3)
    popt_anchors = {"kps": anchor_kps, "bones": anchor_bones, "rots": anchor_rots, "beta": anchor_beta}
    if popt_layer.use_cache:
        popt_layer.update_cache()
    pose_optimizer.zero_grad()           # gradients that came with the checkpoint
    return pose_optimizer, {"popt_anchors": popt_anchors, "popt_layer": popt_layer, "skel_type": skel_type}


def load_poseopt_from_state_dict(state_dict):
    """pose_opt.py:211-240: rebuild the layer from a checkpoint's `poseopt_layer_state_dict`."""
    sd = state_dict["poseopt_layer_state_dict"]
    kp_map = kp_uidxs = None
    if "kp_map" in sd:                                                # multi-view: root bone stored separately
        kp_map, kp_uidxs = sd["kp_map"].cpu().numpy(), sd["kp_uidxs"].cpu().numpy()
    n = sd["pelvis"].shape[0]
    layer = PoseOptLayer(torch.zeros(n, 24, 3), torch.zeros(n, 24, 3), torch.zeros(sd["rest_pose"].shape),
                         use_rot6d=sd["bones"].shape[-1] == 6, kp_map=kp_map, kp_uidxs=kp_uidxs)
    layer.load_state_dict(sd)
    return layer


def load_bones_from_state_dict(state_dict, device="cpu"):
    """pose_opt.py:195-202: the checkpoint's bones as axis-angle"""
    bones = state_dict["poseopt_layer_state_dict"]["bones"]
    return (rot6d_to_axisang(bones) if bones.shape[-1] == 6 else bones).to(device)
