"""Tensor-level wrappers over the C ABI (include/anerf.h).  PyTorch is plumbing here: it owns device
memory and the stream; every op below enqueues hand-written HIP kernels from libanerf_hip.so on
torch's current stream and returns torch tensors that alias the buffers the kernels wrote.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

PARAM_ORDER = [f"pts_linears.{i}" for i in range(8)] + ["alpha_linear", "feature_linear", "views_linears.0", "rgb_linear"]


class PathConfig:
    """Static configuration of the path (reference: create_raycaster, core/raycasters.py:17-184)."""

    def __init__(self, multires=7, multires_views=4, framecode_ch=0, density_scale=1.0, softplus_shift=None,
                 n_joints=24, netdepth=8, netwidth=256, skip=4, cutoff_bones=False):
        self.multires, self.multires_views, self.framecode_ch = multires, multires_views, framecode_ch
        self.cutoff_bones = bool(cutoff_bones)     # --cutoff_bones: the bone-direction block is gated by the distance gate too
        self.density_scale, self.softplus_shift = float(density_scale), softplus_shift
        self.n_joints, self.netdepth, self.netwidth, self.skip = n_joints, netdepth, netwidth, skip
        self.dim_v = n_joints * (1 + 2 * multires)
        self.dim_x = self.dim_v + 3 * n_joints
        self.dim_d = 3 * n_joints * (1 + 2 * multires_views)
        self.x_width = self.dim_x + self.dim_d + (1 if framecode_ch else 0)

    def key(self):
        return (self.multires, self.multires_views, self.framecode_ch, self.n_joints, self.netdepth, self.netwidth, self.skip)

    def c(self):
        return _lib.AnerfConfig(self.n_joints, self.multires, self.multires_views, self.framecode_ch, self.netdepth,
                                self.netwidth, self.skip, 0 if self.softplus_shift is None else 1, self.density_scale,
                                0.0 if self.softplus_shift is None else float(self.softplus_shift), int(self.cutoff_bones))


def _stream():
    """the current HIP stream of the current device as the C ABI takes it.  torch.cuda.current_stream() builds a Stream object through
    three Python layers (~9 us; 23 calls per training step = 0.2 ms of the eager path's host time): ask the C binding directly"""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a float32 CUDA(ROCm) tensor, got {t.dtype} on {t.device}")
    return t if t.is_contiguous() else t.contiguous()


def row_shared(t):
    """True for a tensor whose rows are ONE row seen N times: a stride-0 expand along dim 0 -- what run_nerf.render_path's
    `reuse_input(x, expand)` hands the caster for the frame's pose / cylinder / camera index (run_nerf.py:62-72,84-88:
    `y.expand(expand, *x.shape[1:])`).  Such an input is passed to the kernels as its single row with a zero ray stride;
    `.contiguous()` would write N copies of it (6.3 MB of bone matrices per 4096-ray chunk) and select the per-ray-pose
    prologue of the MLP kernel."""
    return t is not None and t.dim() >= 1 and t.shape[0] > 1 and t.stride(0) == 0


def first_row(t):
    """the one row of a row_shared tensor, [1, ...] (no copy when the row itself is contiguous)"""
    return t[:1] if row_shared(t) else t


_layout_cache = {}
_table_cache = {}


def layout(cfg, which=0):
    k = (cfg.key(), which)
    if k not in _layout_cache:
        L = _lib.AnerfLayout()
        cc = cfg.c()
        _lib.check(_lib.load().anerf_layout(C.byref(cc), which, C.byref(L)), "anerf_layout")
        _layout_cache[k] = (int(L.stream_floats), int(L.aux_floats), int(L.n_stages), int(L.x_width))
    return _layout_cache[k]


def pack_table(cfg, device, which=0):
    """Device copy of the parameter-gather table (built on the host once per config and device)."""
    k = (cfg.key(), which, str(device))
    if k not in _table_cache:
        sf, af, _, _ = layout(cfg, which)
        host = np.empty((2 * sf if which >= 3 else sf) + af, dtype=np.int32)   # bf16x3 images: one entry per bf16 element
        cc = cfg.c()
        _lib.check(_lib.load().anerf_build_pack_table(C.byref(cc), which, host.ctypes.data_as(C.c_void_p)), "anerf_build_pack_table")
        _table_cache[k] = torch.from_numpy(host).to(device)
    return _table_cache[k]


_sched_serial = [0]


class InputSchedule:
    """The --freq_schedule factors of a network's encoded input columns as the library takes them (AnerfNetParams.sched_x / sched_u,
    ABI revision 5): sx [dim_x] for the distance + bone block, su [u_width] for the view (+ frame code) block, device fp32.
    `serial` identifies the VALUES (weight images are cached against it: tensor addresses get reused)."""

    def __init__(self, sx, su):
        self.sx, self.su = _f32c(sx, "sched_x"), _f32c(su, "sched_u")
        _sched_serial[0] += 1
        self.serial = _sched_serial[0]

    def fill(self, st):
        """into an AnerfNetParams / AnerfNetGrads"""
        st.sched_x, st.sched_u = self.sx.data_ptr(), self.su.data_ptr()
        if hasattr(st, "sched_dim_x"):
            st.sched_dim_x, st.sched_dim_u = self.sx.numel(), self.su.numel()


def net_params_struct(params, codes=None, sched=None):
    """params: dict name -> CUDA tensor with the reference's state_dict names.  sched: InputSchedule or None."""
    st = _lib.AnerfNetParams()
    keep = []
    if sched is not None:
        sched.fill(st)
        keep.append(sched)
    for i, n in enumerate(PARAM_ORDER):
        w = _f32c(params[n + ".weight"], n + ".weight")
        b = _f32c(params[n + ".bias"], n + ".bias")
        keep += [w, b]
        st.w[i] = w.data_ptr()
        st.b[i] = b.data_ptr()
    if codes is not None:
        codes = _f32c(codes, "codes")
        keep.append(codes)
        st.codes = codes.data_ptr()
        st.n_codes = codes.shape[0]
    return st, keep


def _check_sched(cfg, sched):
    if sched is not None and (sched.sx.numel() != cfg.dim_x or sched.su.numel() != cfg.dim_d + cfg.framecode_ch):
        raise ValueError(f"InputSchedule of {sched.sx.numel()} / {sched.su.numel()} columns for a network with "
                         f"{cfg.dim_x} / {cfg.dim_d + cfg.framecode_ch}")


def pack_params(cfg, params, which=0, out=None, sched=None):
    """Gather one network's parameters into the packed (stream, aux) images the MLP kernels consume.  sched: InputSchedule
    folded into the columns that consume the encoding, or None."""
    dev = params[PARAM_ORDER[0] + ".weight"].device
    sf, af, _, _ = layout(cfg, which)
    table = pack_table(cfg, dev, which)
    if out is None:
        out = torch.empty(sf + af, dtype=torch.float32, device=dev)
    _check_sched(cfg, sched)
    st, keep = net_params_struct(params, sched=sched)
    if which >= 3:   # 3: bf16x3 W (forward), 4: bf16x3 W^T (backward-data)
        _lib.check(_lib.load().anerf_pack_params_b3(C.byref(st), _p(table), sf, af, _p(out), _stream()), "anerf_pack_params_b3")
    else:
        _lib.check(_lib.load().anerf_pack_params(C.byref(st), _p(table), sf + af, _p(out), _stream()), "anerf_pack_params")
    return out[:sf], out[sf:]


def pack_params_multi(jobs):
    """Gather several weight images in ONE launch (anerf_pack_params_multi).  jobs: list of (cfg, params dict, which, out flat
    tensor of layout(cfg, which) floats[, InputSchedule or None]).  Replaces one k_pack / k_pack_b3 launch per image (4 per fp32
    training step)."""
    if not jobs:
        return
    lib = _lib.load()
    arr, keep = [], []
    for cfg, params, which, out, *rest in jobs:
        dev = out.device
        sf, af, _, _ = layout(cfg, which)
        table = pack_table(cfg, dev, which)
        sched = rest[0] if rest else None
        _check_sched(cfg, sched)
        # params: the state-dict-named tensors, or a prebuilt (AnerfNetParams, keep-alive list) pair (NeRF.params_struct's cache)
        st, k = params if isinstance(params, tuple) else net_params_struct(params, sched=sched)
        keep += list(k) + [table]
        if which >= 3:     # bf16x3 image: hi/lo-split stream part (one table entry per bf16 element) + fp32 aux part
            arr.append(_lib.AnerfPackJob(st, table.data_ptr(), 2 * sf, out.data_ptr(), 1))
            arr.append(_lib.AnerfPackJob(st, table.data_ptr() + 4 * 2 * sf, af, out.data_ptr() + 4 * sf, 0))
        else:
            arr.append(_lib.AnerfPackJob(st, table.data_ptr(), sf + af, out.data_ptr(), 0))
    for i in range(0, len(arr), 8):
        chunk = arr[i:i + 8]
        carr = (_lib.AnerfPackJob * len(chunk))(*chunk)
        _lib.check(lib.anerf_pack_params_multi(carr, len(chunk), _stream()), "anerf_pack_params_multi")


_rng_instances = [0]


class StepBlock:
    """The device-resident AnerfStepBlock of a captured training step (ABI revision 6) and the host-side AnerfStepValues it
    is written from: `write()` is ONE launch whose kernel arguments carry the values (stream-ordered; nothing the host could
    overwrite while an earlier write is still queued).  Inside `with ops.step_block(blk):` every DeviceRng.fill,
    train_forward / backward and adam_step reads its per-step scalars from the block instead of taking them as kernel
    arguments -- same kernels, same arithmetic, so the results are bit-identical; what changes is that a captured hipGraph
    of the step can be replayed without node updates (graph_step.GraphedTrainStep)."""

    def __init__(self, device):
        self.buf = torch.zeros(C.sizeof(_lib.AnerfStepBlock) // 4, dtype=torch.float32, device=device)
        self.values = _lib.AnerfStepValues()
        self.fills = 0            # index of the next DeviceRng.fill inside the current iteration

    @property
    def ptr(self):
        return self.buf.data_ptr()

    def set_rng(self, seed, offset):
        self.values.rng_seed, self.values.rng_offset = int(seed), int(offset)

    def set_tau(self, tau_v, tau_d):
        self.values.tau_v, self.values.tau_d = float(tau_v), float(tau_d)

    def set_adam(self, groups):
        """groups: per optimiser group (lr, beta1, beta2, step, grad_scale); step <= 0 = the group does not step this iteration"""
        if len(groups) > _lib.MAX_ADAM_GROUPS:
            raise ValueError(f"StepBlock: at most {_lib.MAX_ADAM_GROUPS} optimiser groups")
        v = self.values
        v.n_groups = len(groups)
        for g, (lr, b1, b2, step, gs) in enumerate(groups):
            v.lr[g], v.beta1[g], v.beta2[g], v.adam_step[g], v.grad_scale[g] = float(lr), float(b1), float(b2), int(step), float(gs)

    def write(self):
        self.fills = 0
        _lib.check(_lib.load().anerf_step_block_write(C.c_void_p(self.ptr), C.byref(self.values), _stream()), "anerf_step_block_write")

    def read(self):
        """host copy of the device block (a synchronising D2H copy: tests only)"""
        raw = self.buf.cpu().numpy().tobytes()
        return _lib.AnerfStepBlock.from_buffer_copy(raw)


_active_step = None


class step_block:
    """route the per-step scalars of the calls inside through `blk` (see StepBlock)"""

    def __init__(self, blk):
        self.blk = blk

    def __enter__(self):
        global _active_step
        self.prev, _active_step = _active_step, self.blk
        return self.blk

    def __exit__(self, *a):
        global _active_step
        _active_step = self.prev


class DeviceRng:
    """Counter-based generator behind anerf_rand_fill: (key, offset) on the host, one launch per `fill`.  Stands where the
    reference calls torch.rand / torch.randn on the device (ray_utils.py:171-180,240-246; nerf.py:176-182;
    raycasters.py:660,674): same distributions, its own stream (as torch's CPU and GPU generators differ from each other).

    Seeding follows torch's: the key derives from `torch.initial_seed()` and is RE-derived (offset back to 0) whenever that
    seed changes, so `torch.manual_seed(k)` before a run reproduces its jitter / noise draws like it does for the torch.rand
    path this replaces.  The key also mixes a stream id -- the process's rank (RANK / torch.distributed) and a per-instance
    counter -- so data-parallel ranks and deep-copied casters with the same seed draw DIFFERENT numbers for their shards.
    `manual_seed(seed)` pins an explicit seed (no longer follows torch's); `state_dict()` / `load_state_dict()` carry (seed,
    stream id, offset, pinned) for checkpoints (checkpoint.save_nerf stores it under "anerf_rng_state")."""

    def __init__(self, seed=None, stream_id=None):
        if stream_id is None:
            stream_id = (self._rank() << 20) | (_rng_instances[0] & 0xFFFFF)
            _rng_instances[0] += 1
        self.stream_id = int(stream_id)
        self.pinned = seed is not None
        self._torch_seed = None
        self.offset = 0
        self._set(int(torch.initial_seed() if seed is None else seed))

    def _set(self, seed):
        self._torch_seed = int(seed) & (2 ** 64 - 1)
        # splitmix-style mix of the stream id into the Philox key (a plain xor would map (seed, id) pairs onto each other)
        z = (self._torch_seed + 0x9E3779B97F4A7C15 * (self.stream_id + 1)) & (2 ** 64 - 1)
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        self.seed = (z ^ (z >> 31)) & (2 ** 64 - 1)
        self.offset = 0

    def manual_seed(self, seed):
        self.pinned = True
        self._set(seed)
        return self

    def follow_torch_seed(self):
        """torch.manual_seed() was called since the last draw (and no explicit seed is pinned): re-derive the key, offset 0"""
        if not self.pinned and int(torch.initial_seed()) & (2 ** 64 - 1) != self._torch_seed:
            self._set(torch.initial_seed())

    def __deepcopy__(self, memo):
        """a copied caster draws its own stream: same seed policy, fresh stream id"""
        return DeviceRng(seed=self._torch_seed if self.pinned else None)

    def state_dict(self):
        return {"seed": self._torch_seed, "stream_id": self.stream_id, "offset": self.offset, "pinned": self.pinned}

    @staticmethod
    def _rank():
        try:
            import torch.distributed as dist
            return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else int(__import__("os").environ.get("RANK", 0))
        except Exception:
            return 0

    def load_state_dict(self, sd):
        """Continue a saved stream.  The checkpoint is written by ONE rank: the instance part of the stream id (low 20 bits)
        is restored, the rank part is this process's own, so the writing rank resumes its stream exactly and every other
        data-parallel rank keeps drawing different jitter / noise for its different ray shard.  The restored state is
        pinned: a torch seed that differs on resume must not reset the offset at the next draw (follow_torch_seed)."""
        self.stream_id = (self._rank() << 20) | (int(sd["stream_id"]) & 0xFFFFF)
        self._set(int(sd["seed"]))
        self.offset = int(sd["offset"])
        self.pinned = True

    def fill(self, specs, device):
        """specs: list of (shape, kind, scale) with kind "uniform" | "normal" (None entries are skipped and returned as None).
        Returns the tensors, all filled by ONE launch."""
        self.follow_torch_seed()
        outs, jobs = [], []
        for sp in specs:
            if sp is None:
                outs.append(None)
                continue
            shape, kind, scale = sp
            t = torch.empty(*shape, dtype=torch.float32, device=device)
            outs.append(t)
            jobs.append(_lib.AnerfRandJob(t.data_ptr(), t.numel(), 0 if kind == "uniform" else 1, float(scale)))
        if jobs:
            carr = (_lib.AnerfRandJob * len(jobs))(*jobs)
            blk = _active_step
            if blk is not None:       # (seed, offset) from the device-resident step block: the owner writes them per iteration
                _lib.check(_lib.load().anerf_rand_fill_dev(carr, len(jobs), C.c_void_p(blk.ptr), blk.fills, _stream()), "anerf_rand_fill_dev")
                blk.fills += 1
            else:
                _lib.check(_lib.load().anerf_rand_fill(carr, len(jobs), self.seed, self.offset, _stream()), "anerf_rand_fill")
            self.offset += 1
        return outs


def make_ray_batch(rays_o, rays_d, near=0.0, far=1.0, use_viewdirs=True):
    """render()'s ray batch [N, 8 | 11] = (o, d, near, far [, d / |d|]) in one launch (core/trainer.py:116-135)."""
    rays_o, rays_d = _f32c(rays_o, "rays_o"), _f32c(rays_d, "rays_d")
    n = rays_o.shape[0]
    stride = 11 if use_viewdirs else 8
    out = torch.empty(n, stride, dtype=torch.float32, device=rays_o.device)
    _lib.check(_lib.load().anerf_make_ray_batch(_p(rays_o), _p(rays_d), n, float(near), float(far), stride, _p(out), _stream()),
               "anerf_make_ray_batch")
    return out


_circle_cache = {}


def cyl_bbox(cyls, c2ws, hwf, off):
    """Pixel boxes (x0, y0, x1, y1) int32 [F,4] of F projected bounding cylinders, one launch (anerf_cyl_bbox;
    cylinder_to_box_2d, core/utils/skeleton_utils.py:607-690).  cyls [F,5], c2ws [F,3,4], hwf [F,4] = (H, W, fx, fy) float64
    device tensors, off [F,2] int32 (integer principal point)."""
    dev = cyls.device
    if str(dev) not in _circle_cache:
        rads = np.linspace(0.0, 2 * np.pi, 50)
        _circle_cache[str(dev)] = torch.tensor(np.stack([np.cos(rads), np.sin(rads)], -1), dtype=torch.float64, device=dev)
    for t, nm in ((cyls, "cyls"), (c2ws, "c2ws"), (hwf, "hwf")):
        if t.dtype != torch.float64 or not t.is_contiguous():
            raise TypeError(f"cyl_bbox: {nm} must be a contiguous float64 tensor")
    if off.dtype != torch.int32 or not off.is_contiguous():
        raise TypeError("cyl_bbox: off must be a contiguous int32 tensor")
    F = cyls.shape[0]
    out = torch.empty(F, 4, dtype=torch.int32, device=dev)
    _lib.check(_lib.load().anerf_cyl_bbox(_p(cyls), _p(c2ws), _p(hwf), _p(off), _p(_circle_cache[str(dev)]), F, _p(out), _stream()),
               "anerf_cyl_bbox")
    return out


def ray_bounds(rays, cyls):
    """get_near_far_in_cylinder (ray_utils.py:292): rays [N,>=8], cyls [N,5] -> (near_far [N,2] raw, stats: 32 bytes of scratch for coarse_z)."""
    rays, cyls = _f32c(rays, "rays"), _f32c(cyls, "cyls")
    n = rays.shape[0]
    nf = torch.empty(n, 2, dtype=torch.float32, device=rays.device)
    stats = torch.empty(4, dtype=torch.int64, device=rays.device)
    _lib.check(_lib.load().anerf_ray_bounds(_p(rays), rays.shape[1], _p(cyls), n, _p(nf), _p(stats), _stream()), "anerf_ray_bounds")
    return nf, stats


def coarse_z(near_far, stats, rays, n_samples, t_rand=None, lindisp=False):
    rays = _f32c(rays, "rays")
    n = rays.shape[0]
    t_rand = _f32c(t_rand, "t_rand")
    z = torch.empty(n, n_samples, dtype=torch.float32, device=rays.device)
    nf_fixed = torch.empty(n, 2, dtype=torch.float32, device=rays.device)
    _lib.check(_lib.load().anerf_coarse_z(_p(near_far), _p(stats), _p(rays), rays.shape[1], n, n_samples, _p(t_rand),
                                          int(bool(lindisp)), _p(z), _p(nf_fixed), _stream()), "anerf_coarse_z")
    return z, nf_fixed


def mlp_raw(cfg, packed, aux, rays, z_vals, skts, tau_v, tau_d, cut_v, cut_d, cam_idx=None, codes=None, precision="fp32"):
    """Fused encode + MLP: raw [N,S,4].  skts [N,24,4,4] or [1,24,4,4] (shared pose).
    precision "fp32" (image which=0) or "bf16x3" (image which=3: hi/lo-split bf16 MFMAs, fp32 accumulation)."""
    rays, z_vals, skts = _f32c(rays, "rays"), _f32c(z_vals, "z_vals"), _f32c(skts, "skts")
    n, s = z_vals.shape
    if skts.shape[0] not in (1, n):
        raise ValueError(f"skts batch {skts.shape[0]} must be 1 or N={n}")
    stride = 0 if skts.shape[0] == 1 else 16 * cfg.n_joints
    cam_idx, codes = _f32c(cam_idx, "cam_idx"), _f32c(codes, "codes")
    raw = torch.empty(n, s, 4, dtype=torch.float32, device=rays.device)
    cc = cfg.c()
    if precision not in ("fp32", "bf16x3"):
        raise ValueError(f"precision {precision!r}")
    fn = _lib.load().anerf_mlp_raw if precision == "fp32" else _lib.load().anerf_mlp_raw_b3
    _lib.check(fn(C.byref(cc), _p(packed), _p(aux), _p(rays), rays.shape[1], _p(z_vals), _p(skts),
                  stride, _p(cam_idx), _p(codes), 0 if codes is None else codes.shape[0],
                  float(tau_v), float(tau_d), _p(_f32c(cut_v, "cut_v")), _p(_f32c(cut_d, "cut_d")),
                  n, s, _p(raw), _stream()), "anerf_mlp_raw[" + precision + "]")
    return raw


def mlp_forward(cfg, packed, aux, x, codes=None):
    """NeRF.forward seam: x [..., x_width] pre-encoded -> raw [..., 4]."""
    x = _f32c(x, "x")
    if x.shape[-1] != cfg.x_width:
        raise ValueError(f"x last dim {x.shape[-1]} != {cfg.x_width}")
    flat = x.reshape(-1, x.shape[-1])
    codes = _f32c(codes, "codes")
    raw = torch.empty(flat.shape[0], 4, dtype=torch.float32, device=x.device)
    cc = cfg.c()
    _lib.check(_lib.load().anerf_mlp_forward(C.byref(cc), _p(packed), _p(aux), _p(flat), flat.shape[0], _p(codes),
                                             0 if codes is None else codes.shape[0], _p(raw), _stream()), "anerf_mlp_forward")
    return raw.reshape(*x.shape[:-1], 4)


def composite(cfg, raw, z_vals, rays, noise=None, want_depth=False):
    """raw2outputs (nerf.py:150-205)."""
    raw, z_vals, rays, noise = _f32c(raw, "raw"), _f32c(z_vals, "z_vals"), _f32c(rays, "rays"), _f32c(noise, "noise")
    n, s = z_vals.shape
    dev = raw.device
    rgb = torch.empty(n, 3, dtype=torch.float32, device=dev)
    disp = torch.empty(n, dtype=torch.float32, device=dev)
    acc = torch.empty(n, dtype=torch.float32, device=dev)
    w = torch.empty(n, s, dtype=torch.float32, device=dev)
    al = torch.empty(n, s, dtype=torch.float32, device=dev)
    depth = torch.empty(n, dtype=torch.float32, device=dev) if want_depth else None
    cc = cfg.c()
    _lib.check(_lib.load().anerf_composite(C.byref(cc), _p(raw), _p(z_vals), _p(rays), rays.shape[1], _p(noise), n, s,
                                           _p(rgb), _p(disp), _p(acc), _p(w), _p(al), _p(depth), _stream()), "anerf_composite")
    out = {"rgb_map": rgb, "disp_map": disp, "acc_map": acc, "weights": w, "alpha": al}
    if want_depth:
        out["depth_map"] = depth
    return out


def importance(z_vals, weights, n_importance, u=None, single_net=False, want_idx=True):
    """isample_from_lineseg (ray_utils.py:255-289): -> z_samples [N,Ni], z_merged [N,S+Ni], sorted_idx int64."""
    z_vals, weights, u = _f32c(z_vals, "z_vals"), _f32c(weights, "weights"), _f32c(u, "u")
    n, s = z_vals.shape
    dev = z_vals.device
    zs = torch.empty(n, n_importance, dtype=torch.float32, device=dev)
    zm = torch.empty(n, s + n_importance, dtype=torch.float32, device=dev)
    idx = torch.empty(n, s + n_importance, dtype=torch.int64, device=dev) if want_idx else None
    _lib.check(_lib.load().anerf_importance(_p(z_vals), _p(weights), n, s, n_importance, _p(u), int(bool(single_net)),
                                            _p(zs), _p(zm), _p(idx), _stream()), "anerf_importance")
    return zs, zm, idx


def density(cfg, packed, aux, pts, skts, tau_v, cut_v):
    """Density query (raycasters.py:597-648): pts [...,3] under ONE pose skts [1,24,4,4] -> sigma logits [...,1]."""
    pts, skts = _f32c(pts, "pts"), _f32c(skts, "skts")
    if skts.numel() != 16 * cfg.n_joints:
        raise NotImplementedError("density query supports a single pose (skts [1,24,4,4]) as in render_mesh_density")
    flat = pts.reshape(-1, 3)
    out = torch.empty(flat.shape[0], dtype=torch.float32, device=pts.device)
    cc = cfg.c()
    _lib.check(_lib.load().anerf_density(C.byref(cc), _p(packed), _p(aux), _p(flat), _p(skts), float(tau_v),
                                         _p(_f32c(cut_v, "cut_v")), flat.shape[0], _p(out), _stream()), "anerf_density")
    return out.reshape(*pts.shape[:-1], 1)


def gen_rays(H, W, focal, c2w, bbox, center=None, near=0.0, far=1.0):
    """Ray batch [N,11] + valid_idx [N] (int64) for the pixel box bbox = (x0, y0, x1, y1)   (ray_utils.py:6-28,83-136)."""
    c2w = _f32c(c2w, "c2w")[:3, :4].contiguous()
    x0, y0, x1, y1 = [int(b) for b in bbox]
    n = max(0, x1 - x0) * max(0, y1 - y0)
    fx, fy = (focal, focal) if not hasattr(focal, "__len__") else (float(focal[0]), float(focal[1]))
    cx, cy = (W * 0.5, H * 0.5) if center is None else (float(center[0]), float(center[1]))
    rb = torch.empty(n, 11, dtype=torch.float32, device=c2w.device)
    idx = torch.empty(n, dtype=torch.int64, device=c2w.device)
    _lib.check(_lib.load().anerf_gen_rays(H, W, float(fx), float(fy), cx, cy, _p(c2w), x0, y0, x1, y1, float(near), float(far),
                                          _p(rb), _p(idx), _stream()), "anerf_gen_rays")
    return rb, idx


def assemble_frame(rgb_map, acc_map, disp_map, valid_idx, rgb_img, want_acc=False):
    """run_nerf.py:118-131: rgb_img [H*W,3] holds the background and is updated in place; returns (rgb_img, disp_img, acc_img)."""
    n = rgb_map.shape[0]
    dev = rgb_map.device
    disp_img = torch.zeros(rgb_img.shape[0], dtype=torch.float32, device=dev)
    acc_img = torch.zeros(rgb_img.shape[0], dtype=torch.float32, device=dev) if want_acc else None
    _lib.check(_lib.load().anerf_assemble_frame(_p(_f32c(rgb_map, "rgb")), _p(_f32c(acc_map, "acc")), _p(_f32c(disp_map, "disp")),
                                                _p(valid_idx), n, _p(rgb_img), _p(disp_img), _p(acc_img), _stream()),
               "anerf_assemble_frame")
    return rgb_img, disp_img, acc_img


def loss(rgb, acc, target, rgb0=None, acc0=None, bgs=None, loss_type=0, coarse_weight=1.0, want_grads=True, beta=0.1):
    """_compute_nerf_loss + its gradient w.r.t. the rendered maps in one pass (anerf_loss; trainer.py:353-380).
    bgs: None (no background composite), a [3] / [1,3] colour, or per-ray [N,3].
    loss_type 0 MSE, 1 L1, 2 Huber (smooth-L1 with `beta`, trainer.py:57).
    Returns (out4 = [total, fine, coarse, fine mse], grads dict or None)."""
    rgb, acc, target = _f32c(rgb, "rgb"), _f32c(acc, "acc"), _f32c(target, "target")
    rgb0, acc0, bgs = _f32c(rgb0, "rgb0"), _f32c(acc0, "acc0"), _f32c(bgs, "bgs")
    n, dev = rgb.shape[0], rgb.device
    if target.shape != rgb.shape:
        raise ValueError(f"loss: target {tuple(target.shape)} vs rgb {tuple(rgb.shape)}")
    stride = 0
    if bgs is not None:
        if bgs.numel() == 3:
            stride = 0
        elif bgs.shape == rgb.shape:
            stride = 3
        else:
            raise ValueError(f"loss: bgs must have 3 or N*3 elements, got {tuple(bgs.shape)}")
    lib = _lib.load()
    nblk = lib.anerf_loss_blocks(n)
    scratch = torch.empty(4 + 4 * nblk, dtype=torch.float32, device=dev)     # out4 (k_loss_final writes all four) + partials
    out, part = scratch[:4], scratch[4:]
    if n == 0:
        out.zero_()                                   # empty batch: the library enqueues nothing
    g = None
    if want_grads:
        # the four gradient maps share ONE buffer ("flat"): the upstream gradient of the loss scales them with one launch
        sizes = {"rgb": 3 * n, "acc": n if bgs is not None else 0, "rgb0": 3 * n if rgb0 is not None else 0,
                 "acc0": n if (rgb0 is not None and bgs is not None) else 0}
        flat = torch.empty(sum(sizes.values()), dtype=torch.float32, device=dev)
        g, o = {"flat": flat}, 0
        for k, sz in sizes.items():
            g[k] = flat[o:o + sz].view((n, 3) if k.startswith("rgb") else (n,)) if sz else None
            o += sz
    gp = (lambda k: _p(g[k]) if g is not None else None)
    _lib.check(lib.anerf_loss(_p(rgb), _p(acc), _p(rgb0), _p(acc0), _p(target), _p(bgs), stride, n, int(loss_type),
                              float(beta), float(coarse_weight), _p(out), gp("rgb"), gp("acc"), gp("rgb0"), gp("acc0"), _p(part), _stream()),
               "anerf_loss")
    return out, g


def adam_step(params, grads, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0, zero_grads=False,
              n_tensors=0, norms2=None, group=0):
    """torch.optim.Adam's update over one flat fp32 buffer (anerf_adam_step).  norms2: optional [2] tensor that
    receives get_gradnorm's (total_norm, avg_norm).  Inside `with step_block(blk)`: lr / step / grad_scale of optimiser group
    `group` are read from the device-resident block (anerf_adam_step_dev); the arguments given here are ignored."""
    for t, nm in ((params, "params"), (grads, "grads"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != params.numel():
            raise ValueError(f"adam_step: {nm} must be contiguous float32 with {params.numel()} elements")
    lib = _lib.load()
    n = params.numel()
    part = torch.empty(lib.anerf_adam_blocks(n), dtype=torch.float32, device=params.device) if norms2 is not None else None
    if _active_step is not None:
        _lib.check(lib.anerf_adam_step_dev(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, float(beta1), float(beta2), float(eps),
                                           C.c_void_p(_active_step.ptr), int(group), int(bool(zero_grads)), int(n_tensors), _p(part),
                                           _p(norms2), _stream()), "anerf_adam_step_dev")
        return
    _lib.check(lib.anerf_adam_step(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), n, float(lr), float(beta1), float(beta2),
                                   float(eps), int(step), float(grad_scale), int(bool(zero_grads)), int(n_tensors), _p(part),
                                   _p(norms2), _stream()), "anerf_adam_step")


def fk_forward(bones, rest_pose, pelvis=None, want=("l2ws", "skts", "rots", "kp")):
    """PoseOptLayer.calculate_kinematic's math (anerf_fk_forward; pose_opt.py:372-445): bones [U,24,3] axis-angle or
    [U,24,6] 6D rotations (opt_rot6d),
    rest_pose [24,3] / [1,24,3] (shared) or [U,24,3], pelvis [U,3] or None -> dict of the requested outputs."""
    bones, rest_pose, pelvis = _f32c(bones, "bones"), _f32c(rest_pose, "rest_pose"), _f32c(pelvis, "pelvis")
    if bones.dim() != 3 or bones.shape[1] != 24 or bones.shape[2] not in (3, 6):
        raise ValueError(f"fk_forward: bones must be [U,24,3] axis-angle or [U,24,6] rot6d, got {tuple(bones.shape)}")
    u, dev = bones.shape[0], bones.device
    rest = rest_pose.reshape(-1, 24, 3)
    if rest.shape[0] not in (1, u):
        raise ValueError(f"fk_forward: rest_pose must have 1 or {u} poses, got {rest.shape[0]}")
    stride = 0 if rest.shape[0] == 1 else 72
    shapes = {"l2ws": (u, 24, 4, 4), "skts": (u, 24, 4, 4), "rots": (u, 24, 3, 3), "kp": (u, 24, 3)}
    out = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in want}
    _lib.check(_lib.load().anerf_fk_forward(_p(bones), bones.shape[2], _p(pelvis), _p(rest), stride, u, _p(out.get("l2ws")), _p(out.get("skts")),
                                            _p(out.get("rots")), _p(out.get("kp")), _stream()), "anerf_fk_forward")
    return out


def fk_backward(bones, rest_pose, pelvis, g_skts=None, g_l2ws=None, g_kp=None, g_rots=None):
    """-> (g_bones [U,24,3|6], g_pelvis [U,3] or None)   (anerf_fk_backward)"""
    bones, rest_pose, pelvis = _f32c(bones, "bones"), _f32c(rest_pose, "rest_pose"), _f32c(pelvis, "pelvis")
    g_skts, g_l2ws, g_kp, g_rots = _f32c(g_skts, "g_skts"), _f32c(g_l2ws, "g_l2ws"), _f32c(g_kp, "g_kp"), _f32c(g_rots, "g_rots")
    u = bones.shape[0]
    rest = rest_pose.reshape(-1, 24, 3)
    stride = 0 if rest.shape[0] == 1 else 72
    gb = torch.empty_like(bones)
    gp = torch.empty(u, 3, dtype=torch.float32, device=bones.device) if pelvis is not None else None
    _lib.check(_lib.load().anerf_fk_backward(_p(bones), bones.shape[2], _p(pelvis), _p(rest), stride, u, _p(g_skts), _p(g_l2ws), _p(g_kp),
                                             _p(g_rots), _p(gb), _p(gp), _stream()), "anerf_fk_backward")
    return gb, gp


def pose_batch_forward(bones, pelvis, rest_pose, pose_idx, inverse, want_rays=("kp", "bones", "skts", "l2ws", "rots")):
    """anerf_pose_batch_forward: the pose layer's forward for one batch in ONE launch.  bones [P,24,3|6] / pelvis [P,3] = the
    layer's full parameters, pose_idx [U] int64 (distinct poses of the batch), inverse [N] int32 (ray -> slot), rest_pose [24,3].
    Returns (unique dict: kp / bones / rots [U, ...], per-ray dict of the requested rows [N, ...])."""
    bones, pelvis, rest = _f32c(bones, "bones"), _f32c(pelvis, "pelvis"), _f32c(rest_pose, "rest_pose").reshape(-1, 24, 3)
    if rest.shape[0] != 1:
        raise ValueError("pose_batch_forward: one rest pose shared by all poses")
    if pose_idx.dtype != torch.int64 or inverse.dtype != torch.int32 or not pose_idx.is_contiguous() or not inverse.is_contiguous():
        raise TypeError("pose_batch_forward: pose_idx int64 / inverse int32, contiguous")
    u, n, rd, dev = pose_idx.numel(), inverse.numel(), bones.shape[-1], bones.device
    E = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
    uniq = {"kp": E(u, 24, 3), "bones": E(u, 24, rd), "rots": E(u, 24, 3, 3)}
    shapes = {"kp": (n, 24, 3), "bones": (n, 24, rd), "skts": (n, 24, 4, 4), "l2ws": (n, 24, 4, 4), "rots": (n, 24, 3, 3)}
    rays = {k: E(*shapes[k]) for k in want_rays}
    _lib.check(_lib.load().anerf_pose_batch_forward(_p(bones), rd, _p(pelvis), _p(rest), _p(pose_idx), u, _p(inverse), n, _p(uniq["kp"]),
                                                    _p(uniq["bones"]), _p(uniq["rots"]), _p(rays.get("kp")), _p(rays.get("bones")),
                                                    _p(rays.get("skts")), _p(rays.get("l2ws")), _p(rays.get("rots")), _stream()),
               "anerf_pose_batch_forward")
    return uniq, rays


def pose_batch_backward(bones, pelvis, rest_pose, pose_idx, inverse, g_rays, g_uniq, g_bones, g_pelvis, accumulate):
    """anerf_pose_batch_backward.  g_rays / g_uniq: dicts of gradients (missing or None = none) w.r.t. the per-ray rows (kp, bones,
    skts, l2ws, rots) and the unique-level outputs (kp, bones, rots); g_bones [P,24,rd] / g_pelvis [P,3]: the full-size
    parameter gradients, rows pose_idx written (accumulate False; other rows untouched) or added to (True)."""
    bones, pelvis, rest = _f32c(bones, "bones"), _f32c(pelvis, "pelvis"), _f32c(rest_pose, "rest_pose").reshape(-1, 24, 3)
    f = lambda d, k: _f32c(d.get(k), k)
    gr = {k: f(g_rays, k) for k in ("kp", "bones", "skts", "l2ws", "rots")}
    gu = {k: f(g_uniq, k) for k in ("kp", "bones", "rots")}
    lib = _lib.load()
    scratch, sbytes = None, 0
    if any(v is not None for v in gr.values()):
        scratch, sbytes = _workspace(lib.anerf_pose_batch_scratch_size, "anerf_pose_batch_scratch_size", bones.device, pose_idx.numel(),
                                     inverse.numel())
    _lib.check(lib.anerf_pose_batch_backward(_p(bones), bones.shape[-1], _p(pelvis), _p(rest), _p(pose_idx), pose_idx.numel(),
                                             _p(inverse), inverse.numel(), _p(gr["kp"]), _p(gr["bones"]), _p(gr["skts"]),
                                             _p(gr["l2ws"]), _p(gr["rots"]), _p(gu["kp"]), _p(gu["bones"]), _p(gu["rots"]),
                                             _p(g_bones), _p(g_pelvis), int(bool(accumulate)), _p(scratch), sbytes, _stream()),
               "anerf_pose_batch_backward")


class Profile:
    """Caller-owned HIP timing events for AnerfProfile (ABI revision 3): created with hipEventCreate through the HIP
    runtime already loaded in the process, recorded by the library around the MFMA kernels of the one-call training step,
    read back here.  `with ops.profiling(prof): step()` routes the next train_forward / backward calls through it."""

    SLOTS = {"fwd": 0, "bwd": 4, "gemm": 8, "bwd_in": 12}       # ANERF_PROF_* of include/anerf.h; + 2 * pass (0 coarse, 1 fine)

    @staticmethod
    def _hip_runtime():
        """the HIP runtime ALREADY mapped into this process (torch ships its own copy next to the system one: opening the bare
        soname could map a second runtime whose events the library's stream would not know)"""
        try:
            for line in open("/proc/self/maps"):
                path = line.rsplit(" ", 1)[-1].strip()
                if "libamdhip64.so" in path and path.startswith("/"):
                    return C.CDLL(path)
        except OSError:
            pass
        return C.CDLL("libamdhip64.so")

    def __init__(self):
        self.hip = self._hip_runtime()
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.st = _lib.AnerfProfile()
        self.reset()

    def reset(self):
        """Replace every event by a fresh, never-recorded one.  Call before each profiled step: hipEventSynchronize succeeds on
        a stale event and hipEventElapsedTime then returns the PREVIOUS step's value, so a pair the coming step does not record
        (a skipped pass, no input gradients) could not be told from a recorded one; on a never-recorded event
        hipEventElapsedTime fails and ms() returns None.  The caller synchronises before resetting (the library may still
        hold the old handles in a call it has enqueued -- events are recorded at enqueue time, so after the call returned
        they are no longer referenced)."""
        for i in range(_lib.PROF_SLOTS):
            if self.st.ev[i]:
                self.hip.hipEventDestroy(self.st.ev[i])
            ev = C.c_void_p()
            if self.hip.hipEventCreate(C.byref(ev)) != 0:
                raise RuntimeError("hipEventCreate failed")
            self.st.ev[i] = ev.value

    def ms(self, kind, which_pass):
        """elapsed milliseconds of kernel `kind` ("fwd" | "bwd" | "gemm" | "bwd_in") of pass 0 (coarse) / 1 (fine), or None if
        the pair was not recorded since the last reset()"""
        a = self.SLOTS[kind] + 2 * which_pass
        out = C.c_float()
        if self.hip.hipEventSynchronize(self.st.ev[a + 1]) != 0:
            return None
        return float(out.value) if self.hip.hipEventElapsedTime(C.byref(out), self.st.ev[a], self.st.ev[a + 1]) == 0 else None

    def __del__(self):
        try:
            for i in range(_lib.PROF_SLOTS):
                if self.st.ev[i]:
                    self.hip.hipEventDestroy(self.st.ev[i])
        except Exception:
            pass


_active_profile = None


class profiling:
    def __init__(self, prof):
        self.prof = prof

    def __enter__(self):
        global _active_profile
        self.prev, _active_profile = _active_profile, self.prof
        return self.prof

    def __exit__(self, *a):
        global _active_profile
        _active_profile = self.prev


def _forward_io(cfg, net_c, net_f, rays, skts, cyls, n_samples, n_importance, tau_v, tau_d, cut_v, cut_d, cam_idx, codes_c,
                codes_f, t_rand, u_imp, noise, noise_fine, lindisp, single_net, precision, pts_noise=None, pts_noise_is=None):
    """AnerfForwardIO of one caster call + the output dict + the tensors whose pointers it holds"""
    f = lambda t, nm: _f32c(t, nm)
    # stride-0 expanded per-frame inputs (the reference's render_path) travel as their single row
    rays, skts, cyls = f(rays, "rays"), f(first_row(skts), "skts"), f(first_row(cyls), "cyls")
    n, dev, S, Ni = rays.shape[0], rays.device, int(n_samples), int(n_importance)
    cut_v = torch.full((cfg.n_joints,), 0.5, device=dev) if cut_v is None else f(cut_v, "cut_v")
    cut_d = torch.full((cfg.n_joints,), 0.5, device=dev) if cut_d is None else f(cut_d, "cut_d")
    E = lambda *sh: torch.empty(*sh, dtype=torch.float32, device=dev)
    out = {"rgb_map": E(n, 3), "disp_map": E(n), "acc_map": E(n), "alpha": E(n, S + Ni)}
    if Ni > 0:
        out.update({"rgb0": E(n, 3), "disp0": E(n), "acc0": E(n), "alpha0": E(n, S)})
    io = _lib.AnerfForwardIO()
    io.packed_c, io.aux_c = net_c[0].data_ptr(), net_c[1].data_ptr()
    if net_f is not None:
        io.packed_f, io.aux_f = net_f[0].data_ptr(), net_f[1].data_ptr()
    io.rays, io.ray_stride = rays.data_ptr(), rays.shape[1]
    if skts.shape[0] not in (1, n):
        raise ValueError(f"skts must hold one pose or one per ray, got {tuple(skts.shape)} for {n} rays")
    io.skts, io.skt_ray_stride = skts.data_ptr(), 16 * cfg.n_joints if skts.shape[0] == n else 0
    if cyls.shape[0] not in (1, n):
        raise ValueError(f"cyls must hold one cylinder or one per ray, got {tuple(cyls.shape)} for {n} rays")
    io.cyls, io.cyl_shared = cyls.data_ptr(), int(cyls.shape[0] == 1 and n > 1)
    keep = [rays, skts, cyls, cut_v, cut_d, net_c, net_f]
    if pts_noise is not None:
        if tuple(pts_noise.shape) != (n, S, 3) or (Ni > 0 and (pts_noise_is is None or tuple(pts_noise_is.shape) != (n, Ni, 3))):
            raise ValueError("pts_noise must be [N,S,3] and, with importance samples, pts_noise_is [N,Ni,3]")
    for name, t in (("cam_idx", cam_idx), ("codes_c", codes_c), ("codes_f", codes_f), ("t_rand", t_rand), ("u_imp", u_imp),
                    ("noise", noise), ("noise_fine", noise_fine), ("pts_noise", pts_noise),
                    ("pts_noise_is", pts_noise_is if Ni > 0 else None)):
        if t is not None:
            t = f(t, name)
            keep.append(t)
            setattr(io, name, t.data_ptr())
    io.n_codes = 0 if codes_c is None else codes_c.shape[0]
    io.cutoff_v, io.cutoff_d, io.tau_v, io.tau_d = cut_v.data_ptr(), cut_d.data_ptr(), float(tau_v), float(tau_d)
    io.n_rays, io.n_samples, io.n_importance = n, S, Ni
    io.lindisp, io.single_net, io.precision = int(bool(lindisp)), int(bool(single_net)), 1 if precision == "bf16x3" else 0
    for k, v in out.items():
        setattr(io, k, v.data_ptr())
    if _active_profile is not None:
        io.profile = C.pointer(_active_profile.st)
        keep.append(_active_profile)
    if _active_step is not None:      # the training kernels read tau_v / tau_d from the device-resident step block
        io.step = _active_step.ptr
        keep.append(_active_step)
    return io, out, keep


def _workspace(fn, name, dev, *args):
    nbytes = fn(*args)
    if nbytes < 0:
        _lib.check(int(nbytes), name)
    return torch.empty(max(int(nbytes), 16) // 4, dtype=torch.float32, device=dev), int(nbytes)


def forward(cfg, net_c, net_f, rays, skts, cyls, n_samples, n_importance=0, tau_v=20.0, tau_d=20.0, cut_v=None, cut_d=None,
            cam_idx=None, codes_c=None, codes_f=None, t_rand=None, u_imp=None, noise=None, noise_fine=None, lindisp=False,
            single_net=False, precision="fp32", pts_noise=None, pts_noise_is=None):
    """RayCaster.render_rays for one caster call as ONE C call (anerf_forward): every intermediate lives in a single
    workspace tensor; returns the reference's output dict.  net_c / net_f: (packed, aux) from pack_params (which=0 for
    "fp32", which=3 for "bf16x3").  pts_noise / pts_noise_is: sample-point offsets (ray_noise_std > 0)."""
    io, out, keep = _forward_io(cfg, net_c, net_f, rays, skts, cyls, n_samples, n_importance, tau_v, tau_d, cut_v, cut_d,
                                cam_idx, codes_c, codes_f, t_rand, u_imp, noise, noise_fine, lindisp, single_net, precision,
                                pts_noise, pts_noise_is)
    lib, cc = _lib.load(), cfg.c()
    ws, nbytes = _workspace(lib.anerf_workspace_size, "anerf_workspace_size", rays.device, C.byref(cc), io.n_rays,
                            io.n_samples, io.n_importance)
    _lib.check(lib.anerf_forward(C.byref(cc), C.byref(io), _p(ws), nbytes, _stream()), "anerf_forward")
    return out


def train_forward(cfg, net_c, net_f, rays, skts, cyls, n_samples, n_importance=0, tau_v=20.0, tau_d=20.0, cut_v=None,
                  cut_d=None, cam_idx=None, codes_c=None, codes_f=None, t_rand=None, u_imp=None, noise=None, noise_fine=None,
                  lindisp=False, precision="fp32", pts_noise=None, pts_noise_is=None):
    """anerf_train_forward: `forward` with the training kernels.  Returns (output dict, state); `state` owns the workspace
    with the saved activations and every input the backward re-reads -- hand it to `backward` unchanged."""
    io, out, keep = _forward_io(cfg, net_c, net_f, rays, skts, cyls, n_samples, n_importance, tau_v, tau_d, cut_v, cut_d,
                                cam_idx, codes_c, codes_f, t_rand, u_imp, noise, noise_fine, lindisp, False, precision,
                                pts_noise, pts_noise_is)
    lib, cc = _lib.load(), cfg.c()
    ws, nbytes = _workspace(lib.anerf_train_workspace_size, "anerf_train_workspace_size", rays.device, C.byref(cc), io.n_rays,
                            io.n_samples, io.n_importance)
    _lib.check(lib.anerf_train_forward(C.byref(cc), C.byref(io), _p(ws), nbytes, _stream()), "anerf_train_forward")
    return out, {"cfg": cfg, "io": io, "keep": keep, "ws": ws, "ws_bytes": nbytes, "out": out}


def backward(state, g, packed_t_c, packed_t_f, perm, shapes_c, shapes_f, packed_i_c=None, packed_i_f=None, want_skts=False,
             want_codes_c=False, want_codes_f=False, accumulate_into=None, after_fine=None, codes_into=None, sched=None,
             after_coarse_params=None, after_coarse_weights=None):
    """anerf_backward.  g: dict of gradients of the rendered maps (keys as the output dict; rgb_map and, when hierarchical,
    rgb0 are required -- missing ones are taken as zero).  shapes_*: parameter shapes in AnerfNetGrads order (w0, b0, ...).
    accumulate_into: optional (list_c, list_f) of existing gradient tensors the parameter gradients are ADDED to in place
    (then returned as they are) instead of being written to fresh tensors.
    after_fine: optional callable; hierarchical calls are then enqueued as two halves (AnerfBackwardIO.passes = 1, then 2)
    and `after_fine()` runs in between, when everything that produces the FINE network's parameter gradients is on the
    stream -- the data-parallel path starts their all-reduce there, under the coarse pass.
    after_coarse_params: optional callable (with after_fine); runs when everything that produces the COARSE network's parameter
    gradients (weights, biases, frame codes) is on the stream.  With pose gradients requested the coarse pass is enqueued as two
    calls (passes = 4, then 8) and the callable runs in between: the coarse network's all-reduce then runs under the
    pose-gradient tail (k_encode_bwd, k_pose_reduce and the pose layer's own backward), none of which is all-reduced on an
    iteration that does not step the pose group.
    after_coarse_weights: optional callable (with after_fine, input gradients requested); runs when the COARSE network's weight /
    bias gradients are on the stream (passes = 16), in front of its input-gradient kernel: the data-parallel path starts the
    3.46 MB all-reduce there and has only the frame-code table left for after_coarse_params (behind passes = 32).
    sched: the InputSchedule the weight images were packed with (its factors multiply the same gradient columns), or None.
    Returns (grads_c, grads_f, g_skts, g_codes_c, g_codes_f)."""
    cfg, io = state["cfg"], state["io"]
    n, S, Ni = io.n_rays, io.n_samples, io.n_importance
    dev = state["ws"].device
    hier = Ni > 0
    b = _lib.AnerfBackwardIO()
    keep = []

    def gp(key, shape, required):
        t = g.get(key)
        if t is None:
            if not required:
                return None
            t = torch.zeros(shape, dtype=torch.float32, device=dev)
        t = _f32c(t, key)
        keep.append(t)
        return t.data_ptr()

    b.g_rgb, b.g_disp, b.g_acc, b.g_alpha = gp("rgb_map", (n, 3), True), gp("disp_map", (n,), False), gp("acc_map", (n,), False), \
        gp("alpha", (n, S + Ni), False)
    if hier:
        b.g_rgb0, b.g_disp0, b.g_acc0, b.g_alpha0 = gp("rgb0", (n, 3), True), gp("disp0", (n,), False), gp("acc0", (n,), False), \
            gp("alpha0", (n, S), False)
    b.packed_t_c = packed_t_c.data_ptr()
    b.packed_t_f = packed_t_f.data_ptr() if packed_t_f is not None else None
    b.packed_i_c = packed_i_c.data_ptr() if packed_i_c is not None else None
    b.packed_i_f = packed_i_f.data_ptr() if packed_i_f is not None else None
    b.perm_x, b.perm_u = perm[0].data_ptr(), perm[1].data_ptr()
    E = lambda sh: torch.empty(sh, dtype=torch.float32, device=dev)
    if accumulate_into is not None:
        grads_c, grads_f = accumulate_into
        b.accumulate = 1
    else:
        grads_c = [E(sh) for sh in shapes_c]
        grads_f = [E(sh) for sh in shapes_f] if hier else []
    for i in range(12):
        b.grads_c.w[i], b.grads_c.b[i] = grads_c[2 * i].data_ptr(), grads_c[2 * i + 1].data_ptr()
        if hier:
            b.grads_f.w[i], b.grads_f.b[i] = grads_f[2 * i].data_ptr(), grads_f[2 * i + 1].data_ptr()
    if sched is not None:
        _check_sched(cfg, sched)
        sched.fill(b.grads_c)
        sched.fill(b.grads_f)
    g_skts = E((n, cfg.n_joints, 4, 4)) if want_skts else None
    # codes_into (only with accumulate_into): the frame-code tables' gradient tensors, added to in place like the parameters'
    if codes_into is not None and accumulate_into is None:
        raise ValueError("backward: codes_into needs accumulate_into (one `accumulate` flag covers both)")
    if accumulate_into is not None and codes_into is None and (want_codes_c or want_codes_f):
        codes_into = (torch.zeros((io.n_codes, 16), dtype=torch.float32, device=dev) if want_codes_c else None,
                      torch.zeros((io.n_codes, 16), dtype=torch.float32, device=dev) if (want_codes_f and hier) else None)
    if codes_into is not None:
        g_codes_c, g_codes_f = (codes_into[0] if want_codes_c else None), (codes_into[1] if (want_codes_f and hier) else None)
    else:
        g_codes_c = E((io.n_codes, 16)) if want_codes_c else None
        g_codes_f = E((io.n_codes, 16)) if (want_codes_f and hier) else None
    b.g_skts, b.g_codes_c, b.g_codes_f = _p(g_skts), _p(g_codes_c), _p(g_codes_f)
    lib, cc = _lib.load(), cfg.c()
    if _active_profile is not None:
        b.profile = C.pointer(_active_profile.st)
    want_in = int(want_skts or want_codes_c or want_codes_f)
    scratch, sbytes = _workspace(lib.anerf_backward_scratch_size, "anerf_backward_scratch_size", dev, C.byref(cc), n, S, Ni, want_in)
    split = after_fine is not None and hier
    if not split:
        plan = (0,)
    elif after_coarse_weights is not None and want_in:
        plan = (1, 16, 32) + ((8,) if want_skts else ())
    elif after_coarse_params is not None and want_skts:
        plan = (1, 4, 8)
    else:
        plan = (1, 2)
    for passes in plan:
        b.passes = passes
        _lib.check(lib.anerf_backward(C.byref(cc), C.byref(io), C.byref(b), _p(state["ws"]), state["ws_bytes"], _p(scratch), sbytes,
                                      _stream()), "anerf_backward")
        if passes == 1:
            after_fine()
        elif passes == 16:
            after_coarse_weights()
        elif passes in (2, 4, 32) and after_coarse_params is not None:
            after_coarse_params()
    return grads_c, grads_f, g_skts, g_codes_c, g_codes_f
