"""Host-side mirror of core/networks/{nerf,embedding}.py and core/cutoff_embedder.py.

These modules own the PARAMETERS (same names and shapes as the reference's state_dict, so its
checkpoints load unchanged: pts_linears.{0..7}, alpha_linear, feature_linear, views_linears.0,
rgb_linear, framecodes.codes; embedders: cutoff_dist [24] + tau buffer) and the reference's call
signatures.  All arithmetic runs in the HIP library through ops.py -- none of it is torch math.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class SoftplusShift:
    """density_fn for --density_type softplus (core/raycasters.py:230-238); carries the shift for the kernel."""

    def __init__(self, shift):
        self.softplus_shift = float(shift)

    def __call__(self, x):
        return F.softplus(x - self.softplus_shift, beta=1)


def density_shift_of(act_fn):
    """Map the reference's act_fn argument to the kernel's density activation: None -> relu."""
    if act_fn is None or act_fn is F.relu or act_fn is torch.relu:
        return None
    if hasattr(act_fn, "softplus_shift"):
        return act_fn.softplus_shift
    raise NotImplementedError("density_fn must be F.relu or SoftplusShift(shift) for the HIP path")


class Optcodes(nn.Module):
    """Per-frame appearance codes (core/networks/embedding.py:4-45); lookup happens inside the fused kernel."""

    def __init__(self, n_codes, code_ch, idx_map=None, transform_code=False, mean=None, std=None):
        super().__init__()
        if idx_map is not None or transform_code:
            raise NotImplementedError("idx_map / transform_code are not used by any shipped config")
        self.n_codes, self.code_ch = n_codes, code_ch
        self.codes = nn.Embedding(n_codes, code_ch)
        if mean is None:
            nn.init.xavier_normal_(self.codes.weight)
        elif std > 0.:
            nn.init.normal_(self.codes.weight, mean=mean, std=std)
        else:
            nn.init.constant_(self.codes.weight, mean)

    def table_for(self, idx, training):
        """(table, idx) the kernel should use.  Eval with idx < 0 -> the mean code (embedding.py:21-22: the reference
        switches the whole call to the mean code when `idx.max() < 0`; any other negative index would crash its
        nn.Embedding).  Here the rule is applied per ray ON THE DEVICE -- the table gets the mean code as one extra row
        and negative indices are pointed at it -- so no host read of `idx` (a stream sync per caster call) is needed and the
        result is the reference's wherever the reference is defined."""
        if training or idx is None:
            return self.codes.weight, idx
        w = self.codes.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, "_eval_table_key", None) != key:
            with torch.no_grad():
                self._eval_table = torch.cat([w.detach(), w.detach().mean(0, keepdim=True)], 0)
            self._eval_table_key = key
        return self._eval_table, torch.where(idx < 0, torch.full_like(idx, float(self.n_codes)), idx)


class NeRF(nn.Module):
    """Mirror of core/networks/nerf.py:12-205 (constructor signature, parameter names, forward / raw2outputs)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_bones=0, input_ch_views=3, output_ch=4, skips=[4],
                 use_viewdirs=False, use_framecode=False, framecode_ch=16, n_framecodes=0, skel_type=None,
                 density_scale=1.0):
        super().__init__()
        if not use_viewdirs:
            raise NotImplementedError("HIP path is built for use_viewdirs=True (all shipped configs)")
        self.D, self.W = D, W
        self.input_ch, self.input_ch_bones, self.input_ch_views = input_ch, input_ch_bones, input_ch_views
        self.skips, self.use_viewdirs = skips, use_viewdirs
        self.use_framecode, self.framecode_ch, self.n_framecodes = use_framecode, framecode_ch, n_framecodes
        self.cam_ch = 1 if use_framecode else 0
        self.N_joints = 24
        self.output_ch, self.skel_type, self.density_scale = output_ch, skel_type, density_scale
        dnet = input_ch + input_ch_bones
        layers = [nn.Linear(dnet, W)]
        for i in range(D - 1):
            layers += [nn.Linear(W, W)] if i not in skips else [nn.Linear(W + dnet, W)]
        self.pts_linears = nn.ModuleList(layers)
        self.alpha_linear = nn.Linear(W, 1)
        vin = input_ch_views + (framecode_ch if use_framecode else 0) + W
        self.views_linears = nn.ModuleList([nn.Linear(vin, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.rgb_linear = nn.Linear(W // 2, 3)
        if use_framecode:
            self.framecodes = Optcodes(n_framecodes, framecode_ch)
        n_j = self.N_joints
        mv = (input_ch // n_j - 1) // 2
        md = (input_ch_views // (3 * n_j) - 1) // 2
        self.path_cfg = ops.PathConfig(multires=mv, multires_views=md, framecode_ch=framecode_ch if use_framecode else 0,
                                       density_scale=density_scale, netdepth=D, netwidth=W, skip=skips[0])
        if input_ch_bones != 3 * n_j or self.path_cfg.dim_v != input_ch or self.path_cfg.dim_d != input_ch_views:
            raise NotImplementedError("input widths do not match the reldist/reldir/relray encoders the HIP path fuses")
        self._packed = {}

    @property
    def dnet_input(self):
        return self.input_ch + self.input_ch_bones

    @property
    def vnet_input(self):
        return self.input_ch_views + (self.framecode_ch if self.use_framecode else 0) + self.W

    def named_path_params(self):
        """name -> Parameter of the 24 path tensors.  Cached: nn.Module.named_parameters() walks the module tree in Python (~45 us per
        call) and the training step asks ten times per network (weight-image staleness checks, the autograd node's input list): 0.5 ms
        of host time per step at a 2.3 ms step.  Parameter objects keep their identity through .to() / load_state_dict(); the cache is
        dropped whenever the module is converted (_apply) or a submodule is assigned."""
        c = self.__dict__.get("_npp_cache")
        if c is None:
            c = {n: p for n, p in self.named_parameters() if not n.startswith("framecodes")}
            self.__dict__["_npp_cache"] = c
        return c

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_npp_cache", None)
        return super()._apply(fn, *args, **kwargs)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__.pop("_npp_cache", None)
        super().__setattr__(name, value)

    def set_input_schedule(self, sched):
        """sched = None, or ops.InputSchedule: the embedders' --freq_schedule factors of this network's encoded input columns.  The
        kernels produce the UNSCALED encoding; the factors are folded into the weight images the fused paths consume (W diag(s)) and
        into the weight gradients (AnerfNetParams / AnerfNetGrads.sched_x, sched_u).  Set by the caster at every entry (the
        embedders are its children, not this module's)."""
        self.__dict__["_sched"] = sched

    def input_schedule(self):
        return self.__dict__.get("_sched")

    def _image_version(self, P, fold):
        sched = self.__dict__.get("_sched") if fold else None
        return tuple((p.data_ptr(), p._version) for p in P.values()) + ((sched.serial,) if sched is not None else ()), sched

    def packed(self, which=0, fold=True):
        """(stream, aux) weight images for the kernels; re-gathered only when a parameter (or the frequency schedule) changed.
        fold=False: the plain weights -- for callers that bring their own encoded input (NeRF.forward), whose rows already carry
        the schedule."""
        P = self.named_path_params()
        ver, sched = self._image_version(P, fold)
        sf, af, _, _ = ops.layout(self.path_cfg, which)
        hit = self._packed.get(which)
        if hit is None or hit[0] != ver:
            dev = next(iter(P.values())).device
            flat = hit[1] if hit is not None and hit[1].device == dev else torch.empty(sf + af, dtype=torch.float32, device=dev)
            with torch.no_grad():
                ops.pack_params(self.path_cfg, {k: v.detach() for k, v in P.items()}, which, out=flat, sched=sched)
            self._packed[which] = (ver, flat)
        flat = self._packed[which][1]
        return flat[:sf], flat[sf:]

    def params_struct(self, ver, sched):
        """(AnerfNetParams, keep-alive list) of the path parameters, cached while their addresses (and the schedule) stay what
        they are -- building it walks 24 tensors through ctypes, six times per training step otherwise"""
        P = self.named_path_params()
        ptrs = tuple(v[0] for v in ver[:len(P)])
        key = (ptrs, None if sched is None else sched.serial)
        hit = self.__dict__.get("_pstruct")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                st, keep = ops.net_params_struct({k: v.detach() for k, v in P.items()}, sched=sched)
            hit = self.__dict__["_pstruct"] = (key, st, keep)
        return hit[1], hit[2]

    def _stale(self, which, ver_sched=None):
        """(version key, flat buffer to (re)fill, parameters, schedule) when image `which` is out of date, else None.
        ver_sched: this network's _image_version(..., True), computed once by the caller for all its images"""
        P = self.named_path_params()
        ver, sched = ver_sched if ver_sched is not None else self._image_version(P, True)
        hit = self._packed.get(which)
        if hit is not None and hit[0] == ver:
            return None
        sf, af, _, _ = ops.layout(self.path_cfg, which)
        dev = next(iter(P.values())).device
        flat = hit[1] if hit is not None and hit[1].device == dev else torch.empty(sf + af, dtype=torch.float32, device=dev)
        return ver, flat, P, sched

    def codes_table(self, cam_idx):
        if not self.use_framecode:
            return None, None
        return self.framecodes.table_for(cam_idx, self.training)

    def forward(self, x):
        """x [..., input_ch+input_ch_bones+input_ch_views(+1)] -> [..., 4]   (nerf.py:133-148)."""
        stream, aux = self.packed(fold=False)
        codes = self.framecodes.codes.weight if self.use_framecode else None
        if self.use_framecode and not self.training:
            codes, idx = self.framecodes.table_for(x[..., -1], False)
            x = torch.cat([x[..., :-1], idx[..., None]], -1)
        return ops.mlp_forward(self.path_cfg, stream, aux, x, codes)

    def forward_batchify(self, inputs, chunk=1024 * 64, **kwargs):
        # the fused kernel tiles internally; `chunk` is accepted for signature compatibility (nerf.py:90-92)
        return self.forward(inputs)

    def raw2outputs(self, raw, z_vals, rays_d, raw_noise_std=0, pytest=False, B=0.01, rgb_act=torch.sigmoid,
                    act_fn=F.relu, rgb_eps=0.001, **kwargs):
        """nerf.py:150-205.  Noise: N(0,1)*std*B, or numpy-seeded uniform*std when pytest (nerf.py:176-182)."""
        if rgb_act is not torch.sigmoid or rgb_eps != 0.001:
            raise NotImplementedError("HIP composite implements rgb = sigmoid(raw)*1.002 - 0.001")
        cfg = ops.PathConfig(self.path_cfg.multires, self.path_cfg.multires_views, self.path_cfg.framecode_ch,
                             density_scale=B, softplus_shift=density_shift_of(act_fn))
        noise = None
        if raw_noise_std > 0.:
            if pytest:
                np.random.seed(0)
                noise = torch.tensor(np.random.rand(*raw.shape[:-1]) * raw_noise_std, dtype=torch.float32, device=raw.device)
            else:
                noise = torch.randn(raw.shape[:-1], device=raw.device) * (raw_noise_std * B)
        rays = torch.cat([torch.zeros_like(rays_d), rays_d], -1).contiguous()
        out = ops.composite(cfg, raw, z_vals, rays, noise)
        return out


class Embedder(nn.Module):
    """Plain positional-encoding descriptor (core/cutoff_embedder.py:9-58).  The encoding itself is fused into the
    MLP kernel; this object carries out_dim and the (no-op) schedule hooks."""

    def __init__(self, **kwargs):
        super().__init__()
        self.kwargs = kwargs
        d, nf = kwargs["input_dims"], kwargs["num_freqs"]
        self.out_dim = d * ((1 if kwargs["include_input"] else 0) + 2 * nf)
        self.freq_bands = 2. ** torch.linspace(0., kwargs["max_freq_log2"], steps=nf) if nf > 0 else torch.zeros(0)

    def forward(self, inputs, **kwargs):
        raise NotImplementedError("positional encoding is fused into the HIP MLP kernel (anerf_mlp_raw)")

    def update_threshold(self, *args, **kwargs):
        pass

    def update_tau(self, *args, **kwargs):
        pass

    def update_alpha(self, *args, **kwargs):
        pass

    def get_tau(self):
        return 0.0

    def schedule_weights(self):
        """per-band factors of the frequency schedule, or None (the plain embedder has none)"""
        return None

    def column_scale(self):
        """[out_dim] CPU fp32 factors of this embedder's output columns under the frequency schedule (1 for the raw input block)"""
        d, nf = self.kwargs["input_dims"], self.kwargs["num_freqs"]
        w = self.schedule_weights()
        if w is None:
            return torch.ones(self.out_dim)
        return torch.cat([torch.ones(d if self.kwargs["include_input"] else 0), w.repeat_interleave(2).repeat_interleave(d)])


class CutoffEmbedder(Embedder):
    """Cutoff PE state (core/cutoff_embedder.py:61-197): cutoff_dist [cutoff_dim] parameter (frozen), tau buffer,
    tau schedule.  tau and cutoff_dist are runtime kernel arguments."""

    def __init__(self, cutoff_dist=500 * 0.00035, std=0.1, normalize=False, dist_inputs=False, cutoff_inputs=False,
                 opt_cutoff=False, cutoff_dim=24, freq_schedule=False, init_alpha=0., cut_to_cutoff=False,
                 shift_inputs=False, **kwargs):
        super().__init__(**kwargs)
        if normalize or cut_to_cutoff or shift_inputs or not cutoff_inputs:
            raise NotImplementedError("HIP path: cutoff_inputs=True, no normalize / cut_to_dist / cutoff_shift (these change the "
                                      "encoding arithmetic inside the fused kernel; no shipped config uses them)")
        # opt_cutoff is accepted: the reference stores the flag and nothing reads it -- cutoff_dist stays requires_grad=False
        # (cutoff_embedder.py:83,93-94), so the path is the same with or without it
        self.opt_cutoff = opt_cutoff
        self.dist_inputs, self.cutoff_inputs, self.cutoff_dim = dist_inputs, cutoff_inputs, cutoff_dim
        self.cutoff_dist = nn.Parameter(torch.ones(cutoff_dim) * cutoff_dist, requires_grad=False)
        self.init_tau = 20.
        self.register_buffer("tau", torch.tensor(self.init_tau))
        self._tau_host = (None, None, 0.0)     # (tensor identity, version, value): host copy of the device buffer
        # --freq_schedule (cutoff_embedder.py:96-98,185-197): alpha opens the bands one after the other; its factors reach the
        # network folded into the weight images (NeRF.set_input_schedule), the kernel's encoding is unchanged
        self.freq_schedule = freq_schedule
        nf = self.kwargs["num_freqs"]
        self.freq_k = torch.arange(nf, dtype=torch.float32)          # log2 of the bands 2^0 .. 2^(nf-1)
        if freq_schedule:
            self.init_alpha = init_alpha
            self.register_buffer("sched_alpha", torch.tensor(self.init_alpha))
            self._alpha_host = (None, None, 0.0)

    def get_tau(self):
        """tau as a host float (a kernel argument).  The device buffer is only read back when it changed
        (load_state_dict, .to()); a plain `self.tau.item()` here was a host sync in every caster call."""
        key = (id(self.tau), self.tau._version)
        if self._tau_host[:2] != key:
            self._tau_host = key + (float(self.tau.item()),)
        return self._tau_host[2]

    def get_cutoff_dist(self):
        return self.cutoff_dist

    def update_threshold(self, global_step, tau_step, tau_rate, alpha_step, alpha_target):
        self.update_tau(global_step, tau_step, tau_rate)
        self.update_alpha(global_step, alpha_step, alpha_target)

    def get_alpha(self):
        """sched_alpha as a host float, read back from the buffer only when it changed (as get_tau)"""
        key = (id(self.sched_alpha), self.sched_alpha._version)
        if self._alpha_host[:2] != key:
            self._alpha_host = key + (float(self.sched_alpha.item()),)
        return self._alpha_host[2]

    def update_alpha(self, global_step, step, target=None):
        # cutoff_embedder.py:185-189 (target None -> the highest band)
        if not self.freq_schedule:
            return
        if target is None:
            target = float(self.freq_k.max())
        val = torch.tensor(self.init_alpha + (target - self.init_alpha) * global_step / float(step * 1000))
        self.sched_alpha = val.to(self.sched_alpha.device)
        self._alpha_host = (id(self.sched_alpha), self.sched_alpha._version, float(val))

    def schedule_weights(self):
        """cutoff_embedder.py:191-197: w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2 per band k, on the host in fp32"""
        if not self.freq_schedule:
            return None
        diff = torch.clamp(torch.tensor(self.get_alpha(), dtype=torch.float32) - self.freq_k, 0, 1)
        return 0.5 * (1. - torch.cos(np.pi * diff))

    def get_schedule_w(self):
        w = self.schedule_weights()
        return 1. if w is None else w.repeat_interleave(2).view(1, -1, 1)

    def update_tau(self, global_step, step, rate):
        # cutoff_embedder.py:181-183
        val = min(self.init_tau * rate ** (global_step / float(step * 1000)), 2000.)
        self.tau = torch.full_like(self.tau, val)
        self._tau_host = (id(self.tau), self.tau._version, float(np.float32(val)))   # what .item() of the fp32 buffer returns


def prepack(pairs):
    """Refresh every stale weight image among `pairs` = [(NeRF, which), ...] with ONE launch (anerf_pack_params_multi); the
    following `net.packed(which)` calls are cache hits.  A training step needs four images (W and W^T of both networks) right
    after each optimiser step: one launch instead of four."""
    jobs, done, seen, vers = [], [], set(), {}
    for net, which in pairs:
        if net is None or (id(net), which) in seen:
            continue
        seen.add((id(net), which))
        if id(net) not in vers:      # one walk over the network's 24 (address, version) pairs for all of its images
            vers[id(net)] = net._image_version(net.named_path_params(), True)
        st = net._stale(which, vers[id(net)])
        if st is None:
            continue
        ver, flat, P, sched = st
        jobs.append((net.path_cfg, net.params_struct(ver, sched), which, flat, sched))     # (the cached AnerfNetParams)
        done.append((net, which, ver, flat))
    if jobs:
        with torch.no_grad():
            ops.pack_params_multi(jobs)
        for net, which, ver, flat in done:
            net._packed[which] = (ver, flat)


def get_embedder(multires, i=0, input_dims=3, cutoff_kwargs={"cutoff": False}, skel_type=None, kc=False):
    """core/cutoff_embedder.py:199-224."""
    if i == -1:
        return nn.Identity(), input_dims
    embed_kwargs = {"include_input": True, "input_dims": input_dims, "max_freq_log2": multires - 1,
                    "num_freqs": multires, "log_sampling": True, "periodic_fns": [torch.sin, torch.cos],
                    "skel_type": skel_type}
    if cutoff_kwargs["cutoff"]:
        ck = {k: v for k, v in cutoff_kwargs.items() if k != "cutoff"}
        obj = CutoffEmbedder(**ck, **embed_kwargs)
    else:
        obj = Embedder(**embed_kwargs)
    return obj, obj.out_dim
