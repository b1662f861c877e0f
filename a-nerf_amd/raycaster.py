"""Host-side mirror of core/raycasters.py: `create_raycaster`, `RayCaster` (forward / render_rays with the
reference's signatures, state_dict key mapping, embedder schedule), and the train-side wrapper that exposes
`.module` like nn.DataParallel but runs one process per GPU with an RCCL all-reduce of gradients.

Everything numeric happens in libanerf_hip.so (pipeline.py / autograd_path.py); this file is argument plumbing.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, pipeline
from .networks import NeRF, SoftplusShift, density_shift_of, get_embedder

SUPPORTED = dict(pts_tr_type="local", kp_dist_type="reldist", bone_type="reldir", view_type="relray")


class _EncoderTag:
    """Stand-in for the reference's encoder objects in preproc_kwargs (core/encoders.py): the fused kernel
    implements exactly WorldToLocal + RelDist + VecNorm(bones) + VecNorm(rays); the tag lets callers that
    inspect `encoder_name` / `dims` keep working."""

    def __init__(self, name, dims):
        self.encoder_name, self.dims = name, dims

    def __call__(self, *a, **k):
        raise NotImplementedError(f"{self.encoder_name} is fused into the HIP MLP kernel")


def get_density_fn(args):
    if args.density_type == "relu":
        return F.relu
    if args.density_type == "softplus":
        return SoftplusShift(args.softplus_shift)
    raise NotImplementedError(f"density activation {args.density_type} is undefined")


class RayCaster(nn.Module):
    """core/raycasters.py:326-794."""

    def __init__(self, network, embed_fn, embedbones_fn, embeddirs_fn, network_fine=None, joint_coords=None,
                 single_net=False):
        super().__init__()
        self.network = network
        self.network_fine = network_fine
        self.embed_fn = embed_fn
        self.embedbones_fn = embedbones_fn
        self.embeddirs_fn = embeddirs_fn
        if joint_coords is not None:
            n_j = joint_coords.shape[-3]
            self.register_buffer("joint_coords", joint_coords.reshape(-1, n_j, 3, 3))
        self.single_net = single_net
        # "fp32" (exact fp32 MFMA) or "bf16x3" (hi/lo-split bf16 MFMAs, ~5x faster, same 1e-4 RGB bar) for the
        # no-grad render path.
        self.render_precision = "fp32"
        # Training: "fp32", or "bf16x3" = split-bf16 forward (saves fp32 activations), backward-data and weight-gradient
        # GEMM kernels.  Outputs / gradients differ from fp32 by ~1e-6 / ~1e-5 relative.
        self.train_precision = "fp32"
        self.train_route = "one_call"          # "staged": compose the per-stage autograd nodes instead (same kernels; tests)

    @torch.no_grad()
    def forward_eval(self, *args, **kwargs):
        return self.render_rays(*args, **kwargs)

    def forward(self, *args, fwd_type="", **kwargs):
        if fwd_type == "density":
            return self.render_pts_density(*args, **kwargs)
        if fwd_type == "mesh":
            return self.render_mesh_density(*args, **kwargs)
        if fwd_type == "density_color":
            # raycasters.py:352-353, 623-624: `color=True` asserts the network has `texture_linears`; no network class of the
            # reference has them (nerf.py), so the reference's own call ends in this AssertionError, message and all
            raise AssertionError("need to have texture layer!")
        if not self.training:
            return self.forward_eval(*args, **kwargs)
        return self.render_rays(*args, **kwargs)

    def _all_params(self):
        """list(self.parameters()), cached (the module-tree walk costs ~0.1 ms of host time per caster call); dropped on conversion
        (_apply) and when a child module is assigned"""
        c = self.__dict__.get("_params_cache")
        if c is None:
            c = self.__dict__["_params_cache"] = list(self.parameters())
        return c

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_params_cache", None)
        return super()._apply(fn, *args, **kwargs)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)):
            self.__dict__.pop("_params_cache", None)
        super().__setattr__(name, value)

    def _taus(self):
        """(tau_v, tau_d) as the kernels take them.  A plain Embedder has no temperature (its get_tau() reports 0.0, as the
        reference's does); its gate is switched off through the cutoff (_cutoffs), for which any POSITIVE tau will do."""
        tv = self.embed_fn.get_tau() if hasattr(self.embed_fn, "cutoff_dist") else 1.0
        td = self.embeddirs_fn.get_tau() if hasattr(self.embeddirs_fn, "cutoff_dist") else 1.0
        return tv, td

    def _bones_gated(self):
        """--cutoff_bones (core/raycasters.py:54-57): the bone embedder is a CutoffEmbedder fed the joint distances -- the bone
        directions are multiplied by the same gate as the distance encoding.  The kernels take ONE distance gate (tau_v, cut_v):
        the two embedders are built from the same arguments and stepped by the same update_embed_fns call, and a checkpoint that
        makes them differ is refused here rather than rendered wrongly."""
        fb, fv = self.embedbones_fn, self.embed_fn
        if fb is None or not hasattr(fb, "cutoff_dist"):
            return False
        key = (fb.get_tau(), fv.get_tau() if hasattr(fv, "cutoff_dist") else None, fb.cutoff_dist._version, id(fb.cutoff_dist),
               getattr(getattr(fv, "cutoff_dist", None), "_version", None))
        if self.__dict__.get("_bones_gate_ok") != key:
            if not hasattr(fv, "cutoff_dist") or fb.get_tau() != fv.get_tau() or not torch.equal(fb.cutoff_dist, fv.cutoff_dist):
                raise NotImplementedError("cutoff_bones: the bone embedder's tau / cutoff_dist differ from the distance embedder's")
            self.__dict__["_bones_gate_ok"] = key
        return True

    def _cutoffs(self, dev):
        """(cut_v, cut_d) [24] device tensors for the kernels' gates w = 1 - sigmoid(tau * (dist - cutoff)).  A plain Embedder
        (use_cutoff / cutoff_viewdir = False: core/raycasters.py:29-31,66-71; same channels, no gate) is the gate with its cutoff
        at +1e30: tau * (dist - 1e30) underflows exp() to 0, so w = 1 and dw = 0 EXACTLY."""
        out = []
        for f in (self.embed_fn, self.embeddirs_fn):
            if hasattr(f, "cutoff_dist"):
                out.append(f.cutoff_dist.detach())
            else:
                c = self.__dict__.get("_no_cutoff")
                if c is None or c.device != dev:
                    c = self.__dict__["_no_cutoff"] = torch.full((self.network.N_joints,), 1e30, dtype=torch.float32, device=dev)
                out.append(c)
        return out

    def _apply_input_schedule(self, dev):
        """--freq_schedule: hand the embedders' current band factors to the networks (folded into their weight images and weight
        gradients, NeRF.set_input_schedule); rebuilt only when an alpha changed."""
        fv, fd = self.embed_fn, self.embeddirs_fn
        sched = None
        if getattr(fv, "freq_schedule", False) or getattr(fd, "freq_schedule", False):
            key = (fv.get_alpha() if getattr(fv, "freq_schedule", False) else None,
                   fd.get_alpha() if getattr(fd, "freq_schedule", False) else None, str(dev))
            c = self.__dict__.get("_sched_cache")
            if c is None or c[0] != key:
                net = self.network
                sx = torch.cat([fv.column_scale(), torch.ones(net.input_ch_bones)])
                su = torch.cat([fd.column_scale(), torch.ones(net.framecode_ch if net.use_framecode else 0)])
                c = self.__dict__["_sched_cache"] = (key, ops.InputSchedule(sx.to(dev), su.to(dev)))
            sched = c[1]
        for net in (self.network, self.network_fine):
            if net is not None:
                net.set_input_schedule(sched)
        return sched

    def render_rays(self, ray_batch, N_samples, kp_batch, skts=None, cyls=None, bones=None, cams=None,
                    subject_idxs=None, retraw=False, lindisp=False, perturb=0., N_importance=0, network_fine=None,
                    raw_noise_std=0., ray_noise_std=0., verbose=False, ext_scale=0.001, pytest=False,
                    preproc_kwargs={}, nerf_type="nerf"):
        """Same arguments and returned dict as core/raycasters.py:361-474."""
        if skts is None or cyls is None:
            raise ValueError("skts and cyls are required (world->bone transforms and bounding cylinders)")
        if subject_idxs is not None:
            # the reference's own NeRF.forward rejects the extra input column encode_inputs appends for it
            # (torch.split sizes, nerf.py:135-137 vs raycasters.py:545-548): no network in the repository consumes it
            raise NotImplementedError("subject_idxs: no network of the reference consumes the subject column")
        n = ray_batch.shape[0]
        dev = ray_batch.device
        net_c, net_f = self.network, self.network_fine
        cfg = net_c.path_cfg
        B = preproc_kwargs.get("density_scale", net_c.density_scale)
        shift = density_shift_of(preproc_kwargs.get("density_fn", F.relu))
        cfg = ops.PathConfig(cfg.multires, cfg.multires_views, cfg.framecode_ch, density_scale=B, softplus_shift=shift,
                             cutoff_bones=self._bones_gated())
        # randomness is generated here (device tensors) and handed to the kernels as inputs
        t_rand = u_imp = noise = noise_f = pts_noise = pts_noise_is = None
        hier = N_importance > 0
        if pytest:      # the reference's numpy-seeded overrides (ray_utils.py:171-180,240-244; nerf.py:178-182)
            if perturb > 0.:
                t_rand = _rand(True, (n, N_samples), dev)
                if hier:
                    u_imp = _rand(True, (n, N_importance), dev)
            if raw_noise_std > 0.:
                noise = _rand(True, (n, N_samples), dev) * raw_noise_std
                if hier:
                    noise_f = _rand(True, (n, N_samples + N_importance), dev) * raw_noise_std
            if ray_noise_std > 0.:
                pts_noise = torch.randn(n, N_samples, 3, device=dev) * ray_noise_std
                if hier:
                    pts_noise_is = torch.randn(n, N_importance, 3, device=dev) * ray_noise_std
        elif perturb > 0. or raw_noise_std > 0. or ray_noise_std > 0.:
            # every random input of the call in ONE launch (ops.DeviceRng / anerf_rand_fill): uniforms for the stratified
            # jitter and the inverse-CDF draws, N(0,1) * raw_noise_std * B for the density logits (nerf.py:176-177), N(0,1) *
            # ray_noise_std for the sample-point offsets (raycasters.py:660,674: coarse and importance samples)
            sd = raw_noise_std * B
            t_rand, u_imp, noise, noise_f, pts_noise, pts_noise_is = self.rng().fill([
                ((n, N_samples), "uniform", 1.0) if perturb > 0. else None,
                ((n, N_importance), "uniform", 1.0) if perturb > 0. and hier else None,
                ((n, N_samples), "normal", sd) if raw_noise_std > 0. else None,
                ((n, N_samples + N_importance), "normal", sd) if raw_noise_std > 0. and hier else None,
                ((n, N_samples, 3), "normal", ray_noise_std) if ray_noise_std > 0. else None,
                ((n, N_importance, 3), "normal", ray_noise_std) if ray_noise_std > 0. and hier else None], dev)
        tau_v, tau_d = self._taus()
        cut_v, cut_d = self._cutoffs(dev)
        self._apply_input_schedule(dev)
        cam_idx = None if cams is None else cams.reshape(-1).float()
        codes_c, cam_c = net_c.codes_table(cam_idx)
        codes_f, cam_f = (net_f.codes_table(cam_idx) if net_f is not None else (None, None))
        kw = dict(cfg=cfg, ray_batch=ray_batch.contiguous(), skts=skts, cyls=cyls, n_samples=N_samples,
                  n_importance=N_importance, tau_v=tau_v, tau_d=tau_d, cut_v=cut_v, cut_d=cut_d,
                  cam_idx=cam_c if cam_c is not None else cam_idx, t_rand=t_rand, u_imp=u_imp, noise=noise,
                  noise_fine=noise_f, lindisp=lindisp, single_net=self.single_net, pts_noise=pts_noise, pts_noise_is=pts_noise_is)
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in self._all_params()) or skts.requires_grad)
        if needs_grad:
            from . import autograd_path
            return autograd_path.render_rays_train(self, kw)
        which = 3 if self.render_precision == "bf16x3" else 0
        return pipeline.render_rays_forward(net_c=net_c.packed(which), net_f=None if net_f is None else net_f.packed(which),
                                            codes_c=codes_c, codes_f=codes_f, precision=self.render_precision, **kw)

    @torch.no_grad()
    def render_pts_density(self, pts, kps, skts, bones, render_kwargs=None, subject_idxs=None, netchunk=1024 * 64,
                           network=None, color=False, v=None):
        """core/raycasters.py:597-648: raw density alpha_linear(forward_density(embed(pts))) for pts [N,1,3]
        under one pose; uses the fine network when there is one."""
        if color or v is not None or subject_idxs is not None:
            raise NotImplementedError("color / precomputed v / subject_idxs are not used by the shipped configs")
        net = network if network is not None else (self.network_fine if self.network_fine is not None else self.network)
        self._apply_input_schedule(pts.device)
        if network is not None:
            network.set_input_schedule(self.network.input_schedule())
        stream, aux = net.packed()
        tau_v, _ = self._taus()
        c = net.path_cfg
        cfg = ops.PathConfig(c.multires, c.multires_views, c.framecode_ch, density_scale=c.density_scale, cutoff_bones=self._bones_gated())
        return ops.density(cfg, stream, aux, pts, skts, tau_v, self._cutoffs(pts.device)[0])

    @torch.no_grad()
    def render_mesh_density(self, kps, skts, bones, subject_idxs=None, radius=1.0, res=64, render_kwargs=None,
                            netchunk=1024 * 64, v=None):
        """core/raycasters.py:579-595: density on a (res+1)^3 grid centred at the root joint, for marching cubes."""
        t = np.linspace(-radius, radius, res + 1)
        grid = np.stack(np.meshgrid(t, t, t), axis=-1).astype(np.float32)
        sh = grid.shape
        pts = torch.tensor(grid.reshape(-1, 3), device=kps.device) + kps[0, 0]
        raw = self.render_pts_density(pts.reshape(-1, 1, 3), kps, skts, bones, render_kwargs, subject_idxs, netchunk, v=v)
        return raw[..., :1].reshape(*sh[:-1]).transpose(1, 0)

    # ------------------------------------------------------------------ bookkeeping identical to the reference
    def update_embed_fns(self, global_step, args):
        fns = [self.embed_fn, self.embeddirs_fn] + ([self.embedbones_fn] if self.embedbones_fn is not None else [])
        for f in fns:
            f.update_threshold(global_step, args.cutoff_step, args.cutoff_rate, args.freq_schedule_step, args.multires - 1)

    @staticmethod
    def _ckpt_key(k):
        if k.endswith("_fine"):
            return f"{k}_state_dict"
        if k.endswith("_fn"):
            return f"{k.split('_fn')[0]}_state_dict"
        if k == "network":
            return "network_fn_state_dict"
        return f"{k}_state_dict"

    def state_dict(self):
        """Checkpoint layout of the reference (raycasters.py:752-766): one sub-dict per child module."""
        return {self._ckpt_key(k): m.state_dict() for k, m in self._modules.items() if m is not None}

    def load_state_dict(self, ckpt, strict=True):
        for k, m in self._modules.items():
            if m is None:
                continue
            key = self._ckpt_key(k)
            try:
                m.load_state_dict(ckpt[key], strict=strict)
            except (KeyError, RuntimeError):
                if k.startswith("network"):
                    cur = m.state_dict()
                    ok = {n: v for n, v in ckpt[key].items() if n in cur and cur[n].shape == v.shape}
                    m.load_state_dict(ok, strict=False)
                else:
                    print(f"Error occurred when loading state dict for {key}. The entity is not in the state dict?")

    # ---- the device generator behind perturb / raw_noise_std / ray_noise_std (ops.DeviceRng): seeding and checkpointing
    def rng(self):
        if getattr(self, "_rng", None) is None:
            self._rng = ops.DeviceRng()
        return self._rng

    def manual_seed(self, seed):
        """pin this caster's random stream to `seed` (otherwise it follows torch.manual_seed / torch.initial_seed)"""
        self.rng().manual_seed(seed)
        return self

    def rng_state(self):
        return self.rng().state_dict()

    def set_rng_state(self, sd):
        self.rng().load_state_dict(sd)

    def get_embed_fns(self):
        return self.embed_fn, self.embedbones_fn, self.embeddirs_fn

    def get_networks(self):
        return self.network, self.network_fine


def _rand(pytest, shape, dev):
    """torch.rand on the device, or the reference's numpy-seeded override (ray_utils.py:171-180,240-244)."""
    if pytest:
        np.random.seed(0)
        return torch.tensor(np.random.rand(*shape), dtype=torch.float32, device=dev)
    return torch.rand(*shape, device=dev)


class RayParallel(nn.Module):
    """Train-side wrapper standing where the reference puts nn.DataParallel (raycasters.py:157): exposes `.module`
    (trainer.py:265,270,504 reach through it).  Parallelism is one process per GPU: each rank renders its own
    ray batch and `sync_gradients()` all-reduces one flat fp32 bucket over RCCL (see parallel.py)."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def sync_gradients(self, group=None):
        """Average the gradients of every trainable parameter over the ranks with ONE all-reduce of a flat fp32 bucket
        (parallel.GradBucket); call after loss.backward().  No-op in a single process."""
        if getattr(self, "_bucket", None) is None:
            from .parallel import GradBucket
            self._bucket = GradBucket([p for p in self.module.parameters() if p.requires_grad])
        return self._bucket.all_reduce_mean(group)


def get_grad_vars(args, ray_caster):
    network, network_fine = ray_caster.get_networks()
    if getattr(args, "finetune", False) and getattr(args, "fix_layer", 0) > 0:
        for net in (network, network_fine):
            for i, l in enumerate(net.pts_linears):
                if i < args.fix_layer:
                    for p in l.parameters():
                        p.requires_grad = False
    out = []
    mods = [network] + ([network_fine] if not args.single_net else []) + list(ray_caster.get_embed_fns())
    for m in mods:
        if m is not None:
            out += [p for p in m.parameters() if p.requires_grad]
    return out


def create_raycaster(args, data_attrs, device=None):
    """core/raycasters.py:17-184: same arguments, same 6-tuple
    (render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, loaded_ckpt)."""
    for k, v in SUPPORTED.items():
        if getattr(args, k) != v:
            raise NotImplementedError(f"{k}={getattr(args, k)!r}: the HIP path fuses {SUPPORTED} (all shipped configs)")
    if not args.use_viewdirs:
        raise NotImplementedError("HIP path needs use_viewdirs")
    if args.use_cutoff and not args.cutoff_inputs:
        raise NotImplementedError("use_cutoff without cutoff_inputs (raw input ungated, bands gated) is not in the fused encoder")
    if args.multires_bones != 0:
        raise NotImplementedError("multires_bones must be 0 (identity bone embedding)")
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    skel_type = data_attrs["skel_type"]
    n_j = len(skel_type.joint_names)
    n_framecodes = data_attrs["n_views"] if args.n_framecodes is None else args.n_framecodes
    # normalize_cutoff: the reference hands it over under a key its CutoffEmbedder does not read ("normalize_cutoff" vs the
    # constructor's `normalize`, core/raycasters.py:32 / cutoff_embedder.py:64,77): the flag changes nothing there, nor here
    cutoff_kwargs = {"cutoff": args.use_cutoff, "normalize_cutoff": args.normalize_cutoff, "cutoff_dist": args.cutoff_mm * args.ext_scale,
                     "cutoff_inputs": args.cutoff_inputs, "opt_cutoff": args.opt_cutoff, "cutoff_dim": n_j,
                     "freq_schedule": args.freq_schedule, "init_alpha": args.init_freq}
    embed_fn, input_ch = get_embedder(args.multires, args.i_embed, input_dims=n_j, skel_type=skel_type,
                                      cutoff_kwargs=dict(cutoff_kwargs, dist_inputs=False, cut_to_cutoff=args.cut_to_dist,
                                                         shift_inputs=args.cutoff_shift))
    embedbones_fn, input_ch_bones = get_embedder(0, args.i_embed, input_dims=3 * n_j, skel_type=skel_type,
                                                 cutoff_kwargs=dict(cutoff_kwargs, dist_inputs=True) if args.cutoff_bones
                                                 else {"cutoff": False})
    embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed, input_dims=3 * n_j, skel_type=skel_type,
                                                cutoff_kwargs=dict(cutoff_kwargs, dist_inputs=True) if args.cutoff_viewdir
                                                else {"cutoff": False})
    nerf_kwargs = dict(D=args.netdepth, W=args.netwidth, input_ch=input_ch, input_ch_bones=input_ch_bones,
                       input_ch_views=input_ch_views, output_ch=5 if args.N_importance > 0 else 4, skips=[4],
                       use_viewdirs=args.use_viewdirs, use_framecode=args.opt_framecode, framecode_ch=args.framecode_size,
                       n_framecodes=n_framecodes, skel_type=skel_type, density_scale=args.density_scale)
    model = NeRF(**nerf_kwargs)
    model_fine = None
    if args.N_importance > 0:
        model_fine = model if args.single_net else NeRF(**nerf_kwargs)
    ray_caster = RayCaster(model, embed_fn, embedbones_fn, embeddirs_fn, network_fine=model_fine,
                           joint_coords=torch.tensor(data_attrs["joint_coords"]), single_net=args.single_net).to(device)
    grad_vars = get_grad_vars(args, ray_caster)
    optimizer = torch.optim.Adam(params=grad_vars, lr=args.lrate, betas=(0.9, 0.999))
    start, loaded_ckpt = 0, None
    if args.ft_path is not None and args.ft_path != "None":
        ckpts = [args.ft_path]
    else:
        d = os.path.join(args.basedir, args.expname)
        ckpts = [os.path.join(d, f) for f in sorted(os.listdir(d)) if "tar" in f and "pose" not in f] if os.path.isdir(d) else []
    if len(ckpts) > 0 and not args.no_reload:
        loaded_ckpt = torch.load(ckpts[-1], map_location=device)
        start = loaded_ckpt["global_step"]
        ray_caster.load_state_dict(loaded_ckpt)
        if not args.finetune and "optimizer_state_dict" in loaded_ckpt:
            optimizer.load_state_dict(loaded_ckpt["optimizer_state_dict"])
        if args.finetune:
            start = 0
    preproc_kwargs = {"pts_tr_fn": _EncoderTag("W2LEncoder", n_j), "kp_input_fn": _EncoderTag("RelDist", n_j),
                      "view_input_fn": _EncoderTag("VecNorm", 3 * n_j), "bone_input_fn": _EncoderTag("VecNorm", 3 * n_j),
                      "density_scale": args.density_scale, "density_fn": get_density_fn(args)}
    render_kwargs_train = {"ray_caster": RayParallel(ray_caster), "perturb": args.perturb, "N_importance": args.N_importance,
                           "N_samples": args.N_samples, "use_viewdirs": args.use_viewdirs, "raw_noise_std": args.raw_noise_std,
                           "ray_noise_std": args.ray_noise_std, "ext_scale": args.ext_scale, "preproc_kwargs": preproc_kwargs,
                           "lindisp": args.lindisp, "nerf_type": args.nerf_type}
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test.update(ray_caster=ray_caster, preproc_kwargs=dict(preproc_kwargs), perturb=False, raw_noise_std=0.,
                              ray_noise_std=0.)
    optimizer.zero_grad()
    return render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer, loaded_ckpt
