"""CPU oracle for the A-NeRF ray-march hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU fp32 ops, the algorithm of the reference's hot path
(LemonATsu/A-NeRF: core/raycasters.py + core/encoders.py + core/cutoff_embedder.py +
core/networks/nerf.py + core/utils/ray_utils.py).  Each function cites the reference file:line it
follows.  It exists to CHECK the HIP path; it is not a product path and not a fallback:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: PINNED.  tests/golden/gen_golden.py imports the reference itself (in the build
container, where /root/reference exists), runs it on seeded synthetic inputs and commits the outputs
as tests/golden/*.npz; tests/test_oracle_golden.py checks this file against those vectors.

All tensors fp32, differentiable (autograd gives the gradient oracle for the backward kernels).
"""
import numpy as np
import torch
import torch.nn.functional as F

N_JOINTS = 24


class OracleConfig:
    """Static configuration of the path (reference: create_raycaster, raycasters.py:17-184)."""

    def __init__(self, multires=7, multires_views=4, framecode_ch=0, density_scale=1.0,
                 softplus_shift=None, netwidth=256, netdepth=8, skip=4):
        self.multires = multires
        self.multires_views = multires_views
        self.framecode_ch = framecode_ch
        self.density_scale = density_scale
        self.softplus_shift = softplus_shift  # None -> relu density (raycasters.py:230-238)
        self.W, self.D, self.skip = netwidth, netdepth, skip
        self.dim_v = N_JOINTS * (1 + 2 * multires)
        self.dim_r = N_JOINTS * 3
        self.dim_d = N_JOINTS * 3 * (1 + 2 * multires_views)
        self.dim_x = self.dim_v + self.dim_r


# ------------------------------------------------------------------------------------------------
# A2: ray bounds inside the x-z cylinder           (core/utils/ray_utils.py:292-344)
# ------------------------------------------------------------------------------------------------
def ray_bounds(rays_o, rays_d, cyls, near, far):
    """near/far [N,1] overridden by the ray/circle intersection in the ground plane.

    near/far inputs are the placeholder bounds (0 and 1 from render(), trainer.py:83,131).
    Rows whose ray misses the circle (Q = NaN) take the nan-mean over this call's rays (:328-342).
    """
    g = [0, 2]
    p_near = (rays_o + rays_d * near)[:, g]
    p_far = (rays_o + rays_d * far)[:, g]
    nc = cyls[:, :2] - p_near
    nf = p_far - p_near
    nf_len = nf.norm(dim=-1)
    scale = rays_d[:, g].norm(dim=-1, keepdim=True)
    cross = nc[:, 0] * nf[:, 1] - nc[:, 1] * nf[:, 0]
    dist = (cross.abs() / nf_len)[:, None]
    Q = (cyls[:, 2:3] ** 2 - dist ** 2) ** 0.5
    K = ((nc * nf).sum(-1) / nf_len)[:, None]
    inside = (Q < K).float()
    new_near = near + inside * (K - Q) / scale
    new_far = near + (K + Q) / scale
    bad = torch.isnan(Q[:, 0])
    if bool(torch.isnan(new_near).any()):
        ok_n = ~torch.isnan(new_near[:, 0])
        ok_f = ~torch.isnan(new_far[:, 0])
        new_near = new_near.clone()
        new_far = new_far.clone()
        new_near[bad] = new_near[ok_n].mean() if bool(ok_n.any()) else near[bad]
        new_far[bad] = new_far[ok_f].mean() if bool(ok_f.any()) else far[bad]
    return new_near, new_far


# ------------------------------------------------------------------------------------------------
# A3: coarse sample depths                          (core/utils/ray_utils.py:204-251)
# ------------------------------------------------------------------------------------------------
def coarse_z(near, far, n_samples, t_rand=None, lindisp=False):
    """z [N,S]; t_rand [N,S] in [0,1) enables stratified jitter (perturb>0)."""
    t = torch.linspace(0.0, 1.0, n_samples)[None]
    if lindisp:
        z = 1.0 / (1.0 / near * (1.0 - t) + 1.0 / far * t)
    else:
        z = near * (1.0 - t) + far * t
    if t_rand is not None:
        mid = 0.5 * (z[:, 1:] + z[:, :-1])
        hi = torch.cat([mid, z[:, -1:]], -1)
        lo = torch.cat([z[:, :1], mid], -1)
        z = lo + (hi - lo) * t_rand
    return z


# ------------------------------------------------------------------------------------------------
# A4/A5: world->bone transform and skeleton-relative features   (core/encoders.py:8-37,110-122,181-193)
# ------------------------------------------------------------------------------------------------
def bone_features(pts, rays_d, skts):
    """pts [N,S,3], rays_d [N,3], skts [N or 1,24,4,4] ->
    v [N,S,24] (distance to each joint in bone space), r [N,S,72] (unit direction, 3j+c),
    e [N,72] (unit ray direction in bone space, per ray)."""
    R = skts[:, :, :3, :3]                      # [n,24,3,3]
    t = skts[:, :, :3, 3]                       # [n,24,3]
    y = torch.einsum("njab,nsb->nsja", R.expand(pts.shape[0], -1, -1, -1), pts) + t[:, None].expand(pts.shape[0], -1, -1, -1)
    v = y.norm(dim=-1)
    r = F.normalize(y, dim=-1, p=2, eps=1e-12).flatten(start_dim=2)
    dl = torch.einsum("njab,nb->nja", R.expand(rays_d.shape[0], -1, -1, -1), rays_d)
    e = F.normalize(dl, dim=-1, p=2, eps=1e-12).flatten(start_dim=1)
    return v, r, e


# ------------------------------------------------------------------------------------------------
# A6: positional encoding with sigmoid cutoff        (core/cutoff_embedder.py:111-174)
# ------------------------------------------------------------------------------------------------
def schedule_w(alpha, n_freq):
    """--freq_schedule (CutoffEmbedder.get_schedule_w, core/cutoff_embedder.py:191-197): weight of band k = 0..n_freq-1,
    w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2, the same for its sin and its cos channel; alpha follows
    update_alpha (:185-189).  [n_freq] fp32."""
    k = torch.arange(n_freq, dtype=torch.float32)
    diff = torch.clamp(torch.tensor(alpha, dtype=torch.float32) - k, 0, 1)
    return 0.5 * (1. - torch.cos(np.pi * diff))


def schedule_alpha(global_step, step, target, init_alpha=0.):
    """CutoffEmbedder.update_alpha (core/cutoff_embedder.py:185-189); RayCaster.update_embed_fns passes target = multires - 1 to
    BOTH embedders (raycasters.py:731-748)."""
    return float(torch.tensor(init_alpha + (target - init_alpha) * global_step / float(step * 1000)))


def cutoff_pe(x, dist, n_freq, tau, cutoff, per_joint, sched_alpha=None, gated=True):
    """x [...,C] with C = 24*per_joint (per_joint=1 for v, 3 for ray dirs); dist [...,24].
    Returns [..., C*(1+2*n_freq)], channel = k*C + c with k=0 raw, 1+2f sin(2^f x), 2+2f cos(2^f x);
    every band (raw input included: cutoff_inputs=True) is gated by w_j = 1 - sigmoid(tau*(dist_j - c_j)).
    sched_alpha: the frequency schedule's alpha (sin / cos bands times schedule_w, the raw input untouched:
    cutoff_embedder.py:159-163) or None.  gated=False: the plain Embedder of use_cutoff / cutoff_viewdir = False
    (cutoff_embedder.py:9-45: same channels, no gate)."""
    bands = [x]
    sw = None if sched_alpha is None else schedule_w(sched_alpha, n_freq)
    for f in range(n_freq):
        m = 1.0 if sw is None else sw[f]
        bands.append(torch.sin(x * (2.0 ** f)) * m)
        bands.append(torch.cos(x * (2.0 ** f)) * m)
    out = torch.stack(bands, dim=-2)
    if gated:
        w = 1.0 - torch.sigmoid(tau * (dist - cutoff))
        w = w.repeat_interleave(per_joint, dim=-1)
        out = out * w[..., None, :]
    return out.flatten(start_dim=-2)


def encode(cfg, pts, rays_d, skts, tau_v, tau_d, cut_v, cut_d, cam_idx=None, sched_alpha=None, gate_v=True, gate_d=True,
           gate_r=False):
    """MLP input X [N,S,dim_x + dim_d (+1)]  (RayCaster.encode_inputs + run_network cat,
    raycasters.py:476-577).  embedbones_fn is the identity (multires_bones=0), or with gate_r (--cutoff_bones,
    raycasters.py:54-57) a CutoffEmbedder without bands fed the joint distances: r_j * w_j."""
    v, r, e = bone_features(pts, rays_d, skts)
    if gate_r:
        r = cutoff_pe(r, v, 0, tau_v, cut_v, 3)
    V = cutoff_pe(v, v, cfg.multires, tau_v, cut_v, 1, sched_alpha, gate_v)
    eS = e[:, None, :].expand(-1, pts.shape[1], -1)
    Dv = cutoff_pe(eS, v, cfg.multires_views, tau_d, cut_d, 3, sched_alpha, gate_d)
    parts = [V, r, Dv]
    if cam_idx is not None:
        parts.append(cam_idx.view(-1, 1, 1).expand(-1, pts.shape[1], 1))
    return torch.cat(parts, -1)


# ------------------------------------------------------------------------------------------------
# A9: the radiance/density MLP                       (core/networks/nerf.py:94-148)
# ------------------------------------------------------------------------------------------------
def mlp(cfg, P, X, eval_mean_code=False):
    """P: dict 'pts_linears.i.weight' ... (reference state_dict names); X [...,dim_x+dim_d(+1)] -> raw [...,4]."""
    x = X[..., :cfg.dim_x]
    u = X[..., cfg.dim_x:cfg.dim_x + cfg.dim_d]
    h = x
    for i in range(cfg.D):
        h = F.relu(F.linear(h, P[f"pts_linears.{i}.weight"], P[f"pts_linears.{i}.bias"]))
        if i == cfg.skip:
            h = torch.cat([x, h], -1)
    sigma = F.linear(h, P["alpha_linear.weight"], P["alpha_linear.bias"])
    feat = F.linear(h, P["feature_linear.weight"], P["feature_linear.bias"])
    vin = [feat, u]
    if cfg.framecode_ch > 0:
        idx = X[..., cfg.dim_x + cfg.dim_d]
        E = P["framecodes.codes.weight"]
        if eval_mean_code and float(idx.max()) < 0:            # embedding.py:21-22
            code = E.mean(0, keepdim=True).expand(*idx.shape, -1)
        else:
            code = E[idx.long()]
        vin.append(code)
    g = F.relu(F.linear(torch.cat(vin, -1), P["views_linears.0.weight"], P["views_linears.0.bias"]))
    rgb = F.linear(g, P["rgb_linear.weight"], P["rgb_linear.bias"])
    return torch.cat([rgb, sigma], -1)


def density_query(cfg, P, pts, skts, tau_v=20.0, cut_v=None, sched_alpha=None, gate_v=True, gate_r=False):
    """RayCaster.render_pts_density (core/raycasters.py:597-648): pts [N,1,3] under one pose -> alpha_linear(
    forward_density([embed(v), r])) [N,1,1].  sched_alpha / gate_v / gate_r: the embedder variants of encode()."""
    cut_v = torch.full((N_JOINTS,), 0.5) if cut_v is None else cut_v
    v, r, _ = bone_features(pts, torch.ones(pts.shape[0], 3), skts)
    if gate_r:
        r = cutoff_pe(r, v, 0, tau_v, cut_v, 3)
    x = torch.cat([cutoff_pe(v, v, cfg.multires, tau_v, cut_v, 1, sched_alpha, gate_v), r], -1)
    h = x
    for i in range(cfg.D):
        h = F.relu(F.linear(h, P[f"pts_linears.{i}.weight"], P[f"pts_linears.{i}.bias"]))
        if i == cfg.skip:
            h = torch.cat([x, h], -1)
    return F.linear(h, P["alpha_linear.weight"], P["alpha_linear.bias"])


def mesh_density(cfg, P, kps, skts, radius=1.0, res=64, tau_v=20.0, cut_v=None):
    """RayCaster.render_mesh_density (core/raycasters.py:579-595)."""
    t = np.linspace(-radius, radius, res + 1)
    grid = np.stack(np.meshgrid(t, t, t), axis=-1).astype(np.float32)
    pts = torch.tensor(grid.reshape(-1, 3)) + kps[0, 0]
    raw = density_query(cfg, P, pts.reshape(-1, 1, 3), skts, tau_v, cut_v)
    return raw[..., :1].reshape(*grid.shape[:-1]).transpose(1, 0)


# ------------------------------------------------------------------------------------------------
# A10: alpha compositing                             (core/networks/nerf.py:150-205)
# ------------------------------------------------------------------------------------------------
def density_act(cfg, x):
    if cfg.softplus_shift is None:
        return F.relu(x)
    return F.softplus(x - cfg.softplus_shift, beta=1)


def composite(cfg, raw, z, rays_d, noise=None):
    """raw [N,S,4], z [N,S], rays_d [N,3], noise [N,S] or None (already scaled as the caller wants)."""
    B = cfg.density_scale
    dn = rays_d.norm(dim=-1, keepdim=True)
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)], -1) * dn
    rgb = torch.sigmoid(raw[..., :3]) * 1.002 - 0.001
    pre = raw[..., 3] / B
    if noise is not None:
        pre = pre + noise
    alpha = 1.0 - torch.exp(-density_act(cfg, pre) * delta)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)[:, :-1]
    w = alpha * T
    rgb_map = (w[..., None] * rgb).sum(-2)
    depth = (w * z).sum(-1)
    acc = w.sum(-1)
    disp = 1.0 / torch.maximum(torch.full_like(depth, 1e-10), depth / (acc + 1e-10))
    disp = disp * (~torch.isclose(acc, torch.zeros(()))).float()
    return {"rgb_map": rgb_map, "disp_map": disp, "acc_map": torch.minimum(acc, torch.ones(())),
            "weights": w, "alpha": alpha}


# ------------------------------------------------------------------------------------------------
# A11: importance resampling                          (core/utils/ray_utils.py:157-201,255-289)
# ------------------------------------------------------------------------------------------------
def importance_z(z, weights, n_imp, u=None, single_net=False):
    """z [N,S], weights [N,S] -> z_samples [N,Ni] (detached), z_merged [N,S+Ni], sorted_idx [N,S+Ni].
    u [N,Ni] uniform numbers (perturb>0) or None -> linspace(0,1,Ni) (deterministic)."""
    bins = 0.5 * (z[:, 1:] + z[:, :-1])
    if single_net:
        wl, wk, wu = weights[:, :-2], weights[:, 1:-1], weights[:, 2:]
        pw = 0.5 * (torch.maximum(wl, wk) + torch.maximum(wk, wu)) + 0.01
    else:
        pw = weights[:, 1:-1]
    pw = pw + 1e-5
    pdf = pw / pw.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    if u is None:
        u = torch.linspace(0.0, 1.0, n_imp)[None].expand(z.shape[0], -1)
    u = u.contiguous()
    k = torch.searchsorted(cdf.detach(), u, right=True)
    lo = (k - 1).clamp(min=0)
    hi = k.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    zs = (b_lo + (u - c_lo) / den * (b_hi - b_lo)).detach()
    zm, idx = torch.sort(torch.cat([z, zs], -1), -1)
    return zs, zm, idx


# ------------------------------------------------------------------------------------------------
# A1/A13: the whole caster call                       (core/raycasters.py:361-474,711-724)
# ------------------------------------------------------------------------------------------------
def render_rays(cfg, P, P_fine, ray_batch, skts, cyls, n_samples, n_importance=0,
                tau_v=20.0, tau_d=20.0, cut_v=None, cut_d=None, cam_idx=None,
                t_rand=None, u_imp=None, noise=None, noise_fine=None, lindisp=False,
                single_net=False, eval_mean_code=False, return_extras=False, pts_noise=None, pts_noise_is=None,
                sched_alpha=None, gate_v=True, gate_d=True, gate_r=False):
    """ray_batch [N,>=8] = (o3,d3,near,far[,viewdirs3]); returns the reference's output dict.
    pts_noise [N,S,3] / pts_noise_is [N,Ni,3]: `pts + randn_like(pts) * ray_noise_std` of RayCaster.sample_pts /
    sample_pts_is (raycasters.py:650-677), the random part passed in; the merged samples of the fine pass keep their own
    offsets (the reference merges the ENCODINGS of the two point sets by the sort order, raycasters.py:679-709)."""
    cut_v = torch.full((N_JOINTS,), 0.5) if cut_v is None else cut_v
    cut_d = torch.full((N_JOINTS,), 0.5) if cut_d is None else cut_d
    o, d = ray_batch[:, 0:3], ray_batch[:, 3:6]
    near, far = ray_bounds(o, d, cyls, ray_batch[:, 6:7], ray_batch[:, 7:8])
    z = coarse_z(near, far, n_samples, t_rand, lindisp)
    pts = o[:, None] + d[:, None] * z[..., None]
    if pts_noise is not None:
        pts = pts + pts_noise
    enc_kw = dict(sched_alpha=sched_alpha, gate_v=gate_v, gate_d=gate_d, gate_r=gate_r)
    X = encode(cfg, pts, d, skts, tau_v, tau_d, cut_v, cut_d, cam_idx, **enc_kw)
    raw = mlp(cfg, P, X, eval_mean_code)
    out = composite(cfg, raw, z, d, noise)
    extras = {"near": near, "far": far, "z_vals": z, "X": X, "raw": raw, "weights": out["weights"]}
    ret = {"rgb_map": out["rgb_map"], "disp_map": out["disp_map"], "acc_map": out["acc_map"], "alpha": out["alpha"]}
    if n_importance > 0:
        zs, zm, idx = importance_z(z, out["weights"], n_importance, u_imp, single_net)
        pts_f = o[:, None] + d[:, None] * zm[..., None]
        pts_n = o[:, None] + d[:, None] * zs[..., None]
        if pts_noise is not None:
            pts_n = pts_n + pts_noise_is
            pts_f = pts_f + torch.gather(torch.cat([pts_noise, pts_noise_is], 1), 1, idx[..., None].expand(-1, -1, 3))
        Xf = encode(cfg, pts_f, d, skts, tau_v, tau_d, cut_v, cut_d, cam_idx, **enc_kw)
        if single_net:
            # only the new samples go through the (shared) net; raw outputs are merged (raycasters.py:462-469)
            Xn = encode(cfg, pts_n, d, skts, tau_v, tau_d, cut_v, cut_d, cam_idx, **enc_kw)
            raw_n = mlp(cfg, P_fine, Xn, eval_mean_code)
            raw_f = torch.gather(torch.cat([raw, raw_n], 1), 1, idx[..., None].expand(-1, -1, 4))
        else:
            raw_f = mlp(cfg, P_fine, Xf, eval_mean_code)
        fine = composite(cfg, raw_f, zm, d, noise_fine)
        ret = {"rgb_map": fine["rgb_map"], "disp_map": fine["disp_map"], "acc_map": fine["acc_map"],
               "alpha": fine["alpha"], "rgb0": ret["rgb_map"], "disp0": ret["disp_map"],
               "acc0": ret["acc_map"], "alpha0": ret["alpha"]}
        extras.update({"z_samples": zs, "z_fine": zm, "sorted_idx": idx, "raw_fine": raw_f,
                       "weights_fine": fine["weights"]})
    if return_extras:
        ret["_extras"] = extras
    return ret


def make_ray_batch(rays_o, rays_d, near=0.0, far=1.0):
    """render()'s ray-batch assembly [N,11] (core/trainer.py:116-135, use_viewdirs=True)."""
    vd = rays_d / rays_d.norm(dim=-1, keepdim=True)
    ones = torch.ones_like(rays_d[:, :1])
    return torch.cat([rays_o, rays_d, near * ones, far * ones, vd], -1)


def render_chunked(chunk, ray_batch, skts, cyls, cam_idx=None, **kw):
    """batchify_rays (core/trainer.py:64-79): chunk over rays, concatenate dict entries."""
    outs = []
    per_ray = {k: kw.pop(k) for k in ["t_rand", "u_imp", "noise", "noise_fine"] if k in kw and kw[k] is not None}
    for i in range(0, ray_batch.shape[0], chunk):
        sl = slice(i, i + chunk)
        sk = skts[sl] if skts.shape[0] > 1 else skts
        outs.append(render_rays(ray_batch=ray_batch[sl], skts=sk, cyls=cyls[sl],
                                cam_idx=None if cam_idx is None else cam_idx[sl],
                                **{k: v[sl] for k, v in per_ray.items()}, **kw))
    return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0] if k != "_extras"}


# ------------------------------------------------------------------------------------------------
# caller-side loss / PSNR                             (core/trainer.py:8,353-380)
# ------------------------------------------------------------------------------------------------
def nerf_loss(ret, target, bgs, loss="MSE", coarse_weight=1.0):
    fn = (lambda a, b: ((a - b) ** 2).mean()) if loss == "MSE" else (lambda a, b: (a - b).abs().mean())
    pred = ret["rgb_map"] + (1.0 - ret["acc_map"])[..., None] * bgs
    total = fn(pred, target)
    if "rgb0" in ret:
        pred0 = ret["rgb0"] + (1.0 - ret["acc0"])[..., None] * bgs
        total = total + coarse_weight * fn(pred0, target)
    return total, pred


def psnr(pred, target):
    return float(-10.0 * torch.log10(((pred - target) ** 2).mean()))


def params_from_numpy(d, requires_grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=requires_grad) for k, v in d.items()}


# ---------------------------------------------------------------------------------------------------------------
# forward kinematics of the pose-refinement layer
# ---------------------------------------------------------------------------------------------------------------
SMPL_PARENTS = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]   # skeleton_utils.py:98-104


def axis_angle_to_matrix(a):
    """pytorch3d.transforms.axis_angle_to_matrix, restated from the published source (third-party dependency of the
    reference, absent from this image; called by core/utils/skeleton_utils.py:411-412 axisang_to_rot):
    axis_angle_to_quaternion (half-angle, Taylor branch below 1e-6) then quaternion_to_matrix.  Cross-checked in
    tests/golden/gen_golden_fk.py against scipy Rotation.from_rotvec, which the reference itself uses for the same map
    (skeleton_utils.py:349)."""
    ang = torch.norm(a, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), a * k], -1)
    r, i, j, kk = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
                     two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
                     two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(a.shape[:-1] + (3, 3))


def rot6d_to_rotmat(x):
    """core/utils/skeleton_utils.py:420-436: 6D rotation (first two columns of R, row-major) -> R by Gram-Schmidt."""
    sh = x.shape[:-1]
    x = x.reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / a1.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    v = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    b3 = torch.stack([b1[:, 1] * b2[:, 2] - b1[:, 2] * b2[:, 1], b1[:, 2] * b2[:, 0] - b1[:, 0] * b2[:, 2],
                      b1[:, 0] * b2[:, 1] - b1[:, 1] * b2[:, 0]], -1)
    return torch.stack((b1, b2, b3), dim=-1).reshape(*sh, 3, 3)


def fk_chain(bones, rest_pose, pelvis=None):
    """PoseOptLayer.calculate_kinematic (core/pose_opt.py:372-445) / get_kinematic_chain_T (:482-512), axis-angle bones
    [U,24,3] or 6D-rotation bones [U,24,6] (use_rot6d, :391-392), rest_pose [24,3] or [U,24,3], pelvis [U,3] or None
    ->  kp [U,24,3], skts, l2ws [U,24,4,4], rots [U,24,3,3].
    The reference unrolls the SMPL tree by hand (unrolled_kinematic_chain :514-566); the parent loop is the same product."""
    U = bones.shape[0]
    rots = rot6d_to_rotmat(bones) if bones.shape[-1] == 6 else axis_angle_to_matrix(bones)
    rest = rest_pose.expand(U, 24, 3)
    bottom = torch.tensor([0., 0., 0., 1.], dtype=bones.dtype).expand(U, 1, 4)
    l2ws = []
    for j in range(24):
        p = SMPL_PARENTS[j]
        loc = rest[:, j] if j == 0 else rest[:, j] - rest[:, p]
        T = torch.cat([torch.cat([rots[:, j], loc[..., None]], -1), bottom], -2)
        l2ws.append(T if j == 0 else l2ws[p] @ T)
    l2ws = torch.stack(l2ws, 1)
    if pelvis is not None:
        shift = torch.zeros(U, 4, 4, dtype=bones.dtype)
        shift[:, :3, 3] = pelvis
        l2ws = l2ws + shift[:, None]
    skts = torch.inverse(l2ws)
    return l2ws[..., :3, 3], skts, l2ws, rots
