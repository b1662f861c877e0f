"""Forward ray-march pipeline = RayCaster.render_rays (core/raycasters.py:361-474) as a sequence of
HIP kernel launches on one stream, no host synchronisation in between:

  ray_bounds -> coarse_z -> [fused encode+MLP] -> composite
            (-> importance -> [fused encode+MLP on the merged depths, fine net] -> composite)

Without `extras` the whole sequence is ONE C call (anerf_forward) on one workspace tensor; with `extras` the staged entry
points are called one by one so that the intermediates can be returned (tests).

The fine pass re-encodes the merged coarse+fine depths inside the fused kernel instead of
gather-merging 1080-wide encodings (raycasters.py:679-709): same values, no [N,S,1080] tensor.
"""
import torch

from . import ops


def render_rays_forward(cfg, net_c, net_f, ray_batch, skts, cyls, n_samples, n_importance=0,
                        tau_v=20.0, tau_d=20.0, cut_v=None, cut_d=None, cam_idx=None,
                        codes_c=None, codes_f=None, t_rand=None, u_imp=None, noise=None, noise_fine=None,
                        lindisp=False, single_net=False, extras=False, precision="fp32", pts_noise=None, pts_noise_is=None):
    """net_c / net_f: (packed, aux) images from ops.pack_params (which=0 for precision "fp32", which=3 for
    "bf16x3").  Returns the reference's output dict
    (RayCaster._collect_outputs, raycasters.py:711-724); extras adds the intermediates."""
    if not extras:   # production path: one C call, one workspace (anerf_forward); bit-identical to the staged calls below
        return ops.forward(cfg, net_c, net_f, ray_batch, skts, cyls, n_samples, n_importance, tau_v, tau_d, cut_v, cut_d, cam_idx,
                           codes_c, codes_f, t_rand, u_imp, noise, noise_fine, lindisp, single_net, precision, pts_noise, pts_noise_is)
    if pts_noise is not None:
        raise NotImplementedError("sample-point offsets (ray_noise_std > 0) go through the one-call entry points only")
    dev = ray_batch.device
    if cut_v is None:
        cut_v = torch.full((cfg.n_joints,), 0.5, device=dev)
    if cut_d is None:
        cut_d = torch.full((cfg.n_joints,), 0.5, device=dev)
    nf_raw, stats = ops.ray_bounds(ray_batch, cyls)
    z, nf = ops.coarse_z(nf_raw, stats, ray_batch, n_samples, t_rand, lindisp)
    raw = ops.mlp_raw(cfg, net_c[0], net_c[1], ray_batch, z, skts, tau_v, tau_d, cut_v, cut_d, cam_idx, codes_c, precision)
    co = ops.composite(cfg, raw, z, ray_batch, noise)
    ret = {"rgb_map": co["rgb_map"], "disp_map": co["disp_map"], "acc_map": co["acc_map"], "alpha": co["alpha"]}
    ex = {"near_far": nf, "z_vals": z, "raw": raw, "weights": co["weights"]}
    if n_importance > 0:
        zs, zm, idx = ops.importance(z, co["weights"], n_importance, u_imp, single_net, want_idx=True)
        if single_net:
            raw_is = ops.mlp_raw(cfg, net_f[0], net_f[1], ray_batch, zs, skts, tau_v, tau_d, cut_v, cut_d, cam_idx, codes_f, precision)
            raw_f = torch.gather(torch.cat([raw, raw_is], 1), 1, idx[..., None].expand(-1, -1, 4)).contiguous()
        else:
            raw_f = ops.mlp_raw(cfg, net_f[0], net_f[1], ray_batch, zm, skts, tau_v, tau_d, cut_v, cut_d, cam_idx, codes_f, precision)
        fo = ops.composite(cfg, raw_f, zm, ray_batch, noise_fine)
        ret = {"rgb_map": fo["rgb_map"], "disp_map": fo["disp_map"], "acc_map": fo["acc_map"], "alpha": fo["alpha"],
               "rgb0": ret["rgb_map"], "disp0": ret["disp_map"], "acc0": ret["acc_map"], "alpha0": ret["alpha"]}
        ex.update({"z_samples": zs, "z_fine": zm, "sorted_idx": idx, "raw_fine": raw_f, "weights_fine": fo["weights"]})
    if extras:
        ret["_extras"] = ex
    return ret


def make_ray_batch(rays_o, rays_d, near=0.0, far=1.0):
    """render()'s ray batch [N,11] = (o, d, near, far, viewdirs)   (core/trainer.py:116-135)."""
    vd = rays_d / rays_d.norm(dim=-1, keepdim=True)
    ones = torch.ones_like(rays_d[:, :1])
    return torch.cat([rays_o, rays_d, near * ones, far * ones, vd], -1).contiguous()
