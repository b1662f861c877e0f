"""Caller-side glue with the reference's signatures: `render()` / `batchify_rays()` (core/trainer.py:64-145)
and the image losses / PSNR the trainer applies to the returned dict (core/trainer.py:8-61,353-380).

These are thin: ray-batch assembly and dict concatenation only; the chunks go to RayCaster (HIP kernels).
"""
import torch


def batchify_rays(rays_flat, chunk=1024 * 32, ray_caster=None, **kwargs):
    """core/trainer.py:64-79."""
    all_ret = {}
    for i in range(0, rays_flat.shape[0], chunk):
        batch_kwargs = {k: kwargs[k][i:i + chunk] if torch.is_tensor(kwargs[k]) else kwargs[k] for k in kwargs}
        ret = ray_caster(rays_flat[i:i + chunk], **batch_kwargs)
        for k in ret:
            all_ret.setdefault(k, []).append(ret[k])
    return {k: torch.cat(all_ret[k], 0) for k in all_ret}


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, near=0., far=1., center=None, use_viewdirs=False,
           c2w_staticcam=None, **kwargs):
    """core/trainer.py:82-145 (the `rays` form: the reference's c2w form drops into pdb)."""
    if rays is None:
        raise NotImplementedError("render() needs `rays`; build them with synth.camera_rays / the dataset")
    rays_o, rays_d = rays
    sh = rays_d.shape
    rays_o = torch.reshape(rays_o, [-1, 3]).float()
    rays_d = torch.reshape(rays_d, [-1, 3]).float()
    cols = [rays_o, rays_d, near * torch.ones_like(rays_d[..., :1]), far * torch.ones_like(rays_d[..., :1])]
    if use_viewdirs:
        cols.append(rays_d / torch.norm(rays_d, dim=-1, keepdim=True))
    ray_batch = torch.cat(cols, -1)
    all_ret = batchify_rays(ray_batch, chunk, **kwargs)
    for k in all_ret:
        if all_ret[k].dim() >= 4:
            continue
        all_ret[k] = torch.reshape(all_ret[k], list(sh[:-1]) + list(all_ret[k].shape[1:]))
    return all_ret


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def img2l1(x, y):
    return torch.mean((x - y).abs())


def mse2psnr(x):
    return -10. * torch.log10(x)


def nerf_loss(preds, target, bgs=1.0, loss_fn="MSE", coarse_weight=1.0, use_background=True):
    """_compute_nerf_loss for the fine and coarse heads (core/trainer.py:353-380)."""
    fn = img2mse if loss_fn == "MSE" else img2l1
    def comp(rgb, acc):
        return rgb + (1. - acc)[..., None] * bgs if use_background else rgb
    pred = comp(preds["rgb_map"], preds["acc_map"])
    total = fn(pred, target)
    if "rgb0" in preds:
        total = total + coarse_weight * fn(comp(preds["rgb0"], preds["acc0"]), target)
    return total, pred
