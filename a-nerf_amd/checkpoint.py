"""On-disk checkpoint format of the reference: `Trainer.save_nerf` / `save_popt`
(core/trainer.py:485-517) and the reload side in `create_raycaster` / `create_popt` (core/raycasters.py:117-143,
core/pose_opt.py:52-75).

A `.tar` written here is a `torch.save` of exactly the reference's dict

    {'global_step', 'optimizer_state_dict', 'poseopt_layer_state_dict', 'pose_optimizer_state_dict', 'poseopt_anchors',
     'network_fn_state_dict', 'network_fine_state_dict', 'embed_state_dict', 'embedbones_state_dict', 'embeddirs_state_dict'}

with torch-format optimizer states, so the reference loads our checkpoints and we load its (`surreal.tar` etc.).  Host-side
plumbing only: tensors are moved with torch, nothing here computes.
"""
import torch

from .optim import FusedAdam

NERF_KEYS = ("global_step", "optimizer_state_dict", "poseopt_layer_state_dict", "pose_optimizer_state_dict", "poseopt_anchors")


def _unwrap(ray_caster):
    return getattr(ray_caster, "module", ray_caster)


def _optim_states(optimizer, pose_optimizer):
    """(optimizer_state_dict, pose_optimizer_state_dict) in torch.optim.Adam's format.  One FusedAdam that carries the pose
    parameters as its second group (the flat DP bucket, optim.py) is split into the two dicts the reference keeps."""
    if isinstance(optimizer, FusedAdam) and pose_optimizer is None and len(optimizer.param_groups) > 1:
        return optimizer.state_dict(group=0), optimizer.state_dict(group=1)
    osd = optimizer.state_dict(group=0) if isinstance(optimizer, FusedAdam) else optimizer.state_dict()
    return osd, (None if pose_optimizer is None else pose_optimizer.state_dict())


def nerf_state(global_step, ray_caster, optimizer, popt_layer=None, pose_optimizer=None, popt_anchors=None):
    """The dict `Trainer.save_nerf` hands to torch.save (trainer.py:498-505)."""
    osd, psd = _optim_states(optimizer, pose_optimizer)
    if popt_layer is None:
        psd, popt_anchors = None, None
    return {"global_step": global_step,
            "optimizer_state_dict": osd,
            "poseopt_layer_state_dict": None if popt_layer is None else popt_layer.state_dict(),
            "pose_optimizer_state_dict": psd,
            "poseopt_anchors": popt_anchors,
            # ours only (ignored by the reference's loaders): where the caster's counter-based generator stands, so that a resumed
            # run continues the random stream instead of replaying it
            "anerf_rng_state": _unwrap(ray_caster).rng_state() if hasattr(_unwrap(ray_caster), "rng_state") else None,
            **_unwrap(ray_caster).state_dict()}


def save_nerf(path, global_step, ray_caster, optimizer, popt_layer=None, pose_optimizer=None, popt_anchors=None):
    """Trainer.save_nerf (trainer.py:485-506).  ray_caster: the RayCaster or its train-side wrapper (`.module`)."""
    torch.save(nerf_state(global_step, ray_caster, optimizer, popt_layer, pose_optimizer, popt_anchors), path)
    print("Saved checkpoints at", path)


def save_popt(path, global_step, popt_layer, popt_anchors):
    """Trainer.save_popt (trainer.py:508-517)."""
    torch.save({"global_step": global_step, "poseopt_layer_state_dict": popt_layer.state_dict(),
                "poseopt_anchors": popt_anchors}, path)
    print("Saved pose at", path)


def load_nerf(path_or_ckpt, ray_caster, optimizer=None, popt_layer=None, pose_optimizer=None, finetune=False,
              map_location=None):
    """Restore what `create_raycaster` (raycasters.py:117-143) and `create_popt` (pose_opt.py:52-75) restore from a
    checkpoint: networks + embedder state (shape-mismatched tensors skipped like filter_state_dict), the optimizer state
    unless `finetune`, the pose layer, its optimizer and the regularisation anchors.
    Returns {'global_step': start, 'poseopt_anchors': anchors or None, 'ckpt': the loaded dict}."""
    ckpt = torch.load(path_or_ckpt, map_location=map_location, weights_only=False) if isinstance(path_or_ckpt, str) else path_or_ckpt
    missing = [k for k in ("global_step", "network_fn_state_dict") if k not in ckpt]
    if missing:
        raise KeyError(f"not an A-NeRF checkpoint: missing {missing}")
    caster = _unwrap(ray_caster)
    caster.load_state_dict(ckpt)
    if not finetune and ckpt.get("anerf_rng_state") is not None and hasattr(caster, "set_rng_state"):
        caster.set_rng_state(ckpt["anerf_rng_state"])
    start = 0 if finetune else ckpt["global_step"]
    if optimizer is not None and not finetune and ckpt.get("optimizer_state_dict") is not None:
        if isinstance(optimizer, FusedAdam) and len(optimizer.param_groups) > 1:
            if not optimizer.params[0].is_cuda:
                raise RuntimeError("load_nerf: move the model to the GPU before loading a multi-group FusedAdam state")
            optimizer.load_state_dict(ckpt["optimizer_state_dict"], group=0)
        else:
            optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    anchors = None
    if popt_layer is not None and ckpt.get("poseopt_layer_state_dict") is not None:
        popt_layer.load_state_dict(ckpt["poseopt_layer_state_dict"])
        psd = ckpt.get("pose_optimizer_state_dict")
        if psd is not None:
            if pose_optimizer is not None:
                pose_optimizer.load_state_dict(psd)
            elif isinstance(optimizer, FusedAdam) and len(optimizer.param_groups) > 1:
                optimizer.load_state_dict(psd, group=1)
        anchors = ckpt.get("poseopt_anchors")
        if getattr(popt_layer, "use_cache", False):
            popt_layer.update_cache()
    return {"global_step": start, "poseopt_anchors": anchors, "ckpt": ckpt}


def manifest(ckpt):
    """key -> shape / dtype / type summary of a checkpoint dict (what the tests pin against the reference's writer)."""
    def walk(v):
        if torch.is_tensor(v):
            return {"shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
        if isinstance(v, dict):
            return {str(k): walk(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [walk(x) for x in v]
        if v is None or isinstance(v, (bool, int, float, str)):
            return type(v).__name__
        return type(v).__name__
    return walk(ckpt)
