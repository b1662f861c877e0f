"""On-disk dataset layout of the reference and the ray-batch assembly that feeds the hot path.

The reference pre-processes every dataset into ONE `.h5` file (`core/process_spin.py:234-297 write_to_h5py`) and trains
from it through `BaseH5Dataset` + `ray_collate_fn` (`core/dataset.py:20-420, 813-820`).  Layout (N images of H x W):

    img_shape [4] int32 = (N, H, W, 3)        imgs / bkgds [*, H*W, 3] uint8 (flattened pixels)
    masks / sampling_masks [N, H*W, 1] uint8   bkgd_idxs [N] int64
    kp3d [N,24,3]  bones [N,24,3]  skts [N,24,4,4]  cyls [N,5]  rest_pose [24,3]  betas [*,10]   float32
    c2ws [N,4,4]  focals [N] or [N,2]  (centers [N,2], gt_kp3d, kp_idxs / cam_idxs for multi-view sets: optional)

`H5PoseData` reads that layout -- from a real `.h5` when `h5py` is importable (it is not in the build image; the import is
optional and loud), or from an `.npz` twin holding the same keys with the same shapes and dtypes (`write_npz_twin`; what
the tests use) -- and `sample_batch()` assembles exactly the batch dict `ray_collate_fn` hands to `Trainer.train_batch`
(`rays [2,N,3], target_s, kp_idx, kp3d, bones, skts, cyls, cam_idxs, fgs, bgs`, all per-ray replicated), as device tensors.
Host-side I/O only: no arithmetic of the hot path lives here.
"""
import numpy as np
import torch

REQUIRED = ("img_shape", "imgs", "masks", "sampling_masks", "kp3d", "bones", "skts", "cyls", "rest_pose", "c2ws", "focals")
OPTIONAL = ("bkgds", "bkgd_idxs", "betas", "centers", "gt_kp3d", "kp_idxs", "cam_idxs", "ext_scale", "pose_scale")
IMAGE_KEYS = ("imgs", "bkgds", "masks", "sampling_masks")


def _open(path):
    """dict-like read access to the layout: h5py.File for .h5 / .hdf5, numpy's lazy NpzFile for the .npz twin"""
    if str(path).endswith(".npz"):
        return np.load(path, allow_pickle=False)
    try:
        import h5py
    except ImportError as e:
        raise ImportError(f"reading {path} needs h5py, which is not installed here; convert the file with "
                          "dataset.write_npz_twin() on a machine that has it, or install h5py") from e
    return h5py.File(path, "r")


def write_npz_twin(path, data, compressed=True):
    """Write `data` (the dict `write_to_h5py` receives: images as [N,H,W,C]) in the reference's layout as an .npz:
    images flattened to [N, H*W, C], floats as float32, integers as int64 (process_spin.py:246-293)."""
    imgs = np.asarray(data["imgs"])
    n, h, w = imgs.shape[:3]
    out = {"img_shape": np.array(imgs.shape, np.int32)}
    for k, v in data.items():
        if k in ("index", "img_path", "img_shape"):          # `redundants` (process_spin.py:244)
            continue
        v = np.asarray(v)
        if v.ndim == 0:
            out[k] = v
        elif k in IMAGE_KEYS:
            out[k] = v.reshape(v.shape[0], h * w, v.shape[-1])
        elif np.issubdtype(v.dtype, np.floating):
            out[k] = v.astype(np.float32)
        elif np.issubdtype(v.dtype, np.integer):
            out[k] = v.astype(np.int64)
        else:
            raise NotImplementedError(f"unknown datatype for key {k}: {v.dtype}")
    (np.savez_compressed if compressed else np.savez)(path, **out)


class H5PoseData:
    """BaseH5Dataset's in-memory meta (dataset.py:125-183) + per-image pixel reads + batch assembly."""

    def __init__(self, path, device="cuda"):
        self.path, self.device = path, torch.device(device)
        f = _open(path)
        keys = set(f.keys())
        missing = [k for k in REQUIRED if k not in keys]
        if missing:
            raise KeyError(f"{path}: not the A-NeRF dataset layout, missing {missing}")
        self.keys = sorted(keys)
        shp = np.asarray(f["img_shape"][:])
        self.n_images, self.HW = int(shp[0]), (int(shp[1]), int(shp[2]))
        rd = lambda k: np.asarray(f[k][:])
        self.kp3d, self.bones, self.skts, self.cyls = (rd(k).astype(np.float32) for k in ("kp3d", "bones", "skts", "cyls"))
        self.rest_pose = rd("rest_pose").astype(np.float32)
        self.betas = rd("betas").astype(np.float32) if "betas" in keys else None
        self.c2ws, self.focals = rd("c2ws").astype(np.float32), rd("focals").astype(np.float32)
        self.centers = rd("centers").astype(np.float32) if "centers" in keys else None
        self.has_bg = "bkgds" in keys
        if self.has_bg:
            self.bgs = rd("bkgds").reshape(-1, self.HW[0] * self.HW[1], 3)
            self.bg_idxs = rd("bkgd_idxs").astype(np.int64)
        self.kp_idxs = rd("kp_idxs").astype(np.int64) if "kp_idxs" in keys else np.arange(self.n_images)
        self.cam_idxs = rd("cam_idxs").astype(np.int64) if "cam_idxs" in keys else np.arange(self.n_images)
        # per-image pixel reads: h5py slices one row from disk per access; numpy's NpzFile is lazy per KEY, not per row -- every
        # `f[key][idx]` would inflate the whole array again (0.2 s per row on a 39 MB `imgs`, three times per sampled image).
        # The .npz twin's image arrays (uint8) are therefore read ONCE here and indexed in memory as the reference indexes its
        # open h5 file (dataset.py:262-327).
        if str(path).endswith(".npz"):
            self._f = {k: np.asarray(f[k]) for k in ("imgs", "masks", "sampling_masks") if k in keys}
            f.close()
        else:
            self._f = f
        # pre-computed pixel directions (dataset.py:147-163); the first two columns still need the division by focal
        i, j = np.meshgrid(np.arange(self.HW[1], dtype=np.float32), np.arange(self.HW[0], dtype=np.float32), indexing="xy")
        i, j = i.reshape(-1), j.reshape(-1)
        oy, ox = (self.HW[0] * 0.5, self.HW[1] * 0.5) if self.centers is None else (0.0, 0.0)
        self._dirs = np.stack([i - ox, -(j - oy), -np.ones_like(i)], -1)

    def __len__(self):
        return self.n_images

    def data_attrs(self, near=0.0, far=1.0):
        """what create_raycaster / create_popt read (run_nerf.py:520-560): skeleton-independent part"""
        return {"near": near, "far": far, "n_views": int(self.cam_idxs.max()) + 1, "hwf": (self.HW[0], self.HW[1], self.focals),
                "rest_pose": self.rest_pose, "betas": self.betas, "kp3d": self.kp3d, "bones": self.bones,
                "skts": self.skts, "cyls": self.cyls, "c2ws": self.c2ws, "centers": self.centers}

    # ---- per-image pieces, named as in BaseH5Dataset -------------------------------------------------------------
    def get_rays(self, c2w, focal, pixel_idxs, center=None):
        """dataset.py:343-363"""
        dirs = self._dirs[pixel_idxs].copy()
        if center is not None:
            c = np.array(center, np.float32).copy()
            c[1] *= -1
            dirs[..., :2] -= c
        dirs[:, :2] /= focal
        rays_d = dirs if np.isclose(np.eye(3), c2w[:3, :3]).all() else np.sum(dirs[..., None, :] * c2w[:3, :3], -1)
        return np.broadcast_to(c2w[:3, -1], rays_d.shape).copy(), rays_d.copy()

    def sample_pixels(self, idx, n, rng):
        """dataset.py:286-327 for patch_size 1, N_nms 0: n distinct pixels of the sampling mask, ascending"""
        mask = np.asarray(self._f["sampling_masks"][idx]).reshape(-1)      # one row (h5py: one read; .npz twin: resident array)
        valid, = np.where(mask > 0)
        return np.sort(rng.choice(valid, n, replace=False))

    def get_img_data(self, idx, pixel_idxs, mask_img=False):
        """dataset.py:262-284: (rgb in [0,1], foreground mask, background colour or None) at the sampled pixels"""
        fg = np.asarray(self._f["masks"][idx])[pixel_idxs].astype(np.float32)
        img = np.asarray(self._f["imgs"][idx])[pixel_idxs].astype(np.float32) / 255.0
        bg = None
        if self.has_bg:
            bg = self.bgs[self.bg_idxs[idx], pixel_idxs].astype(np.float32) / 255.0
            if mask_img:
                img = img * fg + (1.0 - fg) * bg
        return img, fg, bg

    # ---- the collated batch ---------------------------------------------------------------------------------------
    def sample_batch(self, img_idxs, n_per_image, rng=None, mask_img=False):
        """`ray_collate_fn` over `BaseH5Dataset.__getitem__` for the images `img_idxs` (dataset.py:60-103, 813-820):
        len(img_idxs) * n_per_image rays, every per-pose / per-camera quantity replicated per ray, on `self.device`."""
        rng = np.random.default_rng() if rng is None else rng
        cols = {k: [] for k in ("rays_o", "rays_d", "target_s", "kp_idx", "kp3d", "bones", "skts", "cyls", "cam_idxs", "fgs", "bgs")}
        for idx in np.asarray(img_idxs).reshape(-1):
            idx = int(idx)
            px = self.sample_pixels(idx, n_per_image, rng)
            center = None if self.centers is None else self.centers[idx]
            ro, rd = self.get_rays(self.c2ws[idx], self.focals[idx], px, center)
            rgb, fg, bg = self.get_img_data(idx, px, mask_img)
            kidx = int(self.kp_idxs[idx])
            rep = lambda a: np.repeat(a[kidx:kidx + 1], n_per_image, 0)
            cols["rays_o"].append(ro), cols["rays_d"].append(rd), cols["target_s"].append(rgb), cols["fgs"].append(fg)
            cols["bgs"].append(bg if bg is not None else np.zeros_like(rgb))
            cols["kp_idx"].append(np.full(n_per_image, kidx, np.int64))
            cols["cam_idxs"].append(np.full(n_per_image, int(self.cam_idxs[idx]), np.int64))
            cols["kp3d"].append(rep(self.kp3d)), cols["bones"].append(rep(self.bones)), cols["skts"].append(rep(self.skts))
            cols["cyls"].append(rep(self.cyls))
        batch = {k: torch.as_tensor(np.concatenate(v, 0)).to(self.device) for k, v in cols.items()}
        batch["rays"] = torch.stack([batch["rays_o"], batch["rays_d"]], 0)
        return batch
