"""On-disk dataset layout of the reference and the ray-batch assembly that feeds the hot path.

The reference pre-processes every dataset into ONE `.h5` file (`core/process_spin.py:234-297 write_to_h5py`) and trains
from it through `BaseH5Dataset` + `ray_collate_fn` (`core/dataset.py:20-420, 813-820`).  Layout (N images of H x W):

    img_shape [4] int32 = (N, H, W, 3)        imgs / bkgds [*, H*W, 3] uint8 (flattened pixels)
    masks / sampling_masks [N, H*W, 1] uint8   bkgd_idxs [N] int64
    kp3d [N,24,3]  bones [N,24,3]  skts [N,24,4,4]  cyls [N,5]  rest_pose [24,3]  betas [*,10]   float32
    c2ws [N,4,4]  focals [N] or [N,2]  (centers [N,2], gt_kp3d, kp_idxs / cam_idxs for multi-view sets: optional)

`H5PoseData` reads that layout -- from a real `.h5` when `h5py` is importable (it is not in the build image; the import is
optional and loud), or from an `.npz` twin holding the same keys with the same shapes and dtypes (`write_npz_twin`) -- and
`sample_batch()` assembles exactly the batch dict `ray_collate_fn` hands to `Trainer.train_batch` (`rays [2,N,3], target_s,
kp_idx, kp3d, bones, skts, cyls, cam_idxs, fgs, bgs`, all per-ray replicated), as device tensors (`kp_idx` on the host, where the pose layer reads it).  `kind` selects the
index arithmetic of the reference's dataset classes: "base" (BaseH5Dataset), "surreal" (SurrealDataset: images and cameras
arranged (N_cams, N_kps), poses shared by the cameras; load_surreal.py:302-380), "mixamo" (MixamoDataset: the sorted subset
named by `*selected.npy`, white background; load_mixamo.py:161-199).

Pinned against the reference itself: tests/golden/dataset_*.npz hold the collated batches, `get_meta()` and sampler output of
the reference's own classes run over tests/h5shim.py (tests/golden/gen_golden_dataset.py); tests/test_dataset_layout.py
reproduces them key for key.  The pixel sampler consumes numpy's generator exactly as the reference does (one
`choice(valid, N, replace=False)` and one `random()` per image, dataset.py:298-318), so a seeded run draws the same pixels.
One stated difference: ray directions are float32 here; the reference's are float32 under NumPy 1.x and float64 under
NumPy >= 2 (its `np.int32 * 0.5` image-centre offset becomes a float64 scalar, dataset.py:150-163) -- `render()` casts to
float32 either way (trainer.py:124-125).
Host-side I/O only: no arithmetic of the hot path lives here.
"""
import numpy as np
import torch

REQUIRED = ("img_shape", "imgs", "masks", "sampling_masks", "kp3d", "bones", "skts", "cyls", "rest_pose", "c2ws", "focals")
OPTIONAL = ("bkgds", "bkgd_idxs", "betas", "centers", "gt_kp3d", "img_paths", "ext_scale", "pose_scale")
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
        if v.ndim == 0:                                      # "non-iterable": a scalar dataset of the value's own type
            out[k] = v
        elif k in IMAGE_KEYS:
            out[k] = v.reshape(v.shape[0], h * w, v.shape[-1])
        elif k == "img_paths":                               # byte strings (process_spin.py:277-280)
            out[k] = v.astype("S")
        elif np.issubdtype(v.dtype, np.floating):
            out[k] = v.astype(np.float32)
        elif np.issubdtype(v.dtype, np.integer):
            out[k] = v.astype(np.int64)
        else:
            raise NotImplementedError(f"unknown datatype for key {k}: {v.dtype}")
    (np.savez_compressed if compressed else np.savez)(path, **out)


def per_joint_coords(rest_pose, parents):
    """skeleton_utils.py:493-539 (get_per_joint_coords / create_local_coord): per joint, the frame whose z axis points from the
    joint to its parent in the rest pose -- rotate about y, then about x, until z meets that direction"""
    def rot_y(t):
        return np.array([[np.cos(t), 0, -np.sin(t)], [0, 1, 0], [np.sin(t), 0, np.cos(t)]], dtype=np.float32)

    def rot_x(t):
        return np.array([[1, 0, 0], [0, np.cos(t), -np.sin(t)], [0, np.sin(t), np.cos(t)]], dtype=np.float32)
    acos = lambda a: np.arccos(np.clip(a, -1. + 1e-8, 1. - 1e-8))
    out = []
    for i, j in enumerate(parents):
        vec = rest_pose[j] - rest_pose[i]
        vec = vec / (np.linalg.norm(vec) + 1e-5)
        if np.isclose(np.linalg.norm(vec), 0.):
            out.append(np.eye(3, dtype=np.float32))
            continue
        xz = vec[[0, 2]] / np.linalg.norm(vec[[0, 2]])
        ry = rot_y(acos(xz[-1]) * np.sign(xz[0]))
        v1 = ry @ vec
        yz = v1[1:3] / np.linalg.norm(v1[1:3])
        rx = rot_x(acos(yz[-1]) * np.sign(yz[0]))
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = rx @ ry
        out.append(np.eye(3, dtype=np.float32) @ np.linalg.inv(m)[:3, :3].T)
    return np.array(out)


def image_batches(n, N_images, n_iter, generator=None):
    """RayImageSampler over RandIntGenerator (dataset.py:748-811): `n_iter` sorted batches of `N_images` image indices drawn
    from successive torch.randperm(n) permutations (every image once per epoch), each permutation seeded as the reference
    seeds it -- from torch's global generator (`torch.empty((), int64).random_()`) unless `generator` is given."""
    def perm():
        g = generator
        if g is None:
            g = torch.Generator()
            g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        return iter(torch.randperm(n, generator=g).tolist())
    it, batch = perm(), []
    for _ in range(n_iter):
        while len(batch) < N_images:
            try:
                batch.append(next(it))
            except StopIteration:
                it = perm()
                batch.append(next(it))
        yield np.sort(batch)
        batch = []


class H5PoseData:
    """BaseH5Dataset's in-memory meta (dataset.py:125-183) + per-image pixel reads + batch assembly."""

    PERFCAP_N_VAL = {"weipeng": 230, "nadia": 327}      # MonoPerfCapDataset.n_vals (load_perfcap.py:55)

    def __init__(self, path, device="cuda", kind="base", mask_img=False, N_cams=None, N_rand_kps=None, idx_map=None, N_nms=0.0,
                 patch_size=1, split="full", subject=None, n_val=None):
        """kind: which of the reference's dataset classes' index arithmetic applies -- "base" (BaseH5Dataset), "surreal", "mixamo",
        "h36m" (H36MDataset: split "train" / "val" by the sequence name inside `img_paths`, subjects ending in "c" keep the "-1"
        takes; load_h36m.py:380-421), "perfcap" (MonoPerfCapDataset: the last n_val images are the validation set, camera
        translations divided by 1.05; load_perfcap.py:65-89).  split: "full" (what load_data.py:117 passes unless --use_val),
        "train", "val" (h36m / perfcap)."""
        if kind not in ("base", "surreal", "mixamo", "h36m", "perfcap"):
            raise ValueError(f"H5PoseData: kind {kind!r} (base | surreal | mixamo | h36m | perfcap)")
        if split not in ("full", "train", "val"):
            raise NotImplementedError(f"Split {split} is undefined!")
        if N_nms != 0 or patch_size != 1 or N_rand_kps is not None:
            raise NotImplementedError("N_nms > 0 (P_nms), patch_size > 1 and rand_train_kps are used by no shipped config")
        self.path, self.device, self.kind, self.mask_img = path, torch.device(device), kind, bool(mask_img)
        self.staged_uploads, self.staging_slots, self._staging = True, 3, {}          # see _upload_staged (device "cuda" only)
        f = _open(path)
        keys = set(f.keys())
        missing = [k for k in REQUIRED if k not in keys]
        if missing:
            raise KeyError(f"{path}: not the A-NeRF dataset layout, missing {missing}")
        self.keys = sorted(keys)
        shp = np.asarray(f["img_shape"][:])
        self.n_images, self.HW = int(shp[0]), (int(shp[1]), int(shp[2]))
        rd = lambda k: np.asarray(f[k][:])
        # as stored (float32 on disk); the per-item casts of get_pose_data / get_camera_data are no-ops on them
        self.kp3d, self.bones, self.skts, self.cyls = (rd(k) for k in ("kp3d", "bones", "skts", "cyls"))
        self.gt_kp3d = rd("gt_kp3d") if "gt_kp3d" in keys else None
        self.rest_pose = rd("rest_pose")
        self.betas = rd("betas") if "betas" in keys else None
        self.c2ws, self.focals = rd("c2ws"), rd("focals")
        self.centers = rd("centers") if "centers" in keys else None
        self.has_bg = "bkgds" in keys
        if self.has_bg:
            self.bgs = rd("bkgds").reshape(-1, self.HW[0] * self.HW[1], 3)
            self.bg_idxs = rd("bkgd_idxs").astype(np.int64)
        # ---- which images a queried index means (`_idx_map`) and how it maps to a pose / a camera
        self._idx_map = None if idx_map is None else np.asarray(idx_map)
        self._N_kps = self._N_cams = None
        if kind == "surreal":                       # load_surreal.py:326-364
            n_kps_total, n_cams_total = len(self.kp3d), len(self.c2ws) // len(self.kp3d)
            self._N_kps, self._N_cams = n_kps_total, (n_cams_total if N_cams is None else int(N_cams))
            if self._N_cams != n_cams_total:        # the reference's hard-coded camera subset
                self._idx_map = np.concatenate([np.arange(n_kps_total) + n_kps_total * c for c in (0, 3, 6)])
        elif kind == "mixamo":                      # load_mixamo.py:187-199
            sel = str(path).replace("processed_h5py.h5", "selected.npy").replace("processed_h5py.npz", "selected.npy")
            if idx_map is None:
                if sel == str(path):
                    raise ValueError("H5PoseData(kind='mixamo'): the file name must contain 'processed_h5py' (its subset is read from "
                                     "'<prefix>selected.npy' next to it), or pass idx_map")
                self._idx_map = np.array(sorted(np.load(sel)))
            else:
                self._idx_map = np.array(sorted(self._idx_map))
            self.bgs = np.full((1, self.HW[0] * self.HW[1], 3), 255, dtype=np.uint8)      # "set white bkgd manually"
            self.bg_idxs = np.zeros(self.n_images, dtype=np.int64)
            self.has_bg = True
        elif kind == "h36m":                        # load_h36m.py:380-421
            if "img_paths" not in keys:
                raise KeyError(f"{path}: the H36M layout holds `img_paths` (the splits are read off the sequence names)")
            seqs = [bytes(p).decode().split("/")[1] for p in rd("img_paths")]
            if idx_map is None and subject is not None and str(subject).endswith("c"):
                self._idx_map = np.array([i for i, q in enumerate(seqs) if q.endswith("-1")])
            elif idx_map is None and split != "full":
                is_val = np.array([any(q.startswith(v) for v in ("Greeting-", "Walking-", "Posing-")) for q in seqs])
                self._idx_map = np.nonzero(is_val if split == "val" else ~is_val)[0]
        elif kind == "perfcap":                     # load_perfcap.py:65-89
            if idx_map is None and split != "full":
                nv = n_val if n_val is not None else self.PERFCAP_N_VAL.get(subject)
                if nv is None:
                    raise ValueError("H5PoseData(kind='perfcap', split != 'full'): subject 'weipeng' / 'nadia', or n_val")
                ids = np.arange(self.n_images)
                self._idx_map = ids[-nv:] if split == "val" else ids[:-nv]
            self.c2ws = self.c2ws.copy()
            self.c2ws[..., :3, -1] /= 1.05          # "the estimation for MonoPerfCap is somehow off by a small scale"
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
        oy, ox = (np.float32(self.HW[0] * 0.5), np.float32(self.HW[1] * 0.5)) if self.centers is None else (np.float32(0), np.float32(0))
        self._dirs = np.stack([i - ox, -(j - oy), -np.ones_like(i)], -1)

    def __len__(self):
        return self.n_images if self._idx_map is None else len(self._idx_map)

    # ---- index arithmetic of the dataset classes (dataset.py:396-412, load_surreal.py:366-382) ------------------------
    def get_kp_idx(self, idx, q_idx):
        """(index of the pose data in the file, index the pose is known by -- PoseOptLayer row, `kp_idx` of the batch)"""
        if self.kind == "surreal":
            return idx % len(self.kp3d), q_idx % self._N_kps
        return idx, q_idx

    def get_cam_idx(self, idx, q_idx):
        """(index of the camera data in the file, index of the per-view code -- `cam_idxs` of the batch)"""
        if self.kind == "surreal":
            return idx, q_idx // self._N_kps
        return idx, q_idx

    def _subset_idxs(self):
        """dataset.py:414-431 `_get_subset_idxs`: file indices of the poses / cameras / images the dataset serves"""
        if self._idx_map is not None:
            i_idxs = _k = _c = self._idx_map
            _kq = _cq = np.arange(len(self._idx_map))
        else:
            i_idxs = np.arange(self.n_images)
            _k = _kq = np.arange(len(self.kp3d))
            _c = _cq = np.arange(len(self.c2ws))
        k_idxs, kq_idxs = self.get_kp_idx(_k, _kq)
        c_idxs, cq_idxs = self.get_cam_idx(_c, _cq)
        return k_idxs, c_idxs, i_idxs, kq_idxs, cq_idxs

    def data_attrs(self, skel_type=None):
        """`get_meta()` (dataset.py:433-484, load_surreal.py:384-387): what create_raycaster / create_popt / the trainer read.
        skel_type: an object with `joint_trees` (parent index per joint); default: the 24-joint SMPL tree."""
        from . import synth
        parents = np.asarray(synth.SMPL_PARENTS if skel_type is None else skel_type.joint_trees)
        k_idxs, c_idxs, _, _, _ = self._subset_idxs()
        H, W = np.int32(self.HW[0]), np.int32(self.HW[1])                  # img_shape is stored as int32
        hwf = (np.repeat([H], len(c_idxs), 0), np.repeat([W], len(c_idxs), 0), self.focals[c_idxs])
        betas = self.betas
        if betas is not None:
            if len(betas) > 1:
                betas = betas[k_idxs]
            betas = betas.mean(0, keepdims=True).repeat(len(betas), 0)
        return {"hwf": hwf, "center": None if self.centers is None else self.centers[c_idxs].copy(), "c2ws": self.c2ws[c_idxs],
                "near": 60., "far": 100., "n_views": self._N_cams if self.kind == "surreal" else len(self),
                "skel_type": skel_type, "joint_coords": per_joint_coords(self.rest_pose, parents), "rest_pose": self.rest_pose,
                "gt_kp3d": None if self.gt_kp3d is None else self.gt_kp3d[k_idxs], "kp3d": self.kp3d[k_idxs], "skts": self.skts[k_idxs],
                "bones": self.bones[k_idxs], "betas": betas, "kp_map": None, "kp_uidxs": None}

    RENDER_SUBSET = {"surreal": (1, 15), "mixamo": (40, 15), "h36m": (80, 15), "perfcap": (10, 15)}   # (render_skip, N_render) of the classes

    def render_data(self, render_skip=None, N_render=None):
        """`get_render_data()` (dataset.py:486-541): every render_skip-th image of the served subset, at most N_render of them --
        images / masks / backgrounds as float [*,H,W,C], their cameras and poses: what run_nerf.py hands to render_path for its
        periodic test renders.  Defaults: the class attributes of the reference's dataset class for this `kind`."""
        skip, n = self.RENDER_SUBSET.get(self.kind, (None, None))     # (BaseH5Dataset defines none: pass both)
        skip, n = (skip if render_skip is None else render_skip), (n if N_render is None else N_render)
        if skip is None or n is None:
            raise ValueError("render_data: kind 'base' has no default render subset; pass render_skip and N_render")
        k_idxs, c_idxs, i_idxs, kq_idxs, cq_idxs = self._subset_idxs()
        sub = lambda a: np.asarray(a)[::skip][:n]
        k_idxs, c_idxs, i_idxs, kq_idxs, cq_idxs = sub(k_idxs), sub(c_idxs), sub(i_idxs), sub(kq_idxs), sub(cq_idxs)
        # PoseRefinedDataset (mixamo / h36m / perfcap) reports the QUERIED indices as kp_idxs / cam_idxs (dataset.py:570-584) --
        # the rows of the pose layer and of the frame-code table -- the plain classes the file indices
        refined = self.kind in ("mixamo", "h36m", "perfcap")
        H, W = self.HW
        img = lambda key, ch: np.stack([np.asarray(self._f[key][int(i)]) for i in i_idxs]).reshape(-1, H, W, ch)
        Hs, Ws = np.repeat([np.int32(H)], len(c_idxs), 0), np.repeat([np.int32(W)], len(c_idxs), 0)
        return {"imgs": img("imgs", 3).astype(np.float32) / 255., "fgs": img("masks", 1),
                "bgs": self.bgs.reshape(-1, H, W, 3).astype(np.float32) / 255., "bg_idxs": self.bg_idxs[i_idxs], "bg_idxs_len": len(self.bgs),
                "cam_idxs": cq_idxs.copy() if refined else c_idxs, "cam_idxs_len": len(self.c2ws), "c2ws": self.c2ws[c_idxs],
                "hwf": (Hs, Ws, self.focals[c_idxs]), "center": None if self.centers is None else self.centers[c_idxs].copy(),
                "kp_idxs": kq_idxs.copy() if refined else k_idxs, "kp_idxs_len": len(self.kp3d), "kp3d": self.kp3d[k_idxs], "skts": self.skts[k_idxs], "bones": self.bones[k_idxs]}

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

    def sample_pixels(self, idx, n, rng=None):
        """dataset.py:286-327 for patch_size 1 and N_nms = 0.0: n distinct pixels of image `idx`'s sampling mask, ascending.
        rng: numpy's global generator (None, as the reference), a RandomState, or a Generator.  Consumes it as the reference
        does: the choice, then the one `random()` its N_nms dice throws even at probability 0 (dataset.py:312-318)."""
        rng = np.random if rng is None else rng
        mask = np.asarray(self._f["sampling_masks"][idx]).reshape(-1)      # one row (h5py: one read; .npz twin: resident array)
        valid, = np.where(mask > 0)
        px = rng.choice(valid, n, replace=False)
        rng.random()
        return np.sort(px)

    def get_img_data(self, idx, pixel_idxs, mask_img=None):
        """dataset.py:262-284: (rgb in [0,1], foreground mask, background colour or None) at the sampled pixels"""
        mask_img = self.mask_img if mask_img is None else mask_img
        fg = np.asarray(self._f["masks"][idx, pixel_idxs]).astype(np.float32)
        img = np.asarray(self._f["imgs"][idx, pixel_idxs]).astype(np.float32) / 255.
        bg = None
        if self.has_bg:
            bg = self.bgs[self.bg_idxs[idx], pixel_idxs].astype(np.float32) / 255.
            if mask_img:
                img = img * fg + (1. - fg) * bg
        return img, fg, bg

    # ---- the collated batch ---------------------------------------------------------------------------------------
    def sample_batch(self, q_idxs, n_per_image, rng=None, mask_img=None):
        """`ray_collate_fn` over `__getitem__` for the QUERIED indices `q_idxs` (in [0, len(self)); dataset.py:57-103, 813-820):
        len(q_idxs) * n_per_image rays, every per-pose / per-camera quantity replicated per ray, on `self.device`."""
        cols = {k: [] for k in ("rays_o", "rays_d", "target_s", "kp_idx", "kp3d", "bones", "skts", "cyls", "cam_idxs", "fgs", "bgs")}
        for q in np.asarray(q_idxs).reshape(-1):
            q = int(q)
            idx = q if self._idx_map is None else int(self._idx_map[q])
            cam_real, cam_idx = self.get_cam_idx(idx, q)
            kp_real, kp_idx = self.get_kp_idx(idx, q)
            px = self.sample_pixels(idx, n_per_image, rng)
            center = None if self.centers is None else self.centers[cam_real]
            ro, rd = self.get_rays(self.c2ws[cam_real].astype(np.float32), self.focals[cam_real], px, center)
            rgb, fg, bg = self.get_img_data(idx, px, mask_img)
            rep = lambda a: np.repeat(a[kp_real:kp_real + 1].astype(np.float32), n_per_image, 0)
            cols["rays_o"].append(ro), cols["rays_d"].append(rd), cols["target_s"].append(rgb), cols["fgs"].append(fg)
            if bg is None:
                raise KeyError(f"{self.path}: no `bkgds` (the reference's collate fails on a None background too)")
            cols["bgs"].append(bg)
            cols["kp_idx"].append(np.full(n_per_image, kp_idx, np.int64))
            cols["cam_idxs"].append(np.full(n_per_image, cam_idx, np.int64))
            cols["kp3d"].append(rep(self.kp3d)), cols["bones"].append(rep(self.bones)), cols["skts"].append(rep(self.skts))
            cols["cyls"].append(rep(self.cyls))
        # kp_idx stays on the host: its one consumer, the pose layer, groups the rays by pose there (trainer.py:299 `kp_idx.cpu().numpy()`
        # in the reference) -- a device copy would be read back every iteration, a blocking transfer behind the previous step
        kp_idx_host = torch.as_tensor(np.concatenate(cols.pop("kp_idx"), 0)) if self.device.type == "cuda" else None
        if self.staged_uploads and (self.device.type == "cuda" or self.staged_uploads == "always"):
            batch = self._upload_staged(cols)
        else:
            # plain pageable uploads: 0.21 ms for the 11 tensors of a 1024-ray batch on the MI355X box; `pin_memory()` + non-blocking
            # copies measured 2.2 ms there (pinning a fresh buffer per tensor costs more than the copy; tools/leases/r05_probe_upload.py)
            batch = {k: torch.as_tensor(np.concatenate(v, 0)).to(self.device) for k, v in cols.items()}
        if kp_idx_host is not None:
            batch["kp_idx"] = kp_idx_host
        batch["rays"] = torch.stack([batch["rays_o"], batch["rays_d"]], 0)
        return batch

    def _upload_staged(self, cols):
        """The batch through ONE pinned host arena and ONE asynchronous copy.  A pageable upload blocks the host until the stream
        reaches it, i.e. until the previous training step has finished -- sampling and the GPU step then alternate instead of
        overlapping.  Here the columns are concatenated straight into a pinned arena (allocated once per batch layout), one
        non-blocking copy moves it into a device arena of the same layout, the batch's tensors are typed views of that.  Arenas
        form a ring of `self.staging_slots` (default 3): a batch's tensors stay valid until that many later batches have been
        sampled -- consume (or clone) them before; an event per slot keeps a slot's pinned memory from being rewritten under a
        copy still in flight."""
        parts = {k: (v, sum(a.shape[0] for a in v), v[0].shape[1:], np.result_type(*[a.dtype for a in v])) for k, v in cols.items()}
        cuda = self.device.type == "cuda"            # (staged_uploads = "always" runs the same packing on the host: the CPU test's handle)
        key = tuple((k, n, tail, str(dt)) for k, (_, n, tail, dt) in parts.items())
        ring = self._staging.get(key)
        if ring is None:
            off, layout = 0, {}
            for k, (_, n, tail, dt) in parts.items():
                nbytes = int(n * int(np.prod(tail, dtype=np.int64)) * dt.itemsize)
                layout[k] = (off, nbytes, (n,) + tuple(tail), dt)
                off += (nbytes + 255) // 256 * 256
            slots = []
            for _ in range(max(2, int(self.staging_slots))):
                host = torch.empty(max(off, 1), dtype=torch.uint8, pin_memory=cuda)
                dev = torch.empty(max(off, 1), dtype=torch.uint8, device=self.device)
                hv = {k: host.numpy()[o:o + nb].view(dt).reshape(shape) for k, (o, nb, shape, dt) in layout.items()}
                dv = {k: dev[o:o + nb].view(getattr(torch, np.dtype(dt).name)).reshape(shape) for k, (o, nb, shape, dt) in layout.items()}
                slots.append({"host": host, "dev": dev, "hv": hv, "dv": dv, "event": None})
            ring = self._staging[key] = {"slots": slots, "next": 0}
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % len(ring["slots"])
        if slot["event"] is not None:
            slot["event"].synchronize()
        for k, (v, _, _, _) in parts.items():
            np.concatenate(v, 0, out=slot["hv"][k])
        slot["dev"].copy_(slot["host"], non_blocking=True)
        if cuda:
            if slot["event"] is None:
                slot["event"] = torch.cuda.Event()
            slot["event"].record(torch.cuda.current_stream(self.device))
        return dict(slot["dv"])

    def batches(self, q_batches, n_per_image, rng=None, prefetch=2, mask_img=None):
        """`sample_batch` for every entry of `q_batches` (e.g. `image_batches(...)`), assembled `prefetch` batches ahead by ONE
        background thread -- the role of the reference's `DataLoader(..., num_workers=...)` (load_data.py:71-82).  One producer
        walks the batches in order, so the generator is consumed exactly as by the sequential loop: a seeded run yields the same
        batches.  (While it runs, nothing else in the process should draw from the same generator -- numpy's global one by default.)
        On the GPU the batches' tensors are views of a small ring of upload arenas (`_upload_staged`): use each batch before asking
        for the one `staging_slots` later, or clone what must live longer (`staged_uploads = False`: independent tensors)."""
        if self.staging_slots < int(prefetch or 0) + 3:      # queued + in production + in the consumer's hands must not share an arena
            self.staging_slots, self._staging = int(prefetch) + 3, {}
        if not prefetch:                        # inline: the right choice while the GPU step, not the sampling, bounds the loop -- a
            for qb in q_batches:                # Python producer thread costs the training thread the GIL (E2E run, 1024 rays: 162 -> 148 it/s)
                yield self.sample_batch(qb, n_per_image, rng, mask_img)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=max(1, int(prefetch)))
        stop = threading.Event()

        def produce():
            try:
                if self.device.type == "cuda" and self.device.index is not None:
                    torch.cuda.set_device(self.device)          # (a bare "cuda" = the process's current device, which threads share)
                for qb in q_batches:
                    if stop.is_set():
                        return
                    q.put(("ok", self.sample_batch(qb, n_per_image, rng, mask_img)))
                q.put(("end", None))
            except BaseException as e:          # hand the failure to the consumer instead of dying silently
                q.put(("err", e))
        th = threading.Thread(target=produce, daemon=True)
        th.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    return
                if kind == "err":
                    raise item
                yield item
        finally:
            stop.set()
            while not q.empty():                # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
