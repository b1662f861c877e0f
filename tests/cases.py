"""Seeded test cases shared by the oracle tests (CPU) and the HIP parity tests (GPU).

Each case regenerates the exact inputs tests/golden/gen_golden.py fed to the reference.
pytest-mode random numbers restate the reference's numpy-seeded overrides
(ray_utils.py:171-180, 240-244; nerf.py:178-182).
"""
import importlib
import numpy as np

synth = importlib.import_module("a-nerf_amd.synth")


def pytest_rand(shape):
    np.random.seed(0)
    return np.random.rand(*shape).astype(np.float32)


CASES = {
    # name: (n_rays, pose_seeds, ray_seed, per_ray_pose, S, Ni, seeds(coarse, fine), cfg kwargs)
    "eval_s32": dict(n=96, poses=[0], ray_seed=1, per_ray=False, S=32, Ni=0, seeds=(11, 12), cfg={}),
    "eval_hier": dict(n=64, poses=[1], ray_seed=2, per_ray=False, S=64, Ni=16, seeds=(11, 12), cfg={}),
    # BASELINE config 5's sample counts (64 coarse + 128 importance): 192-sample merged pass, two networks
    "eval_hier128": dict(n=96, poses=[12], ray_seed=9, per_ray=False, S=64, Ni=128, seeds=(11, 12), cfg={}),
    "nan_fallback": dict(n=64, poses=[2], ray_seed=3, per_ray=False, S=16, Ni=0, seeds=(11, 12), cfg={}, cyl_scale=0.45),
    "train_pytest": dict(n=48, poses=[4, 5, 6], ray_seed=4, per_ray=True, S=64, Ni=16, seeds=(11, 12), cfg={}, train=True),
    "mixamo_train": dict(n=40, poses=[7, 8], ray_seed=5, per_ray=True, S=64, Ni=16, seeds=(21, 22),
                         cfg=dict(framecode_ch=16), n_codes=8, train=True, loss="L1"),
    # ray_noise_std > 0 (raycasters.py:660,674): train mode, pytest randomness, per-ray poses, 24 + 8 samples
    "ray_noise": dict(n=40, poses=[13, 14, 15], ray_seed=10, per_ray=True, S=24, Ni=8, seeds=(11, 12), cfg={}, train=True, ray_noise=True),
    "single_net": dict(n=32, poses=[9], ray_seed=6, per_ray=False, S=96, Ni=48, seeds=(31, 31),
                       cfg=dict(multires_views=0), single_net=True),
}


RAY_NOISE_STD = 0.15


def ray_noise_arrays(n, S, Ni, std=RAY_NOISE_STD):
    """the numpy-seeded stand-in for torch.randn_like that tests/golden/gen_golden_raynoise.py fed the reference"""
    a = np.random.RandomState(1000 + S).randn(n, S, 3).astype(np.float32) * np.float32(std)
    b = np.random.RandomState(1000 + Ni).randn(n, Ni, 3).astype(np.float32) * np.float32(std)
    return a, b


def build(name):
    c = CASES[name]
    ro, rd, kp, skts, bones, cyls, which = synth.scene_batch(c["n"], c["poses"], ray_seed=c["ray_seed"],
                                                             per_ray_pose=c["per_ray"])
    if "cyl_scale" in c:
        cyls = cyls.copy()
        cyls[:, 2] *= c["cyl_scale"]
    mv = c["cfg"].get("multires_views", 4)
    fc = c["cfg"].get("framecode_ch", 0)
    nc = c.get("n_codes", 0)
    Pc = synth.make_net_params(c["seeds"][0], 7, mv, fc, nc)
    Pf = synth.make_net_params(c["seeds"][1], 7, mv, fc, nc)
    out = dict(c)
    out.update(rays_o=ro, rays_d=rd, kp=kp, skts=skts, bones=bones, cyls=cyls, Pc=Pc, Pf=Pf)
    if c.get("train"):
        n, S, Ni = c["n"], c["S"], c["Ni"]
        out["t_rand"] = pytest_rand((n, S))
        out["u_imp"] = pytest_rand((n, Ni))
        out["noise"] = pytest_rand((n, S))            # raw_noise_std = 1.0, NOT scaled by B (nerf.py:181)
        out["noise_fine"] = pytest_rand((n, S + Ni))
    if fc:
        out["cams"] = (np.arange(c["n"]) % 8).astype(np.float32)
    if c.get("ray_noise"):
        out["pts_noise"], out["pts_noise_is"] = ray_noise_arrays(c["n"], c["S"], c["Ni"])
    return out


# ---- on-disk dataset layout: the dicts the reference's write_to_h5py receives, numpy-seeded ------------------
DATASET_CASES = {
    # name: images, poses, (H, W), reference dataset class, its kwargs, image batches (QUERIED indices, sorted as RayImageSampler
    # yields them), seed of numpy's global generator before each batch
    "base": dict(n=3, n_poses=3, HW=(24, 32), focal=40.0, cls="BaseH5Dataset", kw={}, batches=[[0, 2], [1, 1, 2]], seed=5),
    "base_mask_img": dict(n=3, n_poses=3, HW=(24, 32), focal=40.0, cls="BaseH5Dataset", kw=dict(mask_img=True), batches=[[0, 1]], seed=6),
    "centers": dict(n=3, n_poses=3, HW=(20, 28), focal=35.0, cls="BaseH5Dataset", kw={}, batches=[[0, 1, 2]], seed=7, centers=True,
                    focal_xy=True),
    # SURREAL: imgs / c2ws arranged (N_cams, N_kps); kp3d holds N_kps poses (load_surreal.py:302-380)
    "surreal_full": dict(n=21, n_poses=3, HW=(16, 20), focal=30.0, cls="SurrealDataset", kw={}, batches=[[0, 4, 11, 20]], seed=8),
    "surreal_3cams": dict(n=21, n_poses=3, HW=(16, 20), focal=30.0, cls="SurrealDataset", kw=dict(N_cams=3), batches=[[1, 3, 8], [0, 5]], seed=9),
    # Mixamo: a sorted subset of the file (selected.npy), white background whatever the file holds (load_mixamo.py:161-199)
    "mixamo": dict(n=10, n_poses=10, HW=(16, 20), focal=30.0, cls="MixamoDataset", kw=dict(subject="james"), batches=[[0, 3], [1, 2, 2]], seed=10,
                   selected=[7, 1, 4, 8], img_paths=True),
    # H36M (load_h36m.py:369-428): train / val split by the sequence name inside img_paths, the "c" subjects keep the "-1" takes
    "h36m_full": dict(n=10, n_poses=10, HW=(16, 20), focal=30.0, cls="H36MDataset", kw=dict(subject="S9", split="full"), batches=[[0, 9], [3, 4]],
                      seed=11, img_paths="h36m"),
    "h36m_train": dict(n=10, n_poses=10, HW=(16, 20), focal=30.0, cls="H36MDataset", kw=dict(subject="S9", split="train"), batches=[[0, 2, 4]],
                       seed=12, img_paths="h36m"),
    "h36m_val": dict(n=10, n_poses=10, HW=(16, 20), focal=30.0, cls="H36MDataset", kw=dict(subject="S9", split="val"), batches=[[1, 3]], seed=13,
                     img_paths="h36m"),
    "h36m_c": dict(n=10, n_poses=10, HW=(16, 20), focal=30.0, cls="H36MDataset", kw=dict(subject="S9c", split="full"), batches=[[0, 1, 2]], seed=14,
                   img_paths="h36m"),
    # MonoPerfCap (load_perfcap.py:54-89): the last n_val images are the validation set; camera translations divided by 1.05
    "perfcap_full": dict(n=8, n_poses=8, HW=(16, 20), focal=30.0, cls="MonoPerfCapDataset", kw=dict(subject="weipeng", split="full"),
                         batches=[[0, 7], [2, 5]], seed=15, n_val=3),
    "perfcap_train": dict(n=8, n_poses=8, HW=(16, 20), focal=30.0, cls="MonoPerfCapDataset", kw=dict(subject="weipeng", split="train"),
                          batches=[[0, 4]], seed=16, n_val=3),
    "perfcap_val": dict(n=8, n_poses=8, HW=(16, 20), focal=30.0, cls="MonoPerfCapDataset", kw=dict(subject="weipeng", split="val"),
                        batches=[[0, 2]], seed=17, n_val=3),
}
H36M_SEQS = ["Directions-1", "Greeting-1", "Walking-2", "Eating-2", "Posing-1", "Sitting-1", "Purchases-2", "Walking-1", "Photo-1", "Smoking-2"]
DATASET_N_SAMPLES = 24


def dataset_dict(name):
    """the dict handed to write_to_h5py (images as [N,H,W,C]) for a DATASET_CASES entry; same arrays on every box"""
    c = DATASET_CASES[name]
    n, (H, W) = c["n"], c["HW"]
    base_names = ["base", "base_mask_img", "centers", "mixamo", "surreal_3cams", "surreal_full"]     # (the first fixtures' seeds stay what they were)
    key = name.replace("_mask_img", "")
    rng = np.random.default_rng(100 + (base_names.index(key) if key in base_names else 50 + sorted(DATASET_CASES).index(key)))
    poses = [synth.make_pose(50 + k) for k in range(c["n_poses"])]
    c2w = synth.default_c2w()
    c2ws = np.stack([c2w] * n).astype(np.float64)
    c2ws[:, :3, 3] += rng.normal(0, 0.05, (n, 3))
    for k in range(n // 2):                      # half of the cameras rotated (get_rays takes its dot-product branch), half identity-like
        a = 0.1 * (k + 1)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        c2ws[2 * k, :3, :3] = R @ c2ws[2 * k, :3, :3]
    masks = (rng.random((n, H, W, 1)) > 0.4).astype(np.uint8)
    d = {"imgs": rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8), "masks": masks,
         "sampling_masks": np.maximum(masks, (rng.random((n, H, W, 1)) > 0.7).astype(np.uint8)),
         "bkgds": rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8), "bkgd_idxs": (np.arange(n) % 2).astype(np.int32),
         "kp3d": np.stack([q["kp"] for q in poses]).astype(np.float64), "gt_kp3d": np.stack([q["kp"] for q in poses]).astype(np.float64),
         "bones": np.stack([q["bones"] for q in poses]), "skts": np.stack([q["skts"] for q in poses]),
         "cyls": np.stack([synth.bounding_cylinder(q["kp"]) for q in poses]),
         "rest_pose": (synth.SMPL_REST_POSE * synth.SURREAL_SCALE), "betas": rng.normal(0, 1, (1 if name.startswith("surreal") else c["n_poses"], 10)),
         "c2ws": c2ws, "focals": (np.stack([np.full(n, c["focal"]), np.full(n, c["focal"] * 1.1)], -1) if c.get("focal_xy")
                                  else np.full(n, c["focal"]) + np.arange(n) * 0.5),
         "ext_scale": 0.001, "index": np.arange(n)}
    if c.get("centers"):
        d["centers"] = np.stack([np.full(n, W * 0.5) + rng.normal(0, 2, n), np.full(n, H * 0.5) + rng.normal(0, 2, n)], -1)
    if c.get("img_paths") == "h36m":
        d["img_paths"] = np.array([f"S9/{H36M_SEQS[k]}/frame{k:04d}.jpg" for k in range(n)])
    elif c.get("img_paths"):
        d["img_paths"] = np.array([f"seq{k // 4}/Image{k % 4 + (2 if k == 6 else 0):04d}.png" for k in range(n)])
    return d
