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
