"""CPU: the reference's dataset layout (core/process_spin.py:234-297) through a-nerf_amd/dataset.py.

A small synthetic dataset is written in that layout (`.npz` twin: same keys, shapes, dtypes as the `.h5`), read back, and
the collated batch (`BaseH5Dataset.__getitem__` + `ray_collate_fn`, core/dataset.py:60-103,813-820) is checked against
independent sources: rays against synth.camera_rays (itself pinned against the reference's get_rays, synth_pins.npz),
pose replication, pixel values, background lookup, dtypes; and the sampled training batch feeds the oracle."""
import importlib

import numpy as np
import pytest
import torch

dataset = importlib.import_module("a-nerf_amd.dataset")
synth = importlib.import_module("a-nerf_amd.synth")


def make_data(n=3, H=24, W=32, focal=40.0):
    rng = np.random.default_rng(0)
    poses = [synth.make_pose(50 + k) for k in range(n)]
    c2w = synth.default_c2w()
    c2ws = np.stack([c2w] * n)
    c2ws[1, :3, 3] += [0.1, -0.05, 0.2]
    masks = (rng.random((n, H, W, 1)) > 0.4).astype(np.uint8)
    return {"imgs": rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8), "masks": masks,
            "sampling_masks": np.maximum(masks, (rng.random((n, H, W, 1)) > 0.7).astype(np.uint8)),
            "bkgds": rng.integers(0, 256, (2, H, W, 3), dtype=np.uint8), "bkgd_idxs": np.array([0, 1, 0]),
            "kp3d": np.stack([q["kp"] for q in poses]).astype(np.float64), "bones": np.stack([q["bones"] for q in poses]),
            "skts": np.stack([q["skts"] for q in poses]), "cyls": np.stack([synth.bounding_cylinder(q["kp"]) for q in poses]),
            "rest_pose": (synth.SMPL_REST_POSE * synth.SURREAL_SCALE), "betas": np.zeros((1, 10)),
            "c2ws": c2ws, "focals": np.full(n, focal), "index": np.arange(n)}, (H, W, focal)


def test_layout_round_trip_and_batch(tmp_path, oracle):
    data, (H, W, focal) = make_data()
    path = str(tmp_path / "tiny_h5py_layout.npz")
    dataset.write_npz_twin(path, dict(data))
    raw = np.load(path)
    assert list(raw["img_shape"]) == [3, H, W, 3] and raw["img_shape"].dtype == np.int32
    assert raw["imgs"].shape == (3, H * W, 3) and raw["imgs"].dtype == np.uint8 and raw["masks"].shape == (3, H * W, 1)
    assert raw["kp3d"].dtype == np.float32 and raw["bkgd_idxs"].dtype == np.int64 and "index" not in raw
    ds = dataset.H5PoseData(path, device="cpu")
    assert len(ds) == 3 and ds.HW == (H, W) and ds.has_bg
    # the .npz twin's image arrays are resident (NpzFile would inflate the whole array on every per-row access)
    assert all(isinstance(ds._f[k], np.ndarray) for k in ("imgs", "masks", "sampling_masks"))
    rng = np.random.default_rng(5)
    b = ds.sample_batch([2, 0], 40, rng=rng)
    assert set(b) == {"rays_o", "rays_d", "target_s", "kp_idx", "kp3d", "bones", "skts", "cyls", "cam_idxs", "fgs", "bgs", "rays"}
    assert b["rays"].shape == (2, 80, 3) and b["skts"].shape == (80, 24, 4, 4) and b["kp_idx"].dtype == torch.int64
    assert b["kp_idx"][:40].eq(2).all() and b["kp_idx"][40:].eq(0).all() and b["cam_idxs"][:40].eq(2).all()
    # re-derive the pixel indices the sampler drew and check every column against independent sources
    rng2 = np.random.default_rng(5)
    for blk, idx in enumerate([2, 0]):
        sl = slice(40 * blk, 40 * blk + 40)
        valid, = np.where(data["sampling_masks"][idx].reshape(-1) > 0)
        px = np.sort(rng2.choice(valid, 40, replace=False))
        ro, rd = synth.camera_rays(H, W, focal, data["c2ws"][idx].astype(np.float32))
        np.testing.assert_allclose(b["rays_d"][sl].numpy(), rd.reshape(-1, 3)[px], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(b["rays_o"][sl].numpy(), ro.reshape(-1, 3)[px], rtol=0, atol=0)
        np.testing.assert_allclose(b["target_s"][sl].numpy(), data["imgs"][idx].reshape(-1, 3)[px] / 255.0, atol=1e-7)
        np.testing.assert_array_equal(b["fgs"][sl].numpy(), data["masks"][idx].reshape(-1, 1)[px].astype(np.float32))
        np.testing.assert_allclose(b["bgs"][sl].numpy(), data["bkgds"][data["bkgd_idxs"][idx]].reshape(-1, 3)[px] / 255.0, atol=1e-7)
        np.testing.assert_allclose(b["skts"][sl].numpy(), np.repeat(data["skts"][idx:idx + 1], 40, 0).astype(np.float32))
        np.testing.assert_allclose(b["cyls"][sl].numpy(), np.repeat(data["cyls"][idx:idx + 1], 40, 0).astype(np.float32))
    # the batch drives the path: oracle render on it is finite (rays of the sampling mask hit the bounding cylinder or
    # take the NaN-mean fallback)
    cfg = oracle.OracleConfig()
    P = oracle.params_from_numpy(synth.make_net_params(11))
    rb = oracle.make_ray_batch(b["rays"][0], b["rays"][1])
    with torch.no_grad():
        out = oracle.render_rays(cfg, P, None, rb, b["skts"], b["cyls"], 16)
    assert torch.isfinite(out["rgb_map"]).all()
    attrs = ds.data_attrs()
    assert attrs["n_views"] == 3 and attrs["rest_pose"].shape == (24, 3)


def test_errors_are_loud(tmp_path):
    p = str(tmp_path / "bad.npz")
    np.savez(p, imgs=np.zeros((1, 4, 3), np.uint8))
    with pytest.raises(KeyError, match="missing"):
        dataset.H5PoseData(p, device="cpu")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            dataset.H5PoseData(str(tmp_path / "x.h5"), device="cpu")
