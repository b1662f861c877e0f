"""CPU: the reference's on-disk dataset layout and its reader against the REFERENCE'S OWN writer and dataset
classes.

tests/golden/gen_golden_dataset.py ran `core.process_spin.write_to_h5py`, `BaseH5Dataset` / `SurrealDataset` / `MixamoDataset`
+ `ray_collate_fn`, `get_meta()` and `RayImageSampler` in the build container (h5py -> tests/h5shim.py) over the numpy-seeded
dicts of tests/cases.py DATASET_CASES and stored what they produced.  Here a-nerf_amd/dataset.py reproduces it:
  * write_npz_twin == the reference writer's file, key for key (dtype, shape, bytes);
  * H5PoseData.sample_batch == the collated batch of the reference's DataLoader iteration, key for key and dtype for dtype,
    from the .npz twin AND through the `.h5` branch of `_open` (h5py -> the same shim);
  * data_attrs == get_meta(); image_batches == RayImageSampler.
"""
import importlib
import json
import os
import sys
import zlib

import numpy as np
import pytest
import torch

import cases

dataset = importlib.import_module("a-nerf_amd.dataset")
synth = importlib.import_module("a-nerf_amd.synth")
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIND = {"BaseH5Dataset": "base", "SurrealDataset": "surreal", "MixamoDataset": "mixamo", "H36MDataset": "h36m", "MonoPerfCapDataset": "perfcap"}


def reader(path, c, **kw):
    """H5PoseData with the arguments that correspond to the reference class's constructor arguments of a DATASET_CASES entry"""
    k = c["kw"]
    return dataset.H5PoseData(path, device="cpu", kind=KIND[c["cls"]], mask_img=k.get("mask_img", False), N_cams=k.get("N_cams"),
                              split=k.get("split", "full"), subject=k.get("subject"), n_val=c.get("n_val"), **kw)


def write_case(name, tmp_path, ext):
    c = cases.DATASET_CASES[name]
    stem = "james_processed_h5py" if c["cls"] == "MixamoDataset" else "synthetic_train_h5py"
    path = str(tmp_path / f"{stem}.{ext}")
    if ext == "npz":
        dataset.write_npz_twin(path, cases.dataset_dict(name))
    else:                                    # an ".h5" whose container is the shim's (.npz inside), written uncompressed
        dataset.write_npz_twin(path + ".npz", cases.dataset_dict(name), compressed=False)
        os.replace(path + ".npz", path)
    if "selected" in c:
        np.save(str(tmp_path / "james_selected.npy"), np.array(c["selected"]))
    return path, c


@pytest.mark.parametrize("name", sorted(cases.DATASET_CASES))
def test_writer_matches_the_reference_writer(name, tmp_path):
    want = json.load(open(os.path.join(GOLDEN, "dataset_layout_manifest.json")))[name]
    path, _ = write_case(name, tmp_path, "npz")
    with np.load(path, allow_pickle=False) as z:
        got = {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape), "crc32": zlib.crc32(np.ascontiguousarray(z[k]).tobytes())}
               for k in sorted(z.files)}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k
    assert "index" not in got and got["img_shape"]["dtype"] == "int32" and got["imgs"]["dtype"] == "uint8"


@pytest.fixture
def h5py_shim():
    """`import h5py` -> tests/h5shim.py for the duration of a test (h5py itself is not in the image)"""
    import h5shim
    had = sys.modules.get("h5py")
    h5shim.install()
    yield h5shim
    if had is None:
        sys.modules.pop("h5py", None)
    else:
        sys.modules["h5py"] = had


@pytest.mark.parametrize("ext", ["npz", "h5"])
@pytest.mark.parametrize("name", sorted(cases.DATASET_CASES))
def test_collated_batch_matches_the_reference_dataset(name, ext, tmp_path, h5py_shim):
    g = dict(np.load(os.path.join(GOLDEN, f"dataset_{name}.npz")))
    path, c = write_case(name, tmp_path, ext)
    ds = reader(path, c)
    if ext == "h5":
        assert isinstance(ds._f, h5py_shim.File)                       # the `.h5` branch of _open ran, rows are read per access
    assert len(ds) == int(g["len"])
    n = cases.DATASET_N_SAMPLES
    for b, q_idxs in enumerate(c["batches"]):
        np.random.seed(c["seed"] + b)                                  # the reference samples from numpy's global generator
        got = ds.sample_batch(q_idxs, n)
        assert sorted(got) == [str(k) for k in g["batch_keys"]]
        for k in got:
            want = g[f"b{b}.{k}"]
            have = got[k].numpy()
            assert have.shape == want.shape, (k, have.shape, want.shape)
            if k in ("rays_d", "rays") and want.dtype == np.float64:
                # the reference under NumPy >= 2: float64 directions (an np.int32 * 0.5 offset promotes them); float32 under
                # NumPy 1.x and here.  Same numbers to float32 rounding of the few operations involved
                assert have.dtype == np.float32
                np.testing.assert_allclose(have, want, rtol=3e-7, atol=3e-7)
            else:
                assert have.dtype == want.dtype, (k, have.dtype, want.dtype)
                np.testing.assert_array_equal(have, want, err_msg=k)
        assert got["kp_idx"].dtype == torch.int64 and got["rays"].shape == (2, n * len(q_idxs), 3)
    # seeded another way: a RandomState handed in draws the same pixels as the global generator with that seed
    np.random.seed(123)
    a = ds.sample_batch(c["batches"][0], n)
    b_ = ds.sample_batch(c["batches"][0], n, rng=np.random.RandomState(123))
    assert all(torch.equal(a[k], b_[k]) for k in a)


@pytest.mark.parametrize("name", sorted(cases.DATASET_CASES))
def test_data_attrs_match_get_meta(name, tmp_path):
    g = dict(np.load(os.path.join(GOLDEN, f"dataset_{name}.npz")))
    path, c = write_case(name, tmp_path, "npz")
    ds = reader(path, c)
    m = ds.data_attrs()
    assert sorted(m) == [str(k) for k in g["meta.keys"]]
    H, W, focals = m["hwf"]
    for key, have in (("H", H), ("W", W), ("focals", focals), ("c2ws", m["c2ws"]), ("rest_pose", m["rest_pose"]), ("kp3d", m["kp3d"]),
                      ("skts", m["skts"]), ("bones", m["bones"]), ("betas", m["betas"]), ("joint_coords", m["joint_coords"])):
        want = g[f"meta.{key}"]
        assert np.asarray(have).shape == want.shape and np.asarray(have).dtype == want.dtype, (key, np.asarray(have).dtype, want.dtype)
        np.testing.assert_array_equal(np.asarray(have), want, err_msg=key)
    assert int(m["n_views"]) == int(g["meta.n_views"]) and [m["near"], m["far"]] == list(g["meta.near_far"])
    if "meta.center" in g:
        np.testing.assert_array_equal(m["center"], g["meta.center"])
    else:
        assert m["center"] is None
    if "meta.gt_kp3d" in g:
        np.testing.assert_array_equal(m["gt_kp3d"], g["meta.gt_kp3d"])


def test_image_batches_follow_the_reference_sampler():
    g = np.load(os.path.join(GOLDEN, "dataset_sampler.npz"))
    torch.manual_seed(int(g["seed"]))
    got = np.stack(list(dataset.image_batches(int(g["n"]), int(g["N_images"]), len(g["batches"]))))
    np.testing.assert_array_equal(got, g["batches"])
    assert all((np.diff(b) >= 0).all() for b in got)
    assert len(set(got.reshape(-1)[:int(g["n"])].tolist())) == int(g["n"])     # every image once per permutation


def test_batch_feeds_the_path(tmp_path, oracle):
    """the sampled batch drives the hot path's oracle: finite output on rays of the sampling mask"""
    path, c = write_case("base", tmp_path, "npz")
    ds = dataset.H5PoseData(path, device="cpu")
    b = ds.sample_batch([2, 0], 40, rng=np.random.default_rng(5))
    assert b["rays"].shape == (2, 80, 3) and b["skts"].shape == (80, 24, 4, 4)
    assert b["kp_idx"][:40].eq(2).all() and b["kp_idx"][40:].eq(0).all() and b["cam_idxs"][:40].eq(2).all()
    assert all(isinstance(ds._f[k], np.ndarray) for k in ("imgs", "masks", "sampling_masks"))     # the twin's image arrays are resident
    P = oracle.params_from_numpy(synth.make_net_params(11))
    rb = oracle.make_ray_batch(b["rays"][0], b["rays"][1])
    with torch.no_grad():
        out = oracle.render_rays(oracle.OracleConfig(), P, None, rb, b["skts"], b["cyls"], 16)
    assert torch.isfinite(out["rgb_map"]).all()


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
    path, _ = write_case("base", tmp_path, "npz")
    with pytest.raises(NotImplementedError, match="shipped config"):
        dataset.H5PoseData(path, device="cpu", patch_size=4)
    with pytest.raises(ValueError, match="kind"):
        dataset.H5PoseData(path, device="cpu", kind="zju")


def test_data_attrs_build_the_caster(tmp_path):
    """run_nerf.py:520-560: `data_attrs = dataset.get_meta()` goes straight into create_raycaster -- ours does too"""
    import argparse
    raycaster = importlib.import_module("a-nerf_amd.raycaster")
    d = json.load(open(os.path.join(GOLDEN, "args_surreal.json")))
    d.pop("_config_file")
    d.update(basedir="/nonexistent")

    class Skel:
        joint_names = ["j%d" % i for i in range(24)]
        joint_trees = np.asarray(synth.SMPL_PARENTS)
    path, c = write_case("surreal_3cams", tmp_path, "npz")
    ds = dataset.H5PoseData(path, device="cpu", kind="surreal", N_cams=3)
    attrs = ds.data_attrs(skel_type=Skel)
    rk_train, rk_test, start, grad_vars, opt, _ = raycaster.create_raycaster(argparse.Namespace(**d), attrs, device="cpu")
    caster = rk_test["ray_caster"]
    assert attrs["n_views"] == 3 and tuple(caster.joint_coords.shape[-3:]) == (24, 3, 3)
    np.testing.assert_array_equal(caster.joint_coords.reshape(24, 3, 3).numpy(), attrs["joint_coords"])


def test_prefetching_generator_yields_the_sequential_batches(tmp_path):
    """H5PoseData.batches: one background producer walks the image batches in order -- same generator consumption, same batches as
    the plain loop (the role of the reference's DataLoader workers); a failure in the producer surfaces in the consumer"""
    path, c = write_case("surreal_full", tmp_path, "npz")
    ds = dataset.H5PoseData(path, device="cpu", kind="surreal")
    torch.manual_seed(11)
    qs = list(dataset.image_batches(len(ds), 4, 6))
    np.random.seed(42)
    want = [ds.sample_batch(q, 10) for q in qs]
    for prefetch in (3, 0):                             # background producer / inline
        np.random.seed(42)
        got = list(ds.batches(qs, 10, prefetch=prefetch))
        assert len(got) == len(want) == 6
        for a, b in zip(want, got):
            assert sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) for k in a)
    with pytest.raises(IndexError):
        list(ds.batches([[0, 1], [10 ** 6]], 10))
    it = ds.batches(qs, 10, prefetch=1)                 # abandoning the generator early must not leave the producer stuck
    next(it)
    it.close()


def test_staged_upload_packs_the_same_batch(tmp_path):
    """_upload_staged (the GPU reader's one-arena, one-copy upload) run on the host: same keys, shapes, dtypes and values as the
    plain path; a batch's tensors live in a ring slot -- intact while fewer than `staging_slots` later batches were sampled,
    reused by the batch that many later; a second batch layout gets its own ring"""
    path, c = write_case("mixamo", tmp_path, "npz")
    plain, staged = reader(path, c), reader(path, c)
    staged.staged_uploads = "always"
    qs, n = c["batches"], cases.DATASET_N_SAMPLES
    got = []
    for b in range(4):
        np.random.seed(5 + b)
        want = plain.sample_batch(qs[0], n)
        np.random.seed(5 + b)
        have = staged.sample_batch(qs[0], n)
        assert sorted(want) == sorted(have)
        for k in want:
            assert have[k].dtype == want[k].dtype and have[k].shape == want[k].shape and torch.equal(have[k], want[k]), k
        base = min(v.data_ptr() for k, v in have.items() if k != "rays")
        assert all((v.data_ptr() - base) % 256 == 0 for k, v in have.items() if k != "rays")      # 256-byte aligned arena offsets
        got.append((have, {k: v.clone() for k, v in want.items()}))
    assert len(staged._staging) == 1 and len(next(iter(staged._staging.values()))["slots"]) == 3
    for b in (1, 2, 3):                                                   # the three latest batches are intact ...
        assert all(torch.equal(got[b][0][k], got[b][1][k]) for k in got[b][1] if k != "rays")
    assert got[0][0]["rays_o"].data_ptr() == got[3][0]["rays_o"].data_ptr()        # ... the first one's slot went to the fourth
    np.random.seed(1)
    other = staged.sample_batch(qs[1], n)                                  # another batch size: another ring
    assert len(staged._staging) == 2 and other["target_s"].shape[0] == n * len(qs[1])
    np.random.seed(9)
    n_before = staged.staging_slots
    list(staged.batches([qs[0]] * 2, n, prefetch=2))                      # a prefetching generator widens the rings first
    assert staged.staging_slots == 5 > n_before


@pytest.mark.gpu
def test_device_reader_uploads_the_reference_batch(tmp_path):
    """the reader on the GPU: the staged upload (pinned arena, one asynchronous copy) delivers the reference's collated batch -- the
    fixture's values and dtypes -- as device tensors, `kp_idx` on the host; pageable uploads (`staged_uploads = False`) the same"""
    g = dict(np.load(os.path.join(GOLDEN, "dataset_mixamo.npz")))
    path, c = write_case("mixamo", tmp_path, "npz")
    n = cases.DATASET_N_SAMPLES
    for staged in (True, False):
        k = c["kw"]
        ds = dataset.H5PoseData(path, device="cuda", kind="mixamo", subject=k.get("subject"))
        ds.staged_uploads = staged
        kept = []
        for rep in range(2):                                  # twice: the second pass reuses arenas that are still in the ring
            for b, q_idxs in enumerate(c["batches"]):
                np.random.seed(c["seed"] + b)
                got = ds.sample_batch(q_idxs, n)
                kept.append((b, got))
                assert got["kp_idx"].device.type == "cpu" and got["kp_idx"].dtype == torch.int64
                for key, v in got.items():
                    want = g[f"b{b}.{key}"]
                    if key != "kp_idx":
                        assert v.device.type == "cuda", key
                    if key in ("rays_d", "rays") and want.dtype == np.float64:
                        np.testing.assert_allclose(v.cpu().numpy(), want, rtol=3e-7, atol=3e-7)
                    else:
                        assert v.cpu().numpy().dtype == want.dtype, key
                        np.testing.assert_array_equal(v.cpu().numpy(), want, err_msg=key)
        assert (len(ds._staging) == 2) == staged              # two batch sizes, two rings
        torch.cuda.synchronize()
        for b, got in kept[-2:]:                              # the latest batches are still what they were
            np.testing.assert_array_equal(got["target_s"].cpu().numpy(), g[f"b{b}.target_s"])


@pytest.mark.parametrize("name", [n for n in sorted(cases.DATASET_CASES) if cases.DATASET_CASES[n]["cls"] != "BaseH5Dataset"])
def test_render_data_matches_get_render_data(name, tmp_path):
    """H5PoseData.render_data == the reference's get_render_data() (dataset.py:486-541): every key, dtype and value of the
    render subset run_nerf.py hands to render_path for its periodic test renders (class defaults for render_skip / N_render)"""
    g = dict(np.load(os.path.join(GOLDEN, f"dataset_{name}.npz")))
    path, c = write_case(name, tmp_path, "npz")
    rd = reader(path, c).render_data()
    assert sorted(rd) == [str(k) for k in g["render.keys"]]
    for k, v in rd.items():
        items = dict(zip(("H", "W", "focals"), v)) if k == "hwf" else {k: v}
        for kk, vv in items.items():
            if vv is None:
                assert f"render.{kk}" not in g
                continue
            want = g[f"render.{kk}"]
            assert np.asarray(vv).shape == want.shape and np.asarray(vv).dtype == want.dtype, (kk, np.asarray(vv).dtype, want.dtype)
            np.testing.assert_array_equal(np.asarray(vv), want, err_msg=kk)
    with pytest.raises(ValueError, match="render subset"):
        dataset.H5PoseData(path, device="cpu").render_data()
