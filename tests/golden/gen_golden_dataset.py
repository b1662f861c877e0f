#!/usr/bin/env python3
"""Dataset fixtures from the REFERENCE's own writer and dataset classes (build container only; the reference never travels).

h5py is not in the image: tests/h5shim.py (a numpy-backed stand-in for the slice of h5py this code uses) is installed as
`h5py`, cv2 / smplx / pytorch3d / imageio are stubbed as in the other generators (nothing on this path calls them).  Everything
that decides the fixture's content is the reference's code:

  * `core.process_spin.write_to_h5py` (process_spin.py:234-297) writes each tests/cases.py DATASET_CASES dict
    -> per-key manifest (dtype, shape, CRC-32 of the bytes): pins a-nerf_amd/dataset.py:write_npz_twin;
  * `BaseH5Dataset` / `SurrealDataset` / `MixamoDataset` / `H36MDataset` / `MonoPerfCapDataset` (dataset.py:20-420,
    load_surreal.py:302-380, load_mixamo.py:161-199, load_h36m.py:369-428, load_perfcap.py:54-89)
    open that file; `ray_collate_fn([ds[q] for q in batch])` -- what DataLoader(batch_sampler=RayImageSampler, collate_fn=
    ray_collate_fn) does per iteration (load_data.py:71-82) -- with numpy's global generator seeded per batch
    -> every key of the collated batch: pins H5PoseData.sample_batch (values AND dtypes);
  * `get_meta()` (dataset.py:433-484) -> the arrays create_raycaster / create_popt read: pins H5PoseData.data_attrs;
    `get_render_data()` (dataset.py:486-541) -> what run_nerf's periodic test renders read: pins H5PoseData.render_data;
  * `RayImageSampler` (dataset.py:774-811) under torch.manual_seed -> the image batches of the first iterations: pins
    dataset.image_batches.

Run:  python tests/golden/gen_golden_dataset.py      (writes tests/golden/dataset_*.npz and dataset_layout_manifest.json)
"""
import importlib
import json
import os
import sys
import tempfile
import zlib
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import gen_golden
import cases
import h5shim


def manifest_of(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: {"dtype": str(z[k].dtype), "shape": list(z[k].shape), "crc32": zlib.crc32(np.ascontiguousarray(z[k]).tobytes())}
                for k in sorted(z.files)}


def main():
    gen_golden.import_reference()
    h5shim.install()
    for m in ["smplx", "imageio", "tensorboard", "deepdish"]:
        sys.modules.setdefault(m, mock.MagicMock(name=m))
    import core.utils.skeleton_utils as su
    oracle = importlib.import_module("oracle.anerf_oracle")
    su.p3dr.axis_angle_to_matrix = oracle.axis_angle_to_matrix            # pytorch3d is absent (as gen_golden_fk.py)
    import core.dataset as ds_mod
    from core.process_spin import write_to_h5py
    from core.load_surreal import SurrealDataset
    from core.load_mixamo import MixamoDataset
    from core.load_h36m import H36MDataset
    from core.load_perfcap import MonoPerfCapDataset
    classes = {"BaseH5Dataset": ds_mod.BaseH5Dataset, "SurrealDataset": SurrealDataset, "MixamoDataset": MixamoDataset,
               "H36MDataset": H36MDataset, "MonoPerfCapDataset": MonoPerfCapDataset}

    manifests = {}
    tmp = tempfile.mkdtemp()
    for name, c in cases.DATASET_CASES.items():
        d = cases.dataset_dict(name)
        os.makedirs(os.path.join(tmp, name), exist_ok=True)
        path = os.path.join(tmp, name, "james_processed_h5py.h5" if c["cls"] == "MixamoDataset" else "synthetic_train_h5py.h5")
        write_to_h5py(path, dict(d))                                       # the reference's writer, through the shim
        manifests[name] = manifest_of(path)
        if "selected" in c:
            np.save(path.replace("processed_h5py.h5", "selected.npy"), np.array(c["selected"]))
        if "n_val" in c:      # the validation-set size is a per-subject table of the class (230 / 327 images): a class attribute, set
            classes[c["cls"]].n_vals = dict(classes[c["cls"]].n_vals, **{c["kw"]["subject"]: c["n_val"]})      # to the fixture's size
        dset = classes[c["cls"]](path, N_samples=cases.DATASET_N_SAMPLES, **c["kw"])
        out = {"len": np.array(len(dset))}
        for b, q_idxs in enumerate(c["batches"]):
            np.random.seed(c["seed"] + b)
            batch = ds_mod.ray_collate_fn([dset[int(q)] for q in q_idxs])
            if b == 0:
                out["batch_keys"] = np.array(sorted(batch))
            for k, v in batch.items():
                out[f"b{b}.{k}"] = v.numpy()
        meta = dset.get_meta()
        H, W, focals = meta["hwf"]
        out.update({"meta.H": np.asarray(H), "meta.W": np.asarray(W), "meta.focals": np.asarray(focals), "meta.c2ws": meta["c2ws"],
                    "meta.n_views": np.array(meta["n_views"]), "meta.joint_coords": np.asarray(meta["joint_coords"]),
                    "meta.rest_pose": meta["rest_pose"], "meta.kp3d": meta["kp3d"], "meta.skts": meta["skts"], "meta.bones": meta["bones"],
                    "meta.betas": meta["betas"], "meta.near_far": np.array([meta["near"], meta["far"]]),
                    "meta.keys": np.array(sorted(meta))})
        if meta["center"] is not None:
            out["meta.center"] = meta["center"]
        if meta["gt_kp3d"] is not None:
            out["meta.gt_kp3d"] = meta["gt_kp3d"]
        # dataset.py:486-541: what run_nerf's test renders read (BaseH5Dataset itself defines no render subset: subclasses only)
        rd = dset.get_render_data() if hasattr(dset, "render_skip") else {}
        if rd:
            out["render.keys"] = np.array(sorted(rd))
        for k, v in rd.items():
            if k == "hwf":
                out["render.H"], out["render.W"], out["render.focals"] = np.asarray(v[0]), np.asarray(v[1]), np.asarray(v[2])
            elif v is not None:
                out[f"render.{k}"] = np.asarray(v)
        np.savez_compressed(os.path.join(HERE, f"dataset_{name}.npz"), **out)
        print(name, "len", len(dset), {k: (v.dtype, v.shape) for k, v in out.items() if k.startswith("b0.")})
    json.dump(manifests, open(os.path.join(HERE, "dataset_layout_manifest.json"), "w"), indent=1, sort_keys=True)

    # RayImageSampler: the first 6 image batches for n = 21 images, N_images = 4, under torch.manual_seed(1234)
    class _Len:
        def __len__(self):
            return 21
    torch.manual_seed(1234)
    smp = ds_mod.RayImageSampler(_Len(), N_images=4, N_iter=8)
    np.savez(os.path.join(HERE, "dataset_sampler.npz"), batches=np.stack([b for b in smp]), n=np.array(21), N_images=np.array(4),
             seed=np.array(1234))
    print("sampler ok")


if __name__ == "__main__":
    main()
