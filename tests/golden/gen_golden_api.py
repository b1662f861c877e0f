#!/usr/bin/env python3
"""The reference's call signatures for the hot path's interface, as data: parameter names, kinds and (repr of) defaults of the
functions / methods a caller of the path touches, read with `inspect.signature` from the imported reference (build container only).
tests/test_api_signatures.py holds our mirror to them: every parameter the reference accepts must be accepted here under the same name,
in the same position, with the same default.

Run:  python tests/golden/gen_golden_api.py      (writes tests/golden/api_signatures.json)
"""
import inspect
import json
import os
import sys
from unittest import mock

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G          # noqa: E402

TARGETS = [
    ("core.raycasters", "create_raycaster"), ("core.raycasters", "RayCaster.__init__"), ("core.raycasters", "RayCaster.forward"),
    ("core.raycasters", "RayCaster.render_rays"), ("core.raycasters", "RayCaster.render_pts_density"),
    ("core.raycasters", "RayCaster.render_mesh_density"), ("core.raycasters", "RayCaster.update_embed_fns"),
    ("core.raycasters", "RayCaster.state_dict"), ("core.raycasters", "RayCaster.load_state_dict"),
    ("core.networks.nerf", "NeRF.__init__"), ("core.networks.nerf", "NeRF.forward"), ("core.networks.nerf", "NeRF.forward_batchify"),
    ("core.networks.nerf", "NeRF.raw2outputs"),
    ("core.cutoff_embedder", "get_embedder"), ("core.cutoff_embedder", "CutoffEmbedder.__init__"),
    ("core.cutoff_embedder", "CutoffEmbedder.update_threshold"), ("core.cutoff_embedder", "CutoffEmbedder.update_tau"),
    ("core.cutoff_embedder", "CutoffEmbedder.update_alpha"),
    ("core.trainer", "render"), ("core.trainer", "batchify_rays"), ("core.trainer", "decay_optimizer_lrate"),
    ("core.trainer", "Trainer.__init__"), ("core.trainer", "Trainer.train_batch"),
    ("core.pose_opt", "PoseOptLayer.__init__"), ("core.pose_opt", "PoseOptLayer.forward"), ("core.pose_opt", "create_popt"),
]


def describe(fn):
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        d = p.default
        if d is inspect.Parameter.empty:
            d = None
        elif type(d) in (int, float, bool, str, type(None), list, dict, tuple):
            d = repr(d)
        elif callable(d):
            d = "callable:" + getattr(d, "__name__", type(d).__name__)        # (torch.sigmoid, F.relu: no addresses in the file)
        else:
            d = "object:" + type(d).__name__                                   # (the SMPL skeleton constant)
        out.append({"name": name, "kind": p.kind.name, "default": d})
    return out


def main():
    G.import_reference()
    for m in ["smplx", "h5py", "imageio", "core.process_spin", "core.load_data", "tensorboard"]:
        sys.modules.setdefault(m, mock.MagicMock(name=m))
    import importlib
    res = {}
    for mod, path in TARGETS:
        obj = importlib.import_module(mod)
        for part in path.split("."):
            obj = getattr(obj, part)
        res[f"{mod}:{path}"] = describe(obj)
    # run_nerf.py is a script (its imports start the whole application): render_path's signature is read from its syntax tree
    import ast
    tree = ast.parse(open(os.path.join(G.REF, "run_nerf.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "render_path"][0]
    pos = fn.args.args
    dfl = [None] * (len(pos) - len(fn.args.defaults)) + [repr(ast.literal_eval(d)) for d in fn.args.defaults]
    res["run_nerf:render_path"] = [{"name": a.arg, "kind": "POSITIONAL_OR_KEYWORD", "default": d} for a, d in zip(pos, dfl)]
    json.dump(res, open(os.path.join(G.OUT, "api_signatures.json"), "w"), indent=1)
    print(len(res), "signatures;", sum(len(v) for v in res.values()), "parameters")


if __name__ == "__main__":
    main()
