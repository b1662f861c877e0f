"""Drop-in at the signature level: tests/golden/api_signatures.json holds the reference's own `inspect.signature` of every function /
method a caller of the hot path touches (tests/golden/gen_golden_api.py, build container).  Our mirror must accept every one of those
parameters under the same name, in the same position and kind, with the same default -- so the reference's scripts call it unchanged
(INTEGRATION.md section 1).  Extra trailing keyword parameters of ours are allowed and listed."""
import importlib
import inspect
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_signatures.json")
# reference module -> ours
MODULES = {"core.raycasters": "a-nerf_amd.raycaster", "core.networks.nerf": "a-nerf_amd.networks", "core.cutoff_embedder": "a-nerf_amd.networks",
           "core.trainer": None, "core.pose_opt": "a-nerf_amd.pose_opt", "run_nerf": "a-nerf_amd.render"}
TRAINER_HOME = {"render": "a-nerf_amd.render", "batchify_rays": "a-nerf_amd.render", "decay_optimizer_lrate": "a-nerf_amd.trainer",
                "Trainer.__init__": "a-nerf_amd.trainer", "Trainer.train_batch": "a-nerf_amd.trainer"}
SIGS = json.load(open(GOLDEN))


def default_repr(d):
    if d is inspect.Parameter.empty:
        return None
    if type(d) in (int, float, bool, str, type(None), list, dict, tuple):
        return repr(d)
    if callable(d):
        return "callable:" + getattr(d, "__name__", type(d).__name__)
    return "object:" + type(d).__name__


def ours(key):
    mod, path = key.split(":")
    home = MODULES[mod] if MODULES[mod] is not None else TRAINER_HOME[path]
    obj = importlib.import_module(home)
    for part in path.split("."):
        obj = getattr(obj, part)
    return obj


@pytest.mark.parametrize("key", sorted(SIGS))
def test_mirror_accepts_the_reference_signature(key):
    ref = SIGS[key]
    mine = list(inspect.signature(ours(key)).parameters.items())
    names = [n for n, _ in mine]
    has_var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for _, p in mine)
    for pos, r in enumerate(ref):
        if r["kind"] in ("VAR_KEYWORD", "VAR_POSITIONAL"):
            assert any(p.kind.name == r["kind"] for _, p in mine), (key, r["name"], "the reference swallows extra arguments here")
            continue
        assert r["name"] in names, (key, r["name"], "parameter missing")
        i = names.index(r["name"])
        p = mine[i][1]
        assert i == pos, (key, r["name"], f"position {i} here, {pos} in the reference")
        assert p.kind.name == r["kind"], (key, r["name"], p.kind.name, r["kind"])
        if r["default"] is not None and r["default"].startswith("object:"):
            continue          # the SMPL skeleton constant: ours takes the caller's (data_attrs["skel_type"]) or None
        assert default_repr(p.default) == r["default"], (key, r["name"], default_repr(p.default), r["default"])
    extra = [n for n, p in mine if n not in [r["name"] for r in ref] and p.kind not in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL)]
    for n in extra:           # additions of ours must not shift or shadow anything: keyword-capable and defaulted
        p = dict(mine)[n]
        assert p.default is not inspect.Parameter.empty, (key, n, "extra parameter without a default")
    assert has_var_kw or not any(r["kind"] == "VAR_KEYWORD" for r in ref)
