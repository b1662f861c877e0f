"""CPU: libanerf_hip.so loads and exports every symbol include/anerf.h declares (no compute calls)."""
import ctypes
import importlib
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound():
    lib_mod = importlib.import_module("a-nerf_amd._lib")
    lib = lib_mod.load()
    hdr = open(os.path.join(ROOT, "include", "anerf.h")).read()
    declared = set(re.findall(r"\b(anerf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in anerf.h but not exported"
    assert declared == set(lib_mod.SIGNATURES), (declared ^ set(lib_mod.SIGNATURES))
    assert lib.anerf_version() == lib_mod.ABI_VERSION == 7


def test_step_block_structs_match_the_compiled_header(tmp_path):
    """ABI revision 6: sizes and field offsets of AnerfStepBlock / AnerfStepValues / AnerfForwardIO as a C compiler lays out
    include/anerf.h, against the ctypes mirrors the host side fills"""
    import subprocess
    lib_mod = importlib.import_module("a-nerf_amd._lib")
    src = tmp_path / "sz.c"
    src.write_text('#include "anerf.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(AnerfStepBlock), sizeof(AnerfStepValues), sizeof(AnerfForwardIO), offsetof(AnerfStepBlock, tau_v), '
                   'offsetof(AnerfStepBlock, adam_grad_scale), offsetof(AnerfStepValues, adam_step), offsetof(AnerfStepValues, grad_scale), '
                   'offsetof(AnerfForwardIO, step));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    B, V, F = lib_mod.AnerfStepBlock, lib_mod.AnerfStepValues, lib_mod.AnerfForwardIO
    assert got == [ctypes.sizeof(B), ctypes.sizeof(V), ctypes.sizeof(F), B.tau_v.offset, B.adam_grad_scale.offset, V.adam_step.offset,
                   V.grad_scale.offset, F.step.offset]


def test_layout_and_pack_table_host_side():
    """Host-only entry points: layout sizes and the gather table are consistent with the layer shapes."""
    ops = importlib.import_module("a-nerf_amd.ops")
    lib_mod = importlib.import_module("a-nerf_amd._lib")
    synth = importlib.import_module("a-nerf_amd.synth")
    for kw, xw in [({}, 1080), ({"framecode_ch": 16}, 1081), ({"multires_views": 0}, 504)]:
        cfg = ops.PathConfig(**kw)
        sf, af, nst, x_width = ops.layout(cfg)
        assert x_width == xw and sf == nst * 8192 and af == 3080
        host = np.empty(sf + af, dtype=np.int32)
        cc = cfg.c()
        assert lib_mod.load().anerf_build_pack_table(ctypes.byref(cc), 0, host.ctypes.data_as(ctypes.c_void_p)) == 0
        used = host[host >= 0]
        ids, offs = used >> 24, used & 0xFFFFFF
        shapes = synth.net_shapes(7, cfg.multires_views, cfg.framecode_ch)
        names = ops.PARAM_ORDER
        # every weight element of every layer appears exactly once across stream + aux
        for i, n in enumerate(names):
            o, k = shapes[n]
            cnt = np.bincount(offs[ids == i], minlength=o * k)
            assert cnt.shape[0] == o * k and (cnt == 1).all(), n
            b = np.bincount(offs[ids == 12 + i], minlength=o)
            assert (b == 1).all(), n + ".bias"
    # bf16x3 image: one table entry per bf16 element (hi and lo of every weight exactly once) + the fp32 aux entries
    cfg = ops.PathConfig()
    sf, af, nst, _ = ops.layout(cfg, 3)
    assert (sf, nst) == ops.layout(cfg, 0)[0:3:2]
    host = np.empty(2 * sf + af, dtype=np.int32)
    cc = cfg.c()
    assert lib_mod.load().anerf_build_pack_table(ctypes.byref(cc), 3, host.ctypes.data_as(ctypes.c_void_p)) == 0
    st = host[:2 * sf]
    used = st[st >= 0]
    part, ids, offs = (used >> 29) & 1, (used >> 24) & 31, used & 0xFFFFFF
    shapes = synth.net_shapes(7, 4, 0)
    for i, n in enumerate(ops.PARAM_ORDER):
        if i in (8, 11):
            continue                      # alpha / rgb heads live in aux (fp32)
        o, k = shapes[n]
        for pp in (0, 1):
            cnt = np.bincount(offs[(ids == i) & (part == pp)], minlength=o * k)
            assert (cnt == 1).all(), (n, pp)
    # bf16x3 W^T image (which=4, backward-data): every hidden-side weight the backward contracts appears once as hi and once
    # as lo: views_linears.0[:, :256], feature, pts_linears.1-4, 6, 7, and pts_linears.5[:, 432:]
    sf4, af4, nst4, _ = ops.layout(cfg, 4)
    assert (sf4, nst4) == ops.layout(cfg, 1)[0:3:2]
    host = np.empty(2 * sf4 + af4, dtype=np.int32)
    assert lib_mod.load().anerf_build_pack_table(ctypes.byref(cc), 4, host.ctypes.data_as(ctypes.c_void_p)) == 0
    used = host[:2 * sf4][host[:2 * sf4] >= 0]
    part, ids, offs = (used >> 29) & 1, (used >> 24) & 31, used & 0xFFFFFF
    for i, n in enumerate(ops.PARAM_ORDER):
        o, k = shapes[n]
        sel = ids == i
        if i in (1, 2, 3, 4, 6, 7, 9):
            cols = np.arange(k)
        elif i == 5:
            cols = np.arange(432, k)
        elif i == 10:
            cols = np.arange(256)
        else:
            assert not sel.any(), n
            continue
        want = (np.arange(o)[:, None] * k + cols[None, :]).ravel()
        for pp in (0, 1):
            got = np.sort(offs[sel & (part == pp)])
            assert np.array_equal(got, np.sort(want)), (n, pp)
    bad = ops.PathConfig(multires=5)
    cc = bad.c()
    L = lib_mod.AnerfLayout()
    assert lib_mod.load().anerf_layout(ctypes.byref(cc), 0, ctypes.byref(L)) == -1
    assert b"unsupported" in lib_mod.load().anerf_last_error()


def test_revision6_entry_points_refuse_bad_arguments_before_any_launch():
    """anerf_step_block_write / anerf_rand_fill_dev / anerf_adam_step_dev validate on the host (error code + message, nothing is
    enqueued): NULL and misaligned blocks, group / n_groups / call_index out of range; an empty update is a no-op."""
    lib_mod = importlib.import_module("a-nerf_amd._lib")
    lib = lib_mod.load()
    E_SHAPE, E_NULL = -2, -3
    vals = lib_mod.AnerfStepValues()
    fake = ctypes.c_void_p(0x7F0000001000)             # never dereferenced: every call below returns before its launch
    assert lib.anerf_step_block_write(None, ctypes.byref(vals), None) == E_NULL
    assert lib.anerf_step_block_write(fake, None, None) == E_NULL
    assert lib.anerf_step_block_write(ctypes.c_void_p(0x7F0000001008), ctypes.byref(vals), None) == E_SHAPE
    assert b"16-byte" in lib.anerf_last_error()
    vals.n_groups = 5
    assert lib.anerf_step_block_write(fake, ctypes.byref(vals), None) == E_SHAPE
    job = (lib_mod.AnerfRandJob * 1)(lib_mod.AnerfRandJob(0x7F0000002000, 16, 0, 1.0))
    assert lib.anerf_rand_fill_dev(job, 1, None, 0, None) == E_NULL
    assert lib.anerf_rand_fill_dev(job, 1, fake, -1, None) == E_SHAPE
    assert lib.anerf_rand_fill_dev(job, 0, fake, 0, None) == 0                      # no jobs: nothing to do
    a = ctypes.c_void_p(0x7F0000003000)
    assert lib.anerf_adam_step_dev(a, a, a, a, 64, 0.9, 0.999, 1e-8, fake, 4, 0, 1, None, None, None) == E_SHAPE
    assert lib.anerf_adam_step_dev(a, a, a, a, 64, 0.9, 0.999, 1e-8, None, 0, 0, 1, None, None, None) == E_NULL
    assert lib.anerf_adam_step_dev(a, a, a, a, 0, 0.9, 0.999, 1e-8, fake, 0, 0, 1, None, None, None) == 0
    assert lib.anerf_adam_step_dev(ctypes.c_void_p(0x7F0000003004), a, a, a, 64, 0.9, 0.999, 1e-8, fake, 0, 0, 1, None, None, None) == E_SHAPE
