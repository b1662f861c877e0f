"""ctypes binding of libanerf_hip.so (the C ABI declared in include/anerf.h).

The library is the product path: if it is missing this module raises at import of the symbols --
there is no CPU fallback (oracle/ is test infrastructure and is never imported from here).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ANERF_LIB", os.path.join(_HERE, "libanerf_hip.so"))   # ANERF_LIB: ablation builds only

c_f32p = C.c_void_p  # device pointers are passed as integers (tensor.data_ptr())


class AnerfConfig(C.Structure):
    _fields_ = [("n_joints", C.c_int32), ("multires", C.c_int32), ("multires_views", C.c_int32),
                ("framecode_ch", C.c_int32), ("netdepth", C.c_int32), ("netwidth", C.c_int32),
                ("skip", C.c_int32), ("density_act", C.c_int32), ("density_scale", C.c_float),
                ("softplus_shift", C.c_float), ("cutoff_bones", C.c_int32)]


class AnerfNetParams(C.Structure):
    _fields_ = [("w", C.c_void_p * 12), ("b", C.c_void_p * 12), ("codes", C.c_void_p), ("n_codes", C.c_int32),
                ("sched_dim_x", C.c_int32), ("sched_dim_u", C.c_int32), ("reserved_", C.c_int32),
                ("sched_x", C.c_void_p), ("sched_u", C.c_void_p)]


class AnerfLayout(C.Structure):
    _fields_ = [("stream_floats", C.c_int64), ("aux_floats", C.c_int64), ("n_stages", C.c_int32),
                ("x_width", C.c_int32)]


class AnerfSaved(C.Structure):
    _fields_ = [("h", C.c_void_p), ("f", C.c_void_p), ("g", C.c_void_p), ("x", C.c_void_p), ("u", C.c_void_p),
                ("p_pad", C.c_int64)]


ABI_VERSION = 7        # revision of include/anerf.h these structures were written for (checked against anerf_version())
PROF_SLOTS = 16


class AnerfProfile(C.Structure):
    _fields_ = [("ev", C.c_void_p * PROF_SLOTS)]


class AnerfPackJob(C.Structure):
    _fields_ = [("params", AnerfNetParams), ("table", C.c_void_p), ("n", C.c_int64), ("out", C.c_void_p), ("kind", C.c_int32)]


class AnerfRandJob(C.Structure):
    _fields_ = [("out", C.c_void_p), ("n", C.c_int64), ("kind", C.c_int32), ("scale", C.c_float)]


MAX_ADAM_GROUPS = 4


class AnerfStepBlock(C.Structure):        # DEVICE layout (ABI revision 6); on the host only its size and field offsets are used
    _fields_ = [("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64), ("tau_v", C.c_float), ("tau_d", C.c_float),
                ("adam_step_size", C.c_float * MAX_ADAM_GROUPS), ("adam_sqrt_bc2", C.c_float * MAX_ADAM_GROUPS),
                ("adam_grad_scale", C.c_float * MAX_ADAM_GROUPS), ("reserved_", C.c_float * 2)]


class AnerfStepValues(C.Structure):
    _fields_ = [("rng_seed", C.c_uint64), ("rng_offset", C.c_uint64), ("tau_v", C.c_float), ("tau_d", C.c_float),
                ("n_groups", C.c_int32), ("lr", C.c_float * MAX_ADAM_GROUPS), ("beta1", C.c_float * MAX_ADAM_GROUPS),
                ("beta2", C.c_float * MAX_ADAM_GROUPS), ("adam_step", C.c_int32 * MAX_ADAM_GROUPS),
                ("grad_scale", C.c_float * MAX_ADAM_GROUPS)]


class AnerfForwardIO(C.Structure):
    _fields_ = [("packed_c", C.c_void_p), ("aux_c", C.c_void_p), ("packed_f", C.c_void_p), ("aux_f", C.c_void_p),
                ("rays", C.c_void_p), ("ray_stride", C.c_int32),
                ("skts", C.c_void_p), ("skt_ray_stride", C.c_int64),
                ("cyls", C.c_void_p), ("cam_idx", C.c_void_p), ("codes_c", C.c_void_p), ("codes_f", C.c_void_p), ("n_codes", C.c_int32),
                ("t_rand", C.c_void_p), ("u_imp", C.c_void_p), ("noise", C.c_void_p), ("noise_fine", C.c_void_p),
                ("cutoff_v", C.c_void_p), ("cutoff_d", C.c_void_p), ("tau_v", C.c_float), ("tau_d", C.c_float),
                ("n_rays", C.c_int32), ("n_samples", C.c_int32), ("n_importance", C.c_int32), ("lindisp", C.c_int32),
                ("single_net", C.c_int32), ("precision", C.c_int32),
                ("rgb_map", C.c_void_p), ("disp_map", C.c_void_p), ("acc_map", C.c_void_p), ("alpha", C.c_void_p),
                ("rgb0", C.c_void_p), ("disp0", C.c_void_p), ("acc0", C.c_void_p), ("alpha0", C.c_void_p),
                ("pts_noise", C.c_void_p), ("pts_noise_is", C.c_void_p), ("profile", C.POINTER(AnerfProfile)), ("cyl_shared", C.c_int32),
                ("step", C.c_void_p)]


class AnerfNetGrads(C.Structure):
    _fields_ = [("w", C.c_void_p * 12), ("b", C.c_void_p * 12), ("sched_x", C.c_void_p), ("sched_u", C.c_void_p)]


class AnerfBackwardIO(C.Structure):
    _fields_ = [("g_rgb", C.c_void_p), ("g_disp", C.c_void_p), ("g_acc", C.c_void_p), ("g_alpha", C.c_void_p),
                ("g_rgb0", C.c_void_p), ("g_disp0", C.c_void_p), ("g_acc0", C.c_void_p), ("g_alpha0", C.c_void_p),
                ("packed_t_c", C.c_void_p), ("packed_t_f", C.c_void_p), ("packed_i_c", C.c_void_p), ("packed_i_f", C.c_void_p),
                ("perm_x", C.c_void_p), ("perm_u", C.c_void_p),
                ("grads_c", AnerfNetGrads), ("grads_f", AnerfNetGrads),
                ("g_skts", C.c_void_p), ("g_codes_c", C.c_void_p), ("g_codes_f", C.c_void_p), ("accumulate", C.c_int32),
                ("passes", C.c_int32), ("profile", C.POINTER(AnerfProfile))]


class AnerfTrainLayout(C.Structure):
    _fields_ = [("p_pad", C.c_int64), ("x_width", C.c_int32), ("u_width", C.c_int32), ("gemm_chunks", C.c_int32),
                ("gemm_ws_floats", C.c_int64)]


# name -> (restype, argtypes); every symbol include/anerf.h declares must be listed here
SIGNATURES = {
    "anerf_last_error": (C.c_char_p, []),
    "anerf_version": (C.c_int, []),
    "anerf_layout": (C.c_int, [C.POINTER(AnerfConfig), C.c_int, C.POINTER(AnerfLayout)]),
    "anerf_build_pack_table": (C.c_int, [C.POINTER(AnerfConfig), C.c_int, C.c_void_p]),
    "anerf_pack_params": (C.c_int, [C.POINTER(AnerfNetParams), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "anerf_ray_bounds": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_coarse_z": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_mlp_raw": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "anerf_mlp_forward": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "anerf_composite": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "anerf_importance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_train_layout": (C.c_int, [C.POINTER(AnerfConfig), C.c_int64, C.POINTER(AnerfTrainLayout)]),
    "anerf_build_perm_tables": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p]),
    "anerf_mlp_raw_train": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(AnerfSaved),
                                      C.c_void_p]),
    "anerf_composite_backward": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_mlp_backward": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AnerfSaved),
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "anerf_weight_grads": (C.c_int, [C.POINTER(AnerfConfig), C.POINTER(AnerfSaved), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(AnerfNetGrads), C.c_void_p,
                                     C.c_int64, C.c_void_p]),
    "anerf_input_grads": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_encode_backward": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_code_grads": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_int32, C.c_void_p, C.c_void_p]),
    "anerf_density": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                C.c_int64, C.c_void_p, C.c_void_p]),
    "anerf_gen_rays": (C.c_int, [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_assemble_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
    "anerf_pack_params_b3": (C.c_int, [C.POINTER(AnerfNetParams), C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "anerf_mlp_raw_b3": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "anerf_build_perm_tables_b3": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p]),
    "anerf_mlp_raw_train_b3": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(AnerfSaved),
                                         C.c_void_p]),
    "anerf_mlp_backward_b3": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(AnerfSaved),
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "anerf_input_grads_b3": (C.c_int, [C.POINTER(AnerfConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_weight_grads_b3": (C.c_int, [C.POINTER(AnerfConfig), C.POINTER(AnerfSaved), C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(AnerfNetGrads), C.c_void_p,
                                        C.c_int64, C.c_void_p]),
    "anerf_workspace_size": (C.c_int64, [C.POINTER(AnerfConfig), C.c_int32, C.c_int32, C.c_int32]),
    "anerf_forward": (C.c_int, [C.POINTER(AnerfConfig), C.POINTER(AnerfForwardIO), C.c_void_p, C.c_int64, C.c_void_p]),
    "anerf_loss_blocks": (C.c_int, [C.c_int32]),
    "anerf_loss": (C.c_int, [C.c_void_p] * 6 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float] + [C.c_void_p] * 7),
    "anerf_fk_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 5),
    "anerf_fk_backward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32] + [C.c_void_p] * 7),
    "anerf_pose_batch_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] +
                                 [C.c_void_p] * 9),
    "anerf_pose_batch_backward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32] +
                                  [C.c_void_p] * 10 + [C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "anerf_pose_batch_scratch_size": (C.c_int64, [C.c_int32, C.c_int32]),
    "anerf_train_workspace_size": (C.c_int64, [C.POINTER(AnerfConfig), C.c_int32, C.c_int32, C.c_int32]),
    "anerf_backward_scratch_size": (C.c_int64, [C.POINTER(AnerfConfig), C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "anerf_train_forward": (C.c_int, [C.POINTER(AnerfConfig), C.POINTER(AnerfForwardIO), C.c_void_p, C.c_int64, C.c_void_p]),
    "anerf_backward": (C.c_int, [C.POINTER(AnerfConfig), C.POINTER(AnerfForwardIO), C.POINTER(AnerfBackwardIO), C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "anerf_pack_params_multi": (C.c_int, [C.POINTER(AnerfPackJob), C.c_int32, C.c_void_p]),
    "anerf_rand_fill": (C.c_int, [C.POINTER(AnerfRandJob), C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p]),
    "anerf_make_ray_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "anerf_cyl_bbox": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_void_p, C.c_void_p]),
    "anerf_kp_loss": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "anerf_kp_loss_add": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_adam_blocks": (C.c_int, [C.c_int64]),
    "anerf_adam_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32,
                                  C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "anerf_step_block_write": (C.c_int, [C.c_void_p, C.POINTER(AnerfStepValues), C.c_void_p]),
    "anerf_rand_fill_dev": (C.c_int, [C.POINTER(AnerfRandJob), C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "anerf_adam_step_dev": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` "
                           "(make -C a-nerf_amd/csrc).  There is no CPU fallback for the hot path.")
    # torch first: it ships its own HIP runtime, and the one mapped FIRST is the one both sides must share -- loading this library
    # before torch maps the system runtime for it and torch's own next to it, and the library's launches then fail with "no
    # ROCm-capable device is detected" (seen when build() and smoke() ran in one process)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    # the ctypes structures above mirror ONE revision of include/anerf.h: a library of another revision would read fields
    # past the end of (or at other offsets in) the caller's structs -- refuse it instead of running on garbage
    if lib.anerf_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports ABI revision {lib.anerf_version()}, these bindings are for revision {ABI_VERSION}: "
                           "rebuild the library (make -C a-nerf_amd/csrc)")
    _lib = lib
    return lib


class AnerfError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().anerf_last_error().decode()
        raise AnerfError(f"{what} failed with code {rc}: {msg}")
