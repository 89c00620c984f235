"""ctypes mirror of include/sdempc.h (struct layouts + library loader)."""
import ctypes as C
import os

NX = 13
NNOISE = 6
MAX_MOTORS = 8
HID = 32
BLOB_MAGIC = 0x31454453
BLOB_HEADER_INTS = 16
BLOB_FLOATS = 2120

_F8 = C.c_float * MAX_MOTORS
_F3 = C.c_float * 3


class SdempcCfg(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("horizon", C.c_int32),
        ("num_particles", C.c_int32),
        ("num_motors", C.c_int32),
        ("time_steps", C.POINTER(C.c_float)),
        ("discount", C.c_float),
        ("uref", _F8),
        ("uerr", C.c_float),
        ("perr", _F3), ("verr", _F3), ("qerr", _F3), ("werr", _F3),
        ("res_mult", C.c_float),
        ("u_slew_coeff", C.c_float),
        ("has_slew_constr", C.c_int32),
        ("u_slew_lo", _F8), ("u_slew_hi", _F8),
        ("u_slew_constr_coeff", C.c_float),
        ("u_lo", _F8), ("u_hi", _F8),
        ("max_iter", C.c_int32),
        ("max_no_improvement_iter", C.c_int32),
        ("use_moment_scale", C.c_int32),
        ("moment_scale", C.c_float),
        ("beta_init", C.c_float),
        ("atol", C.c_float), ("rtol", C.c_float),
        ("stepsize", C.c_float),
        ("ls_init_stepsize", C.c_float), ("ls_max_stepsize", C.c_float), ("ls_coef", C.c_float),
        ("ls_decrease_factor", C.c_float), ("ls_increase_factor", C.c_float),
        ("ls_reset_option", C.c_int32),
        ("ls_maxls", C.c_int32),
        ("mlp_dtype", C.c_int32),
        ("math_mode", C.c_int32),
        ("num_state_constr", C.c_int32),
        ("state_id", C.c_int32 * NX),
        ("state_w", C.c_float * NX), ("state_lo", C.c_float * NX), ("state_hi", C.c_float * NX),
    ]


class SdempcInfo(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "avg_linesearch", "stepsize", "num_steps", "grad_sqr", "avg_stepsize", "init_cost", "opt_cost",
        "num_ls_trials")]


INFO_FIELDS = [f[0] for f in SdempcInfo._fields_]

# execution options of a handle (include/sdempc.h, SDEMPC_OPT_*)
OPTIONS = {"lane": 1, "coop": 2, "spec": 3, "pk": 4, "ustg": 5, "coop_launch": 6, "coop_fence": 7, "coop_spin_us": 8, "device_cus": 9, "duo": 10, "hex": 11, "test_absent_wg": 12}

ABI_VERSION = 2          # include/sdempc.h: SDEMPC_ABI_VERSION (the layout of SdempcCfg / SdempcInfo below)

_LIB = None


def lib_path():
    """csrc/libsdempc.so; SDEMPC_LIB may name another build of the same library (kernel A/B runs)."""
    override = os.environ.get("SDEMPC_LIB")
    if override:
        return os.path.abspath(override)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libsdempc.so")


def load_library():
    """Load csrc/libsdempc.so. Fails loudly: there is no CPU fallback for the product path."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C sde4mbrl_px4_amd/csrc). sde4mbrl_px4_amd has no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, fp = C.c_void_p, C.c_int32, C.POINTER(C.c_float)
    lib.sdempc_create.argtypes = [C.POINTER(SdempcCfg), vp, C.c_size_t, i32, C.POINTER(vp)]
    lib.sdempc_create.restype = C.c_int
    lib.sdempc_destroy.argtypes = [vp]
    lib.sdempc_destroy.restype = None
    lib.sdempc_last_error.argtypes = [vp]
    lib.sdempc_last_error.restype = C.c_char_p
    lib.sdempc_abi_version.restype = C.c_int
    got = lib.sdempc_abi_version()
    if got != ABI_VERSION:       # a stale build (or an SDEMPC_LIB override made against an older header) would read SdempcCfg past its end
        raise RuntimeError(f"{path} reports ABI version {got}, this package's struct layouts are version {ABI_VERSION} "
                           "(include/sdempc.h: SDEMPC_ABI_VERSION): rebuild the library (make -C sde4mbrl_px4_amd/csrc)")
    lib.sdempc_build_flags.restype = C.c_int
    lib.sdempc_set_device.argtypes = [vp, i32]
    lib.sdempc_device_ready.argtypes = [vp]
    lib.sdempc_device_ready.restype = C.c_int
    lib.sdempc_set_option.argtypes = [vp, i32, i32]
    lib.sdempc_set_option.restype = C.c_int
    lib.sdempc_get_option.argtypes = [vp, i32, C.POINTER(i32)]
    lib.sdempc_get_option.restype = C.c_int
    lib.sdempc_reset.argtypes = [vp, fp, fp, fp, C.POINTER(SdempcInfo)]
    lib.sdempc_rollout_batch.argtypes = [vp, i32, fp, fp, fp, fp, fp, fp, fp]
    lib.sdempc_grad_batch.argtypes = [vp, i32, fp, fp, fp, fp, fp, fp]
    lib.sdempc_solve_batch.argtypes = [vp, i32, fp, fp, fp, fp, fp, fp, fp, C.POINTER(SdempcInfo)]
    lib.sdempc_noise_dev_floats.argtypes = [vp, i32]
    lib.sdempc_noise_dev_floats.restype = C.c_size_t
    lib.sdempc_traj_dev_floats.argtypes = [vp, i32]
    lib.sdempc_traj_dev_floats.restype = C.c_size_t
    lib.sdempc_noise_to_device_layout.argtypes = [vp, i32, fp, fp]
    lib.sdempc_noise_to_device_layout_dev.argtypes = [vp, i32, vp, vp, vp]
    lib.sdempc_traj_to_canonical_dev.argtypes = [vp, i32, vp, vp]
    lib.sdempc_solve_batch_dev.argtypes = [vp, i32] + [vp] * 9
    lib.sdempc_rollout_batch_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.sdempc_grad_batch_dev.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    u32p = C.POINTER(C.c_uint32)
    lib.sdempc_noise_from_keys_dev.argtypes = [vp, i32, u32p, vp, vp]
    lib.sdempc_noise_from_keys.argtypes = [vp, i32, u32p, fp]
    lib.sdempc_solve_batch_keys.argtypes = [vp, i32, fp, fp, u32p, fp, fp, fp, fp, C.POINTER(SdempcInfo)]
    lib.sdempc_solve_status.argtypes = [vp]
    lib.sdempc_solve_status.restype = C.c_int
    lib.sdempc_layout_fallbacks.argtypes = [vp]
    lib.sdempc_layout_fallbacks.restype = C.c_int32
    lib.sdempc_work_counters.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    lib.sdempc_work_counters.restype = C.c_int
    lib.sdempc_last_kernel_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.sdempc_last_kernel_name.restype = C.c_int
    lib.sdempc_last_kernel_ms.argtypes = [vp]
    lib.sdempc_last_kernel_ms.restype = C.c_float
    for name in ("sdempc_set_device", "sdempc_reset", "sdempc_rollout_batch", "sdempc_grad_batch", "sdempc_solve_batch",
                 "sdempc_noise_to_device_layout", "sdempc_solve_batch_dev", "sdempc_rollout_batch_dev",
                 "sdempc_grad_batch_dev", "sdempc_noise_to_device_layout_dev", "sdempc_traj_to_canonical_dev",
                 "sdempc_noise_from_keys_dev", "sdempc_noise_from_keys", "sdempc_solve_batch_keys"):
        getattr(lib, name).restype = C.c_int
    _LIB = lib
    return lib


EXPORTED_SYMBOLS = [
    "sdempc_create", "sdempc_destroy", "sdempc_last_error", "sdempc_abi_version", "sdempc_build_flags", "sdempc_set_device", "sdempc_device_ready", "sdempc_set_option", "sdempc_get_option", "sdempc_reset",
    "sdempc_rollout_batch", "sdempc_grad_batch", "sdempc_solve_batch", "sdempc_noise_dev_floats",
    "sdempc_traj_dev_floats", "sdempc_noise_to_device_layout", "sdempc_solve_batch_dev", "sdempc_rollout_batch_dev",
    "sdempc_grad_batch_dev", "sdempc_last_kernel_ms", "sdempc_last_kernel_name", "sdempc_work_counters", "sdempc_solve_status", "sdempc_layout_fallbacks", "sdempc_noise_to_device_layout_dev", "sdempc_traj_to_canonical_dev",
    "sdempc_noise_from_keys_dev", "sdempc_noise_from_keys", "sdempc_solve_batch_keys",
]
