"""ctypes binding of libmarlin_hip.so (the C ABI in include/marlin_hip.h).

Fails loudly when the HIP library is missing: there is no CPU fallback in the product path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmarlin_hip.so")


class MrlDomain(C.Structure):
    _fields_ = [
        ("dim", C.c_int32),
        ("n", C.c_int64 * 3),
        ("min", C.c_double * 3),
        ("max", C.c_double * 3),
        ("device", C.c_int32),
        ("nranks", C.c_int32),
        ("rank", C.c_int32),
        ("weights", C.POINTER(C.c_int64)),
        ("spectrum", C.c_int32),
        ("stream", C.c_void_p),
        ("flags", C.c_int32),
    ]


class MrlChParams(C.Structure):
    _fields_ = [
        ("family", C.c_int32),
        ("coef", C.c_double * 4),
        ("mobility", C.c_double),
        ("kappa", C.c_double),
        ("parsed", C.c_void_p),
    ]


class MrlMechParams(C.Structure):
    _fields_ = [
        ("l_tol", C.c_double),
        ("l_max_its", C.c_int64),
        ("nl_rel_tol", C.c_double),
        ("nl_abs_tol", C.c_double),
        ("nl_max_its", C.c_int32),
    ]


class MrlTiming(C.Structure):
    _fields_ = [
        ("kernel_classes", C.c_int32),
        ("launches", C.c_int64),
        ("device_ms", C.c_double),
        ("algorithmic_bytes", C.c_double),
        ("dominant", C.c_char_p),
        ("dominant_ms", C.c_double),
    ]


class MrlMechStats(C.Structure):
    _fields_ = [
        ("newton_its", C.c_int32),
        ("cg_its_total", C.c_int32),
        ("cg_its", C.c_int32 * 64),
        ("last_anorm", C.c_double),
        ("last_rnorm", C.c_double),
        ("Fn", C.c_double),
    ]


_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes): every symbol include/marlin_hip.h declares
SIGNATURES = {
    "mrl_abi_version": (_i32, []),
    "mrl_ctx_create": (_i32, [C.POINTER(_vp), C.POINTER(MrlDomain)]),
    "mrl_ctx_destroy": (None, [_vp]),
    "mrl_last_error": (C.c_char_p, [_vp]),
    "mrl_sync": (_i32, [_vp]),
    "mrl_set_stream": (_i32, [_vp, _vp]),
    "mrl_local_shape": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_reciprocal_axis": (_i32, [_i64, _dbl, _i32, C.POINTER(_dbl)]),
    "mrl_partition": (_i32, [_i64, C.c_int32, C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_pencil_factors": (_i32, [C.c_int32, C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mrl_pencil_grid": (_i32, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mrl_pencil_layout": (_i32, [C.c_int32, C.c_int32, C.POINTER(_i64)] + [C.POINTER(_i64)] * 8),
    "mrl_ctx_reciprocal_axis": (_i32, [_vp, _i32, C.POINTER(_dbl), _i64]),
    "mrl_fft_r2c": (_i32, [_vp, _vp, _vp, _i64, _i32]),
    "mrl_fft_c2r": (_i32, [_vp, _vp, _vp, _i64, _i32]),
    "mrl_slab_counts": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_slab_fwd_local": (_i32, [_vp, _vp, _vp]),
    "mrl_slab_fwd_finish": (_i32, [_vp, _vp, _vp]),
    "mrl_slab_inv_local": (_i32, [_vp, _vp, _vp]),
    "mrl_slab_inv_finish": (_i32, [_vp, _vp, _vp]),
    "mrl_ch_spec_elems": (_i64, [_vp]),
    "mrl_ch_spec_layout": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_ch_mu": (_i32, [_vp, C.POINTER(MrlChParams), _vp, _vp, _i64]),
    "mrl_ch_substep": (_i32, [_vp, C.POINTER(MrlChParams), _vp, _vp, _vp, _pp, _i32, _dbl, _vp, _vp, _i32]),
    "mrl_ch_substeps": (_i32, [_vp, C.POINTER(MrlChParams), _vp, _vp, _pp, _i32, C.POINTER(_i32), C.POINTER(_i32), _i32, _i32, _i32,
                        _dbl, _vp]),
    "mrl_ch_spec_elems_f32": (_i64, [_vp]),
    "mrl_ch_spec_layout_f32": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_ch_substeps_f32": (_i32, [_vp, C.POINTER(MrlChParams), _vp, _vp, _pp, _i32, C.POINTER(_i32), C.POINTER(_i32), _i32, _i32, _i32, _dbl]),
    "mrl_kspace_abm": (_i32, [_vp, _vp, _vp, _pp, C.POINTER(_dbl), _i32, _vp, _dbl, _i64]),
    "mrl_kspace_coupled": (_i32, [_vp, _i32, _pp, _pp, _pp, C.POINTER(_dbl), C.POINTER(_i32), _pp, _dbl, _i32, _i64]),
    "mrl_slab_fast_path": (_i32, [_vp]),
    "mrl_slab_gamma_counts": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_slab_gamma_row_fwd": (_i32, [_vp, _i32, _vp, _vp]),
    "mrl_slab_gamma_row_mid": (_i32, [_vp, _vp, _dbl]),
    "mrl_slab_gamma_row_inv": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "mrl_slab_gamma_dot": (_i32, [_vp, C.POINTER(_dbl)]),
    "mrl_cg_update": (_i32, [_vp, _dbl, _vp, _vp, _vp, _vp, _i64, C.POINTER(_dbl)]),
    "mrl_cg_update_r": (_i32, [_vp, _dbl, _vp, _vp, _i64, C.POINTER(_dbl)]),
    "mrl_slab_gamma_tangent_fusable": (_i32, [_vp]),
    "mrl_slab_gamma_tangent_z_fwd": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp, _dbl]),
    "mrl_mech_tangent_dir_fm": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _dbl, _vp]),
    "mrl_mech_stress_fm": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "mrl_mech_tangent_apply_fm": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mrl_mech_displacements": (_i32, [_vp, _vp, _vp]),
    "mrl_mech_von_mises": (_i32, [_vp, _vp, _vp]),
    "mrl_qs_elasticity": (_i32, [_vp, _vp, _dbl, _dbl, _dbl, _pp]),
    "mrl_elastic_chemical_potential": (_i32, [_vp, _vp, _pp, _dbl, _dbl, _dbl, _vp]),
    "mrl_broyden_init": (_i32, [_vp, _i32, _dbl, _vp, _i64]),
    "mrl_broyden_residual": (_i32, [_vp, _i32, _pp, _pp, _pp, _pp, _dbl, _vp, C.POINTER(_dbl), _i64]),
    "mrl_broyden_predict": (_i32, [_vp, _i32, _vp, _vp, _pp, _dbl, _vp, _pp, _i64]),
    "mrl_broyden_update": (_i32, [_vp, _i32, _vp, _vp, _vp, _pp, _pp, _pp, _pp, _dbl, C.POINTER(_dbl), _i64]),
    "mrl_histogram": (_i32, [_vp, _vp, _i64, C.POINTER(_dbl), _i32, C.POINTER(_i64)]),
    "mrl_secant_begin": (_i32, [_vp, _vp, _vp, _vp, _dbl, _dbl, _vp, _vp, C.POINTER(_dbl), _i64]),
    "mrl_secant_iterate": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _vp, C.POINTER(_dbl), _i64]),
    "mrl_slab_ch_spec_pitch": (_i64, [_vp]),
    "mrl_slab_ch_k_pitch": (_i64, [_vp, _i32, _i32]),
    "mrl_slab_ch_counts": (_i32, [_vp, _i32, _i32, _i32, _i32, C.POINTER(_i64), C.POINTER(_i64)]),
    "mrl_slab_ch_z_fwd": (_i32, [_vp, C.POINTER(MrlChParams), _vp, _vp, _i32]),
    "mrl_slab_ch_x_fwd": (_i32, [_vp, _i32, _i32, _vp, _i32]),
    "mrl_slab_ch_kspace": (_i32, [_vp, C.POINTER(MrlChParams), _i32, _i32, _vp, _vp, _vp, _pp, _i32, _dbl, _vp, _i32]),
    "mrl_slab_ch_x_inv": (_i32, [_vp, _i32, _i32, _vp]),
    "mrl_slab_ch_z_inv": (_i32, [_vp, _vp]),
    "mrl_slab_ch_z_inv_fwd": (_i32, [_vp, C.POINTER(MrlChParams), _vp, _i32]),
    "mrl_gamma_apply": (_i32, [_vp, _vp, _vp]),
    "mrl_mech_stress": (_i32, [_vp, _vp, _vp, _vp, _vp]),
    "mrl_mech_tangent_apply": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "mrl_slab_gamma_project": (_i32, [_vp, _vp, _dbl]),
    "mrl_relayout": (_i32, [_vp, _i32, _vp, _vp, _i64, C.c_int32]),
    "mrl_axpby": (_i32, [_vp, _dbl, _vp, _dbl, _vp, _vp, _i64]),
    "mrl_axpy": (_i32, [_vp, _dbl, _vp, _vp, _i64]),
    "mrl_mech_newton_cg": (_i32, [_vp, C.POINTER(MrlMechParams), _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(MrlMechStats)]),
    "mrl_mech_small_strain": (_i32, [_vp, C.POINTER(MrlMechParams), _vp, _vp, _vp, _vp, _vp, C.POINTER(MrlMechStats)]),
    "mrl_parsed_create": (_i32, [_vp, C.POINTER(_vp), C.c_char_p, _i32, C.POINTER(C.c_char_p), C.POINTER(_i32), _i32,
                                 C.POINTER(C.c_char_p), C.POINTER(_dbl), _i32, C.POINTER(C.c_char_p), _i32, _i32]),
    "mrl_parsed_destroy": (None, [_vp]),
    "mrl_parsed_is_complex": (_i32, [_vp]),
    "mrl_parsed_string": (C.c_char_p, [_vp]),
    "mrl_parsed_source": (C.c_char_p, [_vp]),
    "mrl_parsed_eval": (_i32, [_vp, _pp, _vp, _i64, _dbl]),
    "mrl_dot": (_i32, [_vp, _vp, _vp, _i64, C.POINTER(_dbl)]),
    "mrl_norm2": (_i32, [_vp, _vp, _i64, C.POINTER(_dbl)]),
    "mrl_sum": (_i32, [_vp, _vp, _i64, C.POINTER(_dbl)]),
    "mrl_minmax": (_i32, [_vp, _vp, _i64, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "mrl_reciprocal_laplacian": (_i32, [_vp, _i32, _dbl, _vp]),
    "mrl_average": (_i32, [_vp, _vp, _i64, C.POINTER(_dbl)]),
    "mrl_ctx_set_option": (_i32, [_vp, _i32, _i64]),
    "mrl_ctx_get_option": (_i64, [_vp, _i32]),
    "mrl_comm_create": (_i32, [C.POINTER(_vp), C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mrl_comm_destroy": (None, [_vp]),
    "mrl_comm_last_error": (C.c_char_p, [_vp]),
    "mrl_comm_transport": (_i32, [_vp]),
    "mrl_comm_set_transport": (_i32, [_vp, C.c_int32]),
    "mrl_comm_set_timeout": (_i32, [_vp, _dbl]),
    "mrl_comm_reset_error": (_i32, [_vp]),
    "mrl_comm_bootstrap_selftest": (_i32, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32]),
    "mrl_comm_barrier": (_i32, [_vp]),
    "mrl_comm_allreduce": (_i32, [_vp, C.POINTER(_dbl), C.c_int32, C.c_int32]),
    "mrl_comm_stats": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_dbl)]),
    "mrl_comm_describe": (_i32, [_vp, C.c_char_p, C.c_size_t]),
    "mrl_comm_rccl_preflight": (_i32, [_vp]),
    "mrl_ctx_attach_comm": (_i32, [_vp, _vp]),
    "mrl_h5_create": (_i32, [C.c_char_p, C.POINTER(_vp)]),
    "mrl_h5_write": (_i32, [_vp, C.c_char_p, _i32, _i32, C.POINTER(_i64), _vp]),
    "mrl_h5_flush": (_i32, [_vp]),
    "mrl_h5_close": (_i32, [_vp]),
    "mrl_h5_last_error": (C.c_char_p, [_vp]),
    "mrl_timer_start": (_i32, [_vp]),
    "mrl_timer_stop": (_i32, [_vp, C.POINTER(C.c_float)]),
    "mrl_set_profiling": (_i32, [_vp, _i32]),
    "mrl_get_profile": (_i32, [_vp, _i32, C.POINTER(C.c_char_p), C.POINTER(_dbl), C.POINTER(_i64), C.POINTER(_dbl)]),
    "mrl_get_timing": (_i32, [_vp, C.POINTER(MrlTiming)]),
}

_lib = None


def load():
    """Load the HIP library; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.mrl_abi_version() != 3:
        raise RuntimeError("libmarlin_hip.so ABI version mismatch")
    _lib = lib
    return lib
