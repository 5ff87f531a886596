"""ctypes binding of the C ABI declared in include/rustpde_hip.h.

`Lib(path)` binds one shared library explicitly; there is no search and no fallback.  The
product package binds the in-tree HIP library (rustpde_mpi_amd/librustpde_hip.so); the test
suite may bind the host emulation build of the same sources to check program logic without a
GPU (tests/emu) -- that library is never loaded from here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes): every symbol of include/rustpde_hip.h
SIGNATURES = {
    "rpde_last_error": (C.c_char_p, []),
    "rpde_version": (C.c_char_p, []),
    "rpde_is_device_build": (C.c_int, []),
    "rpde_device_count": (C.c_int, [_ip]),
    "rpde_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "rpde_device_trim": (C.c_int, [C.c_int, C.POINTER(C.c_size_t)]),
    "rpde_arena_check": (C.c_int, [C.POINTER(C.c_long)]),
    "rpde_arena_selftest": (C.c_int, []),
    "rpde_navier2d_create_confined": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                C.c_double, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_navier2d_create_periodic": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                C.c_double, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_navier2d_create_sharded": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                               C.c_double, C.c_char_p, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                               C.POINTER(_vp)]),
    "rpde_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "rpde_rccl_alltoallv_once": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "rpde_navier2d_create_sharded_rccl": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                    C.c_double, C.c_double, C.c_char_p, C.c_int, C.c_int,
                                                    C.c_int, C.c_char_p, C.POINTER(_vp)]),
    "rpde_navier2d_comm_stats": (C.c_int, [_vp, _dp, _ip]),
    "rpde_navier2d_destroy": (C.c_int, [_vp]),
    "rpde_adjoint2d_create_confined": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_adjoint2d_create_periodic": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_adjoint2d_destroy": (C.c_int, [_vp]),
    "rpde_adjoint2d_set_velocity": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "rpde_adjoint2d_set_temperature": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "rpde_adjoint2d_reset_time": (C.c_int, [_vp]),
    "rpde_adjoint2d_spectral_shape": (C.c_int, [_vp, C.c_char_p, _ip, _ip, _ip]),
    "rpde_adjoint2d_set_field": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp, C.c_size_t]),
    "rpde_adjoint2d_get_field": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp, C.c_size_t]),
    "rpde_adjoint2d_update": (C.c_int, [_vp, C.c_int]),
    "rpde_adjoint2d_time": (C.c_int, [_vp, _dp]),
    "rpde_adjoint2d_dt": (C.c_int, [_vp, _dp]),
    "rpde_adjoint2d_param": (C.c_int, [_vp, C.c_char_p, _dp]),
    "rpde_adjoint2d_exit": (C.c_int, [_vp, _ip]),
    "rpde_adjoint2d_div_norm": (C.c_int, [_vp, _dp]),
    "rpde_adjoint2d_norm_residual": (C.c_int, [_vp, _dp]),
    "rpde_adjoint2d_write": (C.c_int, [_vp, C.c_char_p]),
    "rpde_adjoint2d_read": (C.c_int, [_vp, C.c_char_p]),
    "rpde_navier2d_create_confined_with_spectrum": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                                              C.c_char_p, C.c_int, _dp, C.c_size_t, C.POINTER(_vp)]),
    "rpde_poisson_x_spectrum": (C.c_int, [C.c_int, C.c_int, C.c_double, _dp, C.c_size_t]),
    "rpde_poisson_x_eigenbasis_from_spectrum": (C.c_int, [C.c_int, C.c_int, C.c_double, _dp, C.c_size_t, _dp, _dp, _dp]),
    "rpde_poisson_create_with_spectrum": (C.c_int, [_vp, C.c_double, C.c_double, _dp, C.c_size_t, C.POINTER(_vp)]),
    "rpde_lnse2d_create_confined": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                              C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_lnse2d_create_periodic": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                              C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_lnse2d_destroy": (C.c_int, [_vp]),
    "rpde_lnse2d_set_velocity": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "rpde_lnse2d_set_temperature": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "rpde_lnse2d_reset_time": (C.c_int, [_vp]),
    "rpde_lnse2d_spectral_shape": (C.c_int, [_vp, C.c_char_p, _ip, _ip, _ip]),
    "rpde_lnse2d_set_field": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp, C.c_size_t]),
    "rpde_lnse2d_get_field": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp, C.c_size_t]),
    "rpde_lnse2d_set_mean": (C.c_int, [_vp, C.c_char_p, _dp, C.c_size_t]),
    "rpde_lnse2d_get_mean": (C.c_int, [_vp, C.c_char_p, _dp, C.c_size_t]),
    "rpde_lnse2d_update": (C.c_int, [_vp, C.c_int]),
    "rpde_lnse2d_time": (C.c_int, [_vp, _dp]),
    "rpde_lnse2d_dt": (C.c_int, [_vp, _dp]),
    "rpde_lnse2d_param": (C.c_int, [_vp, C.c_char_p, _dp]),
    "rpde_lnse2d_exit": (C.c_int, [_vp, _ip]),
    "rpde_lnse2d_div_norm": (C.c_int, [_vp, _dp]),
    "rpde_lnse2d_write": (C.c_int, [_vp, C.c_char_p]),
    "rpde_lnse2d_read": (C.c_int, [_vp, C.c_char_p]),
    "rpde_nonlin2d_create_confined": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                                C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_nonlin2d_create_periodic": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_char_p,
                                                C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "rpde_lnse2d_update_direct": (C.c_int, [_vp, C.c_int]),
    "rpde_lnse2d_history_len": (C.c_int, [_vp, C.POINTER(C.c_long)]),
    "rpde_lnse2d_clear_history": (C.c_int, [_vp]),
    "rpde_lnse2d_update_adjoint": (C.c_int, [_vp, C.c_int]),
    "rpde_lnse2d_integrate": (C.c_int, [_vp, C.c_double, C.POINTER(C.c_long)]),
    "rpde_lnse2d_energy": (C.c_int, [_vp, C.c_double, C.c_double, _dp, _dp, _dp, C.c_size_t, _dp]),
    "rpde_lnse2d_grad_adjoint": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, _dp, _dp, _dp, C.c_size_t, C.c_char_p,
                                           _dp, _dp, _dp, _dp, C.POINTER(C.c_long)]),
    "rpde_lnse2d_callback_from_filename": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_int, C.c_double]),
    "rpde_lnse2d_diagnostics": (C.c_int, [_vp, _dp]),
    "rpde_lnse2d_grad_fd": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, _ip, C.c_long, C.c_size_t, C.c_char_p, _dp, _dp, _dp]),
    "rpde_lnse2d_grad_fd_save": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double, _ip, C.c_long, C.c_size_t, C.c_char_p, _dp, _dp, _dp]),
    "rpde_l2_norm": (C.c_int, [C.c_size_t, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double, C.c_double, _dp]),
    "rpde_steepest_descent_energy_constrained": (C.c_int, [C.c_size_t, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_double,
                                                           C.c_double, C.c_double]),
    "rpde_hholtz_create": (C.c_int, [_vp, C.c_double, C.c_double, C.POINTER(_vp)]),
    "rpde_hholtz_solve": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_hholtz_destroy": (C.c_int, [_vp]),
    "rpde_navier2d_set_velocity": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "rpde_navier2d_set_temperature": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double]),
    "rpde_navier2d_init_random": (C.c_int, [_vp, C.c_double, C.c_uint64]),
    "rpde_navier2d_reset_time": (C.c_int, [_vp]),
    "rpde_navier2d_spectral_shape": (C.c_int, [_vp, C.c_char_p, _ip, _ip, _ip]),
    "rpde_navier2d_set_field": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp, C.c_size_t]),
    "rpde_navier2d_get_field": (C.c_int, [_vp, C.c_char_p, C.c_int, _dp, C.c_size_t]),
    "rpde_navier2d_get_grid": (C.c_int, [_vp, C.c_int, _dp, C.c_size_t]),
    "rpde_navier2d_update": (C.c_int, [_vp, C.c_int]),
    "rpde_navier2d_last_update_ms": (C.c_int, [_vp, _dp]),
    "rpde_navier2d_profile": (C.c_int, [_vp, C.c_int, C.c_char_p, C.c_size_t]),
    "rpde_navier2d_describe_step": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "rpde_navier2d_trace_launch": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_size_t]),
    "rpde_navier2d_set_timed_tag": (C.c_int, [_vp, C.c_char_p]),
    "rpde_navier2d_get_timed": (C.c_int, [_vp, _dp, C.POINTER(C.c_long)]),
    "rpde_navier2d_time": (C.c_int, [_vp, _dp]),
    "rpde_navier2d_dt": (C.c_int, [_vp, _dp]),
    "rpde_navier2d_param": (C.c_int, [_vp, C.c_char_p, _dp]),
    "rpde_navier2d_exit": (C.c_int, [_vp, _ip]),
    "rpde_navier2d_div_norm": (C.c_int, [_vp, _dp]),
    "rpde_navier2d_diagnostics": (C.c_int, [_vp, _dp, _dp, _dp]),
    "rpde_navier2d_write": (C.c_int, [_vp, C.c_char_p]),
    "rpde_navier2d_read": (C.c_int, [_vp, C.c_char_p]),
    "rpde_navier2d_set_write_intervall": (C.c_int, [_vp, C.c_double]),
    "rpde_navier2d_callback": (C.c_int, [_vp]),
    "rpde_navier2d_callback_from_filename": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_int, C.c_double]),
    "rpde_navier2d_statistics_enable": (C.c_int, [_vp, C.c_double, C.c_double]),
    "rpde_navier2d_statistics_attach": (C.c_int, [_vp, C.c_int]),
    "rpde_navier2d_statistics_update": (C.c_int, [_vp]),
    "rpde_navier2d_statistics_write": (C.c_int, [_vp, C.c_char_p]),
    "rpde_navier2d_statistics_read": (C.c_int, [_vp, C.c_char_p]),
    "rpde_navier2d_statistics_get": (C.c_int, [_vp, C.c_char_p, _dp, C.c_size_t]),
    "rpde_navier2d_statistics_scalars": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_longlong)]),
    "rpde_h5_shape": (C.c_int, [C.c_char_p, C.c_char_p, _ip, C.POINTER(C.c_uint64)]),
    "rpde_h5_read": (C.c_int, [C.c_char_p, C.c_char_p, _dp, C.c_size_t]),
    "rpde_h5_write": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), _dp]),
    "rpde_h5_list": (C.c_int, [C.c_char_p, C.c_char_p, C.c_size_t]),
    "rpde_navier2d_integrate": (C.c_int, [_vp, C.c_double, C.c_int, C.POINTER(C.c_long)]),
    "rpde_space2_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "rpde_space2_destroy": (C.c_int, [_vp]),
    "rpde_space2_shape": (C.c_int, [_vp, C.c_int, _ip, _ip, _ip]),
    "rpde_space2_forward": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_space2_backward": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_space2_to_ortho": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_space2_from_ortho": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_space2_gradient": (C.c_int, [_vp, _dp, C.c_size_t, C.c_int, C.c_int, C.c_double, C.c_double,
                                       _dp, C.c_size_t]),
    "rpde_hholtz_adi_create": (C.c_int, [_vp, C.c_double, C.c_double, C.POINTER(_vp)]),
    "rpde_hholtz_adi_solve": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_hholtz_adi_destroy": (C.c_int, [_vp]),
    "rpde_poisson_create": (C.c_int, [_vp, C.c_double, C.c_double, C.POINTER(_vp)]),
    "rpde_poisson_solve": (C.c_int, [_vp, _dp, C.c_size_t, _dp, C.c_size_t]),
    "rpde_poisson_destroy": (C.c_int, [_vp]),
    "rpde_poisson_eigenbasis": (C.c_int, [_vp, _dp, _dp, _dp, C.c_size_t]),
    "rpde_navier2d_poisson_eigenbasis": (C.c_int, [_vp, _dp, _dp, _dp, C.c_size_t]),
    "rpde_transpose": (C.c_int, [_dp, C.c_int, C.c_int, C.c_int, _dp, C.c_int]),
    "rpde_dct_line_backward": (C.c_int, [C.c_int, C.c_int, _dp, C.c_int, _dp, C.c_int]),
    "rpde_dct_line_gradient": (C.c_int, [C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp, C.c_int]),
    "rpde_dct_line_forward": (C.c_int, [C.c_int, _dp, C.c_int, C.c_int, _dp, C.c_int]),
    "rpde_conv_line": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_int, _dp, C.c_int]),
    "rpde_gemm": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, C.c_int, _dp, C.c_int]),
    "rpde_microbench": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp]),
}


ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, _vp, _vp, C.POINTER(C.c_int64), _vp, C.POINTER(C.c_int64))


class RpdeError(RuntimeError):
    """Raised for every non-zero status of the C ABI (the reference panics in these places)."""


class Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise RpdeError(
                f"native library not found: {path}\n"
                "build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
        self.path = path
        self.dll = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.dll, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args

    def call(self, name, *args):
        rc = getattr(self.dll, name)(*args)
        if rc != 0:
            raise RpdeError(self.dll.rpde_last_error().decode())

    @property
    def version(self) -> str:
        return self.dll.rpde_version().decode()

    @property
    def is_device_build(self) -> bool:
        return bool(self.dll.rpde_is_device_build())


def as_f64(a, shape=None):
    """Contiguous float64 view of a real or complex array (complex -> interleaved pairs)."""
    a = np.asarray(a)
    if np.iscomplexobj(a):
        a = np.ascontiguousarray(a, dtype=np.complex128).view(np.float64)
    else:
        a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def ptr(a):
    return a.ctypes.data_as(_dp)
