"""krylov.jl_amd -- host-side mirror of Krylov.jl's workspace / solver / operator API over the
MI355X-native C ABI (include/krylov_hip.h, libkrylov_hip.so).

The directory name contains a dot, so import it through the repo-root shim::

    import krylov_jl_amd as K

Naming follows the reference with Python's trailing-underscore convention for Julia's `!`:
`cg!` -> `cg_`, `kaxpy!` -> `kaxpy_`, ...  (src/krylov_utils.jl:305-349, src/cg.jl:120,
src/gmres.jl:121, src/bicgstab.jl:125, src/block_gmres.jl:110).

There is NO CPU fallback: every call goes to hand-written gfx950 kernels; loading fails loudly if
the extension is missing and context creation fails loudly if no GPU is visible.

PyTorch is optional plumbing.  If it is used in the same process (torch.distributed launchers),
import torch BEFORE this package so that both share one HIP runtime (same SONAME libamdhip64.so.7).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KHIP_LIBRARY") or os.path.join(_HERE, "libkrylov_hip.so")

c_double_p = C.POINTER(C.c_double)
c_void_pp = C.POINTER(C.c_void_p)

APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
CALLBACK_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
GROW_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p)      # khip_grow_fn: push!(V, similar(x)) of an adopted workspace


class KhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libkrylov_hip error {code}: {msg}")
        self.code = code


class COperator(C.Structure):
    _fields_ = [("csr", C.c_void_p), ("apply", APPLY_FN), ("self", C.c_void_p)]


class COptions(C.Structure):
    _fields_ = [("atol", C.c_double), ("rtol", C.c_double), ("itmax", C.c_int), ("timemax", C.c_double),
                ("history", C.c_int), ("radius", C.c_double), ("linesearch", C.c_int), ("restart", C.c_int),
                ("reorthogonalization", C.c_int), ("fused", C.c_int), ("callback", CALLBACK_FN),
                ("callback_data", C.c_void_p), ("variant", C.c_int), ("verbose", C.c_int), ("log_fd", C.c_int)]


class CStats(C.Structure):
    _fields_ = [("niter", C.c_int), ("solved", C.c_int), ("inconsistent", C.c_int), ("indefinite", C.c_int),
                ("npcCount", C.c_int), ("timer", C.c_double), ("status", C.c_char * 96),
                ("residuals", c_double_p), ("nres", C.c_int), ("error", C.c_char * 160), ("allocation_timer", C.c_double)]


# every symbol include/krylov_hip.h declares: name -> (restype, argtypes)
_i64, _int, _dbl, _vp, _sz = C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_size_t
SIGNATURES = {
    "khip_last_error": (C.c_char_p, []),
    "khip_version": (None, [C.POINTER(_int), C.POINTER(_int)]),
    "khip_ctx_create": (_int, [_int, _vp, c_void_pp]),
    "khip_ctx_destroy": (_int, [_vp]),
    "khip_ctx_sync": (_int, [_vp]),
    "khip_ctx_stream": (_vp, [_vp]),
    "khip_ctx_set_option": (_int, [_vp, C.c_char_p, _int]),
    "khip_ctx_get_option": (_int, [_vp, C.c_char_p, C.POINTER(_int)]),
    "khip_malloc": (_int, [_vp, _sz, c_void_pp]),
    "khip_free": (_int, [_vp, _vp]),
    "khip_memcpy_h2d": (_int, [_vp, _vp, _vp, _sz]),
    "khip_memcpy_d2h": (_int, [_vp, _vp, _vp, _sz]),
    "khip_memcpy_d2d": (_int, [_vp, _vp, _vp, _sz]),
    "khip_mem_info": (_int, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "khip_csr_create": (_int, [_vp, _i64, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, c_void_pp]),
    "khip_csr_create_dist": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _int, _vp, _vp, _int, _int, c_void_pp]),
    "khip_csr_destroy": (_int, [_vp]),
    "khip_csr_transpose": (_int, [_vp, _vp, c_void_pp]),
    "khip_csr_compress": (_int, [_vp, _vp, C.POINTER(_int)]),
    "khip_spmv_bytes_stored": (_int, [_vp, C.POINTER(_i64)]),
    "khip_csr_shape": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "khip_gen_stencil": (_int, [_vp, _int, _int, _int, _int, _i64, _i64, c_void_pp, c_void_pp, c_void_pp,
                                C.POINTER(_i64)]),
    "khip_gen_banded_random": (_int, [_vp, _i64, _int, _int, C.c_uint64, _int, _int, _i64, _i64, c_void_pp, c_void_pp, c_void_pp,
                                      C.POINTER(_i64)]),
    "khip_spmv": (_int, [_vp, _vp, _vp, _vp]),
    "khip_spmm": (_int, [_vp, _vp, _vp, _vp, _int]),
    "khip_spmv_bytes": (_int, [_vp, C.POINTER(_i64)]),
    "khip_csr_code_info": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "khip_csr_sell_info": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64)]),
    "khip_csr_sell32_info": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64)]),
    "khip_csr_sell_narrow": (_int, [_vp, C.POINTER(C.c_int)]),
    "khip_spmv_kernel_info": (_int, [_vp, _vp, C.POINTER(C.c_int)]),
    "khip_csr_delta_info": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_i64)]),
    "khip_csr_tile_info": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "khip_csr_halo_info": (_int, [_vp, C.POINTER(C.c_int), C.POINTER(_i64), C.POINTER(_i64)]),
    "khip_profile_spmv": (_int, [_vp, C.POINTER(_i64), C.POINTER(_dbl)]),
    "khip_profile_kernels": (_int, [_vp, _int, C.POINTER(_i64), C.POINTER(_dbl)]),
    "khip_dot": (_int, [_vp, _i64, _vp, _vp, c_double_p]),
    "khip_nrm2": (_int, [_vp, _i64, _vp, c_double_p]),
    "khip_scal": (_int, [_vp, _i64, _dbl, _vp]),
    "khip_div": (_int, [_vp, _i64, _vp, _dbl]),
    "khip_copy": (_int, [_vp, _i64, _vp, _vp]),
    "khip_scalcopy": (_int, [_vp, _i64, _vp, _dbl, _vp]),
    "khip_divcopy": (_int, [_vp, _i64, _vp, _vp, _dbl]),
    "khip_axpy": (_int, [_vp, _i64, _dbl, _vp, _vp]),
    "khip_axpby": (_int, [_vp, _i64, _dbl, _vp, _dbl, _vp]),
    "khip_fill": (_int, [_vp, _i64, _vp, _dbl]),
    "khip_ref": (_int, [_vp, _i64, _vp, _vp, _dbl, _dbl]),
    "khip_vmul": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "khip_vdiv": (_int, [_vp, _i64, _vp, _vp, _vp]),
    "khip_csr_diagonal": (_int, [_vp, _vp, _vp]),
    "khip_jacobi_create": (_int, [_vp, _vp, C.POINTER(COperator)]),
    "khip_jacobi_destroy": (_int, [C.POINTER(COperator)]),
    "khip_ilu0_create": (_int, [_vp, _vp, C.POINTER(COperator)]),
    "khip_ilu0_destroy": (_int, [C.POINTER(COperator)]),
    "khip_ilu0_info": (_int, [C.POINTER(COperator), C.POINTER(_i64), C.POINTER(_i64), c_void_pp]),
    "khip_ilu0_set_graph": (_int, [C.POINTER(COperator), _int]),
    "khip_ilu0_block_info": (_int, [C.POINTER(COperator), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_int)]),
    "khip_test_ilu_blocks_host": (_int, [_i64, C.POINTER(_i64), C.POINTER(C.c_int32), _int, C.POINTER(_i64)]),
    "khip_spmv_dot": (_int, [_vp, _vp, _vp, _vp, c_double_p]),
    "khip_axpy2_dot": (_int, [_vp, _i64, _dbl, _vp, _vp, _vp, _vp, c_double_p]),
    "khip_waxpy": (_int, [_vp, _i64, _vp, _vp, _dbl, _vp]),
    "khip_axpy_sqnorm": (_int, [_vp, _i64, _dbl, _vp, _vp, c_double_p]),
    "khip_cg_setup": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, c_double_p]),
    "khip_cg_update": (_int, [_vp, _i64, _dbl, _dbl, _vp, _vp, _vp]),
    "khip_spmv_dotw": (_int, [_vp, _vp, _vp, _vp, _vp, c_double_p]),
    "khip_spmv_dot2": (_int, [_vp, _vp, _vp, _vp, c_double_p]),
    "khip_bicgstab_sx": (_int, [_vp, _i64, _dbl, _vp, _vp, _vp, _vp, _vp]),
    "khip_bicgstab_xr": (_int, [_vp, _i64, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, c_double_p]),
    "khip_bicgstab_p": (_int, [_vp, _i64, _dbl, _dbl, _vp, _vp, _vp]),
    "khip_dot2": (_int, [_vp, _i64, _vp, _vp, c_double_p]),
    "khip_mgs": (_int, [_vp, _i64, _int, c_void_pp, _vp, c_double_p, c_double_p, _int]),
    "khip_hermitian_lanczos": (_int, [_vp, C.POINTER(COperator), _i64, _vp, _int, _int, _int, _vp, _i64, c_double_p, c_double_p]),
    "khip_arnoldi": (_int, [_vp, C.POINTER(COperator), _i64, _vp, _int, _int, _int, _vp, _i64, c_double_p, c_double_p]),
    "khip_golub_kahan": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), _i64, _i64, _vp, _int, _int, _vp, _i64, _vp, _i64,
                                c_double_p, c_double_p]),
    "khip_nonhermitian_lanczos": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), _i64, _vp, _vp, _int, _int, _vp, _i64, _vp, _i64,
                                         c_double_p, c_double_p, c_double_p, c_double_p]),
    "khip_saunders_simon_yip": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), _i64, _i64, _vp, _vp, _int, _int, _vp, _i64, _vp, _i64,
                                       c_double_p, c_double_p, c_double_p, c_double_p]),
    "khip_montoison_orban": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), _i64, _i64, _vp, _vp, _int, _int, _int, _vp, _i64, _vp, _i64,
                                    c_double_p, c_double_p, c_double_p, c_double_p]),
    "khip_multi_axpy": (_int, [_vp, _i64, _int, c_double_p, c_void_pp, _vp]),
    "khip_panel_rows": (_int, [_i64, C.POINTER(_i64)]),
    "khip_panel_from_colmajor": (_int, [_vp, _i64, _int, _vp, _vp]),
    "khip_panel_to_colmajor": (_int, [_vp, _i64, _int, _vp, _vp]),
    "khip_panel_gemm_tn": (_int, [_vp, _i64, _int, _vp, _vp, c_double_p]),
    "khip_panel_gemm_nn": (_int, [_vp, _i64, _int, _dbl, _vp, c_double_p, _dbl, _vp]),
    "khip_panel_mgs": (_int, [_vp, _i64, _int, _int, c_void_pp, _vp, c_double_p, _int]),
    "khip_panel_qr": (_int, [_vp, _i64, _int, _vp, c_double_p]),
    "khip_panel_qr_tau": (_int, [_vp, _i64, _int, _vp, c_double_p, c_double_p]),
    "khip_panel_multi_nn": (_int, [_vp, _i64, _int, _int, C.POINTER(_vp), c_double_p, _dbl, _vp]),
    "khip_panel_norm": (_int, [_vp, _i64, _int, _vp, c_double_p]),
    "khip_comm_unique_id": (_int, [_vp]),
    "khip_comm_init": (_int, [_vp, _int, _int, _vp]),
    "khip_comm_init_local": (_int, [_vp, _int, _int, _int]),
    "khip_comm_rank": (_int, [_vp, C.POINTER(_int), C.POINTER(_int)]),
    "khip_comm_barrier": (_int, [_vp]),
    "khip_device_count": (_int, [C.POINTER(_int)]),
    "khip_device_pci_id": (_int, [_int, C.c_char_p, _sz]),
    "khip_comm_info": (_int, [_vp] + [C.POINTER(_int)] * 5),
    "khip_default_options": (COptions, []),
    "khip_cg_workspace_create": (_int, [_vp, _i64, _i64, c_void_pp]),
    "khip_cg_workspace_adopt": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, c_void_pp]),
    "khip_cg_workspace_adopt_vector": (_int, [_vp, C.c_char_p, _vp]),
    "khip_cg_workspace_destroy": (_int, [_vp]),
    "khip_cg_warm_start": (_int, [_vp, _vp]),
    "khip_cg_solve": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), _vp, C.POINTER(COptions)]),
    "khip_cg_solution": (_vp, [_vp]),
    "khip_cg_stats": (C.POINTER(CStats), [_vp]),
    "khip_cg_last_path": (_int, [_vp]),
    "khip_cg_vector": (_vp, [_vp, C.c_char_p]),
    "khip_cg_workspace_bytes": (_sz, [_vp]),
    "khip_gmres_workspace_create": (_int, [_vp, _i64, _i64, _int, c_void_pp]),
    "khip_gmres_workspace_adopt": (_int, [_vp, _i64, _i64, _int, _vp, _vp, c_void_pp, c_void_pp]),
    "khip_gmres_workspace_adopt_vector": (_int, [_vp, C.c_char_p, _vp]),
    "khip_gmres_workspace_adopt_basis": (_int, [_vp, _int, c_void_pp]),
    "khip_gmres_workspace_set_grow": (_int, [_vp, GROW_FN, _vp]),
    "khip_gmres_host_state": (_int, [_vp, _int, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(_int), C.POINTER(_int)]),
    "khip_gmres_workspace_destroy": (_int, [_vp]),
    "khip_gmres_warm_start": (_int, [_vp, _vp]),
    "khip_gmres_solve": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), C.POINTER(COperator), _vp,
                                C.POINTER(COptions)]),
    "khip_gmres_solution": (_vp, [_vp]),
    "khip_gmres_stats": (C.POINTER(CStats), [_vp]),
    "khip_gmres_last_path": (_int, [_vp]),
    "khip_gmres_workspace_bytes": (_sz, [_vp]),
    "khip_bicgstab_workspace_create": (_int, [_vp, _i64, _i64, c_void_pp]),
    "khip_bicgstab_workspace_adopt": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, c_void_pp]),
    "khip_bicgstab_workspace_adopt_vector": (_int, [_vp, C.c_char_p, _vp]),
    "khip_bicgstab_workspace_destroy": (_int, [_vp]),
    "khip_bicgstab_warm_start": (_int, [_vp, _vp]),
    "khip_bicgstab_solve": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), C.POINTER(COperator), _vp, _vp,
                                   C.POINTER(COptions)]),
    "khip_bicgstab_solution": (_vp, [_vp]),
    "khip_bicgstab_stats": (C.POINTER(CStats), [_vp]),
    "khip_bicgstab_last_path": (_int, [_vp]),
    "khip_bicgstab_workspace_bytes": (_sz, [_vp]),
    "khip_block_gmres_workspace_bytes": (_sz, [_vp, C.POINTER(C.c_size_t)]),
    "khip_test_gen_banded_random_host": (_int, [_i64, _int, _int, C.c_uint64, _int, _int, _i64, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_double_p, C.POINTER(_i64)]),
    "khip_test_small_dense": (_int, [_int, _int, _int, _int, c_double_p, c_double_p, c_double_p]),
    "khip_test_deflating_chol": (_int, [_int, c_double_p, C.c_double, _int, C.c_uint, c_double_p, C.POINTER(_int), C.POINTER(C.c_uint)]),
    "khip_test_householder_r": (_int, [_int, _int, c_double_p, c_double_p]),
    "khip_test_householder_signs": (_int, [_int, _i64, c_double_p, c_double_p, c_double_p]),
    "khip_test_optional_build_failures": (_int, [C.POINTER(_int)]),
    "khip_test_set_halo_self": (_int, [_vp, _int]),
    "khip_test_sym_givens": (_int, [C.c_double, C.c_double, c_double_p, c_double_p, c_double_p]),
    "khip_test_roots_quadratic": (_int, [C.c_double, C.c_double, C.c_double, _int, c_double_p, c_double_p]),
    "khip_test_to_boundary": (_int, [_vp, _i64, _vp, _vp, C.c_double, _int, c_double_p, c_double_p]),
    "khip_block_gmres_workspace_create": (_int, [_vp, _i64, _i64, _int, _int, c_void_pp]),
    "khip_block_gmres_workspace_adopt": (_int, [_vp, _i64, _i64, _int, _int, _vp, _vp, c_void_pp, c_void_pp]),
    "khip_block_gmres_workspace_adopt_panel": (_int, [_vp, C.c_char_p, _vp]),
    "khip_block_gmres_workspace_adopt_basis": (_int, [_vp, _int, c_void_pp]),
    "khip_block_gmres_workspace_set_grow": (_int, [_vp, GROW_FN, _vp]),
    "khip_block_gmres_warm_start_panel": (_int, [_vp, _vp]),
    "khip_block_gmres_solve_panel": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), C.POINTER(COperator), _vp,
                                     C.POINTER(COptions)]),
    "khip_block_gmres_workspace_destroy": (_int, [_vp]),
    "khip_block_gmres_warm_start": (_int, [_vp, _vp]),
    "khip_block_gmres_solve": (_int, [_vp, C.POINTER(COperator), C.POINTER(COperator), C.POINTER(COperator), _vp,
                               C.POINTER(COptions)]),
    "khip_block_gmres_get_X": (_int, [_vp, _vp]),
    "khip_block_gmres_stats": (C.POINTER(CStats), [_vp]),
    "khip_block_gmres_last_path": (_int, [_vp]),
    # host-only helpers (partition / halo plan logic, testable without a GPU)
    "khip_ghost_columns_host": (_int, [_vp, _vp, _i64, _i64, _vp, _i64, C.POINTER(_i64)]),
    "khip_halo_plan_host": (_int, [_int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libkrylov_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    script = os.path.join(_HERE, "build.sh")
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs.append(os.path.join(_HERE, "..", "include", "krylov_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if force or stale:
        subprocess.check_call(["bash", script])
    return LIB_PATH


def lib():
    """Load the native library (fails loudly when it is missing: there is no fallback path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); krylov.jl_amd has no CPU fallback")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)      # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    # khip_options / khip_stats are passed by pointer without a size field: a binding must have been written against the ABI
    # version the library reports; this mirror follows include/krylov_hip.h of version 0.4 (adopt entries, log_fd, last_path)
    major, minor = C.c_int(), C.c_int()
    L.khip_version(C.byref(major), C.byref(minor))
    if (major.value, minor.value) != (0, 4):
        raise ImportError(f"{LIB_PATH} reports ABI {major.value}.{minor.value}; krylov.jl_amd/__init__.py binds 0.4: rebuild (build.sh)")
    _lib = L
    return L


def _ck(rc):
    if rc != 0:
        raise KhipError(rc, lib().khip_last_error().decode("utf-8", "replace"))


def gpu_available() -> bool:
    return os.path.exists("/dev/kfd")


def device_count() -> int:
    n = C.c_int()
    lib().khip_device_count(C.byref(n))
    return n.value


def device_pci_id(device: int = 0) -> str:
    """PCI bus id of the physical GPU behind a visible device index."""
    buf = C.create_string_buffer(64)
    _ck(lib().khip_device_pci_id(device, buf, 64))
    return buf.value.decode()


# --------------------------------------------------------------------------- context / vectors

class Context:
    """Device + HIP stream + reduction scratch (+ RCCL communicator when distributed)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        _ck(lib().khip_ctx_create(device, stream, C.byref(self._h)))
        self.device = device
        self.rank, self.nranks = 0, 1

    def close(self):
        if self._h:
            lib().khip_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _ck(lib().khip_ctx_sync(self._h))

    def set_option(self, key: str, value: int):
        _ck(lib().khip_ctx_set_option(self._h, key.encode(), int(value)))

    def test_set_halo_self(self, enable: int = 1):
        """TEST-ONLY measurement hook (include/krylov_hip_test.h): a one-rank RCCL communicator exchanges its halo with itself."""
        _ck(lib().khip_test_set_halo_self(self._h, int(enable)))

    def get_option(self, key: str) -> int:
        v = C.c_int()
        _ck(lib().khip_ctx_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def mem_info(self):
        f, t = C.c_size_t(), C.c_size_t()
        _ck(lib().khip_mem_info(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    @property
    def stream(self):
        return lib().khip_ctx_stream(self._h)

    def profile_spmv(self):
        """(launches, total_ms) of the SpMV launches recorded since the last call (option profile_spmv=1)."""
        n, ms = C.c_int64(), C.c_double()
        _ck(lib().khip_profile_spmv(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    PROFILE_TAGS = ("spmv", "spmm", "panel_gemm_tn", "panel_nn_tn", "panel_multi_nn", "panel_gemm_nn", "panel_qr_scale_gram",
                    "halo_pack", "halo_transfer", "dot_allgather_combine", "spmv_boundary")

    def profile_kernels(self):
        """{family: (launches, total_ms)} of the HIP-event brackets recorded since the last call (option profile_spmv = 1):
        khip_profile_kernels -- SpMV, SpMM, and the panel kernels of block_gmres!."""
        k = len(self.PROFILE_TAGS)
        n, ms = (C.c_int64 * k)(), (C.c_double * k)()
        _ck(lib().khip_profile_kernels(self._h, k, n, ms))
        return {t: (int(n[i]), float(ms[i])) for i, t in enumerate(self.PROFILE_TAGS)}

    # --- multi-GPU (one process per GPU) ---
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _ck(lib().khip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank: int, nranks: int, unique_id: bytes):
        assert len(unique_id) == 128
        buf = C.create_string_buffer(unique_id, 128)
        _ck(lib().khip_comm_init(self._h, rank, nranks, buf))
        self.rank, self.nranks = rank, nranks

    def comm_init_local(self, rank: int, nranks: int, hub_id: int = 0):
        """In-process communicator: the ranks are contexts of this process, one host thread each."""
        _ck(lib().khip_comm_init_local(self._h, rank, nranks, hub_id))
        self.rank, self.nranks = rank, nranks

    def comm_info(self) -> dict:
        v = [C.c_int() for _ in range(5)]
        _ck(lib().khip_comm_info(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("rank", "nranks", "rccl_ranks", "local_backend", "halo_comm_separate"), (x.value for x in v)))

    def barrier(self):
        _ck(lib().khip_comm_barrier(self._h))

    # --- allocation helpers ---
    def empty(self, n: int) -> "DeviceVector":
        return DeviceVector(self, n)

    def zeros(self, n: int) -> "DeviceVector":
        v = DeviceVector(self, n)
        kfill_(v, 0.0)
        return v

    def array(self, host) -> "DeviceVector":
        a = np.ascontiguousarray(host, dtype=np.float64).ravel()
        v = DeviceVector(self, a.size)
        v.copy_from_host(a)
        return v


class DeviceVector:
    """Float64 vector in HBM: the storage type `S` of the workspaces (`S(undef, n)`, `similar`,
    `length`; docs/src/custom_workspaces.md:107)."""

    def __init__(self, ctx: Context, n: int, ptr: int | None = None, owner=None):
        self.ctx, self.n = ctx, int(n)
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            # pad to a multiple of 2 doubles so 16-byte lane accesses never straddle the allocation
            _ck(lib().khip_malloc(ctx._h, 8 * max(2, (self.n + 1) & ~1), C.byref(p)))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = ptr
            self._owned = False

    def __len__(self):
        return self.n

    def similar(self):
        return DeviceVector(self.ctx, self.n)

    def slice(self, lo: int, hi: int) -> "DeviceVector":
        return DeviceVector(self.ctx, hi - lo, ptr=self.ptr + 8 * lo, owner=self)

    def copy_from_host(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).ravel()
        assert a.size == self.n
        _ck(lib().khip_memcpy_h2d(self.ctx._h, self.ptr, a.ctypes.data, 8 * self.n))
        return self

    def to_host(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        _ck(lib().khip_memcpy_d2h(self.ctx._h, out.ctypes.data, self.ptr, 8 * self.n))
        return out

    def __del__(self):
        try:
            if self._owned and self.ptr and self.ctx._h:
                lib().khip_free(self.ctx._h, self.ptr)
        except Exception:
            pass


def _p(v):
    """device pointer of a DeviceVector / raw int / object with data_ptr() (torch tensor)."""
    if v is None:
        return None
    if isinstance(v, DeviceVector):
        return v.ptr
    if isinstance(v, int):
        return v
    if hasattr(v, "data_ptr"):
        return v.data_ptr()
    raise TypeError(f"not a device buffer: {type(v)}")


# --------------------------------------------------------------------------- k* primitives
# Signatures follow src/krylov_utils.jl:305-349 (n first, scalars as T, return the mutated vector).

def kdot(n, x, y) -> float:
    r = C.c_double()
    _ck(lib().khip_dot(x.ctx._h, n, _p(x), _p(y), C.byref(r)))
    return r.value


kdotr = kdot    # real part; identical for Float64 (src/krylov_utils.jl:313-314)


def knorm(n, x) -> float:
    r = C.c_double()
    _ck(lib().khip_nrm2(x.ctx._h, n, _p(x), C.byref(r)))
    return r.value


def knorm_elliptic(n, x, y) -> float:   # src/krylov_utils.jl:319
    return knorm(n, x) if x is y else math.sqrt(kdotr(n, x, y))


def kscal_(n, s, x):
    _ck(lib().khip_scal(x.ctx._h, n, s, _p(x)))
    return x


def kdiv_(n, x, s):
    _ck(lib().khip_div(x.ctx._h, n, _p(x), s))
    return x


def kcopy_(n, y, x):
    _ck(lib().khip_copy(y.ctx._h, n, _p(y), _p(x)))
    return y


def kscalcopy_(n, y, s, x):
    _ck(lib().khip_scalcopy(y.ctx._h, n, _p(y), s, _p(x)))
    return y


def kdivcopy_(n, y, x, s):
    _ck(lib().khip_divcopy(y.ctx._h, n, _p(y), _p(x), s))
    return y


def kaxpy_(n, s, x, y):
    _ck(lib().khip_axpy(y.ctx._h, n, s, _p(x), _p(y)))
    return y


def kaxpby_(n, s, x, t, y):
    _ck(lib().khip_axpby(y.ctx._h, n, s, _p(x), t, _p(y)))
    return y


def kfill_(x, val):
    _ck(lib().khip_fill(x.ctx._h, x.n, _p(x), val))
    return x


def kref_(n, x, y, c, s):
    _ck(lib().khip_ref(x.ctx._h, n, _p(x), _p(y), c, s))
    return x, y


def kmul_(y, A, x):
    """kmul!(y, A, x) = mul!(y, A, x) (src/krylov_utils.jl:305); A: CsrMatrix, a callable operator, or None (= I,
    the unguarded mul!(v, I, q) of src/bicgstab.jl:222)."""
    if A is None:
        return kcopy_(len(x), y, x)
    if isinstance(A, CsrMatrix):
        _ck(lib().khip_spmv(A.ctx._h, A._h, _p(x), _p(y)))
        return y
    A(x, y)
    return y


def kvmul_(n, w, x, y):
    """w = x .* y (diagonal operator)."""
    _ck(lib().khip_vmul(w.ctx._h, n, _p(w), _p(x), _p(y)))
    return w


def kvdiv_(n, w, x, y):
    """w = x ./ y (Jacobi: z = r ./ diag(A))."""
    _ck(lib().khip_vdiv(w.ctx._h, n, _p(w), _p(x), _p(y)))
    return w


class Jacobi:
    """M = Diagonal(diag(A))^-1 on the device, usable as the M / N argument of cg_ / gmres_ / bicgstab_
    (the reference's Jacobi examples: test/test_gmres.jl:105-128, docs/src/gpu.md)."""

    def __init__(self, A: "CsrMatrix"):
        self.ctx, self.n = A.ctx, A.m
        self.op = COperator()
        _ck(lib().khip_jacobi_create(A.ctx._h, A._h, C.byref(self.op)))

    def __call__(self, x, y):
        return self._apply(x, y)

    def _apply(self, x, y):
        rc = self.op.apply(self.op.self, _p(x), _p(y))
        if rc:
            raise KhipError(rc, lib().khip_last_error().decode())
        return y

    def __del__(self):
        try:
            if self.ctx._h:
                lib().khip_jacobi_destroy(C.byref(self.op))
        except Exception:
            pass


# fused accelerators
def spmv_dot(A, x, y) -> float:
    r = C.c_double()
    _ck(lib().khip_spmv_dot(A.ctx._h, A._h, _p(x), _p(y), C.byref(r)))
    return r.value


def axpy2_dot(n, a, p, q, x, r) -> float:
    out = C.c_double()
    _ck(lib().khip_axpy2_dot(x.ctx._h, n, a, _p(p), _p(q), _p(x), _p(r), C.byref(out)))
    return out.value


def spmv_dotw(A, x, y, w) -> float:
    """y = A x ; returns w . y (src/bicgstab.jl:221-223)."""
    r = C.c_double()
    _ck(lib().khip_spmv_dotw(A.ctx._h, A._h, _p(x), _p(y), _p(w), C.byref(r)))
    return r.value


def spmv_dot2(A, x, y):
    """y = A x ; returns (x . y, y . y) (src/bicgstab.jl:228-230)."""
    out = (C.c_double * 2)()
    _ck(lib().khip_spmv_dot2(A.ctx._h, A._h, _p(x), _p(y), out))
    return out[0], out[1]


def bicgstab_sx_(n, alpha, r, v, y, s, x):
    _ck(lib().khip_bicgstab_sx(x.ctx._h, n, alpha, _p(r), _p(v), _p(y), _p(s), _p(x)))


def bicgstab_xr_(n, omega, s, t, z, c, x, r):
    out = (C.c_double * 2)()
    _ck(lib().khip_bicgstab_xr(x.ctx._h, n, omega, _p(s), _p(t), _p(z), _p(c), _p(x), _p(r), out))
    return out[0], out[1]


def bicgstab_p_(n, omega, beta, v, r, p):
    _ck(lib().khip_bicgstab_p(p.ctx._h, n, omega, beta, _p(v), _p(r), _p(p)))


def axpy_sqnorm(n, a, x, y) -> float:
    """y += a x ; returns y . y (src/cg.jl:240,242)."""
    out = C.c_double()
    _ck(lib().khip_axpy_sqnorm(y.ctx._h, n, a, _p(x), _p(y), C.byref(out)))
    return out.value


def cg_setup_(n, b, x, r, p) -> float:
    """x = 0 ; r = b ; p = b ; returns b . b in one pass (the set-up of cg!, src/cg.jl:153-162 with M = I, no warm start)."""
    out = C.c_double()
    _ck(lib().khip_cg_setup(x.ctx._h, n, _p(b), _p(x), _p(r), _p(p), C.byref(out)))
    return out.value


def cg_update_(n, a, b, r, p, x):
    """x += a p ; p = r + b p in one pass (src/cg.jl:239,259)."""
    _ck(lib().khip_cg_update(x.ctx._h, n, a, b, _p(r), _p(p), _p(x)))
    return p, x


def waxpy_(n, w, x, b, y):
    _ck(lib().khip_waxpy(w.ctx._h, n, _p(w), _p(x), b, _p(y)))
    return w


def dot2(n, x, y):
    out = (C.c_double * 2)()
    _ck(lib().khip_dot2(x.ctx._h, n, _p(x), _p(y), out))
    return out[0], out[1]


def mgs_(n, V, q, accumulate_into=None, want_norm=True):
    k = len(V)
    ptrs = (C.c_void_p * max(k, 1))(*[_p(v) for v in V])
    h = (C.c_double * max(k, 1))()
    if accumulate_into is not None:
        for i in range(k):
            h[i] = accumulate_into[i]
    nrm = C.c_double()
    _ck(lib().khip_mgs(q.ctx._h, n, k, ptrs, _p(q), h, C.byref(nrm) if want_norm else None,
                       1 if accumulate_into is not None else 0))
    return [h[i] for i in range(k)], (nrm.value if want_norm else None)


def multi_axpy_(n, y, V, x):
    k = len(V)
    ptrs = (C.c_void_p * max(k, 1))(*[_p(v) for v in V])
    coef = (C.c_double * max(k, 1))(*[float(c) for c in y])
    _ck(lib().khip_multi_axpy(x.ctx._h, n, k, coef, ptrs, _p(x)))
    return x


# --------------------------------------------------------------------------- CSR operator

class CsrMatrix:
    """CSR operator resident in HBM; `size`, `eltype`, `kmul_` = the operator contract
    (docs/src/matrix_free.md:32-34)."""

    eltype = np.float64

    def __init__(self, ctx: Context, handle, keep=None):
        self.ctx, self._h, self._keep = ctx, handle, keep
        m, n, nnz = C.c_int64(), C.c_int64(), C.c_int64()
        _ck(lib().khip_csr_shape(handle, C.byref(m), C.byref(n), C.byref(nnz)))
        self.m, self.n, self.nnz = m.value, n.value, nnz.value

    @property
    def shape(self):
        return (self.m, self.n)

    @property
    def spmv_bytes(self) -> int:
        b = C.c_int64()
        _ck(lib().khip_spmv_bytes(self._h, C.byref(b)))
        return b.value

    @property
    def spmv_bytes_stored(self) -> int:
        b = C.c_int64()
        _ck(lib().khip_spmv_bytes_stored(self._h, C.byref(b)))
        return b.value

    @property
    def code_info(self):
        """(bits, diagonals) of the column stream the staged SpMV reads: (32, 0) = plain int32 columns, (8 | 16, T) =
        dictionary-coded diagonals (csrc/colcode.hip), built by the first product that can use it."""
        b, t = C.c_int(), C.c_int()
        _ck(lib().khip_csr_code_info(self._h, C.byref(b), C.byref(t)))
        return b.value, t.value

    @property
    def sell_info(self):
        """(state, units_per_slice, total_units) of the sliced form of a coded operator (khip_csr_sell_info, csrc/colcode.hip
        csr_build_sell): state 1 = built and read by spmv_sell_kernel, 0 = not tried, -1 = not usable."""
        st, u, t = C.c_int(), C.c_int(), C.c_int64()
        _ck(lib().khip_csr_sell_info(self._h, C.byref(st), C.byref(u), C.byref(t)))
        return st.value, u.value, t.value

    @property
    def sell_narrow(self) -> bool:
        """True when the sliced form keeps eight 4-bit codes per row in one 32-bit word (khip_csr_sell_narrow)."""
        v = C.c_int()
        _ck(lib().khip_csr_sell_narrow(self._h, C.byref(v)))
        return bool(v.value)

    @property
    def sell32_info(self):
        """(state, units_per_slice, total_units) of the sliced form with int32 columns (khip_csr_sell32_info)."""
        st, u, t = C.c_int(), C.c_int(), C.c_int64()
        _ck(lib().khip_csr_sell32_info(self._h, C.byref(st), C.byref(u), C.byref(t)))
        return st.value, u.value, t.value

    @property
    def spmv_kernel_choice(self):
        """The SpMV kernel this handle takes under the context's current options (khip_spmv_kernel_info): 1 stream, 2 vector,
        3 ordered, 4 staged rows, 5 row templates, 6 wave-private windows."""
        k = C.c_int()
        _ck(lib().khip_spmv_kernel_info(self.ctx._h, self._h, C.byref(k)))
        return k.value

    @property
    def delta_info(self):
        """(bits, rows_per_block, escapes) of the block-delta column stream the stream SpMV reads (csrc/coldelta.hip):
        (32, 0, 0) = plain int32 columns."""
        b, r, e = C.c_int(), C.c_int(), C.c_int64()
        _ck(lib().khip_csr_delta_info(self._h, C.byref(b), C.byref(r), C.byref(e)))
        return b.value, r.value, e.value

    @property
    def tile_info(self):
        """dict(state, window, grid_tiles, groups, direct_groups, reuse): the p = 16 SpMM kernel of this handle
        (csrc/spmm_tile.hip); state 1 = wave-private LDS windows, -1 = not usable, 0 = no 16-column product yet."""
        st, w, gt = C.c_int(), C.c_int(), C.c_int()
        g, d, r = C.c_int64(), C.c_int64(), C.c_double()
        _ck(lib().khip_csr_tile_info(self._h, C.byref(st), C.byref(w), C.byref(gt), C.byref(g), C.byref(d), C.byref(r)))
        return dict(state=st.value, window=w.value, grid_tiles=gt.value, groups=g.value, direct_groups=d.value, reuse=r.value)

    @property
    def halo_info(self):
        """(gather_mode, n_ghost, n_send) of a distributed handle: how the remote part of x is fetched before a product."""
        g, ng, ns = C.c_int(), C.c_int64(), C.c_int64()
        _ck(lib().khip_csr_halo_info(self._h, C.byref(g), C.byref(ng), C.byref(ns)))
        return g.value, ng.value, ns.value

    def transpose(self) -> "CsrMatrix":
        """A' as its own handle: `At.matvec(x, y)` is `mul!(y, A', x)` (docs/src/matrix_free.md:36-42)."""
        h = C.c_void_p()
        _ck(lib().khip_csr_transpose(self.ctx._h, self._h, C.byref(h)))
        return CsrMatrix(self.ctx, h)

    def compress(self) -> int:
        """Re-encode as row templates when the operator repeats few (column - row, value) rows (stencils);
        returns the number of templates, 0 if it stays CSR.  SpMV results are bit-identical either way."""
        t = C.c_int()
        _ck(lib().khip_csr_compress(self.ctx._h, self._h, C.byref(t)))
        return t.value

    @classmethod
    def from_host(cls, ctx, rowptr, col, val, shape, index_base=0, dist_rows=None, n_global=None):
        rowptr = np.ascontiguousarray(rowptr)
        bits = 64 if rowptr.dtype == np.int64 else 32
        if bits == 32:
            rowptr = rowptr.astype(np.int32, copy=False)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = C.c_void_p()
        if dist_rows is None:
            _ck(lib().khip_csr_create(ctx._h, shape[0], shape[1], val.size, rowptr.ctypes.data, bits, col.ctypes.data,
                                      val.ctypes.data, index_base, 0, C.byref(h)))
        else:
            _ck(lib().khip_csr_create_dist(ctx._h, n_global, dist_rows[0], dist_rows[1] - dist_rows[0], val.size,
                                           rowptr.ctypes.data, bits, col.ctypes.data, val.ctypes.data, index_base, 0,
                                           C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_scipy(cls, ctx, S):
        S = S.tocsr()
        S.sort_indices()
        return cls.from_host(ctx, S.indptr, S.indices, S.data, S.shape)

    @classmethod
    def banded_random(cls, ctx, n, half_band=13, links=3, seed=1, unsym=False, dense_rows=0, rows=None, distributed=False):
        """The "banded + random, fixed seed" benchmark operator (csrc/gen_irregular.cpp): the non-stencil stand-in for the
        SuiteSparse matrices of benchmark/cg_bmark.jl:29-54 -- a thinned symmetric band plus seeded long-range links,
        diagonally dominant; unsym halves the entries above the diagonal, dense_rows adds rows of 3000 more entries."""
        r0, r1 = rows if rows is not None else (0, n)
        rp, cl, vl, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        _ck(lib().khip_gen_banded_random(ctx._h, n, half_band, links, seed, 1 if unsym else 0, dense_rows, r0, r1 - r0,
                                         C.byref(rp), C.byref(cl), C.byref(vl), C.byref(nnz)))
        h = C.c_void_p()
        try:
            if distributed:
                _ck(lib().khip_csr_create_dist(ctx._h, n, r0, r1 - r0, nnz.value, rp, 32, cl, vl, 0, 1, C.byref(h)))
            else:
                if (r0, r1) != (0, n):
                    raise ValueError("a row slice needs distributed=True")
                _ck(lib().khip_csr_create(ctx._h, n, n, nnz.value, rp, 32, cl, vl, 0, 1, C.byref(h)))
        finally:
            for p in (rp, cl, vl):
                lib().khip_free(ctx._h, p)
        return cls(ctx, h)

    @classmethod
    def stencil(cls, ctx, kind: str, n1, n2=None, n3=None, rows=None, distributed=False):
        """Device-side generator of the benchmark operators: 'poisson' = get_div_grad(n1,n2,n3)
        (test/get_div_grad.jl:8-25), 'kron_unsymmetric' (test/test_utils.jl:160-169), 'stencil27' (cfg 5)."""
        kinds = {"poisson": 0, "kron_unsymmetric": 1, "stencil27": 2}
        n2 = n2 or n1
        n3 = n3 or n1
        n = n1 * n2 * n3
        r0, r1 = rows if rows is not None else (0, n)
        rp, cl, vl, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
        _ck(lib().khip_gen_stencil(ctx._h, kinds[kind], n1, n2, n3, r0, r1 - r0, C.byref(rp), C.byref(cl),
                                   C.byref(vl), C.byref(nnz)))
        h = C.c_void_p()
        try:
            if distributed:
                _ck(lib().khip_csr_create_dist(ctx._h, n, r0, r1 - r0, nnz.value, rp, 32, cl, vl, 0, 1, C.byref(h)))
            else:
                if (r0, r1) != (0, n):
                    raise ValueError("a row slice needs distributed=True")
                _ck(lib().khip_csr_create(ctx._h, n, n, nnz.value, rp, 32, cl, vl, 0, 1, C.byref(h)))
        finally:
            for p in (rp, cl, vl):
                lib().khip_free(ctx._h, p)
        return cls(ctx, h)

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                lib().khip_csr_destroy(self._h)
        except Exception:
            pass

    def diagonal(self) -> DeviceVector:
        d = DeviceVector(self.ctx, self.m)
        _ck(lib().khip_csr_diagonal(self.ctx._h, self._h, d.ptr))
        return d

    def matvec(self, x: DeviceVector, y: DeviceVector | None = None) -> DeviceVector:
        y = y if y is not None else DeviceVector(self.ctx, self.m)
        return kmul_(y, self, x)


def gen_stencil_arrays(ctx, kind, n1, n2=None, n3=None, rows=None):
    """Raw (rowptr, col, val) of the device generator copied to host -- used by the parity tests."""
    kinds = {"poisson": 0, "kron_unsymmetric": 1, "stencil27": 2}
    n2 = n2 or n1
    n3 = n3 or n1
    n = n1 * n2 * n3
    r0, r1 = rows if rows is not None else (0, n)
    rp, cl, vl, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    _ck(lib().khip_gen_stencil(ctx._h, kinds[kind], n1, n2, n3, r0, r1 - r0, C.byref(rp), C.byref(cl), C.byref(vl),
                               C.byref(nnz)))
    m = r1 - r0
    rowptr = np.empty(m + 1, dtype=np.int32)
    col = np.empty(nnz.value, dtype=np.int32)
    val = np.empty(nnz.value, dtype=np.float64)
    _ck(lib().khip_memcpy_d2h(ctx._h, rowptr.ctypes.data, rp, 4 * (m + 1)))
    if nnz.value:
        _ck(lib().khip_memcpy_d2h(ctx._h, col.ctypes.data, cl, 4 * nnz.value))
        _ck(lib().khip_memcpy_d2h(ctx._h, val.ctypes.data, vl, 8 * nnz.value))
    for p in (rp, cl, vl):
        lib().khip_free(ctx._h, p)
    return rowptr, col, val


def gen_banded_random_arrays(ctx, n, half_band=13, links=3, seed=1, unsym=False, dense_rows=0, rows=None):
    """Raw (rowptr, col, val) of khip_gen_banded_random copied to host -- used by the parity tests."""
    r0, r1 = rows if rows is not None else (0, n)
    rp, cl, vl, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
    _ck(lib().khip_gen_banded_random(ctx._h, n, half_band, links, seed, 1 if unsym else 0, dense_rows, r0, r1 - r0,
                                     C.byref(rp), C.byref(cl), C.byref(vl), C.byref(nnz)))
    m = r1 - r0
    rowptr = np.empty(m + 1, dtype=np.int32)
    col = np.empty(nnz.value, dtype=np.int32)
    val = np.empty(nnz.value, dtype=np.float64)
    _ck(lib().khip_memcpy_d2h(ctx._h, rowptr.ctypes.data, rp, 4 * (m + 1)))
    if nnz.value:
        _ck(lib().khip_memcpy_d2h(ctx._h, col.ctypes.data, cl, 4 * nnz.value))
        _ck(lib().khip_memcpy_d2h(ctx._h, val.ctypes.data, vl, 8 * nnz.value))
    for p in (rp, cl, vl):
        lib().khip_free(ctx._h, p)
    return rowptr, col, val


# --------------------------------------------------------------------------- solvers

class SimpleStats:
    """SimpleStats (src/krylov_stats.jl:24-44)."""

    def __init__(self, st: CStats):
        self.niter = st.niter
        self.solved = bool(st.solved)
        self.inconsistent = bool(st.inconsistent)
        self.indefinite = bool(st.indefinite)
        self.npcCount = st.npcCount
        self.timer = st.timer
        self.allocation_timer = st.allocation_timer
        self.status = st.status.decode("utf-8")
        self.residuals = np.array([st.residuals[i] for i in range(st.nres)]) if st.nres else np.zeros(0)
        self.error = st.error.decode("utf-8")

    def __repr__(self):
        return f"SimpleStats(niter={self.niter}, solved={self.solved}, status={self.status!r})"


class Ilu0:
    """ILU(0) of A on its own pattern as the operator y = U \\ (L \\ x); for SPD A this is IC(0).  Usable as the
    M / N argument of cg_ / gmres_ / bicgstab_ (the reference's ic02 / ilu02 recipes, docs/src/gpu.md:74-163)."""

    def __init__(self, A: "CsrMatrix", graph: bool = True):
        self.ctx, self.n, self.A = A.ctx, A.m, A
        self.op = COperator()
        _ck(lib().khip_ilu0_create(A.ctx._h, A._h, C.byref(self.op)))
        if not graph:
            _ck(lib().khip_ilu0_set_graph(C.byref(self.op), 0))

    def __call__(self, x, y):
        rc = self.op.apply(self.op.self, _p(x), _p(y))
        if rc:
            raise KhipError(rc, lib().khip_last_error().decode())
        return y

    @property
    def levels(self):
        lo, up = _i64(), _i64()
        _ck(lib().khip_ilu0_info(C.byref(self.op), C.byref(lo), C.byref(up), None))
        return lo.value, up.value

    def block_info(self):
        """(grid dims the pattern was recognised as, or (0, 0, 0); blocks per triangle -- 0: level scheduling, > 0 with dims
        (0, 0, 0): blocks from the level-sorted row sequence; 1 if a bounded spin of the block schedule ever gave up).
        Synchronises."""
        dims, nb, failed = (_i64 * 3)(), _i64(), _int()
        _ck(lib().khip_ilu0_block_info(C.byref(self.op), dims, C.byref(nb), C.byref(failed)))
        return tuple(dims), nb.value, failed.value

    def values(self):
        """Factor values on A's pattern (host copy)."""
        p = C.c_void_p()
        _ck(lib().khip_ilu0_info(C.byref(self.op), None, None, C.byref(p)))
        return DeviceVector(self.ctx, self.A.nnz, ptr=p.value, owner=self).to_host()

    def __del__(self):
        try:
            if self.ctx._h:
                lib().khip_ilu0_destroy(C.byref(self.op))
        except Exception:
            pass


def _make_operator(ctx, op, n, keep):
    """CsrMatrix | callable(x: DeviceVector, y: DeviceVector) | None -> POINTER(COperator) or None."""
    if op is None:
        return None
    if isinstance(op, (Jacobi, Ilu0)):  # native operator: no Python in the loop
        keep.append(op)
        return C.byref(op.op)
    co = COperator()
    if isinstance(op, CsrMatrix):
        co.csr = op._h
        co.apply = C.cast(None, APPLY_FN)
    else:
        def thunk(_self, xp, yp):
            try:
                op(DeviceVector(ctx, n, ptr=xp), DeviceVector(ctx, n, ptr=yp))
                return 0
            except Exception as e:  # never propagate across the C boundary
                sys.stderr.write(f"operator callback failed: {e}\n")
                return 1
        fn = APPLY_FN(thunk)
        keep.append(fn)
        co.csr = None
        co.apply = fn
    keep.append(co)
    return C.byref(co)


# The reference's entry points forward EVERY keyword to the in-place method explicitly, their own defaults included:
# `cg(A, b)`, `krylov_solve(Val(:cg), A, b)`, `krylov_solve!(ws, A, b)` and the x0 forms all call
# `cg!(ws, A, b; M, ldiv, ..., callback, iostream)` with `callback = workspace -> false` (src/cg.jl:110, src/interface.jl:146-154,
# 266-275, 331, 336-345).  A binding that takes its device-resident loop only when NO callback was passed would therefore never take
# it from those entry points: the default callback has to be recognised for what it is.  `default_callback` below is
# this mirror's `workspace -> false`; `_user_callback` maps it (and None) to "no callback", exactly as `user_callback` in
# julia/KrylovHIP/src/KrylovHIP.jl maps the anonymous functions the reference's generated methods forward.
def default_callback(workspace) -> bool:          # `callback = workspace -> false`
    return False


def _user_callback(callback):
    return None if (callback is None or callback is default_callback) else callback


def _log_fd(iostream) -> int:
    """The reference's `iostream` keyword (default kstdout) as khip_options.log_fd: 0 = stdout, else the caller's descriptor."""
    if iostream is None or iostream is sys.stdout:
        return 0
    if isinstance(iostream, int):
        return iostream
    iostream.flush()
    return iostream.fileno()


def _make_options(atol=None, rtol=None, itmax=0, timemax=None, history=False, radius=0.0, linesearch=False,
                  restart=False, reorthogonalization=False, fused=True, callback=None, keep=None, ws=None, variant=0, verbose=0, log_fd=0,
                  ldiv=False, iostream=None):
    if ldiv:        # `ldiv = true` applies M / N with ldiv! (src/krylov_utils.jl:307): a factorisation object of the HOST language
        raise KhipError(-4, "ldiv = true has no meaning across the C ABI: pass M / N as operators z <- M r (Jacobi, Ilu0, a callable)")
    if iostream is not None:
        log_fd = _log_fd(iostream)
    if timemax is not None:
        # Inf (the reference's default) = no limit = the C default; a limit the entry point has already used up
        # (`timemax -= elapsed_time`, src/interface.jl:151) stays a limit: the C side reads <= 0 as "none"
        timemax = None if math.isinf(timemax) else max(timemax, sys.float_info.min)
    callback = _user_callback(callback)
    o = lib().khip_default_options()
    if atol is not None:
        o.atol = atol
    if rtol is not None:
        o.rtol = rtol
    o.itmax = int(itmax)
    if timemax is not None:
        o.timemax = timemax
    o.history = int(history)
    o.radius = radius
    o.linesearch = int(linesearch)
    o.restart = int(restart)
    o.reorthogonalization = int(reorthogonalization)
    o.fused = 2 if fused is True else int(fused)     # True = everything that is bit-identical: fused kernels + device-resident scalars
    o.variant = int(variant)
    o.verbose = int(verbose)
    o.log_fd = int(log_fd)          # the reference's `iostream`: 0 = stdout, else a file descriptor
    if callback is not None:
        def cb(_ws, _ud):
            try:
                return 1 if callback(ws) else 0
            except Exception as e:
                sys.stderr.write(f"callback failed: {e}\n")
                return 1
        fn = CALLBACK_FN(cb)
        keep.append(fn)
        o.callback = fn
    return o


# Workspaces own their vectors on THIS side, as the reference's do on the Julia side (`x, r, p, Ap :: S`,
# src/krylov_workspaces.jl:236-291), and hand the device pointers to the library (khip_*_workspace_adopt): the solvers run on the
# caller's vectors, `solution(ws) is ws.x`.  This is the executed twin of julia/KrylovHIP/src/KrylovHIP.jl.  `adopt=False` (or
# KHIP_PY_WORKSPACES=create in the environment) takes the library-owned workspaces of khip_*_workspace_create instead.
def _adopt_default() -> bool:
    return os.environ.get("KHIP_PY_WORKSPACES", "adopt") != "create"


class _Workspace:
    _prefix = ""
    _fields = ()

    def _fn(self, name):
        return getattr(lib(), f"khip_{self._prefix}_{name}")

    def __del__(self):
        try:
            if self._h and self.ctx._h:
                self._fn("workspace_destroy")(self._h)
        except Exception:
            pass

    def _adopt_vector(self, name: str, v: "DeviceVector"):
        """allocate_if(..., workspace, :name, S, workspace.x) followed by the hand-over of the pointer."""
        self._vec[name] = v
        _ck(self._fn("workspace_adopt_vector")(self._h, name.encode(), v.ptr))
        return v

    def _allocate_if(self, cond: bool, name: str):
        """src/krylov_utils.jl:281-288: allocate the lazily-held vector `name` on first need."""
        if cond and self.adopted and name not in self._vec:
            t0 = time.perf_counter()
            v = self.ctx.empty(self.n)
            self._alloc_s += time.perf_counter() - t0            # stats.allocation_timer counts the lazy allocations too
            self._adopt_vector(name, v)

    @property
    def x(self) -> DeviceVector:
        """solution(workspace) === workspace.x (src/workspace_accessors.jl:151, test/test_interface.jl:260)."""
        if self.adopted:
            return self._vec["x"]
        return DeviceVector(self.ctx, self.n, ptr=self._fn("solution")(self._h), owner=self)

    @property
    def stats(self) -> SimpleStats:
        st = SimpleStats(self._fn("stats")(self._h).contents)
        if self.adopted:                                  # the vectors were allocated (and timed) on this side
            st.allocation_timer += self._alloc_s
        st.timer += getattr(self, "_timer_extra", 0.0)    # `workspace.stats.timer += elapsed_time`, src/interface.jl:153
        return st

    @property
    def last_path(self) -> int:
        """Which loop the last solve ran (khip_*_last_path): 2 device-resident / look-ahead, 1 host-driven fused, 0 one launch per
        primitive, -1 none yet.  What KrylovHIP.NATIVE_SOLVES / LAST_PATH report on the Julia side."""
        return self._fn("last_path")(self._h)

    def warm_start_(self, x0: DeviceVector):
        """warm_start!(workspace, x0) (src/workspace_accessors.jl:193-200): allocate_if(true, ws, :Δx, ...); kcopy!(n, ws.Δx, x0)."""
        if self.adopted:
            self._allocate_if(True, "dx")
            kcopy_(self.n, self._vec["dx"], x0)
            _ck(self._fn("warm_start")(self._h, self._vec["dx"].ptr))     # same pointer: only sets the flag
        else:
            _ck(self._fn("warm_start")(self._h, _p(x0)))
        return self

    @property
    def nbytes(self) -> int:
        return self._fn("workspace_bytes")(self._h)


class CgWorkspace(_Workspace):
    """CgWorkspace(m, n, S) (src/krylov_workspaces.jl:236-291)."""
    _prefix = "cg"

    def __init__(self, ctx: Context, m: int, n: int, adopt: bool | None = None):
        self.ctx, self.m, self.n = ctx, m, n
        self.adopted = _adopt_default() if adopt is None else bool(adopt)
        self._h = C.c_void_p()
        if self.adopted:
            t0 = time.perf_counter()
            self._vec = {k: ctx.empty(n) for k in ("x", "r", "p", "Ap")}     # S(undef, n) x 4; dx, npc_dir, z stay empty
            self._alloc_s = time.perf_counter() - t0
            v = self._vec
            _ck(lib().khip_cg_workspace_adopt(ctx._h, m, n, v["x"].ptr, v["r"].ptr, v["p"].ptr, v["Ap"].ptr, C.byref(self._h)))
        else:
            _ck(lib().khip_cg_workspace_create(ctx._h, m, n, C.byref(self._h)))

    def vector(self, name: str):
        if self.adopted and name in self._vec:
            return self._vec[name]
        p = lib().khip_cg_vector(self._h, name.encode())
        return DeviceVector(self.ctx, self.n, ptr=p, owner=self) if p else None


class GmresWorkspace(_Workspace):
    """GmresWorkspace(m, n, S; memory = 20) (src/krylov_workspaces.jl:2857-2924)."""
    _prefix = "gmres"

    def __init__(self, ctx: Context, m: int, n: int, memory: int = 20, adopt: bool | None = None):
        self.ctx, self.m, self.n, self.memory = ctx, m, n, memory
        self.adopted = _adopt_default() if adopt is None else bool(adopt)
        self._h = C.c_void_p()
        if self.adopted:
            mem = 20 if memory <= 0 else memory
            mem = min(m, mem)                                                  # memory = min(m, memory), :2900
            t0 = time.perf_counter()
            self._vec = {k: ctx.empty(n) for k in ("x", "w")}
            self.V = [ctx.empty(n) for _ in range(mem)]                        # V = S[S(undef, n) for i = 1 : memory]
            self._alloc_s = time.perf_counter() - t0
            ptrs = (C.c_void_p * max(mem, 1))(*[v.ptr for v in self.V])
            _ck(lib().khip_gmres_workspace_adopt(ctx._h, m, n, mem, self._vec["x"].ptr, self._vec["w"].ptr, ptrs, C.byref(self._h)))

            def grow(_ud):                                                     # push!(V, similar(x)), src/gmres.jl:319-324
                try:
                    self.V.append(self.ctx.empty(self.n))
                    return self.V[-1].ptr
                except Exception as e:
                    sys.stderr.write(f"grow callback failed: {e}\n")
                    return None
            self._grow = GROW_FN(grow)
            _ck(lib().khip_gmres_workspace_set_grow(self._h, self._grow, None))
        else:
            _ck(lib().khip_gmres_workspace_create(ctx._h, m, n, memory, C.byref(self._h)))

    def host_state(self):
        """(c, s, z, R, inner_iter) of the last solve: the host fields of the reference's workspace (:2866-2871)."""
        ln, it = C.c_int(), C.c_int()
        _ck(lib().khip_gmres_host_state(self._h, 0, None, None, None, None, C.byref(ln), C.byref(it)))
        k = ln.value
        c, s_, z, R = np.zeros(k), np.zeros(k), np.zeros(k), np.zeros(k * (k + 1) // 2)
        _ck(lib().khip_gmres_host_state(self._h, k, c.ctypes.data_as(c_double_p), s_.ctypes.data_as(c_double_p),
                                        z.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p), C.byref(ln), C.byref(it)))
        return c, s_, z, R, it.value


class BicgstabWorkspace(_Workspace):
    """BicgstabWorkspace(m, n, S) (src/krylov_workspaces.jl:1568-1629)."""
    _prefix = "bicgstab"

    def __init__(self, ctx: Context, m: int, n: int, adopt: bool | None = None):
        self.ctx, self.m, self.n = ctx, m, n
        self.adopted = _adopt_default() if adopt is None else bool(adopt)
        self._h = C.c_void_p()
        if self.adopted:
            t0 = time.perf_counter()
            self._vec = {k: ctx.empty(n) for k in ("x", "r", "p", "v", "s", "qd")}
            self._alloc_s = time.perf_counter() - t0
            v = self._vec
            _ck(lib().khip_bicgstab_workspace_adopt(ctx._h, m, n, v["x"].ptr, v["r"].ptr, v["p"].ptr, v["v"].ptr, v["s"].ptr,
                                                    v["qd"].ptr, C.byref(self._h)))
        else:
            _ck(lib().khip_bicgstab_workspace_create(ctx._h, m, n, C.byref(self._h)))


def _finish(ws, rc):
    ws._timer_extra = 0.0          # an entry point that did work of its own before the solve adds it afterwards
    if rc != 0:
        st = ws.stats
        raise KhipError(rc, st.error or lib().khip_last_error().decode("utf-8", "replace"))
    return ws


def cg_(ws: CgWorkspace, A, b: DeviceVector, M=None, **kw):
    """cg!(workspace, A, b; M, radius, linesearch, atol, rtol, itmax, timemax, history, callback)
    (src/cg.jl:120-291).  Returns the workspace."""
    keep = []
    ws._allocate_if(M is not None, "z")                                                   # src/cg.jl:142
    ws._allocate_if(bool(kw.get("linesearch")) or kw.get("radius", 0.0) > 0, "npc_dir")    # :143
    opts = _make_options(keep=keep, ws=ws, **kw)
    rc = lib().khip_cg_solve(ws._h, _make_operator(ws.ctx, A, ws.n, keep), _make_operator(ws.ctx, M, ws.n, keep),
                             _p(b), C.byref(opts))
    return _finish(ws, rc)


def gmres_(ws: GmresWorkspace, A, b: DeviceVector, M=None, N=None, **kw):
    """gmres!(workspace, A, b; M, N, restart, reorthogonalization, ...) (src/gmres.jl:121-384)."""
    keep = []
    ws._allocate_if(M is not None, "q")                                                   # src/gmres.jl:142-144
    ws._allocate_if(N is not None, "p")
    ws._allocate_if(bool(kw.get("restart")), "dx")
    opts = _make_options(keep=keep, ws=ws, **kw)
    rc = lib().khip_gmres_solve(ws._h, _make_operator(ws.ctx, A, ws.n, keep), _make_operator(ws.ctx, M, ws.n, keep),
                                _make_operator(ws.ctx, N, ws.n, keep), _p(b), C.byref(opts))
    return _finish(ws, rc)


def bicgstab_(ws: BicgstabWorkspace, A, b: DeviceVector, c: DeviceVector | None = None, M=None, N=None, **kw):
    """bicgstab!(workspace, A, b; c, M, N, ...) (src/bicgstab.jl:125-277)."""
    keep = []
    ws._allocate_if(M is not None, "t")                                                   # src/bicgstab.jl:148-149
    ws._allocate_if(N is not None, "yz")
    opts = _make_options(keep=keep, ws=ws, **kw)
    rc = lib().khip_bicgstab_solve(ws._h, _make_operator(ws.ctx, A, ws.n, keep),
                                   _make_operator(ws.ctx, M, ws.n, keep), _make_operator(ws.ctx, N, ws.n, keep),
                                   _p(b), _p(c), C.byref(opts))
    return _finish(ws, rc)


def _local_rows(A):
    return A.m if isinstance(A, CsrMatrix) else None


# ---- the generated entry points of src/interface.jl, with what they FORWARD -----------------------------------------------------
# def_kwargs_cg / _gmres / _bicgstab / _block_gmres (src/cg.jl:101-112, src/gmres.jl:96-110, src/bicgstab.jl:105-116,
# src/block_gmres.jl:85-99) as this mirror spells them (M / N: None = I; iostream: None = kstdout; timemax: inf).  Every
# out-of-place and generic entry below builds the complete keyword set from these tables and passes ALL of it on, as the reference's
# generated methods do -- tests/test_abi.py compares the tables with the reference's, and test_gpu_adopt.py asserts that the solve
# they lead to ran the device-resident loop (`last_path == 2`).
_SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)
FORWARDED_DEFAULTS = {
    "cg": dict(M=None, ldiv=False, radius=0.0, linesearch=False, atol=_SQRT_EPS, rtol=_SQRT_EPS, itmax=0, timemax=math.inf,
               verbose=0, history=False, callback=default_callback, iostream=None),
    "gmres": dict(M=None, N=None, ldiv=False, restart=False, reorthogonalization=False, atol=_SQRT_EPS, rtol=_SQRT_EPS, itmax=0,
                  timemax=math.inf, verbose=0, history=False, callback=default_callback, iostream=None),
    "bicgstab": dict(c=None, M=None, N=None, ldiv=False, atol=_SQRT_EPS, rtol=_SQRT_EPS, itmax=0, timemax=math.inf, verbose=0,
                     history=False, callback=default_callback, iostream=None),
    "block_gmres": dict(M=None, N=None, ldiv=False, restart=False, reorthogonalization=False, atol=_SQRT_EPS, rtol=_SQRT_EPS, itmax=0,
                        timemax=math.inf, verbose=0, history=False, callback=default_callback, iostream=None),
}
WORKSPACE_KWARGS = {"gmres": dict(memory=20), "block_gmres": dict(memory=5)}      # kwargs_workspace_gmres, _block_gmres


def _forward(method: str, kw: dict) -> dict:
    """The keyword set a generated entry point passes to the in-place method: its defaults, overridden by what the caller gave.
    Keywords of this library that the reference does not have (fused, variant) pass through."""
    full = dict(FORWARDED_DEFAULTS[method])
    full.update(kw)
    return full


def krylov_workspace(method: str, *args, ctx: Context | None = None, **kw):
    """krylov_workspace(Val(method), m, n, S; memory) / (Val(method), A, b; memory) (src/interface.jl:117-141, 237-244)."""
    cls = {"cg": CgWorkspace, "gmres": GmresWorkspace, "bicgstab": BicgstabWorkspace, "block_gmres": BlockGmresWorkspace}[method]
    if len(args) == 2 and isinstance(args[1], DeviceVector):                  # (A, b)
        A, b = args
        return cls(b.ctx, len(b), len(b), **kw)
    if method == "block_gmres":
        if len(args) == 2:                                                       # (A, B) with a host n x p array B
            B = np.asarray(args[1])
            return cls(ctx or args[0].ctx, B.shape[0], B.shape[0], B.shape[1], **kw)
        m, n, p = args
        return cls(ctx, m, n, p, **kw)
    m, n = args
    return cls(ctx, m, n, **kw)


_INPLACE = {}      # filled below: CgWorkspace -> ("cg", cg_), ...


def krylov_solve_(ws, A, b, x0=None, **kw):
    """krylov_solve!(workspace, A, b[, x0]; kwargs...) (src/interface.jl:331-345, 306-320): dispatch on the workspace type; the x0
    form warm-starts first and charges that time to the solve, as the reference does."""
    method, inplace = _INPLACE[type(ws)]
    full = _forward(method, kw)
    elapsed = 0.0
    if x0 is not None:
        t0 = time.perf_counter()
        ws.warm_start_(x0)
        elapsed = time.perf_counter() - t0
        full["timemax"] = full["timemax"] - elapsed
    inplace(ws, A, b, **full)
    ws._timer_extra = elapsed
    return ws


def krylov_solve(method: str, A, b, x0=None, ctx: Context | None = None, **kw):
    """krylov_solve(Val(method), A, b[, x0]; kwargs...) = method(A, b[, x0]; kwargs...) (src/interface.jl:146-199): a fresh
    workspace (its creation charged to `timemax` and `stats.timer`), every keyword forwarded.  Returns (x, stats, workspace)."""
    wkw = {k: kw.pop(k) for k in list(kw) if k in WORKSPACE_KWARGS.get(method, {})}
    t0 = time.perf_counter()
    if method == "block_gmres":
        B = np.asarray(b, dtype=np.float64)
        ctx = ctx or A.ctx
        ws = krylov_workspace(method, A, B, ctx=ctx, **wkw)
        b = ctx.array(np.asfortranarray(B).ravel(order="F"))
    else:
        ws = krylov_workspace(method, A, b, **wkw)
    if x0 is not None:
        ws.warm_start_(x0)
    elapsed = time.perf_counter() - t0
    full = _forward(method, kw)
    full["timemax"] = full["timemax"] - elapsed
    _INPLACE[type(ws)][1](ws, A, b, **full)
    ws._timer_extra = elapsed
    return (ws.X if method == "block_gmres" else ws.x), ws.stats, ws


def cg(A, b: DeviceVector, x0=None, **kw):
    """Out-of-place cg(A, b[, x0]; kwargs...) -> (x, stats, workspace) (src/interface.jl:146-154, 160-170)."""
    return krylov_solve("cg", A, b, x0, **kw)


def gmres(A, b: DeviceVector, x0=None, **kw):
    return krylov_solve("gmres", A, b, x0, **kw)


def bicgstab(A, b: DeviceVector, x0=None, **kw):
    return krylov_solve("bicgstab", A, b, x0, **kw)


# --------------------------------------------------------------------------- Krylov processes
# src/krylov_processes.jl: same names, arguments and return values; the bases live in HBM.

class DeviceMatrix:
    """Dense column-major n x ncols Float64 matrix in HBM (`M(undef, n, k+1)`, src/krylov_processes.jl:52):
    column j is the DeviceVector `col(j)`; the leading dimension is n rounded up to 32 doubles."""

    def __init__(self, ctx: Context, n: int, ncols: int):
        self.ctx, self.n, self.ncols = ctx, int(n), int(ncols)
        self.ld = max(32, (self.n + 31) & ~31)
        self.buf = DeviceVector(ctx, self.ld * self.ncols)

    @property
    def ptr(self):
        return self.buf.ptr

    @property
    def shape(self):
        return (self.n, self.ncols)

    def col(self, j: int) -> DeviceVector:
        return self.buf.slice(j * self.ld, j * self.ld + self.n)

    def to_host(self) -> np.ndarray:
        return np.asfortranarray(self.buf.to_host().reshape(self.ncols, self.ld)[:, :self.n].T)


def _tridiag_from_nzval(k, nz):
    """The (k+1) x k SparseMatrixCSC of src/krylov_processes.jl:35-48 as scipy CSC."""
    import scipy.sparse as sp
    colptr = np.zeros(k + 1, dtype=np.int64)
    rowval = np.zeros(3 * k - 1, dtype=np.int64)
    for i in range(1, k + 1):
        pos = colptr[i - 1]
        colptr[i] = 3 * i - 1
        if i == 1:
            rowval[pos], rowval[pos + 1] = 0, 1
        else:
            rowval[pos], rowval[pos + 1], rowval[pos + 2] = i - 2, i - 1, i
    return sp.csc_matrix((nz, rowval, colptr), shape=(k + 1, k))


def _bidiag_from_nzval(k, nz):
    """The (k+1) x (k+1) lower bidiagonal SparseMatrixCSC of src/krylov_processes.jl:331-347 as scipy CSC."""
    import scipy.sparse as sp
    colptr = np.zeros(k + 2, dtype=np.int64)
    rowval = np.zeros(2 * k + 1, dtype=np.int64)
    for i in range(1, k + 2):
        pos = colptr[i - 1]
        if i <= k:
            colptr[i] = pos + 2
            rowval[pos], rowval[pos + 1] = i - 1, i
        else:
            colptr[i] = pos + 1
            rowval[pos] = i - 1
    return sp.csc_matrix((nz, rowval, colptr), shape=(k + 1, k + 1))


def _basis(ctx, n, ncols, out):
    if out is None:
        return DeviceMatrix(ctx, n, ncols)
    if out.shape != (n, ncols):
        raise ValueError(f"basis storage is {out.shape}, need {(n, ncols)}")
    return out


def hermitian_lanczos(A, b: DeviceVector, k: int, allow_breakdown=False, reorthogonalization=False, V=None):
    """V, beta1, T = hermitian_lanczos(A, b, k; allow_breakdown, reorthogonalization) (src/krylov_processes.jl:28-102).
    V: DeviceMatrix n x (k+1) (pass `V=` to reuse storage); T: scipy CSC (k+1) x k."""
    ctx, n, keep = b.ctx, len(b), []
    V = _basis(ctx, n, k + 1, V)
    beta, nz = C.c_double(), np.zeros(3 * k - 1)
    _ck(lib().khip_hermitian_lanczos(ctx._h, _make_operator(ctx, A, n, keep), n, b.ptr, k, int(allow_breakdown),
                                     int(reorthogonalization), V.ptr, V.ld, C.byref(beta), nz.ctypes.data_as(c_double_p)))
    return V, beta.value, _tridiag_from_nzval(k, nz)


def arnoldi(A, b: DeviceVector, k: int, allow_breakdown=False, reorthogonalization=False, V=None):
    """V, beta, H = arnoldi(A, b, k; allow_breakdown, reorthogonalization) (src/krylov_processes.jl:250-296).
    V: DeviceMatrix n x (k+1) (pass `V=` to reuse storage); H: dense (k+1) x k numpy array."""
    ctx, n, keep = b.ctx, len(b), []
    V = _basis(ctx, n, k + 1, V)
    beta, H = C.c_double(), np.zeros((k + 1, k), order="F")
    _ck(lib().khip_arnoldi(ctx._h, _make_operator(ctx, A, n, keep), n, b.ptr, k, int(allow_breakdown),
                           int(reorthogonalization), V.ptr, V.ld, C.byref(beta), H.ctypes.data_as(c_double_p)))
    return V, beta.value, H


def golub_kahan(A, b: DeviceVector, k: int, allow_breakdown=False, At=None, n=None, V=None, U=None):
    """V, U, beta1, L = golub_kahan(A, b, k; allow_breakdown) (src/krylov_processes.jl:323-398).  A: CsrMatrix (its
    adjoint is built with A.transpose() unless `At` is given) or a callable together with a callable `At` and the
    column count `n` (`size(A, 2)`)."""
    ctx, m, keep = b.ctx, len(b), []
    if At is None:
        if not isinstance(A, CsrMatrix):
            raise TypeError("golub_kahan: a callable A needs the adjoint callable At")
        At = A.transpose()
    if isinstance(A, CsrMatrix):
        n = A.n
    elif n is None:
        raise TypeError("golub_kahan: a callable A needs n = size(A, 2)")
    V, U = _basis(ctx, n, k + 1, V), _basis(ctx, m, k + 1, U)
    beta, nz = C.c_double(), np.zeros(2 * k + 1)
    # callbacks receive (x, y) sized for their own direction
    opA = _make_operator(ctx, A, n, keep) if isinstance(A, CsrMatrix) else _make_operator_mn(ctx, A, n, m, keep)
    opAt = _make_operator(ctx, At, m, keep) if isinstance(At, CsrMatrix) else _make_operator_mn(ctx, At, m, n, keep)
    _ck(lib().khip_golub_kahan(ctx._h, opA, opAt, m, n, b.ptr, k, int(allow_breakdown), V.ptr, V.ld, U.ptr, U.ld,
                               C.byref(beta), nz.ctypes.data_as(c_double_p)))
    return V, U, beta.value, _bidiag_from_nzval(k, nz)


def _op_mn(ctx, op, nin, nout, keep):
    return _make_operator(ctx, op, nin, keep) if isinstance(op, CsrMatrix) else _make_operator_mn(ctx, op, nin, nout, keep)


def nonhermitian_lanczos(A, b: DeviceVector, c: DeviceVector, k: int, allow_breakdown=False, At=None, V=None, U=None):
    """V, beta1, T, U, gamma1, Tt = nonhermitian_lanczos(A, b, c, k; allow_breakdown) (src/krylov_processes.jl:133-222)."""
    ctx, n, keep = b.ctx, len(b), []
    if At is None:
        if not isinstance(A, CsrMatrix):
            raise TypeError("nonhermitian_lanczos: a callable A needs the adjoint callable At")
        At = A.transpose()
    V, U = _basis(ctx, n, k + 1, V), _basis(ctx, n, k + 1, U)
    beta, gamma, nt, nh = C.c_double(), C.c_double(), np.zeros(3 * k - 1), np.zeros(3 * k - 1)
    _ck(lib().khip_nonhermitian_lanczos(ctx._h, _op_mn(ctx, A, n, n, keep), _op_mn(ctx, At, n, n, keep), n, b.ptr, c.ptr, k,
                                        int(allow_breakdown), V.ptr, V.ld, U.ptr, U.ld, C.byref(beta), C.byref(gamma),
                                        nt.ctypes.data_as(c_double_p), nh.ctypes.data_as(c_double_p)))
    return V, beta.value, _tridiag_from_nzval(k, nt), U, gamma.value, _tridiag_from_nzval(k, nh)


def saunders_simon_yip(A, b: DeviceVector, c: DeviceVector, k: int, allow_breakdown=False, At=None, V=None, U=None):
    """V, beta1, T, U, gamma1, Tt = saunders_simon_yip(A, b, c, k; allow_breakdown) (src/krylov_processes.jl:431-524).
    A is m x n, b has length m, c has length n."""
    ctx, m, n, keep = b.ctx, len(b), len(c), []
    if At is None:
        if not isinstance(A, CsrMatrix):
            raise TypeError("saunders_simon_yip: a callable A needs the adjoint callable At")
        At = A.transpose()
    V, U = _basis(ctx, m, k + 1, V), _basis(ctx, n, k + 1, U)
    beta, gamma, nt, nh = C.c_double(), C.c_double(), np.zeros(3 * k - 1), np.zeros(3 * k - 1)
    _ck(lib().khip_saunders_simon_yip(ctx._h, _op_mn(ctx, A, n, m, keep), _op_mn(ctx, At, m, n, keep), m, n, b.ptr, c.ptr, k,
                                      int(allow_breakdown), V.ptr, V.ld, U.ptr, U.ld, C.byref(beta), C.byref(gamma),
                                      nt.ctypes.data_as(c_double_p), nh.ctypes.data_as(c_double_p)))
    return V, beta.value, _tridiag_from_nzval(k, nt), U, gamma.value, _tridiag_from_nzval(k, nh)


def montoison_orban(A, B, b: DeviceVector, c: DeviceVector, k: int, allow_breakdown=False, reorthogonalization=False,
                    V=None, U=None):
    """V, beta, H, U, gamma, F = montoison_orban(A, B, b, c, k; allow_breakdown, reorthogonalization)
    (src/krylov_processes.jl:553-632).  A is m x n, B is n x m, b has length m, c has length n."""
    ctx, m, n, keep = b.ctx, len(b), len(c), []
    V, U = _basis(ctx, m, k + 1, V), _basis(ctx, n, k + 1, U)
    beta, gamma = C.c_double(), C.c_double()
    H, F = np.zeros((k + 1, k), order="F"), np.zeros((k + 1, k), order="F")
    _ck(lib().khip_montoison_orban(ctx._h, _op_mn(ctx, A, n, m, keep), _op_mn(ctx, B, m, n, keep), m, n, b.ptr, c.ptr, k,
                                   int(allow_breakdown), int(reorthogonalization), V.ptr, V.ld, U.ptr, U.ld,
                                   C.byref(beta), C.byref(gamma), H.ctypes.data_as(c_double_p), F.ctypes.data_as(c_double_p)))
    return V, beta.value, H, U, gamma.value, F


def _make_operator_mn(ctx, op, nin, nout, keep):
    """callable(x: DeviceVector[nin], y: DeviceVector[nout]) -> POINTER(COperator) for rectangular operators."""
    def thunk(_self, xp, yp):
        try:
            op(DeviceVector(ctx, nin, ptr=xp), DeviceVector(ctx, nout, ptr=yp))
            return 0
        except Exception as e:
            sys.stderr.write(f"operator callback failed: {e}\n")
            return 1
    fn = APPLY_FN(thunk)
    co = COperator()
    co.csr, co.apply = None, fn
    keep.extend([fn, co])
    return C.byref(co)


# --------------------------------------------------------------------------- host-only partition helpers

def row_partition(n: int, nranks: int):
    """Contiguous 1-D row partition: rank g owns rows [g*n//G, (g+1)*n//G)."""
    return [(g * n) // nranks for g in range(nranks + 1)]


def ghost_columns_host(rowptr, col, row0):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    m = rowptr.size - 1
    cnt = C.c_int64()
    out = np.empty(max(col.size, 1), dtype=np.int32)
    _ck(lib().khip_ghost_columns_host(rowptr.ctypes.data, col.ctypes.data, m, row0, out.ctypes.data, out.size,
                                      C.byref(cnt)))
    return out[: cnt.value].copy()


def halo_plan_host(rank, nranks, row_starts, ghost_lists):
    row_starts = np.ascontiguousarray(row_starts, dtype=np.int64)
    off = np.zeros(nranks + 1, dtype=np.int64)
    for r in range(nranks):
        off[r + 1] = off[r] + len(ghost_lists[r])
    allg = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int32) for g in ghost_lists])
                                if off[-1] else np.zeros(1, dtype=np.int32), dtype=np.int32)
    recv_off = np.zeros(nranks + 1, dtype=np.int64)
    send_off = np.zeros(nranks + 1, dtype=np.int64)
    cap = int(off[-1]) + 1
    send_idx = np.zeros(cap, dtype=np.int32)
    _ck(lib().khip_halo_plan_host(rank, nranks, row_starts.ctypes.data, allg.ctypes.data, off.ctypes.data,
                                  recv_off.ctypes.data, send_off.ctypes.data, send_idx.ctypes.data, cap))
    return recv_off, send_off, send_idx[: send_off[-1]].copy()


# --------------------------------------------------------------------------- block-GMRES (panels)

def panel_rows(n: int) -> int:
    out = C.c_int64()
    _ck(lib().khip_panel_rows(n, C.byref(out)))
    return out.value


class Panel:
    """n x p block in HBM, ROW-MAJOR with the row count padded to 16 (csrc/panel.hip).  The reference's
    `SM(undef, n, p)` storage (src/block_krylov_workspaces.jl:137-163) in the layout the MFMA kernels want."""

    def __init__(self, ctx: Context, n: int, p: int):
        self.ctx, self.n, self.p = ctx, n, p
        self.n_pad = panel_rows(n)
        self.buf = ctx.zeros(self.n_pad * p)

    @classmethod
    def from_host(cls, ctx, M):
        M = np.asarray(M, dtype=np.float64)
        P = cls(ctx, M.shape[0], M.shape[1])
        col = ctx.array(np.asfortranarray(M).ravel(order="F"))
        _ck(lib().khip_panel_from_colmajor(ctx._h, P.n, P.p, col.ptr, P.buf.ptr))
        ctx.sync()
        return P

    def to_host(self) -> np.ndarray:
        col = self.ctx.empty(self.n * self.p)
        _ck(lib().khip_panel_to_colmajor(self.ctx._h, self.n, self.p, self.buf.ptr, col.ptr))
        return col.to_host().reshape(self.p, self.n).T.copy()


def panel_gemm_tn(V: Panel, Q: Panel) -> np.ndarray:
    """Psi = V' * Q  (mul!(R, V', Q), src/block_gmres.jl:245)."""
    out = np.zeros((V.p, V.p), order="F")
    _ck(lib().khip_panel_gemm_tn(V.ctx._h, V.n, V.p, V.buf.ptr, Q.buf.ptr, out.ctypes.data_as(c_double_p)))
    return out


def panel_gemm_nn_(alpha, V: Panel, Psi, beta, Q: Panel) -> Panel:
    """Q = beta Q + alpha V Psi  (mul!(Q, V, Psi, alpha, beta), src/block_gmres.jl:246)."""
    Pf = np.asfortranarray(Psi, dtype=np.float64)
    _ck(lib().khip_panel_gemm_nn(V.ctx._h, V.n, V.p, alpha, V.buf.ptr, Pf.ctypes.data_as(c_double_p), beta, Q.buf.ptr))
    return Q


def panel_mgs_(V, Q: Panel, accumulate_into=None):
    """Block Gram-Schmidt sweep of Q against the panels V[0..k) (src/block_gmres.jl:244-247) -> list of k p x p blocks."""
    k, p = len(V), Q.p
    ptrs = (C.c_void_p * max(k, 1))(*[v.buf.ptr for v in V])
    out = np.zeros((k, p * p)) if accumulate_into is None else np.ascontiguousarray(
        np.stack([np.asarray(b, dtype=np.float64).ravel(order="F") for b in accumulate_into]))
    _ck(lib().khip_panel_mgs(Q.ctx._h, Q.n, p, k, ptrs, Q.buf.ptr, out.ctypes.data_as(c_double_p),
                             0 if accumulate_into is None else 1))
    return [out[i].reshape(p, p, order="F").copy() for i in range(k)]


def panel_qr_(Q: Panel) -> np.ndarray:
    """Reduced QR in place (householder!(Q, R, tau), src/block_krylov_utils.jl:201-208); returns R."""
    R = np.zeros((Q.p, Q.p), order="F")
    _ck(lib().khip_panel_qr(Q.ctx._h, Q.n, Q.p, Q.buf.ptr, R.ctypes.data_as(c_double_p)))
    return R


def panel_multi_nn_(Vs, Ys, beta: float, X: Panel) -> Panel:
    """X <- beta X + sum_i V_i Y_i in the order i = 0..k-1 (src/block_gmres.jl:324-326 in one pass); Ys: k host p x p blocks."""
    k = len(Vs)
    ptrs = (C.c_void_p * max(k, 1))(*[v.buf.ptr for v in Vs])
    Y = np.ascontiguousarray(np.concatenate([np.asfortranarray(y, dtype=np.float64).ravel(order="F") for y in Ys])) if k else np.zeros(1)
    _ck(lib().khip_panel_multi_nn(X.ctx._h, X.n, X.p, k, ptrs, Y.ctypes.data_as(c_double_p), float(beta), X.buf.ptr))
    return X


def panel_qr_tau_(Q: Panel):
    """householder!(Q, R, tau): Q, R and tau as kgeqrf! + korgqr! leave them (LAPACK signs); returns (R, tau)."""
    R = np.zeros((Q.p, Q.p), order="F")
    tau = np.zeros(Q.p)
    _ck(lib().khip_panel_qr_tau(Q.ctx._h, Q.n, Q.p, Q.buf.ptr, R.ctypes.data_as(c_double_p), tau.ctypes.data_as(c_double_p)))
    return R, tau


def panel_norm(Q: Panel) -> float:
    r = C.c_double()
    _ck(lib().khip_panel_norm(Q.ctx._h, Q.n, Q.p, Q.buf.ptr, C.byref(r)))
    return r.value


def spmm_(A: CsrMatrix, X: Panel, Y: Panel) -> Panel:
    _ck(lib().khip_spmm(A.ctx._h, A._h, X.buf.ptr, Y.buf.ptr, X.p))
    return Y


class BlockGmresWorkspace(_Workspace):
    """BlockGmresWorkspace(m, n, p, SV, SM; memory = 5) (src/block_krylov_workspaces.jl:115-171).  Adopted form: the tall
    blocks X, W, V[i] are Panels of this side (the HIPMatrix of julia/KrylovHIP), the small blocks live in the library."""
    _prefix = "block_gmres"

    def __init__(self, ctx: Context, m: int, n: int, p: int, memory: int = 5, adopt: bool | None = None):
        self.ctx, self.m, self.n, self.p, self.memory = ctx, m, n, p, memory
        self.adopted = _adopt_default() if adopt is None else bool(adopt)
        self._h = C.c_void_p()
        if self.adopted:
            mem = 5 if memory <= 0 else memory
            mem = max(1, min(n // p, mem))                                     # memory = min(div(n, p), memory), :138
            t0 = time.perf_counter()
            self._pan = {k: Panel(ctx, n, p) for k in ("X", "W")}
            self.V = [Panel(ctx, n, p) for _ in range(mem)]
            self._alloc_s = time.perf_counter() - t0
            ptrs = (C.c_void_p * mem)(*[v.buf.ptr for v in self.V])
            _ck(lib().khip_block_gmres_workspace_adopt(ctx._h, m, n, p, mem, self._pan["X"].buf.ptr, self._pan["W"].buf.ptr, ptrs,
                                                       C.byref(self._h)))

            def grow(_ud):                                                     # push!(V, SM(undef, n, p)), src/block_gmres.jl:300-305
                try:
                    self.V.append(Panel(self.ctx, self.n, self.p))
                    return self.V[-1].buf.ptr
                except Exception as e:
                    sys.stderr.write(f"grow callback failed: {e}\n")
                    return None
            self._grow = GROW_FN(grow)
            _ck(lib().khip_block_gmres_workspace_set_grow(self._h, self._grow, None))
        else:
            _ck(lib().khip_block_gmres_workspace_create(ctx._h, m, n, p, memory, C.byref(self._h)))

    def _allocate_panel_if(self, cond: bool, name: str):
        if cond and self.adopted and name not in self._pan:
            t0 = time.perf_counter()
            P = Panel(self.ctx, self.n, self.p)
            self._alloc_s += time.perf_counter() - t0
            self._pan[name] = P
            _ck(lib().khip_block_gmres_workspace_adopt_panel(self._h, name.encode(), P.buf.ptr))

    @property
    def X(self) -> np.ndarray:
        if self.adopted:
            return self._pan["X"].to_host()
        col = self.ctx.empty(self.n * self.p)
        _ck(lib().khip_block_gmres_get_X(self._h, col.ptr))
        return col.to_host().reshape(self.p, self.n).T.copy()

    x = X

    def warm_start_(self, X0):
        if self.adopted:                                                       # allocate_if(true, ws, :ΔX, ...); copyto!(ws.ΔX, X0)
            self._allocate_panel_if(True, "dX")
            P0 = Panel.from_host(self.ctx, np.asarray(X0, dtype=np.float64))
            kcopy_(P0.n_pad * P0.p, self._pan["dX"].buf, P0.buf)
            _ck(lib().khip_block_gmres_warm_start_panel(self._h, self._pan["dX"].buf.ptr))
            return self
        col = self.ctx.array(np.asfortranarray(np.asarray(X0, dtype=np.float64)).ravel(order="F"))
        _ck(lib().khip_block_gmres_warm_start(self._h, col.ptr))
        return self

    @property
    def nbytes(self):
        """Workspace bytes as test/test_allocations.jl:734-761 counts them (includes `nbytes_extra`)."""
        return lib().khip_block_gmres_workspace_bytes(self._h, None)

    @property
    def nbytes_extra(self):
        e = C.c_size_t()
        lib().khip_block_gmres_workspace_bytes(self._h, C.byref(e))
        return e.value


def _make_block_operator(ctx, op, n, p, keep):
    """CsrMatrix | callable(X: Panel, Y: Panel) | None for the block solver (callbacks see row-major panels)."""
    if op is None or isinstance(op, CsrMatrix):
        return _make_operator(ctx, op, n, keep)

    def thunk(_self, xp, yp):
        try:
            X, Y = Panel.__new__(Panel), Panel.__new__(Panel)
            for P_, ptr in ((X, xp), (Y, yp)):
                P_.ctx, P_.n, P_.p, P_.n_pad = ctx, n, p, panel_rows(n)
                P_.buf = DeviceVector(ctx, P_.n_pad * p, ptr=ptr)
            op(X, Y)
            return 0
        except Exception as e:
            sys.stderr.write(f"block operator callback failed: {e}\n")
            return 1
    fn = APPLY_FN(thunk)
    co = COperator()
    co.csr, co.apply = None, fn
    keep.extend([fn, co])
    return C.byref(co)


def block_gmres_(ws: BlockGmresWorkspace, A, B_colmajor: DeviceVector, M=None, N=None, **kw):
    """block_gmres!(workspace, A, B; restart, reorthogonalization, atol, rtol, itmax, history, ...)
    (src/block_gmres.jl:110-358); B is an n x p column-major device array."""
    keep = []
    opts = _make_options(keep=keep, ws=ws, **kw)
    ops = (_make_block_operator(ws.ctx, A, ws.n, ws.p, keep), _make_block_operator(ws.ctx, M, ws.n, ws.p, keep),
           _make_block_operator(ws.ctx, N, ws.n, ws.p, keep))
    if ws.adopted:
        # the binding's matrix type IS a panel (HIPMatrix(B::Matrix) converts once, at construction): B is read in place
        ws._allocate_panel_if(M is not None, "Q")                                          # src/block_gmres.jl:146-147
        ws._allocate_panel_if(N is not None, "P")
        ws._allocate_panel_if(bool(kw.get("restart")), "dX")
        if isinstance(B_colmajor, Panel):
            Bp = B_colmajor
        else:
            Bp = Panel(ws.ctx, ws.n, ws.p)
            _ck(lib().khip_panel_from_colmajor(ws.ctx._h, ws.n, ws.p, _p(B_colmajor), Bp.buf.ptr))
        keep.append(Bp)
        rc = lib().khip_block_gmres_solve_panel(ws._h, *ops, Bp.buf.ptr, C.byref(opts))
    else:
        if isinstance(B_colmajor, Panel):            # a library-owned workspace takes B column-major (khip_block_gmres_solve)
            col = ws.ctx.empty(ws.n * ws.p)
            _ck(lib().khip_panel_to_colmajor(ws.ctx._h, ws.n, ws.p, B_colmajor.buf.ptr, col.ptr))
            keep.append(col)
            B_colmajor = col
        rc = lib().khip_block_gmres_solve(ws._h, *ops, _p(B_colmajor), C.byref(opts))
    return _finish(ws, rc)


def block_gmres(A, B, X0=None, ctx=None, **kw):
    """Out-of-place block_gmres(A, B[, X0]; memory, kwargs...) for a host n x p array B -> (X, stats, workspace)
    (src/interface.jl:247-290)."""
    return krylov_solve("block_gmres", A, B, X0, ctx=ctx, **kw)


_INPLACE.update({CgWorkspace: ("cg", cg_), GmresWorkspace: ("gmres", gmres_), BicgstabWorkspace: ("bicgstab", bicgstab_),
                 BlockGmresWorkspace: ("block_gmres", block_gmres_)})
