"""ctypes binding of the CPU oracle (oracle/krylov_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never from krylov.jl_amd/.  The oracle restates the
reference's CPU path (src/cg.jl, src/gmres.jl, src/bicgstab.jl, src/block_gmres.jl,
src/krylov_utils.jl) -- see krylov_oracle.h for the per-function citations.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libkrylov_oracle.so")

c_double_p = C.POINTER(C.c_double)
MATVEC = C.CFUNCTYPE(None, c_double_p, c_double_p, C.c_void_p)
BLOCK_MATVEC = C.CFUNCTYPE(None, c_double_p, c_double_p, C.c_int, C.c_void_p)
CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)


class Csr(C.Structure):
    _fields_ = [("n", C.c_int64), ("nnz", C.c_int64), ("rowptr", C.POINTER(C.c_int64)),
                ("col", C.POINTER(C.c_int32)), ("val", c_double_p)]


class Options(C.Structure):
    _fields_ = [("atol", C.c_double), ("rtol", C.c_double), ("itmax", C.c_int),
                ("timemax", C.c_double), ("history", C.c_int), ("radius", C.c_double),
                ("linesearch", C.c_int), ("restart", C.c_int), ("reorthogonalization", C.c_int),
                ("ldiv", C.c_int), ("callback", CALLBACK), ("callback_data", C.c_void_p)]


class Stats(C.Structure):
    _fields_ = [("niter", C.c_int), ("solved", C.c_int), ("inconsistent", C.c_int),
                ("indefinite", C.c_int), ("npcCount", C.c_int), ("timer", C.c_double),
                ("status", C.c_char * 96), ("residuals", c_double_p), ("nres", C.c_int),
                ("cap", C.c_int), ("error", C.c_char * 160)]


class CgWs(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64)] + \
               [(k, c_double_p) for k in ("dx", "x", "r", "npc_dir", "p", "Ap", "z")] + \
               [("warm_start", C.c_int), ("stats", Stats)]


class GmresWs(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("mem", C.c_int), ("nV", C.c_int)] + \
               [(k, c_double_p) for k in ("dx", "x", "w", "p", "q")] + \
               [("V", C.POINTER(c_double_p))] + \
               [(k, c_double_p) for k in ("c", "s", "z", "R")] + \
               [("capR", C.c_int), ("capcs", C.c_int), ("capz", C.c_int), ("inner_iter", C.c_int),
                ("warm_start", C.c_int), ("stats", Stats)]


class BicgstabWs(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64)] + \
               [(k, c_double_p) for k in ("dx", "x", "r", "p", "v", "s", "qd", "yz", "t")] + \
               [("warm_start", C.c_int), ("stats", Stats)]


class BlockGmresWs(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("p", C.c_int), ("mem", C.c_int), ("nV", C.c_int)] + \
               [(k, c_double_p) for k in ("dX", "X", "W", "P", "Q", "C", "D")] + \
               [(k, C.POINTER(c_double_p)) for k in ("V", "Z", "R", "H", "tau")] + \
               [("nR", C.c_int), ("nH", C.c_int), ("warm_start", C.c_int), ("stats", Stats)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (and, where /root/reference exists, oracle/_ref)."""
    src = os.path.join(_HERE, "krylov_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "krylov_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp, i64 = c_double_p, C.c_int64
    sig = {
        "ko_csr_poisson3d": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Csr)]),
        "ko_csr_kron_unsymmetric": (C.c_int, [C.c_int, C.POINTER(Csr)]),
        "ko_csr_stencil27_unsym": (C.c_int, [C.c_int, C.POINTER(Csr)]),
        "ko_csr_banded_random": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(Csr)]),
        "ko_csr_tridiag": (C.c_int, [C.c_int, C.c_double, C.c_double, C.c_double, C.POINTER(Csr)]),
        "ko_csr_free": (None, [C.POINTER(Csr)]),
        "ko_csr_row_slice": (C.c_int, [C.POINTER(Csr), i64, i64, C.POINTER(Csr)]),
        "ko_spmv": (None, [C.POINTER(Csr), dp, dp]),
        "ko_ilu0": (C.c_int, [C.POINTER(Csr), dp, C.POINTER(C.c_int64)]),
        "ko_ilu0_solve": (None, [C.POINTER(Csr), dp, C.POINTER(C.c_int64), dp, dp]),
        "ko_spmv_omp": (None, [C.POINTER(Csr), dp, dp]),
        "ko_spmm": (None, [C.POINTER(Csr), dp, dp, C.c_int]),
        "ko_dot": (C.c_double, [i64, dp, dp]),
        "ko_dot_omp": (C.c_double, [i64, dp, dp]),
        "ko_nrm2": (C.c_double, [i64, dp]),
        "ko_scal": (None, [i64, C.c_double, dp]),
        "ko_div": (None, [i64, dp, C.c_double]),
        "ko_copy": (None, [i64, dp, dp]),
        "ko_scalcopy": (None, [i64, dp, C.c_double, dp]),
        "ko_divcopy": (None, [i64, dp, dp, C.c_double]),
        "ko_axpy": (None, [i64, C.c_double, dp, dp]),
        "ko_axpby": (None, [i64, C.c_double, dp, C.c_double, dp]),
        "ko_axpy_omp": (None, [i64, C.c_double, dp, dp]),
        "ko_axpby_omp": (None, [i64, C.c_double, dp, C.c_double, dp]),
        "ko_fill": (None, [i64, dp, C.c_double]),
        "ko_ref": (None, [i64, dp, dp, C.c_double, C.c_double]),
        "ko_set_threads": (None, [C.c_int]),
        "ko_set_dot_mode": (None, [C.c_int]),
        "ko_get_dot_mode": (C.c_int, []),
        "ko_get_threads": (C.c_int, []),
        "ko_sym_givens": (None, [C.c_double, C.c_double, dp, dp, dp]),
        "ko_roots_quadratic": (C.c_int, [C.c_double, C.c_double, C.c_double, C.c_int, dp, dp]),
        "ko_to_boundary": (C.c_int, [i64, dp, dp, C.c_double, C.c_int, C.c_double, C.c_double, dp, dp]),
        "ko_default_options": (Options, []),
        "ko_cg_workspace_create": (C.POINTER(CgWs), [i64, i64]),
        "ko_gmres_workspace_create": (C.POINTER(GmresWs), [i64, i64, C.c_int]),
        "ko_bicgstab_workspace_create": (C.POINTER(BicgstabWs), [i64, i64]),
        "ko_block_gmres_workspace_create": (C.POINTER(BlockGmresWs), [i64, i64, C.c_int, C.c_int]),
        "ko_cg_workspace_free": (None, [C.POINTER(CgWs)]),
        "ko_gmres_workspace_free": (None, [C.POINTER(GmresWs)]),
        "ko_bicgstab_workspace_free": (None, [C.POINTER(BicgstabWs)]),
        "ko_block_gmres_workspace_free": (None, [C.POINTER(BlockGmresWs)]),
        "ko_cg_warm_start": (None, [C.POINTER(CgWs), dp]),
        "ko_gmres_warm_start": (None, [C.POINTER(GmresWs), dp]),
        "ko_bicgstab_warm_start": (None, [C.POINTER(BicgstabWs), dp]),
        "ko_block_gmres_warm_start": (None, [C.POINTER(BlockGmresWs), dp]),
        "ko_cg": (C.c_int, [C.POINTER(CgWs), MATVEC, MATVEC, C.c_void_p, dp, C.POINTER(Options)]),
        "ko_gmres": (C.c_int, [C.POINTER(GmresWs), MATVEC, MATVEC, MATVEC, C.c_void_p, dp, C.POINTER(Options)]),
        "ko_bicgstab": (C.c_int, [C.POINTER(BicgstabWs), MATVEC, MATVEC, MATVEC, C.c_void_p, dp, dp,
                                  C.POINTER(Options)]),
        "ko_block_gmres": (C.c_int, [C.POINTER(BlockGmresWs), BLOCK_MATVEC, BLOCK_MATVEC, BLOCK_MATVEC,
                                     C.c_void_p, dp, C.POINTER(Options)]),
        "ko_cg_stencil7": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Options), C.POINTER(Stats), C.c_int,
                                     C.POINTER(C.c_int64), dp]),
        "ko_stats_init": (None, [C.POINTER(Stats)]),
        "ko_stats_free": (None, [C.POINTER(Stats)]),
        "ko_cg_bench": (C.c_double, [C.POINTER(Csr), C.c_int, C.c_int, dp]),
        "ko_geqrf": (None, [C.c_int, C.c_int, dp, C.c_int, dp]),
        "ko_orgqr": (None, [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]),
        "ko_ormqr_LT": (None, [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, dp, C.c_int]),
        "ko_householder": (None, [C.c_int, C.c_int, dp, dp, dp, C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    # expose the C matvec adaptors as raw function pointers usable as MATVEC arguments
    L.csr_matvec = C.cast(L.ko_csr_matvec, MATVEC)
    L.csr_matvec_omp = C.cast(L.ko_csr_matvec_omp, MATVEC)
    L.stencil7_matvec = C.cast(L.ko_stencil7_matvec, MATVEC)
    L.stencil7_matvec_omp = C.cast(L.ko_stencil7_matvec_omp, MATVEC)
    L.csr_block_matvec = C.cast(L.ko_csr_block_matvec, BLOCK_MATVEC)
    L.csr_block_matvec_omp = C.cast(L.ko_csr_block_matvec_omp, BLOCK_MATVEC)
    _lib = L
    return L


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] or a.flags["F_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


# True: CsrMatrix operators are applied with the row-parallel OpenMP loops (ko_spmv_omp).  Rows are independent, so
# every y value is the one the serial loop produces; only the full-size goldens (tests/golden/make_scale_golden.py)
# switch this on.  dot / axpy / QR arithmetic is unaffected.
PARALLEL_MATVEC = False

NULL_MATVEC = C.cast(None, MATVEC)
NULL_BLOCK_MATVEC = C.cast(None, BLOCK_MATVEC)


class CsrMatrix:
    """Owning wrapper of a ko_csr; .rowptr/.col/.val are numpy views of the C arrays."""

    def __init__(self, c: Csr, keep=None):
        self.c = c
        self._keep = keep          # numpy arrays backing the struct (from_arrays); None: the C side owns them
        self.n, self.nnz = int(c.n), int(c.nnz)
        self.rowptr = np.ctypeslib.as_array(c.rowptr, shape=(self.n + 1,))
        self.col = np.ctypeslib.as_array(c.col, shape=(max(self.nnz, 1),))[: self.nnz]
        self.val = np.ctypeslib.as_array(c.val, shape=(max(self.nnz, 1),))[: self.nnz]

    def __del__(self):
        try:
            if self._keep is None:
                lib().ko_csr_free(C.byref(self.c))
        except Exception:
            pass

    @classmethod
    def from_arrays(cls, rowptr, col, val):
        """Square CSR over caller-provided arrays (0-based; int64 row pointers, int32 columns)."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        c = Csr()
        c.n, c.nnz = rowptr.size - 1, int(rowptr[-1])
        c.rowptr = rowptr.ctypes.data_as(C.POINTER(C.c_int64))
        c.col = col.ctypes.data_as(C.POINTER(C.c_int32))
        c.val = val.ctypes.data_as(c_double_p)
        return cls(c, keep=(rowptr, col, val))

    def ptr(self):
        return C.cast(C.pointer(self.c), C.c_void_p)

    def matvec(self, x):
        y = np.empty(self.n)
        lib().ko_spmv(C.byref(self.c), _dp(np.ascontiguousarray(x, dtype=np.float64)), _dp(y))
        return y

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val.copy(), self.col.copy(), self.rowptr.copy()), shape=(self.n, self.n))

    def row_slice(self, r0, r1):
        out = Csr()
        rc = lib().ko_csr_row_slice(C.byref(self.c), r0, r1, C.byref(out))
        assert rc == 0
        return CsrMatrix(out)


class Ilu0:
    """ILU(0) / IC(0) of a CsrMatrix (krylov_oracle.h): .lu values on A's pattern, .diag positions; solve(x) = U\\(L\\x)."""

    def __init__(self, A: CsrMatrix):
        self.A = A
        self.lu = np.empty(max(A.nnz, 1))
        self.diag = np.empty(max(A.n, 1), dtype=np.int64)
        rc = lib().ko_ilu0(C.byref(A.c), _dp(self.lu), self.diag.ctypes.data_as(C.POINTER(C.c_int64)))
        if rc != 0:
            raise ZeroDivisionError(f"ilu0: missing or zero pivot in row {-rc - 1}")
        self.lu = self.lu[: A.nnz]

    def solve(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.A.n)
        lu = np.ascontiguousarray(self.lu)
        lib().ko_ilu0_solve(C.byref(self.A.c), _dp(lu), self.diag.ctypes.data_as(C.POINTER(C.c_int64)), _dp(x), _dp(y))
        return y


def _gen(fn, *args):
    c = Csr()
    rc = fn(*args, C.byref(c))
    if rc != 0:
        raise MemoryError(f"oracle generator failed rc={rc}")
    return CsrMatrix(c)


def poisson3d(n1, n2=None, n3=None):
    """get_div_grad(n1,n2,n3) -- test/get_div_grad.jl:8-25."""
    return _gen(lib().ko_csr_poisson3d, n1, n2 or n1, n3 or n1)


def kron_unsymmetric(n1):
    """kron_unsymmetric(n) matrix -- test/test_utils.jl:160-169 (b = A*ones)."""
    return _gen(lib().ko_csr_kron_unsymmetric, n1)


def stencil27_unsym(n1):
    return _gen(lib().ko_csr_stencil27_unsym, n1)


def banded_random(n, half_band=13, links=3, seed=1, unsym=False, dense_rows=0):
    """The "banded + random, fixed seed" non-stencil benchmark operator (krylov.jl_amd/csrc/gen_irregular.cpp)."""
    return _gen(lib().ko_csr_banded_random, n, half_band, links, seed, 1 if unsym else 0, dense_rows)


def tridiag(n, lo, di, up):
    return _gen(lib().ko_csr_tridiag, n, lo, di, up)


def make_options(atol=None, rtol=None, itmax=0, timemax=None, history=False, radius=0.0,
                 linesearch=False, restart=False, reorthogonalization=False, callback=None):
    o = lib().ko_default_options()
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
    if callback is not None:
        o.callback = callback
    return o


class Result:
    def __init__(self, x, st: Stats, rc: int):
        self.x = x
        self.rc = rc
        self.niter = st.niter
        self.solved = bool(st.solved)
        self.inconsistent = bool(st.inconsistent)
        self.indefinite = bool(st.indefinite)
        self.npcCount = st.npcCount
        self.status = st.status.decode("utf-8")
        self.error = st.error.decode("utf-8")
        self.timer = st.timer
        self.residuals = np.array([st.residuals[i] for i in range(st.nres)]) if st.nres else np.zeros(0)


def _wrap_matvec(op):
    """op: CsrMatrix | python callable(x)->y | None  ->  (MATVEC, userdata, keepalive)."""
    if op is None:
        return NULL_MATVEC, None, None
    if isinstance(op, CsrMatrix):
        return lib().csr_matvec, op.ptr(), op
    raise TypeError("use _PyOp for python callables")


class _PyOps:
    """Bundle python callables A, M, N behind one userdata-less set of C callbacks."""

    def __init__(self, n, A, M=None, N=None):
        self.n = n

        def mk(f):
            if f is None:
                return NULL_MATVEC

            def cb(xp, yp, _ud):
                x = np.ctypeslib.as_array(xp, shape=(n,))
                y = np.ctypeslib.as_array(yp, shape=(n,))
                y[:] = f(x)
            return MATVEC(cb)
        if isinstance(A, CsrMatrix):
            self.A, self.ud, self.keep = (lib().csr_matvec_omp if PARALLEL_MATVEC else lib().csr_matvec), A.ptr(), A
        else:
            self.A, self.ud, self.keep = mk(A), None, None
        # when A is a CsrMatrix the userdata is the csr; python M/N ignore userdata
        self.M, self.N = mk(M), mk(N)


def cg(A, b, M=None, x0=None, **kw):
    """cg!(ws, A, b; M, atol, rtol, itmax, radius, linesearch, history) -- src/cg.jl:120-291."""
    L = lib()
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = b.size
    ops = _PyOps(n, A, M)
    ws = L.ko_cg_workspace_create(n, n)
    if x0 is not None:
        L.ko_cg_warm_start(ws, _dp(np.ascontiguousarray(x0, dtype=np.float64)))
    o = make_options(**kw)
    rc = L.ko_cg(ws, ops.A, ops.M, ops.ud, _dp(b), C.byref(o))
    x = np.ctypeslib.as_array(ws.contents.x, shape=(n,)).copy()
    res = Result(x, ws.contents.stats, rc)
    L.ko_cg_workspace_free(ws)
    return res


class Stencil7(C.Structure):
    """ko_stencil7: the matrix-free get_div_grad(n1,n2,n3) operator (test/get_div_grad.jl:8-25)."""
    _fields_ = [("n1", C.c_int), ("n2", C.c_int), ("n3", C.c_int)]

    def matvec(self, x, parallel=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        (lib().ko_stencil7_matvec_omp if parallel else lib().ko_stencil7_matvec)(_dp(x), _dp(y), C.cast(C.pointer(self), C.c_void_p))
        return y


def cg_stencil7(n1, n2=None, n3=None, x_index=(), progress=None, **kw):
    """cg! on the MATRIX-FREE get_div_grad(n1,n2,n3), b = ones (ko_cg_stencil7: ko_cg itself, the operator computed
    from the grid indices) -- the cfg-4 oracle.  Returns a Result whose .x holds only the entries x_index.
    progress: callable(iteration) invoked once per iteration (through the solver's callback)."""
    L = lib()
    o = make_options(**kw)
    keep = None
    if progress is not None:
        count = [0]

        def cb(_ws, _data):
            count[0] += 1
            progress(count[0])
            return 0
        keep = CALLBACK(cb)
        o.callback = keep
    st = Stats()
    L.ko_stats_init(C.byref(st))
    idx = np.ascontiguousarray(x_index, dtype=np.int64)
    xs = np.zeros(max(idx.size, 1))
    rc = L.ko_cg_stencil7(n1, n2 or n1, n3 or n1, C.byref(o), C.byref(st), idx.size,
                          idx.ctypes.data_as(C.POINTER(C.c_int64)), _dp(xs))
    res = Result(xs[: idx.size], st, rc)
    L.ko_stats_free(C.byref(st))
    return res


def gmres(A, b, M=None, N=None, x0=None, memory=20, **kw):
    """gmres!(ws, A, b; M, N, restart, reorthogonalization, ...) -- src/gmres.jl:121-384."""
    L = lib()
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = b.size
    ops = _PyOps(n, A, M, N)
    ws = L.ko_gmres_workspace_create(n, n, memory)
    if x0 is not None:
        L.ko_gmres_warm_start(ws, _dp(np.ascontiguousarray(x0, dtype=np.float64)))
    o = make_options(**kw)
    rc = L.ko_gmres(ws, ops.A, ops.M, ops.N, ops.ud, _dp(b), C.byref(o))
    x = np.ctypeslib.as_array(ws.contents.x, shape=(n,)).copy()
    res = Result(x, ws.contents.stats, rc)
    L.ko_gmres_workspace_free(ws)
    return res


def bicgstab(A, b, c=None, M=None, N=None, x0=None, **kw):
    """bicgstab!(ws, A, b; c, M, N, ...) -- src/bicgstab.jl:125-277."""
    L = lib()
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = b.size
    ops = _PyOps(n, A, M, N)
    ws = L.ko_bicgstab_workspace_create(n, n)
    if x0 is not None:
        L.ko_bicgstab_warm_start(ws, _dp(np.ascontiguousarray(x0, dtype=np.float64)))
    o = make_options(**kw)
    cptr = _dp(np.ascontiguousarray(c, dtype=np.float64)) if c is not None else None
    rc = L.ko_bicgstab(ws, ops.A, ops.M, ops.N, ops.ud, _dp(b), cptr, C.byref(o))
    x = np.ctypeslib.as_array(ws.contents.x, shape=(n,)).copy()
    res = Result(x, ws.contents.stats, rc)
    L.ko_bicgstab_workspace_free(ws)
    return res


def block_gmres(A, B, X0=None, memory=5, M=None, N=None, **kw):
    """block_gmres!(ws, A, B; M, N, restart, reorthogonalization, ...) -- src/block_gmres.jl:110-358.
    B is n-by-p (any layout; converted to column-major); M / N are python callables on n-by-p arrays."""
    L = lib()
    Bf = np.asfortranarray(B, dtype=np.float64)
    n, p = Bf.shape

    def mk(f):
        if f is None:
            return NULL_BLOCK_MATVEC

        def cb(Xp, Yp, pp, _ud):
            X = np.ctypeslib.as_array(Xp, shape=(pp, n)).T
            Y = np.ctypeslib.as_array(Yp, shape=(pp, n)).T
            Y[:, :] = f(X)
        return BLOCK_MATVEC(cb)
    if isinstance(A, CsrMatrix):
        fa, ud = (L.csr_block_matvec_omp if PARALLEL_MATVEC else L.csr_block_matvec), A.ptr()
    else:
        fa, ud = mk(A), None
    fm, fn = mk(M), mk(N)
    ws = L.ko_block_gmres_workspace_create(n, n, p, memory)
    if X0 is not None:
        L.ko_block_gmres_warm_start(ws, _dp(np.asfortranarray(X0, dtype=np.float64)))
    o = make_options(**kw)
    rc = L.ko_block_gmres(ws, fa, fm, fn, ud, _dp(Bf), C.byref(o))
    X = np.ctypeslib.as_array(ws.contents.X, shape=(p, n)).T.copy()
    res = Result(X, ws.contents.stats, rc)
    L.ko_block_gmres_workspace_free(ws)
    return res


# ---- thin numpy-facing primitives (used by the parity tests) -------------------

def dot(x, y):
    return lib().ko_dot(x.size, _dp(x), _dp(y))


def nrm2(x):
    return lib().ko_nrm2(x.size, _dp(x))


def axpy(s, x, y):
    lib().ko_axpy(x.size, s, _dp(x), _dp(y))
    return y


def axpby(s, x, t, y):
    lib().ko_axpby(x.size, s, _dp(x), t, _dp(y))
    return y


def sym_givens(a, b):
    c, s, r = C.c_double(), C.c_double(), C.c_double()
    lib().ko_sym_givens(a, b, C.byref(c), C.byref(s), C.byref(r))
    return c.value, s.value, r.value


def roots_quadratic(q2, q1, q0, nitref=1):
    r1, r2 = C.c_double(), C.c_double()
    rc = lib().ko_roots_quadratic(q2, q1, q0, nitref, C.byref(r1), C.byref(r2))
    if rc:
        raise ValueError("The quadratic `q` doesn't have real roots.")
    return r1.value, r2.value


def to_boundary(x, d, radius, flip=False, xNorm2=0.0, dNorm2=0.0):
    s1, s2 = C.c_double(), C.c_double()
    rc = lib().ko_to_boundary(x.size, _dp(x), _dp(d), radius, int(flip), xNorm2, dNorm2,
                              C.byref(s1), C.byref(s2))
    if rc:
        raise ValueError(f"to_boundary rc={rc}")
    return s1.value, s2.value
