"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/krylov_hip.h declares, fails loudly without a GPU, and the host-only partition / halo-plan
helpers are correct.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def K():
    import krylov_jl_amd as K
    if not os.path.exists(K.LIB_PATH):
        K.build()
    K.lib()
    return K


def _declared_symbols(headers=("krylov_hip.h", "krylov_hip_test.h")):
    out = set()
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(khip_[a-z0-9_]+)\s*\(", src))
    return sorted(out)


def test_test_hooks_are_not_in_the_public_header():
    """khip_test_* are test exports, declared in include/krylov_hip_test.h, not in the shipped ABI."""
    pub = _declared_symbols(("krylov_hip.h", "krylov_hip_ext.h"))
    assert not [s for s in pub if s.startswith("khip_test_")]
    hooks = _declared_symbols(("krylov_hip_test.h",))
    assert len([s for s in hooks if s.startswith("khip_test_")]) >= 9
    glue = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "khip_test_" not in glue


def test_library_exports_every_declared_symbol(K):
    import ctypes
    L = ctypes.CDLL(K.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) > 60
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    # typedef'd callback names are not functions
    fn_types = {"khip_apply_fn", "khip_callback_fn"}
    bound = set(K.SIGNATURES)
    assert (set(declared) - fn_types) <= bound | fn_types, sorted(set(declared) - bound - fn_types)


def test_no_cpu_fallback(K):
    if K.gpu_available():
        pytest.skip("GPU present")
    with pytest.raises(K.KhipError) as e:
        K.Context()
    assert "no CPU fallback" in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "krylov.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "krylov_oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)


def test_row_partition(K):
    assert K.row_partition(10, 3) == [0, 3, 6, 10]
    p = K.row_partition(512 ** 3, 8)
    assert p[0] == 0 and p[-1] == 512 ** 3 and all(b - a == 512 ** 3 // 8 for a, b in zip(p, p[1:]))


def test_halo_plan_poisson_slabs(K, oracle):
    """Slab partition of the 7-point grid: each rank needs exactly the neighbouring planes
    (SURVEY.md section 8e), and the send lists mirror the peers' receive lists."""
    n1, G = 8, 4
    A = oracle.poisson3d(n1)
    n = A.n
    starts = K.row_partition(n, G)
    ghosts = []
    for g in range(G):
        sl = A.row_slice(starts[g], starts[g + 1])
        ghosts.append(K.ghost_columns_host(sl.rowptr, sl.col, starts[g]))
    plane = n1 * n1
    for g in range(G):
        exp = []
        if g > 0:
            exp += list(range(starts[g] - plane, starts[g]))
        if g < G - 1:
            exp += list(range(starts[g + 1], starts[g + 1] + plane))
        assert ghosts[g].tolist() == exp
    plans = [K.halo_plan_host(g, G, starts, ghosts) for g in range(G)]
    for g in range(G):
        recv_off, send_off, send_idx = plans[g]
        assert recv_off[-1] == len(ghosts[g])
        for r in range(G):
            # what g sends to r == what r expects from g, as global indices
            sent = send_idx[send_off[r]:send_off[r + 1]] + starts[g]
            r_recv_off = plans[r][0]
            expected = ghosts[r][r_recv_off[g]:r_recv_off[g + 1]]
            assert sent.tolist() == expected.tolist()


def test_halo_plan_random_matrix(K):
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    n, G = 97, 3
    S = (sp.random(n, n, density=0.08, random_state=3, format="csr") + sp.identity(n)).tocsr()
    S.sort_indices()
    starts = K.row_partition(n, G)
    ghosts = []
    for g in range(G):
        blk = S[starts[g]:starts[g + 1]]
        ghosts.append(K.ghost_columns_host(blk.indptr.astype(np.int64), blk.indices.astype(np.int32), starts[g]))
        ref = np.unique(blk.indices[(blk.indices < starts[g]) | (blk.indices >= starts[g + 1])])
        assert ghosts[g].tolist() == ref.tolist()
    x = rng.standard_normal(n)
    y = np.zeros(n)
    plans = [K.halo_plan_host(g, G, starts, ghosts) for g in range(G)]
    # emulate the exchange + local product with [owned | ghost] column numbering
    for g in range(G):
        recv_off, _, _ = plans[g]
        ghost_vals = np.zeros(len(ghosts[g]))
        for r in range(G):
            _, so, si = plans[r]
            seg = x[starts[r]:starts[r + 1]][si[so[g]:so[g + 1]]]
            ghost_vals[recv_off[r]:recv_off[r + 1]] = seg
        blk = S[starts[g]:starts[g + 1]].tocsr()
        m = starts[g + 1] - starts[g]
        xe = np.concatenate([x[starts[g]:starts[g + 1]], ghost_vals])
        cols = blk.indices.copy()
        own = (cols >= starts[g]) & (cols < starts[g + 1])
        loc = np.where(own, cols - starts[g], m + np.searchsorted(ghosts[g], cols))
        yl = np.zeros(m)
        np.add.at(yl, np.repeat(np.arange(m), np.diff(blk.indptr)), blk.data * xe[loc])
        y[starts[g]:starts[g + 1]] = yl
    assert np.allclose(y, S @ x, atol=1e-13)


# ---- the PRODUCT's scalar helpers against the reference's exact known answers (test/test_aux.jl:3-79) --------------
# (host code of libkrylov_hip.so, reached through test-only exports: no GPU needed, no oracle involved)

def _sym_givens(L, a, b):
    c, s, r = C.c_double(), C.c_double(), C.c_double()
    assert L.khip_test_sym_givens(a, b, C.byref(c), C.byref(s), C.byref(r)) == 0
    return c.value, s.value, r.value


def _roots(L, q2, q1, q0, nitref=1):
    r1, r2 = C.c_double(), C.c_double()
    rc = L.khip_test_roots_quadratic(q2, q1, q0, nitref, C.byref(r1), C.byref(r2))
    if rc != 0:
        raise ValueError(L.khip_last_error().decode())
    return r1.value, r2.value


def test_product_sym_givens_known_answers():
    """test/test_aux.jl:3-34 against csrc/solvers.cpp's sym_givens (the copy gmres! runs)."""
    import krylov_jl_amd as K
    L = K.lib()
    g = lambda a, b: _sym_givens(L, a, b)
    assert g(0.0, 0.0) == (1.0, 0.0, 0.0)
    a = 3.14
    assert g(a, 0.0) == (1.0, 0.0, a)
    assert g(-a, 0.0) == (-1.0, 0.0, a)
    assert g(0.0, a) == (0.0, 1.0, a)
    assert g(0.0, -a) == (0.0, -1.0, a)
    for (x, y) in [(3.0, 4.0), (-4.0, 3.0), (1e-3, -7.0), (5.0, -1e-9), (-2.0, -2.0)]:
        c, s, r = g(x, y)
        assert abs(c * x + s * y - r) <= 1e-15 * abs(r)          # [c s; s -c] [x; y] = [r; 0]
        assert abs(s * x - c * y) <= 1e-15 * abs(r)
        assert abs(c * c + s * s - 1) <= 4e-16


def test_product_roots_quadratic_known_answers():
    """test/test_aux.jl:36-79 against csrc/solvers.cpp's roots_quadratic (the copy cg!'s trust region runs)."""
    import math
    import pytest
    import krylov_jl_amd as K
    L = K.lib()
    rq = lambda *a, **k: _roots(L, *a, **k)
    assert rq(0.0, 0.0, 0.0) == (0.0, 0.0)
    with pytest.raises(ValueError):
        rq(0.0, 0.0, 1.0)
    assert rq(0.0, 3.14, -1.0) == (1.0 / 3.14, 1.0 / 3.14)
    with pytest.raises(ValueError):
        rq(1.0, 0.0, 1.0)
    assert rq(1.0, 0.0, 0.0) == (0.0, 0.0)
    r = rq(1.0, 3.0, 2.0)
    assert math.isclose(r[0], -2.0) and math.isclose(r[1], -1.0)
    with pytest.raises(ValueError):
        rq(1.0e8, 1.0, 1.0)
    assert rq(-1.0e-8, 1.0e5, 1.0, nitref=0) == (1.0e13, 0.0)              # ill-conditioned quadratic
    assert rq(-1.0e-8, 1.0e5, 1.0, nitref=1) == (1.0e13, -1.0e-05)         # "iterative refinement is crucial!"
    for nit in (0, 1):
        r = rq(-1.0e-7, 1.0, 1.0, nitref=nit)
        assert math.isclose(r[0], 1.0e7, rel_tol=1e-6) and math.isclose(r[1], -1.0, rel_tol=1e-6)


# ---- the Julia glue of INTEGRATION.md against the C header --------------------------------------------

def _c_declarations():
    """name -> (return type, [parameter types]) of every function declared in include/*.h, normalised: comments and
    parameter names dropped, `const` dropped, whitespace collapsed ("double *", "void * *", "int64_t")."""
    import re
    decls = {}
    for hdr in ("krylov_hip.h", "krylov_hip_ext.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", " ", txt)
        for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\**)\s*(khip_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", txt, flags=re.S):
            ret, name, params = m.group(1), m.group(2), m.group(3)

            def norm(t, is_param):
                t = re.sub(r"\bconst\b", " ", t)
                t = t.replace("*", " * ")
                toks = t.split()
                if is_param and toks and toks[-1] != "*" and len(toks) > 1 and re.fullmatch(r"[A-Za-z_][A-Za-z0-9_]*", toks[-1]):
                    toks = toks[:-1]                                   # the parameter's name
                return " ".join(toks)
            plist = [] if params.strip() in ("", "void") else [norm(p, True) for p in params.split(",")]
            decls[name] = (norm(ret, False), plist)
    return decls


JULIA_SRC = os.path.join(ROOT, "julia", "KrylovHIP", "src", "KrylovHIP.jl")
JULIA_TEST = os.path.join(ROOT, "julia", "KrylovHIP", "test", "runtests.jl")


def _julia_ccalls(paths=None):
    """(symbol, return type, [argument types], number of actual arguments) of every ccall((:sym, lib), ...) in the shipped Julia
    package (julia/KrylovHIP/src/KrylovHIP.jl) and in the remaining snippets of INTEGRATION.md (the multi-GPU set-up)."""
    import re
    txt = "\n".join(open(f).read() for f in (paths or (JULIA_SRC, os.path.join(ROOT, "julia", "KrylovHIP", "examples", "mpi_krylov_hip.jl"),
                                                      os.path.join(ROOT, "INTEGRATION.md"))))
    out = []
    for m in re.finditer(r"ccall\(\(:(khip_[a-z0-9_]+),\s*lib\),\s*([A-Za-z0-9_{}]+),\s*\(", txt):
        sym, ret = m.group(1), m.group(2)
        i = m.end()                                                   # just inside the argument-type tuple
        depth, j = 1, i
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[j], 0)
            j += 1
        tup = txt[i:j - 1]
        types, cur, d = [], "", 0
        for ch in tup:
            if ch in "{(":
                d += 1
            elif ch in "})":
                d -= 1
            if ch == "," and d == 0:
                types.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            types.append(cur.strip())
        # the actual arguments up to the ccall's closing parenthesis
        depth, k = 1, j
        while depth:
            depth += {"(": 1, ")": -1, "[": 1, "]": -1}.get(txt[k], 0)
            k += 1
        args, cur, d = [], "", 0
        for ch in txt[j:k - 1]:
            if ch in "([{":
                d += 1
            elif ch in ")]}":
                d -= 1
            if ch == "," and d == 0:
                args.append(cur.strip()); cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur.strip())
        out.append((sym, ret, types, len([a for a in args if a])))
    return out


_JULIA_TO_C = {
    "Cint": {"int"}, "Int64": {"int64_t"}, "Csize_t": {"size_t"}, "Cdouble": {"double"}, "Cstring": {"char *"}, "Cvoid": {"void"},
    "Ptr{Cdouble}": {"double *"}, "Ptr{Float64}": {"double *"}, "Ref{Cdouble}": {"double *"}, "Ref{Cint}": {"int *"},
    "Ptr{Int32}": {"int32_t *"}, "Ref{Int64}": {"int64_t *"}, "Ptr{Ptr{Cdouble}}": {"double * *"},
    "Ref{Ptr{Cvoid}}": {"POINTER_TO_POINTER"}, "Ptr{Cvoid}": {"ANY_POINTER"},
    "Ref{Operator}": {"khip_operator *"}, "Ptr{Operator}": {"khip_operator *"}, "Ref{Options}": {"khip_options *"},
    "Ptr{Stats}": {"khip_stats *"},
}
_C_FUNCTION_POINTER_TYPEDEFS = {"khip_grow_fn", "khip_apply_fn", "khip_callback_fn"}     # passed as Ptr{Cvoid} (@cfunction)


def test_integration_md_ccalls_match_header():
    """Every ccall of the Julia glue names an exported function and passes what its declaration takes: same arity, the same
    return type, and argument types that map onto the C parameter types (Ptr{Cvoid} = any object pointer, Ref{Ptr{Cvoid}} = any
    pointer to a pointer).  Catches exactly what an executed binding would: a misspelt symbol, a missing or swapped argument,
    an Int64 where the ABI takes an int."""
    decls = _c_declarations()
    calls = _julia_ccalls()
    assert len(calls) >= 80 and len(decls) >= 100, (len(calls), len(decls))
    in_package = {c[0] for c in _julia_ccalls((JULIA_SRC,))}
    # the package reaches the adopt entries and the solver loops of all four methods, not only the primitives
    for sym in ("khip_cg_workspace_adopt", "khip_cg_solve", "khip_gmres_workspace_adopt", "khip_gmres_workspace_adopt_basis",
                "khip_gmres_workspace_set_grow", "khip_gmres_solve", "khip_gmres_host_state", "khip_bicgstab_workspace_adopt",
                "khip_bicgstab_solve", "khip_block_gmres_workspace_adopt", "khip_block_gmres_solve_panel", "khip_cg_stats"):
        assert sym in in_package, f"{sym} is not called by julia/KrylovHIP/src/KrylovHIP.jl"
    import krylov_jl_amd as K
    L = K.lib()
    for sym, ret, types, nargs in calls:
        assert sym in decls, f"{sym}: not declared in include/*.h"
        assert hasattr(L, sym), f"{sym}: not exported by libkrylov_hip.so"
        cret, cparams = decls[sym]
        assert ret in _JULIA_TO_C, (sym, ret)
        assert cret in _JULIA_TO_C[ret] or (ret == "Cstring" and cret == "char *"), (sym, ret, cret)
        assert len(types) == len(cparams), f"{sym}: {len(types)} argument types in the ccall, {len(cparams)} parameters in the header"
        assert nargs == len(types), f"{sym}: {nargs} arguments for {len(types)} argument types"
        for pos, (jt, ct) in enumerate(zip(types, cparams)):
            assert jt in _JULIA_TO_C, (sym, pos, jt)
            want = _JULIA_TO_C[jt]
            if "ANY_POINTER" in want:
                ok = (ct.endswith("*") and not ct.endswith("* *")) or ct in _C_FUNCTION_POINTER_TYPEDEFS
            elif "POINTER_TO_POINTER" in want:
                ok = ct.endswith("* *")
            else:
                ok = ct in want
            assert ok, f"{sym}: argument {pos + 1} is {jt} in the ccall but `{ct}` in the header"


@pytest.mark.parametrize("n,half_band,links,seed,unsym,dense_rows", [(20000, 13, 3, 3, False, 0), (70001, 13, 3, 5, True, 4),
                                                                    (5000, 4, 1, 9, False, 2), (1 << 15, 20, 5, 2, True, 0)])
def test_products_irregular_generator_equals_the_oracles_on_the_host(K, oracle, n, half_band, links, seed, unsym, dense_rows):
    """khip_gen_banded_random's host build (csrc/gen_irregular.cpp, through the host-only test export) against the oracle's
    independent restatement of the definition (oracle/krylov_oracle.c ko_csr_banded_random): row pointers, columns and values
    equal entry for entry -- for the whole operator and for a row slab as a rank of a partition would ask for it."""
    L = K.lib()
    A = oracle.banded_random(n, half_band=half_band, links=links, seed=seed, unsym=unsym, dense_rows=dense_rows)
    for r0, r1 in ((0, n), (n // 3, n // 3 + n // 5)):
        m = r1 - r0
        rp = np.zeros(m + 1, dtype=np.int32)
        nnz = C.c_int64()
        args = (n, half_band, links, seed, 1 if unsym else 0, dense_rows, r0, m)
        assert L.khip_test_gen_banded_random_host(*args, rp.ctypes.data_as(C.POINTER(C.c_int32)), None, None, C.byref(nnz)) == 0
        col, val = np.zeros(nnz.value, dtype=np.int32), np.zeros(nnz.value)
        assert L.khip_test_gen_banded_random_host(*args, rp.ctypes.data_as(C.POINTER(C.c_int32)), col.ctypes.data_as(C.POINTER(C.c_int32)),
                                                  val.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nnz)) == 0
        a, b = int(A.rowptr[r0]), int(A.rowptr[r1])
        assert nnz.value == b - a
        assert np.array_equal(rp, (np.asarray(A.rowptr[r0:r1 + 1]) - a).astype(np.int32))
        assert np.array_equal(col, A.col[a:b]) and np.array_equal(val, A.val[a:b])


REFERENCE_SRC = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SRC), reason="the reference tree is only present in the build container")
def test_julia_glue_defines_what_the_reference_solvers_call():
    """The drop-in promise of INTEGRATION.md, checked mechanically where the reference sources are at hand: every Krylov.k*
    primitive that cg! / gmres! / bicgstab! call on the workspace's vector type (src/cg.jl, gmres.jl, bicgstab.jl) has a method for
    HIPVector in the glue, and every operation block_gmres! applies to its matrix type (src/block_gmres.jl) has one for
    HIPMatrix.  (ktypeof is defined for both; kdisplay is host-only bookkeeping and not a primitive of the vector type.)"""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    glue = open(JULIA_SRC).read()
    used = set()
    for f in ("cg.jl", "gmres.jl", "bicgstab.jl"):
        src = re.sub(r"#.*", "", open(os.path.join(REFERENCE_SRC, f)).read())
        used |= set(re.findall(r"\b(k[a-z_]+!?)\(", src))
    used -= {"kdisplay", "ktimer", "ktypeof"}
    assert {"kdot", "knorm", "kaxpy!", "kaxpby!", "kcopy!", "kfill!", "kmul!", "kdivcopy!"} <= used, sorted(used)
    for name in sorted(used):
        pat = r"(?m)^\s*(?:function\s+)?(?:Krylov\.)?" + re.escape(name) + r"\((?:[^)]*HIPVector)"
        assert re.search(pat, glue), f"no HIPVector method of {name} in julia/KrylovHIP/src/KrylovHIP.jl"
    assert re.search(r"Krylov\.ktypeof\(::HIPVector\)", glue) and re.search(r"Krylov\.ktypeof\(::HIPMatrix\)", glue)
    # block_gmres!: the calls made on SM objects
    bsrc = re.sub(r"#.*", "", open(os.path.join(REFERENCE_SRC, "block_gmres.jl")).read())
    wanted = {"mul!": r"LinearAlgebra\.mul!\(\w+::HIPMatrix", "copyto!": r"Base\.copyto!\(\w+::HIPMatrix", "fill!": r"Base\.fill!\(\w+::HIPMatrix",
              "norm": r"LinearAlgebra\.norm\(\w+::HIPMatrix", "ldiv!": r"LinearAlgebra\.ldiv!\(\w+::UpperTriangular\{Float64,HIPMatrix\}",
              "householder!": r"Krylov\.householder!\(\w+::HIPMatrix", "kormqr!": r"Krylov\.kormqr!\([^)]*::HIPMatrix", "view": r"Base\.view\(\w+::HIPMatrix"}
    for call, pat in wanted.items():
        assert re.search(r"\b" + re.escape(call) + r"\(", bsrc), f"{call} is not called by block_gmres.jl (test out of date)"
        assert re.search(pat, glue), f"no HIPMatrix method of {call} in julia/KrylovHIP/src/KrylovHIP.jl"
    # the three forms of mul! the solver uses: A * P, V' * Q, and the 5-argument update
    assert re.search(r"mul!\(\w+::HIPMatrix, \w+::HIPCsr, \w+::HIPMatrix\)", glue)
    assert re.search(r"mul!\(\w+::HIPMatrix, \w+::Adjoint\{Float64,HIPMatrix\}, \w+::HIPMatrix\)", glue)
    assert re.search(r"mul!\(\w+::HIPMatrix, \w+::HIPMatrix, \w+::HIPMatrix, \w+::Number, \w+::Number\)", glue)
    # the workspace constructor is called as the reference defines it: (m, n, p, SV, SM; memory)
    ws = open(os.path.join(REFERENCE_SRC, "block_krylov_workspaces.jl")).read()
    assert re.search(r"function BlockGmresWorkspace\(m::Integer, n::Integer, p::Integer, SV::Type, SM::Type; memory", ws)
    assert re.search(r"BlockGmresWorkspace\(\w+, \w+, p, Vector\{Float64\}, (?:HIPMatrix|M); memory = \d+\)", open(JULIA_TEST).read())
    # the specialised solver methods: one per in-place method of this path, on the reference's own workspace types, each with the
    # reference's keyword list (src/cg.jl:101-112, src/gmres.jl:95-107, src/bicgstab.jl:105-116, src/block_gmres.jl:85-97) and a
    # fallback to the generic method
    for fn, ws_t, kwfile, kwvar in (("cg!", "CgWs", "cg.jl", "kwargs_cg"), ("gmres!", "GmresWs", "gmres.jl", "kwargs_gmres"),
                                    ("bicgstab!", "BicgstabWs", "bicgstab.jl", "kwargs_bicgstab"),
                                    ("block_gmres!", "BlockGmresWs", "block_gmres.jl", "kwargs_block_gmres")):
        m = re.search(r"function Krylov\." + re.escape(fn) + r"\(ws::" + ws_t + r", A::HIPCsr, \w+::HIP(?:Vector|Matrix);(.*?)\)\n", glue, flags=re.S)
        assert m, f"no specialised method of {fn}"
        ref = open(os.path.join(REFERENCE_SRC, kwfile)).read()
        kws = re.search(r"^" + kwvar + r" = \((.*?)\)", ref, flags=re.M).group(1)
        for kw in re.findall(r":(\w+)", kws):
            assert re.search(r"\b" + kw + r"\b", m.group(1)), f"{fn}: keyword {kw} of the reference is missing"
        assert re.search(r"invoke\(Krylov\." + re.escape(fn), glue), f"{fn}: no fallback to the generic method"
    for ws_t, ref_t in (("CgWs", "CgWorkspace{Float64,Float64,HIPVector}"), ("GmresWs", "GmresWorkspace{Float64,Float64,HIPVector}"),
                        ("BicgstabWs", "BicgstabWorkspace{Float64,Float64,HIPVector}"),
                        ("BlockGmresWs", "BlockGmresWorkspace{Float64,Float64,Vector{Float64},HIPMatrix}")):
        assert f"const {ws_t} = {ref_t}" in glue
    # fields the methods read are fields of the reference's workspaces
    wsrc = open(os.path.join(REFERENCE_SRC, "krylov_workspaces.jl")).read()
    for struct, fields in (("CgWorkspace", ("Δx", "x", "r", "npc_dir", "p", "Ap", "z", "warm_start", "stats")),
                           ("GmresWorkspace", ("Δx", "x", "w", "p", "q", "V", "c", "s", "z", "R", "warm_start", "inner_iter", "stats")),
                           ("BicgstabWorkspace", ("Δx", "x", "r", "p", "v", "s", "qd", "yz", "t", "warm_start", "stats"))):
        body = re.search(r"mutable struct " + struct + r"\{T,FC,S\}.*?\nend", wsrc, flags=re.S).group(0)
        for f in fields:
            assert re.search(r"^\s+" + re.escape(f) + r"\s+::", body, flags=re.M), (struct, f)


# ---- every entry point of the reference reaches the library's loop ("What's weak" 1) -------------------------------
_SOLVERS = (("cg", "CgWs", "cg.jl", "CgWorkspace", "AbstractVector{Float64}"), ("gmres", "GmresWs", "gmres.jl", "GmresWorkspace", "AbstractVector{Float64}"),
            ("bicgstab", "BicgstabWs", "bicgstab.jl", "BicgstabWorkspace", "AbstractVector{Float64}"),
            ("block_gmres", "BlockGmresWs", "block_gmres.jl", "BlockGmresWorkspace", "AbstractMatrix{Float64}"))


def _reference_forwarded_keywords(method, src_file):
    """What the reference's generated entry points pass to `method!`: (names in order, {name: default expression}) from
    `kwargs_<method>` and `def_kwargs_<method>` (src/<method>.jl), after checking in src/interface.jl that EVERY generated method calls
    the in-place one with `$(kwargs...)`, i.e. forwards the whole list, its own defaults included."""
    ref = open(os.path.join(REFERENCE_SRC, src_file)).read()
    names = re.findall(r":(\w+)", re.search(r"^kwargs_" + method + r" = \((.*?)\)", ref, flags=re.M).group(1))
    table = re.search(r"^def_kwargs_" + method + r" = \((.*?)\)\n\n", ref, flags=re.M | re.S).group(1)
    defaults = {}
    for m in re.finditer(r":\(;\s*(\w+)(?:::[^=]+?)?\s*=\s*(.*?)\s*\)\s*[,)]?\s*$", table, flags=re.M):
        defaults[m.group(1)] = m.group(2)
    assert list(defaults) == names, (list(defaults), names)
    iface = re.sub(r"#.*", "", open(os.path.join(REFERENCE_SRC, "interface.jl")).read())
    calls = re.findall(r"\$\(krylov!\)\(workspace, \$\(args\.\.\.\); (.*?)\)\n", iface)
    assert len(calls) >= 14 and set(calls) == {"$(kwargs...)"}, set(calls)          # out-of-place (4 forms), krylov_solve! and the x0 forms, vector and block
    return names, defaults


def _julia_gate(glue, fn, ws_t):
    """(keyword part of the signature, condition of the first `if` = the fallback gate, body up to the solve) of a specialised method."""
    m = re.search(r"function Krylov\." + re.escape(fn) + r"!\(ws::" + ws_t + r", A::HIPCsr, \w+::HIP(?:Vector|Matrix);(.*?)\)\n(?:  [^\n]*\n){0,3}?  if (.*?)\n(.*?)\nend\n", glue, flags=re.S)
    assert m, fn
    return m.group(1), m.group(2), m.group(3)


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SRC), reason="the reference tree is only present in the build container")
def test_forwarded_defaults_reach_the_native_loop():
    """The reference's `cg(A, b)`, `krylov_solve(Val(:cg), A, b)`, `krylov_solve!(ws, A, b)` and the x0 forms forward EVERY keyword to
    `cg!` explicitly, `callback = workspace -> false` included (src/interface.jl:146-154, 160-170, 306-347).  For that keyword set the
    fallback gate of each specialised method in julia/KrylovHIP/src/KrylovHIP.jl must be FALSE (the native branch is taken), the
    callback must be recognised as the default (-> NULL in khip_options: the device-resident loop), and the same holds for the Python
    mirror's tables, which are executed on the GPU (tests/test_gpu_adopt.py)."""
    glue = open(JULIA_SRC).read()
    # the helper predicates the gates use, as defined in the package (a change there must be re-read here)
    assert "native_precond(M) = M === I || M isa HIPOperator" in glue
    assert "native_log(verbose, io) = verbose <= 0 || logfd(io) >= 0" in glue
    assert "default_callback(cb) = parentmodule(typeof(cb)) === Krylov && Base.issingletontype(typeof(cb))" in glue
    assert "user_callback(cb) = (cb === nothing || default_callback(cb)) ? nothing : cb" in glue
    I, kstdout = object(), object()

    class Panel:                                     # a tall HIPMatrix
        pass
    env_fns = {"native_precond": lambda M: M is I, "native_log": lambda verbose, io: verbose <= 0 or io is kstdout, "tall": lambda P: isinstance(P, Panel)}

    def julia_value(expr, names):
        expr = expr.strip()
        table = {"I": I, "false": False, "true": True, "zero(T)": 0.0, "√eps(T)": float(np.sqrt(np.finfo(np.float64).eps)), "Inf": float("inf"),
                 "kstdout": kstdout, "b": "b"}
        if expr == "workspace -> false":
            return "DEFAULT_CALLBACK"
        if expr in table:
            return table[expr]
        return int(expr)

    import krylov_jl_amd as K
    for method, ws_t, src_file, ws_struct, rhs_t in _SOLVERS:
        names, defaults = _reference_forwarded_keywords(method, src_file)
        assert defaults["callback"] == "workspace -> false"
        forwarded = {k: julia_value(v, names) for k, v in defaults.items()}
        sig, gate, body = _julia_gate(glue, method, ws_t)
        # (i) the gate never looks at the callback, and is false for the forwarded defaults
        assert "callback" not in gate, f"{method}!: the fallback gate tests the callback -- cg(A, b) forwards one ALWAYS ({gate})"
        py = gate.replace("||", " or ").replace("&&", " and ").replace("!==", " is not ").replace("===", " is ")
        py = re.sub(r"!(?=[\w(])", " not ", py)
        py = py.replace("ws.X", "wsX")
        scope = dict(env_fns)
        scope.update({k: v for k, v in forwarded.items() if k != "callback"})
        scope.update({"B": Panel(), "wsX": Panel(), "I": I})
        assert eval(py, {"__builtins__": {}}, scope) is False, f"{method}!: forwarded defaults do not reach the native branch: {gate}"
        # ... and true for what the library cannot take
        for k, v in (("ldiv", True), ("M", object()), ("verbose", 1)):
            s2 = dict(scope)
            s2[k] = v
            if k == "verbose":
                s2["iostream"] = object()                   # an IOBuffer: no file descriptor
            assert eval(py, {"__builtins__": {}}, s2) is True, (method, k, gate)
        # (ii) the callback reaches khip_options only through user_callback, and the native branch records itself
        assert "callback_args(user_callback(callback), ws, sp, history)" in body
        assert "NATIVE_SOLVES[] += 1" in body and f"khip_{method}_last_path" in body
        assert re.search(r"callback = nothing", sig)
        # (iii) the fallback names argument types that are a subtype of the generic signature: the fully parametrised alias
        # (T, FC and the storage type fixed) and the element type FC = Float64 on the right-hand side (src/cg.jl:120: FC is shared)
        tuples = re.findall(r"invoke\(Krylov\." + method + r"!, Tuple\{((?:[^{}]|\{[^{}]*\})*)\}", glue)
        assert tuples == [f"{ws_t},Any,{rhs_t}"], tuples
        alias = re.search(r"const " + ws_t + r" = (\w+)\{(.*)\}\n", glue)
        assert alias.group(1) == ws_struct
        nparams = len(re.search(r"mutable struct " + ws_struct + r"\{([^}]*)\}", open(os.path.join(
            REFERENCE_SRC, "block_krylov_workspaces.jl" if method == "block_gmres" else "krylov_workspaces.jl")).read()).group(1).split(","))
        depth, parts, cur = 0, [], ""
        for ch in alias.group(2):
            if ch == "," and depth == 0:
                parts.append(cur); cur = ""
                continue
            depth += ch == "{"
            depth -= ch == "}"
            cur += ch
        parts.append(cur)
        assert len(parts) == nparams and parts[:2] == ["Float64", "Float64"], (parts, nparams)
        gen = re.search(r"function " + method + r"!\(workspace :: " + ws_struct + r"\{[^}]*\}, \$\(def_args_" + method + r"\.\.\.\)", open(os.path.join(REFERENCE_SRC, src_file)).read())
        assert gen, "the generic signature moved"
        # (iv) the Python mirror forwards the same names with the same defaults
        mine = K.FORWARDED_DEFAULTS[method]
        assert list(mine) == names, (list(mine), names)
        for k in names:
            want = forwarded[k]
            if want is I or want is kstdout or want == "b":
                assert mine[k] is None, (method, k)
            elif want == "DEFAULT_CALLBACK":
                assert mine[k] is K.default_callback and K._user_callback(mine[k]) is None
            else:
                assert mine[k] == want and type(mine[k]) is type(want), (method, k, mine[k], want)
    # workspace keywords of the out-of-place entries (src/gmres.jl:110, src/block_gmres.jl:99)
    assert K.WORKSPACE_KWARGS == {"gmres": {"memory": 20}, "block_gmres": {"memory": 5}}
    ref_g = open(os.path.join(REFERENCE_SRC, "gmres.jl")).read()
    ref_b = open(os.path.join(REFERENCE_SRC, "block_gmres.jl")).read()
    assert "def_kwargs_workspace_gmres = (:(; memory::Int = 20),)" in ref_g and "def_kwargs_workspace_block_gmres = (:(; memory::Int = 5),)" in ref_b
    # the package's own test asserts the same on a GPU, and its `invoke` helpers are parametrised too
    rt = open(JULIA_TEST).read()
    assert "Tuple{typeof(ws),Any,AbstractVector{Float64}}" in rt and "Tuple{typeof(ws2),Any,AbstractMatrix{Float64}}" in rt
    assert not re.search(r"Tuple\{\w+Workspace,Any,Abstract(?:Vector|Matrix)\}", rt + glue), "an unparametrised invoke signature is left"
    for entry in ("cg(A_gpu, b_gpu)", "krylov_solve(Val(:cg), A_gpu, b)", "krylov_solve!(ws, A_gpu, b)", "cg!(ws, A_gpu, b, x0)", "krylov_solve!(ws, A_gpu, b, x0)",
                  "cg(A_gpu, b, x0)", "gmres(U_gpu, b)", "bicgstab(U_gpu, b)", "block_gmres(U_gpu, B)"):
        assert re.search(r"native\(\d\) do; " + re.escape(entry), rt), f"runtests.jl does not assert the native path after {entry}"


def test_adjoint_of_the_julia_operator_is_cached_and_finalised():
    """ADVICE / VERDICT r05: `A'` built a new transposed operator per call, without a finalizer."""
    glue = open(JULIA_SRC).read()
    body = re.search(r"function Base\.adjoint\(A::HIPCsr\)\n(.*?)\nend\n", glue, flags=re.S).group(1)
    assert "A.adj === nothing || return A.adj" in body and "finalizer(destroy_csr" in body and "A.adj = At" in body
    assert body.count("khip_csr_transpose") == 1


# ---- a structural check of the Julia sources (no Julia here): brackets and block keywords balance ------------------------------
def _julia_strip(src):
    """Julia source without string / character literals and comments (postfix ' = adjoint stays out of the way)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith('"""', i):
            j = src.find('"""', i + 3); i = (j + 3) if j >= 0 else n; out.append('""'); continue
        if c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            out.append('""'); i = j + 1; continue
        if c == "'":
            prev = out[-1] if out else " "
            if prev.isalnum() or prev in "_)]}'":
                out.append(" "); i += 1; continue
            j = i + 1
            if j < n and src[j] == "\\":
                j += 1
            j = src.find("'", j + 1); out.append("' '"); i = j + 1; continue
        if c == "#":
            if src.startswith("#=", i):
                j = src.find("=#", i + 2); i = (j + 2) if j >= 0 else n; continue
            j = src.find("\n", i); i = j if j >= 0 else n; continue
        out.append(c); i += 1
    return "".join(out)


@pytest.mark.parametrize("rel", ["src/KrylovHIP.jl", "test/runtests.jl", "examples/mpi_krylov_hip.jl"])
def test_julia_sources_are_structurally_balanced(rel):
    """The package cannot be parsed by Julia here.  What can be checked without it: every bracket closes in order, and the block
    openers (function / do / begin / let / struct / module / try / quote / macro anywhere; if / for / while at the start of a
    statement -- comprehensions and generators have no `end`) are as many as the `end`s.  Catches a lost `end` or parenthesis of an
    edit; it is not a parser."""
    s = _julia_strip(open(os.path.join(ROOT, "julia", "KrylovHIP", rel)).read())
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ln, line in enumerate(s.split("\n"), 1):
        for ch in line:
            if ch in "([{":
                stack.append((ch, ln))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], f"{rel}:{ln}: unbalanced {ch}"
                stack.pop()
    assert not stack, f"{rel}: unclosed {stack[-1]}"
    s2 = re.sub(r"\bmutable\s+struct\b", "struct", s)
    opens = 0
    for line in s2.split("\n"):
        opens += len(re.findall(r"\b(function|do|begin|let|struct|module|try|quote|macro)\b", line))
        opens += len(re.findall(r"(?:^|;|=|\breturn\b|\bbegin\b|\bdo\b|\belse\b)\s*(if|for|while)\b", line))
    ends = len(re.findall(r"\bend\b", s2))
    assert opens == ends, f"{rel}: {opens} block openers, {ends} `end`"
