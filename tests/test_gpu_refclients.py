"""The reference's OWN C clients (interfaces/test/C/*.c, interfaces/examples/C/*.c -- compiled from where they
lie by `make -C oracle refhip`, only possible where /root/reference exists) driving the HIP path through
krylov.jl_amd/csrc/capi_compat.cpp (libkrylov_hip_capi.so).  The prebuilt binaries travel to the GPU box inside oracle/_ref/."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref")
OUT_OF_SCOPE = ("MINRES", "Float32", "DQGMRES", "block_minres")


def _run(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        pytest.skip(f"{name} not built (needs /root/reference at build time: make -C oracle refhip)")
    return subprocess.run([path], capture_output=True, text=True, timeout=300)


def test_hip_basic_cg_example():
    out = _run("hip_basic_cg")                 # interfaces/examples/C/basic_cg.c:13-15: niter 3, x = ones
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Solved: yes" in out.stdout and "niter: 3" in out.stdout
    assert "x = [ 1.00 1.00 1.00 1.00 1.00 ]" in out.stdout


def test_hip_test_api_client():
    out = _run("hip_test_api")
    fails = [l for l in out.stdout.splitlines() if "FAIL" in l]
    bad = [l for l in fails if not any(k in l for k in OUT_OF_SCOPE)]
    assert not bad, (bad, out.stderr[-500:])
    assert len(fails) == 5 and "36 checks passed" in out.stdout        # same outcome as against the CPU oracle


def test_hip_test_all_solvers_client():
    out = _run("hip_test_all_solvers")
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if "..." in l}
    for s in ("cg", "gmres", "bicgstab"):
        assert "PASS" in lines[s], (lines[s], out.stderr[-500:])


def test_hip_test_block_client():
    out = _run("hip_test_block")
    sections, cur = {}, None
    for l in out.stdout.splitlines():
        if l.endswith("..."):
            cur = l
            sections[cur] = []
        elif "FAIL" in l and cur:
            sections[cur].append(l)
    assert sections, out.stdout + out.stderr
    for name, fails in sections.items():
        if "block_minres" in name:
            continue
        assert not fails, (name, fails)
    ex = _run("hip_block_gmres_example")
    assert "Block solved: yes" in ex.stdout


def test_hip_fortran_clients():
    """The reference's Fortran tests (interfaces/test/Fortran/*.f90 + its krylov.f90 include) on the HIP path."""
    out = _run("hip_f_test_all_solvers")
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if "..." in l}
    for s in ("cg", "gmres", "bicgstab"):
        assert "PASS" in lines[s], (lines[s], out.stderr[-500:])
    out = _run("hip_f_test_block")
    sections, cur = {}, None
    for l in out.stdout.splitlines():
        if l.rstrip().endswith("...") and "FAIL" not in l and "PASS" not in l:
            cur = l.strip()
            sections[cur] = []
        elif "FAIL" in l and cur:
            sections[cur].append(l)
    assert any("block_gmres" in k for k in sections), out.stdout + out.stderr
    for name, fails in sections.items():
        if "block_minres" in name:
            continue
        assert not fails, (name, fails)


def test_plain_c_example_against_the_abi():
    """examples/cg_poisson.c: a C program on include/krylov_hip.h alone (no Python, no shim) -- built by
    __graft_entry__.build(), run here.  64^3 Poisson, rtol 1e-8: 159 iterations, residual < 1e-6."""
    exe = os.path.join(ROOT, "examples", "cg_poisson")
    if not os.path.exists(exe):
        pytest.skip("examples/cg_poisson not built")
    out = subprocess.run([exe, "64"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Solved: yes" in out.stdout and "niter: 159" in out.stdout
    assert "solution good enough given atol and rtol" in out.stdout


def test_capi_device_mode():
    """tests/c/capi_device.c: the KRYLOV_HIP device enumerator of libkrylov_hip_capi.so (include/krylov_hip_ext.h) --
    device right-hand sides, the built-in CSR operator with matvec_A = NULL, a device callback, the block interface."""
    out = _run("hip_capi_device")
    assert out.returncode == 0 and "0 failure(s)" in out.stdout, out.stdout + out.stderr[-800:]
    assert "FAIL" not in out.stdout


def test_block_gmres_primitive_sequence_equals_the_solver():
    """tests/c/block_primitive_sequence.c: block_gmres! replayed ONE C-ABI call per reference line (src/block_gmres.jl:155-330)
    -- khip_spmm, khip_panel_gemm_tn / _nn, khip_panel_qr_tau, host kormqr on the 2p x p blocks: what a Julia HIPMatrix
    would issue -- against khip_block_gmres_solve: same iteration count, history and solution, with and
    without restart.  Built by __graft_entry__.build()."""
    exe = os.path.join(ROOT, "tests", "c", "block_primitive_sequence")
    if not os.path.exists(exe):
        pytest.skip("tests/c/block_primitive_sequence not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "0 failure(s)" in out.stdout, out.stdout + out.stderr[-800:]
    assert out.stdout.count("PASS") == 2 and "FAIL" not in out.stdout


@pytest.mark.parametrize("sizes", [("64",), ("512",)])
def test_adopted_workspaces_equal_the_owned_ones(sizes):
    """tests/c/adopt_sequence.c: cg! / gmres! / bicgstab! / block_gmres! on CALLER-OWNED vectors (khip_*_workspace_adopt -- what
    the cg!(ws::CgWorkspace{..,HIPVector}, ...) methods of julia/KrylovHIP forward to) against the library-owned workspaces:
    iteration counts, statuses, residual histories and solutions bit for bit, the solution in the caller's x, lazily
    allocated vectors handed over late, the basis grown through the caller's push!.  64^3: every solver; 512^3 (cfg 2, the
    bench workload): the full cg! solve to rtol 1e-8."""
    exe = os.path.join(ROOT, "tests", "c", "adopt_sequence")
    if not os.path.exists(exe):
        pytest.skip("tests/c/adopt_sequence not built")
    out = subprocess.run([exe, *sizes], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "0 failure(s)" in out.stdout and "PASS" in out.stdout, out.stdout[-3000:] + out.stderr[-800:]
    assert "FAIL" not in out.stdout
    assert "bit-identical" in out.stdout
