"""CPU: shim/libhifir.so is a drop-in for the reference's libhifir at the SYMBOL level -- it exports every function
libhifir/include/libhifir.h declares (the list is committed as tests/golden/libhifir_symbols.txt; where the
reference tree is present it is re-derived from the header) -- and the parts that need no GPU behave like
libhifir.cpp: parameter helpers, one-shot error message, non-owning matrix handles, NULL-safety, status codes.
No compute call is made without a GPU; Create must FAIL without one (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import hifir_amd
import shim_util as su
from util import load_hier

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_HDR = "/root/reference/libhifir/include/libhifir.h"

pytestmark = pytest.mark.skipif(not su.available(), reason="shim/libhifir.so not built (needs the reference headers)")


def _required():
    return open(os.path.join(ROOT, "tests", "golden", "libhifir_symbols.txt")).read().split()


def test_exports_every_reference_symbol():
    req = _required()
    assert len(req) == 91
    out = subprocess.check_output(["nm", "-D", "--defined-only", su.SHIM_PATH], text=True)
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert not [s for s in req if s not in have]
    # the additive entry points of include/libhifir_amd_ext.h
    ext = re.findall(r"\b(lhf\w+)\s*\(", open(os.path.join(ROOT, "include", "libhifir_amd_ext.h")).read())
    assert len(set(ext)) == 16 and not [s for s in ext if s not in have]
    # all four type families, incl. the mixed-precision ones
    for s in ("lhfsCreate", "lhfcApply", "lhfsdApply", "lhfczSolve", "lhfsdUpdate", "lhfEnableWarning"):
        assert s in have


@pytest.mark.skipif(not os.path.exists(REF_HDR), reason="reference tree absent")
def test_symbol_list_matches_the_reference_header():
    declared = sorted(set(re.findall(r"\b(lhf[A-Za-z]+)\(", open(REF_HDR).read())))
    assert declared == sorted(_required())


def test_versions_params_and_error_message():
    L = su.lib()
    v = (C.c_int * 3)()
    L.lhfGetVersions(v)
    assert list(v) == [0, 2, 0]  # HIFIR v0.2.0
    p = su.default_params(verbose=1)
    # hif_get_default_options (src/hif/Options.h:135-164) through the LHF_* slots (libhifir.h:94-117)
    assert list(p)[:6] == [1e-4, 1e-4, 3.0, 3.0, 10.0, 10.0]
    assert p[10] == 0.0 and p[12] == 1e3 and p[15] == 0.65 and p[16] == 2000 and p[8] == -2
    L.lhfSetDroptol(1e-2, p), L.lhfSetAlpha(3.0, p), L.lhfSetKappa(5.0, p)
    assert (p[0], p[1], p[4], p[5], p[2], p[3]) == (1e-2, 1e-2, 3.0, 3.0, 5.0, 5.0)
    L.lhfEnableWarning(), L.lhfDisableWarning()  # declared by the reference, defined nowhere there: no-ops here
    assert su.errmsg() is None
    h = L.lhfsCreate(None, None, p)  # a NULL matrix leaves an EMPTY handle behind (libhifir.cpp:383-396), every family
    assert h
    x = np.zeros(3, dtype=np.float32)
    assert L.lhfsSolve(h, su._ptr(x), su._ptr(x.copy())) == su.LHF_HIFIR_ERROR  # ... which cannot solve ...
    m = su.errmsg()
    assert m and "empty" in m  # ... and says why
    assert su.errmsg() is None  # returned once, then cleared (libhifir.cpp:224-229)
    assert L.lhfsDestroy(h) == su.LHF_SUCCESS


def test_matrix_handles_alias_and_null_safety():
    L = su.lib()
    levels, d = load_hier("p2d_5")
    for t in "dszc":
        A = su.Matrix(t, d["A_indptr"], d["A_indices"], d["A_vals"])
        f = lambda name: getattr(L, f"lhf{t}{name}")
        assert f("GetMatrixSize")(A.h) == 25 and f("GetMatrixNnz")(A.h) == len(d["A_vals"])
        assert f("GetMatrixSize")(None) == 0
        empty = f("CreateMatrix")(1, 0, None, None, None)  # "the last three entries can be NULL" (libhifir.h:323)
        assert empty and f("GetMatrixSize")(empty) == 0
        assert f("WrapMatrix")(empty, 25, su._ptr(A.indptr), su._ptr(A.indices), su._ptr(A.vals)) == su.LHF_SUCCESS
        assert f("GetMatrixSize")(empty) == 25
        assert f("WrapMatrix")(None, 25, su._ptr(A.indptr), su._ptr(A.indices), su._ptr(A.vals)) == su.LHF_NULL_OBJ
        assert f("WrapMatrix")(empty, 25, None, None, None) == su.LHF_NULL_OBJ
        assert f("DestroyMatrix")(empty) == su.LHF_SUCCESS and f("DestroyMatrix")(None) == su.LHF_SUCCESS
        # NULL HIF handles: status, not a crash; size getters return 0 (libhifir.cpp:519-533)
        b = np.zeros(25, dtype=A.vals.dtype)
        assert f("Solve")(None, su._ptr(b), su._ptr(b.copy())) == su.LHF_NULL_OBJ
        assert f("Apply")(None, su.LHF_S, su._ptr(b), 1, None, -2, su._ptr(b.copy()), None) == su.LHF_NULL_OBJ
        assert f("Update")(None, A.h) == su.LHF_NULL_OBJ and f("Setup")(None, A.h, None, None) == su.LHF_NULL_OBJ
        assert f("Destroy")(None) == su.LHF_SUCCESS
        assert f("GetNnz")(None) == f("GetLevels")(None) == f("GetSchurSize")(None) == f("GetSchurRank")(None) == 0
        st = (C.c_size_t * 9)(*([7] * 9))
        assert f("GetStats")(None, st) == su.LHF_SUCCESS and list(st) == [0] * 9
        A.close()
    assert L.lhfsdSolve(None, None, None) == su.LHF_NULL_OBJ and L.lhfczUpdate(None, None) == su.LHF_NULL_OBJ


def test_create_fails_loudly_without_a_gpu():
    if hifir_amd.lib().hifamd_device_count() > 0:
        pytest.skip("a GPU is present")
    levels, d = load_hier("p2d_5")
    A = su.Matrix("d", d["A_indptr"], d["A_indices"], d["A_vals"])
    M = su.Hif("d", A, None, su.default_params())  # the host factorization succeeds; shipping it cannot
    assert M.h is None
    m = su.errmsg()
    assert m and "no CPU fallback" in m
    ids = (C.c_int * 1)(0)
    assert su.lib().lhfSetDevices(ids, 1) == su.LHF_MISMATCHED_SIZES  # no such device
    assert su.lib().lhfSetDevices(None, 0) == su.LHF_SUCCESS
    A.close()
