"""GPU: the single-precision (s, c) and mixed (sd, cz) families of libhifir through shim/libhifir.so, against the
reference's OWN libhifir -- libhifir/src/libhifir.cpp compiled where it lies into oracle/_ref/libhifir_ref.so (checker only).

What the shim does with such a handle (shim/libhifir_amd_shim.cpp, header): the reference factorizes in single precision on
the host (same templates, same bits as the checker's factorization), the factors are widened EXACTLY to fp64 and applied by
the fp64 kernels.  Consequences, asserted below with the tolerance next to the reason:
  * lhfsd* / lhfcz* (single hierarchy, double vectors, libhifir.cpp:1192-1284): the reference's sparse stages already run
    in double (float factor x double vector); only its dense last level computes in float.  Agreement to a few float
    epsilons of the solution's size: <= 2e-5 relative (measured 1e-7 ... 2e-6).
  * lhfs* / lhfc*: the reference carries every stage in float; the GPU result is the widened computation rounded once.
    <= 2e-4 relative (measured 1e-6 ... 1e-5).  (Both differ from the exact action of the float hierarchy by float
    rounding: the reference's in every stage, the shim's only in the final rounding of x.)
  * queries (levels, nnz, ranks, Schur size): identical -- the same host factorization."""
import ctypes as C

import numpy as np
import pytest

import shim_util as su
from util import load_hier, relerr

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not su.available(), reason="shim/libhifir.so not built"),
              pytest.mark.skipif(not su.ref_available(), reason="oracle/_ref/libhifir_ref.so not built")]
TOL_MIXED, TOL_SINGLE = 2e-5, 2e-4


def _params(L):
    p = (C.c_double * su.LHF_NUMBER_PARAMS)()
    assert L.lhfSetDefaultParams(p) == su.LHF_SUCCESS
    p[su.LHF_VERBOSE] = 0
    return p


@pytest.fixture(scope="module", params=[("demo_A", "s"), ("young1c", "c"), ("p2d_64_deep", "s")], ids=lambda p: "%s-%s" % p)
def pair(request):
    name, t = request.param
    levels, d = load_hier(name)
    out = []
    for L in (su.lib(), su.ref_lib()):
        A = su.Matrix(t, d["A_indptr"], d["A_indices"], d["A_vals"], L=L)
        M = su.Hif(t, A, None, _params(L), L=L)
        assert M.h, su.errmsg()
        Aw = su.Matrix("d" if t == "s" else "z", d["A_indptr"], d["A_indices"], d["A_vals"], L=L)
        out.append((A, M, Aw))
    yield d, t, out[0], out[1]
    for A, M, Aw in out:
        M.close()
        A.close()
        Aw.close()


def test_queries_are_the_host_factorization(pair):
    d, t, (A, M, Aw), (Ar, Mr, Awr) = pair
    assert M.stats() == Mr.stats()
    for q in ("GetNnz", "GetLevels", "GetSchurSize", "GetSchurRank"):
        assert M._f(q)(M.h) == Mr._f(q)(Mr.h)


def test_single_precision_vectors(pair):
    d, t, (A, M, Aw), (Ar, Mr, Awr) = pair
    b = d["b"].astype(M.dt)
    st, x = M.solve(b)
    str_, xr = Mr.solve(b)
    assert st == su.LHF_SUCCESS and str_ == su.LHF_SUCCESS and x.dtype == M.dt
    assert relerr(x, xr) <= TOL_SINGLE
    for op in (su.LHF_SH, su.LHF_M, su.LHF_MH):  # transpose solve, product, transposed product (libhifir.cpp:447-472)
        st, y = M.apply(op, b)
        str_, yr = Mr.apply(op, b)
        assert st == su.LHF_SUCCESS and str_ == su.LHF_SUCCESS
        assert relerr(y, yr) <= TOL_SINGLE, op


def test_mixed_double_vectors_on_a_single_hierarchy(pair):
    d, t, (A, M, Aw), (Ar, Mr, Awr) = pair
    b = d["b"]
    st, x = M.mixed_solve(b)
    str_, xr = Mr.mixed_solve(b)
    assert st == su.LHF_SUCCESS and str_ == su.LHF_SUCCESS
    assert relerr(x, xr) <= TOL_MIXED
    for op in (su.LHF_SH, su.LHF_M, su.LHF_MH):
        st, y = M.mixed_apply(op, b)
        str_, yr = Mr.mixed_apply(op, b)
        assert st == su.LHF_SUCCESS and str_ == su.LHF_SUCCESS
        assert relerr(y, yr) <= TOL_MIXED, op
    # iterative refinement with the DOUBLE matrix of lhfsdUpdate / lhfczUpdate (Ad / Az, :1185-1190, :1206-1213)
    assert M.mixed("Apply")(M.h, su.LHF_S, su._ptr(b), 3, None, su.LHF_DEFAULT_RANK, su._ptr(b.copy()), None) == su.LHF_NULL_OBJ
    assert M.mixed("Update")(M.h, Aw.h) == su.LHF_SUCCESS and Mr.mixed("Update")(Mr.h, Awr.h) == su.LHF_SUCCESS
    st, xi = M.mixed_apply(su.LHF_S, b, nirs=3)
    str_, xir = Mr.mixed_apply(su.LHF_S, b, nirs=3)
    assert st == su.LHF_SUCCESS and str_ == su.LHF_SUCCESS
    assert relerr(xi, xir) <= 10 * TOL_MIXED  # (three sweeps amplify the dense level's float rounding of the reference)
    # ... and refinement on single-precision vectors uses the handle's OWN matrix again (the devices switch matrices)
    bs = b.astype(M.dt)
    st, xs = M.apply(su.LHF_S, bs, nirs=2)
    str_, xsr = Mr.apply(su.LHF_S, bs, nirs=2)
    assert st == su.LHF_SUCCESS and str_ == su.LHF_SUCCESS and relerr(xs, xsr) <= 10 * TOL_SINGLE
    st, xi2 = M.mixed_apply(su.LHF_S, b, nirs=3)
    assert st == su.LHF_SUCCESS and np.array_equal(xi2, xi)


def test_errors_of_the_single_families():
    L = su.lib()
    assert L.lhfsSolve(None, None, None) == su.LHF_NULL_OBJ
    assert L.lhfsdSolve(None, None, None) == su.LHF_NULL_OBJ and L.lhfczApply(None, 0, None, 1, None, 0, None, None) == su.LHF_NULL_OBJ
    assert L.lhfsdUpdate(None, None) == su.LHF_NULL_OBJ
    h = L.lhfcCreate(None, None, _params(L))  # (a NULL matrix leaves an empty handle behind, libhifir.cpp:383-396)
    assert h
    x = np.zeros(4, dtype=np.complex64)
    assert L.lhfcSolve(h, su._ptr(x), su._ptr(x.copy())) == su.LHF_HIFIR_ERROR and "empty" in su.errmsg()
    assert L.lhfcDestroy(h) == su.LHF_SUCCESS
