"""CPU, only where the compiled reference (oracle/_ref/libhifref.so) exists: pins the C restatement
to the real reference on fresh inputs -- bitwise on the sparse kernels, 1e-12 on the dense level,
including rank-deficient QRCP truncation (the ?laic1 path, small_scale/QRCP.hpp:333-364)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import orc, ref
from util import poisson2d, relerr

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not ref.available(), reason="compiled reference not present")]


def _rand_tri(n, density, lower, rng):
    A = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="csc")
    A = sp.tril(A, -1, format="csc") if lower else sp.triu(A, 1, format="csc")
    A.sort_indices()
    return A


@pytest.mark.parametrize("n", [5, 64, 300])
def test_ccs_kernels_bitwise(n):
    # the reference's tests/test_cs_tri.cpp uses n in [5,300]; here bit-for-bit, not 1e-10
    rng = np.random.default_rng(n)
    for cplx in (False, True):
        for op, lower in [(0, True), (1, False)]:
            A = _rand_tri(n, 0.2, lower, rng)
            v = A.data + (1j * rng.uniform(-1, 1, A.nnz) if cplx else 0)
            x = rng.uniform(-1, 1, n) + (1j * rng.uniform(-1, 1, n) if cplx else 0)
            y0 = ref.ccs_kernel(op, n, n, A.indptr, A.indices, v, x)
            y1 = orc.ccs_kernel(op, n, n, A.indptr, A.indices, v, x)
            assert np.array_equal(y0, y1)
            # conjugate-transpose solves (CompressedStorage.hpp:2307-2324, :2399-2413)
            y0 = ref.ccs_kernel(op + 3, n, n, A.indptr, A.indices, v, x)
            y1 = orc.ccs_kernel(op + 3, n, n, A.indptr, A.indices, v, x)
            assert np.array_equal(y0, y1)
        R = sp.random(n + 3, n, density=0.3, random_state=np.random.RandomState(n), format="csc")
        R.sort_indices()
        v = R.data + (1j * rng.uniform(-1, 1, R.nnz) if cplx else 0)
        x = rng.uniform(-1, 1, n) + (1j * rng.uniform(-1, 1, n) if cplx else 0)
        assert np.array_equal(ref.ccs_kernel(2, n + 3, n, R.indptr, R.indices, v, x),
                              orc.ccs_kernel(2, n + 3, n, R.indptr, R.indices, v, x))
        xt = rng.uniform(-1, 1, n + 3) + (1j * rng.uniform(-1, 1, n + 3) if cplx else 0)
        assert np.array_equal(ref.ccs_kernel(5, n + 3, n, R.indptr, R.indices, v, xt),  # multiply_t_low :2161
                              orc.ccs_kernel(5, n + 3, n, R.indptr, R.indices, v, xt))
        C = R.T.tocsr()
        C.sort_indices()
        xs = rng.uniform(-1, 1, n + 3) + (1j * rng.uniform(-1, 1, n + 3) if cplx else 0)
        vv = C.data + (1j * rng.uniform(-1, 1, C.nnz) if cplx else 0)
        assert np.array_equal(ref.spmv(C.indptr, C.indices, vv, xs), orc.crs_mv(C.indptr, C.indices, vv, xs))
    A = poisson2d(9)
    x = rng.uniform(-1, 1, 81)
    assert np.array_equal(ref.spmv(A.indptr, A.indices, A.data, x), orc.crs_mv(A.indptr, A.indices, A.data, x))


@pytest.mark.parametrize("cplx", [False, True])
def test_qrcp_full_and_deficient_rank(cplx):
    rng = np.random.default_rng(11)
    n = 40
    A = rng.normal(size=(n, n)) + (1j * rng.normal(size=(n, n)) if cplx else 0)
    b = rng.normal(size=n) + (1j * rng.normal(size=n) if cplx else 0)
    x0, r0 = ref.qrcp(A.ravel(order="F"), b)
    x1, r1 = orc.qrcp(A.ravel(order="F"), b)
    assert r0 == r1 == n and relerr(x1, x0) <= 1e-12
    # exact rank deficiency: 7 dependent columns -> truncation through ?laic1
    A2 = A.copy()
    A2[:, n - 7:] = A2[:, :7] @ rng.normal(size=(7, 7))
    x0, r0 = ref.qrcp(A2.ravel(order="F"), b)
    x1, r1 = orc.qrcp(A2.ravel(order="F"), b)
    assert r0 == r1 == n - 7
    assert relerr(x1, x0) <= 1e-9  # minimum-norm-like solution of a singular system: conditioning-limited
    # graded singular values with a user cond threshold
    U, _ = np.linalg.qr(rng.normal(size=(n, n)))
    V, _ = np.linalg.qr(rng.normal(size=(n, n)))
    A3 = (U * np.logspace(0, -12, n)) @ V.T + (0j if cplx else 0)
    for cond in (1e4, 1e8):
        x0, r0 = ref.qrcp(A3.ravel(order="F"), b, rrqr_cond=cond)
        x1, r1 = orc.qrcp(A3.ravel(order="F"), b, rrqr_cond=cond)
        assert r0 == r1 and 0 < r0 < n
        assert relerr(x1, x0) <= 1e-8
    # explicit rank argument (prec_solve passes last_dim, QRCP.hpp:376-377)
    x0, _ = ref.qrcp(A.ravel(order="F"), b, rank=10)
    x1, _ = orc.qrcp(A.ravel(order="F"), b, rank=10)
    assert relerr(x1, x0) <= 1e-12
    y0, _ = ref.qrcp(A.ravel(order="F"), b, op=1)
    y1, _ = orc.qrcp(A.ravel(order="F"), b, op=1)
    assert relerr(y1, y0) <= 1e-12
    # A^H products (_multiply_t, QRCP.hpp:502-541)
    for rk in (0, 10):
        z0, _ = ref.qrcp(A.ravel(order="F"), b, op=3, rank=rk)
        z1, _ = orc.qrcp(A.ravel(order="F"), b, op=3, rank=rk)
        assert relerr(z1, z0) <= 1e-12
    # A^H solves (_solve_t, QRCP.hpp:413-452): full rank, explicit rank, rank-deficient
    for mat, rk, tol in ((A, 0, 1e-12), (A, 10, 1e-12), (A2, 0, 1e-9)):
        z0, _ = ref.qrcp(mat.ravel(order="F"), b, op=2, rank=rk)
        z1, _ = orc.qrcp(mat.ravel(order="F"), b, op=2, rank=rk)
        assert relerr(z1, z0) <= tol


@pytest.mark.parametrize("nx,params", [(40, None), (150, (1e-2, 5.0, 3.0)), (200, None)])
def test_fresh_hierarchies(nx, params):
    A = poisson2d(nx)
    P = None if params is None else ref.make_params(tau=params[0], kappa=params[1], alpha=params[2])
    M = ref.RefHIF(A.indptr, A.indices, A.data, P)
    O = orc.Oracle(M.levels())
    rng = np.random.default_rng(nx)
    b = rng.uniform(-1, 1, A.shape[0])
    x0, x1 = M.solve(b), O.solve(b)
    assert relerr(x1, x0) <= 1e-12
    if M.levels()[-1]["dense_n"] == 0:
        assert np.array_equal(x0, x1)
    assert relerr(O.mmultiply(x0), M.mmultiply(x0)) <= 1e-10
    assert relerr(O.mmultiply(x0, trans=True), M.mmultiply(x0, trans=True)) <= 1e-10
    t0, t1 = M.solve(b, trans=True), O.solve(b, trans=True)
    assert relerr(t1, t0) <= 1e-12
    if M.levels()[-1]["dense_n"] == 0:
        assert np.array_equal(t0, t1)


@pytest.mark.skipif(not ref.available_lup(), reason="reference build with HIF_DENSE_MODE=0 not present")
@pytest.mark.parametrize("cplx", [False, True])
def test_fresh_lup_hierarchies(cplx):
    """The reference compiled with HIF_DENSE_MODE=0 (last level = LU with partial pivoting, small_scale/LUP.hpp) against
    the restatement: solve and conjugate-transpose solve (LUP passes 'T' to ?getrs also for complex data, LUP.hpp:150).
    That build cannot instantiate HIF::mmultiply (LUP.hpp:181 vs prec_prod.hpp:85), so the product is not pinned."""
    import scipy.sparse as sp

    A = poisson2d(40)
    if cplx:
        A = (A.astype(np.complex128) - (0.3 + 0.2j) * sp.identity(A.shape[0])).tocsr()
        A.sort_indices()
    M = ref.RefHIF(A.indptr, A.indices, A.data, None, lup=True)
    levels = M.levels()
    assert levels[-1].get("dense_lup") == 1 and levels[-1]["dense_n"] > 0
    O = orc.Oracle(levels)
    rng = np.random.default_rng(3)
    b = rng.uniform(-1, 1, A.shape[0]) + (1j * rng.uniform(-1, 1, A.shape[0]) if cplx else 0)
    assert relerr(O.solve(b), M.solve(b)) <= 1e-12
    assert relerr(O.solve(b, trans=True), M.solve(b, trans=True)) <= 1e-12
    assert relerr(O.solve(b, rank=5), M.solve(b, rank=5)) <= 1e-12  # the rank is ignored (LUP.hpp:141)


@pytest.mark.parametrize("kind", ["real", "indefinite", "hermitian", "spd"])
def test_fresh_symmetric_hierarchies(kind):
    """is_symm factorizations of the real reference (symm_factor.hpp; last level = SYEIG, small_scale/SYEIG.hpp)
    against the restatement: solve, conjugate-transpose solve (the same operator), product, run-time rank."""
    import scipy.sparse as sp

    A = poisson2d(40)
    spd = 0
    if kind == "indefinite":
        A = (A - 1.3 * sp.identity(A.shape[0])).tocsr()
    elif kind == "hermitian":
        nx = 40
        S = sp.diags([-1.0, 1.0], [-1, 1], shape=(nx, nx), format="csr")
        A = (A.astype(np.complex128) + 0.3j * sp.kron(sp.identity(nx), S)).tocsr()
    elif kind == "spd":
        spd = 1
    A.sort_indices()
    M = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(is_symm=1, spd=spd))
    levels = M.levels()
    assert levels[-1].get("dense_symm") == 1 and levels[-1]["dense_n"] > 0
    O = orc.Oracle(levels)
    assert O.dense_rank == levels[-1]["dense_rank"]
    rng = np.random.default_rng(7)
    b = rng.uniform(-1, 1, A.shape[0])
    if kind == "hermitian":
        b = b + 1j * rng.uniform(-1, 1, A.shape[0])
    x0 = M.solve(b)
    assert relerr(O.solve(b), x0) <= 1e-11
    assert relerr(O.solve(b, trans=True), M.solve(b, trans=True)) <= 1e-11
    assert relerr(O.mmultiply(x0), M.mmultiply(x0)) <= 1e-10
    assert relerr(O.mmultiply(x0, trans=True), M.mmultiply(x0, trans=True)) <= 1e-10
    assert relerr(O.solve(b, rank=7), M.solve(b, rank=7)) <= 1e-11


@pytest.mark.parametrize("nx,params,rtol,maxit,restart", [(40, None, 1e-10, 200, 30), (100, (1e-2, 5.0, 3.0), 1e-8, 200, 10),
                                                         (100, (1e-2, 5.0, 3.0), 1e-12, 7, 30)])
def test_gmres_restatement(nx, params, rtol, maxit, restart):
    # the numpy restatement of examples/advanced/gmres.hpp:19-123 around the oracle's apply vs the real driver
    A = poisson2d(nx)
    P = None if params is None else ref.make_params(tau=params[0], kappa=params[1], alpha=params[2], dense_thres=100)
    M = ref.RefHIF(A.indptr, A.indices, A.data, P)
    O = orc.Oracle(M.levels())
    rng = np.random.default_rng(nx)
    b = rng.uniform(-1, 1, A.shape[0])
    x0, f0, i0 = M.gmres(b, restart=restart, rtol=rtol, maxit=maxit)
    x1, f1, i1 = orc.gmres(O, A.indptr, A.indices, A.data, b, restart=restart, rtol=rtol, maxit=maxit)
    assert (f0, i0) == (f1, i1)
    assert relerr(x1, x0) <= 1e-8
    # the flexible variant (fgmres_hifir, gmres.hpp:127-231): refinement sweeps as the preconditioner
    y0, g0, j0, m0 = M.fgmres(b, restart=restart, rtol=rtol, maxit=maxit)
    y1, g1, j1, m1 = orc.fgmres(O, A.indptr, A.indices, A.data, b, restart=restart, rtol=rtol, maxit=maxit)
    assert (g0, j0, m0) == (g1, j1, m1)
    assert relerr(y1, y0) <= 1e-8
