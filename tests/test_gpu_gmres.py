"""GPU (-m gpu): the batched right-preconditioned GMRES driver (hifamd_gmres_batch, the caller of the hot
path) against the numpy restatement of the reference's examples/advanced/gmres.hpp:19-123 around the
oracle's apply (oracle/orc.py gmres, pinned to the real driver in tests/test_oracle_vs_ref.py) and,
where the compiled reference travelled, against the real driver itself."""
import numpy as np
import pytest
import scipy.sparse as sp

import hifir_amd
from oracle import orc, ref
from util import load_hier, relerr

pytestmark = pytest.mark.gpu

CASES = [  # fixture, restart, rtol, maxit
    ("p2d_30", 30, 1e-10, 200),
    ("p2d_100_tuned", 30, 1e-8, 200),   # 40+ inner iterations: crosses a restart
    ("p2d_100_tuned", 10, 1e-8, 200),
    ("cd2d_48", 30, 1e-10, 200),        # nonsymmetric
    ("demo_A", 30, 1e-10, 200),
    ("p2d_100_tuned", 30, 1e-14, 7),    # runs into maxit -> flag 2
]


@pytest.fixture(scope="module")
def cache():
    return {}


def _get(cache, name):
    if name not in cache:
        levels, d = load_hier(name)
        M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
        M.set_matrix(d["A_indptr"], d["A_indices"], d["A_vals"])
        cache[name] = (levels, d, M, orc.Oracle(levels))
    return cache[name]


@pytest.mark.parametrize("name,restart,rtol,maxit", CASES)
def test_gmres_single_rhs(cache, name, restart, rtol, maxit):
    levels, d, M, O = _get(cache, name)
    b = d["b"]
    n = len(b)
    A = sp.csr_matrix((d["A_vals"], d["A_indices"], d["A_indptr"]), shape=(n, n))
    x, flag, it = M.gmres(b, restart=restart, rtol=rtol, maxit=maxit)
    xo, fo, io = orc.gmres(O, d["A_indptr"], d["A_indices"], d["A_vals"], b, restart=restart, rtol=rtol, maxit=maxit)
    assert (flag, it) == (fo, io)
    assert relerr(x, xo) <= 1e-8
    if flag == 0:
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) <= 10 * rtol
    if ref.available():
        R = ref.RefHIF(d["A_indptr"], d["A_indices"], d["A_vals"], None if not d["params"].any() else d["params"])
        xr, fr, ir = R.gmres(b, restart=restart, rtol=rtol, maxit=maxit)
        assert (flag, it) == (fr, ir)
        assert relerr(x, xr) <= 1e-8


@pytest.mark.parametrize("name", ["p2d_100_tuned", "cd2d_48"])
def test_gmres_batch_columns_are_independent(cache, name):
    # columns of very different difficulty in one lock-step batch: each must behave like its own solve
    torch = pytest.importorskip("torch")
    levels, d, M, O = _get(cache, name)
    n = len(d["b"])
    rng = np.random.default_rng(23)
    B = rng.uniform(-1, 1, size=(n, 5))
    B[:, 1] = 0.0                                   # zero right-hand side: quick return, 0 iterations
    B[:, 2] = d["b"]
    B[:, 3] = sp.csr_matrix((d["A_vals"], d["A_indices"], d["A_indptr"]), shape=(n, n)) @ np.ones(n)
    B[:, 4] *= 1e-30                                # scaling must not matter for a relative test
    X, fl, it = M.gmres(B, restart=20, rtol=1e-9, maxit=100)
    for k in range(B.shape[1]):
        xo, fo, io = orc.gmres(O, d["A_indptr"], d["A_indices"], d["A_vals"], B[:, k].copy(), restart=20, rtol=1e-9, maxit=100)
        assert (int(fl[k]), int(it[k])) == (fo, io), k
        assert relerr(X[:, k], xo) <= 1e-7 or np.abs(xo).max() == 0.0
    assert int(it[1]) == 0 and not X[:, 1].any()
    # device-pointer entry: same results
    Xd, fl2, it2 = M.gmres(torch.from_numpy(B).cuda(), restart=20, rtol=1e-9, maxit=100)
    assert np.array_equal(fl2, fl) and np.array_equal(it2, it)
    assert np.array_equal(Xd.cpu().numpy(), X)


def test_gmres_wide_batch_and_errors(cache):
    levels, d, M, O = _get(cache, "p2d_30")
    n = len(d["b"])
    rng = np.random.default_rng(5)
    B = rng.uniform(-1, 1, size=(n, 70))  # two 64-column tiles
    X, fl, it = M.gmres(B, restart=30, rtol=1e-10, maxit=60)
    A = sp.csr_matrix((d["A_vals"], d["A_indices"], d["A_indptr"]), shape=(n, n))
    assert not fl.any()
    assert (np.linalg.norm(A @ X - B, axis=0) / np.linalg.norm(B, axis=0)).max() <= 1e-9
    with pytest.raises(hifir_amd.HifAmdError):
        M.gmres(B, restart=0)


@pytest.mark.parametrize("name,restart,rtol,maxit", [("p2d_100_tuned", 10, 1e-8, 200), ("cd2d_48", 30, 1e-10, 100),
                                                      ("p2d_100_tuned", 8, 1e-14, 9)])
def test_flexible_gmres(cache, name, restart, rtol, maxit):
    # fgmres_hifir (gmres.hpp:127-231): refinement sweeps (1, 2, 4, ... per outer cycle) as the preconditioner
    levels, d, M, O = _get(cache, name)
    n = len(d["b"])
    rng = np.random.default_rng(31)
    B = rng.uniform(-1, 1, size=(n, 3))
    B[:, 0] = d["b"]
    X, fl, it, mv = M.fgmres(B, restart=restart, rtol=rtol, maxit=maxit)
    for k in range(3):
        xo, fo, io, mo = orc.fgmres(O, d["A_indptr"], d["A_indices"], d["A_vals"], B[:, k].copy(), restart=restart,
                                    rtol=rtol, maxit=maxit)
        assert (int(fl[k]), int(it[k]), int(mv[k])) == (fo, io, mo), k
        assert relerr(X[:, k], xo) <= 1e-7
    if ref.available():
        R = ref.RefHIF(d["A_indptr"], d["A_indices"], d["A_vals"], None if not d["params"].any() else d["params"])
        xr, fr, ir, mr = R.fgmres(d["b"], restart=restart, rtol=rtol, maxit=maxit)
        assert (int(fl[0]), int(it[0]), int(mv[0])) == (fr, ir, mr)
        assert relerr(X[:, 0], xr) <= 1e-7


def _perturbed(d, amp, seed=7):
    """the fixture's matrix with a random complex diagonal added: the hierarchy becomes a mediocre preconditioner,
    so that the solve needs tens of iterations and crosses restarts"""
    n = len(d["b"])
    A = sp.csr_matrix((d["A_vals"], d["A_indices"], d["A_indptr"]), shape=(n, n))
    if amp:
        rng = np.random.default_rng(seed)
        A = (A + sp.diags(amp * abs(A).max() * (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)))).tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("name,amp,restart,rtol,maxit", [("young1c", 0.0, 30, 1e-10, 200), ("young1c", 0.05, 12, 1e-9, 300),
                                                          ("kkt_26", 0.0, 5, 1e-6, 100), ("kkt_26", 0.05, 12, 1e-9, 300),
                                                          ("young1c", 0.05, 6, 1e-13, 11)])
def test_complex_gmres(name, amp, restart, rtol, maxit):
    # complex handles: Hermitian inner product h = sum conj(q) v (the example's hif::inner(v, q) is its conjugate and
    # does not orthogonalize, oracle/orc.py gmres), rotations with the conjugates of gmres.hpp:75-83; checked against
    # the numpy restatement around the oracle's apply and by the true residual
    levels, d = load_hier(name)
    A = _perturbed(d, amp)
    n = A.shape[0]
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=8)
    M.set_matrix(A.indptr, A.indices, A.data)
    O = orc.Oracle(levels)
    rng = np.random.default_rng(3)
    B = rng.uniform(-1, 1, size=(n, 5)) + 1j * rng.uniform(-1, 1, size=(n, 5))
    B[:, 0] = d["b"]
    B[:, 3] = 0.0                      # quick return inside a batch
    B[:, 4] = 1e-3 * B[:, 1]           # same Krylov space, other scale
    X, fl, it = M.gmres(B, restart=restart, rtol=rtol, maxit=maxit)
    assert X.dtype == np.complex128
    for k in range(5):
        xo, fo, io = orc.gmres(O, A.indptr, A.indices, A.data, B[:, k].copy(), restart=restart, rtol=rtol, maxit=maxit)
        assert (int(fl[k]), int(it[k])) == (fo, io), k
        if k == 3:
            assert not X[:, 3].any()
            continue
        assert relerr(X[:, k], xo) <= 1e-8, k
        if fo == 0:
            assert np.linalg.norm(A @ X[:, k] - B[:, k]) / np.linalg.norm(B[:, k]) <= 10 * rtol
    # one column alone behaves like the same column in the batch; device tensors in, device tensors out
    import torch

    x1, f1, i1 = M.gmres(B[:, 1].copy(), restart=restart, rtol=rtol, maxit=maxit)
    assert (f1, i1) == (int(fl[1]), int(it[1])) and relerr(x1, X[:, 1]) <= 1e-10
    Xd, fl2, it2 = M.gmres(torch.from_numpy(B).cuda(), restart=restart, rtol=rtol, maxit=maxit)
    assert np.array_equal(fl2, fl) and np.array_equal(it2, it) and relerr(Xd.cpu().numpy(), X) <= 1e-12
    M.close()


@pytest.mark.parametrize("name,amp", [("young1c", 0.05), ("kkt_26", 0.05)])
def test_complex_flexible_gmres(name, amp):
    levels, d = load_hier(name)
    A = _perturbed(d, amp)
    n = A.shape[0]
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=8)
    M.set_matrix(A.indptr, A.indices, A.data)
    O = orc.Oracle(levels)
    rng = np.random.default_rng(4)
    B = rng.uniform(-1, 1, size=(n, 2)) + 1j * rng.uniform(-1, 1, size=(n, 2))
    X, fl, it, mv = M.fgmres(B, restart=6, rtol=1e-9, maxit=60)
    for k in range(2):
        xo, fo, io, mo = orc.fgmres(O, A.indptr, A.indices, A.data, B[:, k].copy(), restart=6, rtol=1e-9, maxit=60)
        assert (int(fl[k]), int(it[k]), int(mv[k])) == (fo, io, mo), k
        assert relerr(X[:, k], xo) <= 1e-7
        if fo == 0:
            assert np.linalg.norm(A @ X[:, k] - B[:, k]) / np.linalg.norm(B[:, k]) <= 1e-8
    M.close()
