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
    lz, dz = load_hier("young1c")
    Mz = hifir_amd.HIF.from_levels(lz, max_nrhs=4)
    Mz.set_matrix(dz["A_indptr"], dz["A_indices"], dz["A_vals"])
    with pytest.raises(hifir_amd.HifAmdError):  # real-valued driver only
        Mz.gmres(dz["b"])


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
