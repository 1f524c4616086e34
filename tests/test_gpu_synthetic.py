"""GPU (-m gpu): synthetic hierarchies that no factorization would produce, to reach the corners of
the band planner and of the apply driver: very deep chains (1 row per wavefront), a thin run whose
block inverse blows up (must fall back to the sequential, backward-stable scheme), random
multi-level hierarchies with random permutations/scalings, empty blocks (m = 0, m = n, F absent)."""
import numpy as np
import pytest
import scipy.sparse as sp

import hifir_amd
from oracle import orc
from util import relerr

pytestmark = pytest.mark.gpu


def _ccs(A):
    A = sp.csc_matrix(A)
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data.astype(np.float64)


def _level(m, n, L, U, E, F, rng, with_F=True):
    lv = dict(m=m, n=n)
    for k, M in (("L", L), ("U", U), ("E", E), ("F", F)):
        cp, ri, v = _ccs(M)
        lv[k + "_colptr"], lv[k + "_rowind"], lv[k + "_vals"] = cp, ri, v
    if not with_F:
        lv["F_colptr"], lv["F_rowind"], lv["F_vals"] = np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0)
    lv["d"] = rng.uniform(0.5, 2.0, m) * rng.choice([-1.0, 1.0], m)
    lv["s"], lv["t"] = rng.uniform(0.5, 2.0, n), rng.uniform(0.5, 2.0, n)
    lv["p"] = rng.permutation(n).astype(np.int32)
    lv["q"] = rng.permutation(n).astype(np.int32)
    lv["p_inv"] = np.argsort(lv["p"]).astype(np.int32)
    lv["q_inv"] = np.argsort(lv["q"]).astype(np.int32)
    return lv


def _rand_tri(m, density, lower, rng, scale=0.3):
    A = sp.random(m, m, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="csr")
    A.data = rng.uniform(-scale, scale, A.nnz)
    return sp.tril(A, -1) if lower else sp.triu(A, 1)


def _check(levels, nrhs, exact_expected, tol=1e-12, seed=0):
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=min(nrhs, 64))
    O = orc.Oracle(levels)
    n = int(levels[0]["n"])
    B = np.random.default_rng(seed).uniform(-1, 1, size=(n, nrhs))
    X, Xo = M.solve_mrhs(B), O.solve_batch(B, threads=4)
    if exact_expected:
        assert np.array_equal(X, Xo), relerr(X, Xo)
    assert relerr(X, Xo) <= tol
    # the other three operators of lhf?Apply on the same (possibly degenerate) hierarchy, first columns
    k = min(nrhs, 3)
    Bk = np.ascontiguousarray(B[:, :k])
    XH = M.solve_mrhs(Bk, trans=True)
    Y, YH = M.mmultiply(Bk), M.mmultiply(Bk, trans=True)
    for c in range(k):
        assert relerr(XH[:, c], O.solve(Bk[:, c].copy(), trans=True)) <= tol * 10
        assert relerr(Y[:, c], O.mmultiply(Bk[:, c].copy(), rank=-1)) <= 1e-10
        assert relerr(YH[:, c], O.mmultiply(Bk[:, c].copy(), rank=-1, trans=True)) <= 1e-10
    return M


@pytest.mark.parametrize("coef,exact", [(-2.0, True), (-0.5, False)])
def test_deep_chain_and_growth_fallback(coef, exact):
    # L = I + coef * subdiagonal: 400 wavefronts of one row each.  coef = -2: the block inverse has
    # entries up to 2^399 -> the planner must keep the sequential scheme (then bit-exact);
    # coef = -0.5: benign -> block-dense path (tolerance).
    m = 400
    rng = np.random.default_rng(1)
    L = sp.diags([np.full(m - 1, coef)], [-1], shape=(m, m))
    # same for U (coef = -2: |x| grows to ~2^800 < 1.8e308, still finite)
    U = sp.diags([np.full(m - 1, coef)], [1], shape=(m, m)) if coef == -2.0 else \
        sp.diags([rng.uniform(-0.4, 0.4, m - 1)], [1], shape=(m, m))
    lv = _level(m, m, L, U, sp.csr_matrix((0, m)), sp.csr_matrix((m, 0)), rng)
    lv["d"] = np.ones(m)
    _check([lv], 64, exact_expected=exact)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_two_level_hierarchy_with_dense_tail(seed):
    rng = np.random.default_rng(seed)
    n0, m0 = 3000, 2300
    n1, m1 = n0 - m0, 520
    nd = n1 - m1
    lv0 = _level(m0, n0, _rand_tri(m0, 0.004, True, rng), _rand_tri(m0, 0.004, False, rng),
                 sp.random(n0 - m0, m0, density=0.01, random_state=np.random.RandomState(seed), format="csr"),
                 sp.random(m0, n0 - m0, density=0.01, random_state=np.random.RandomState(seed + 7), format="csr"), rng)
    lv1 = _level(m1, n1, _rand_tri(m1, 0.02, True, rng), _rand_tri(m1, 0.02, False, rng),
                 sp.random(nd, m1, density=0.05, random_state=np.random.RandomState(seed + 1), format="csr"),
                 sp.random(m1, nd, density=0.05, random_state=np.random.RandomState(seed + 2), format="csr"), rng)
    D = rng.normal(size=(nd, nd)) + 4.0 * np.eye(nd)
    lv1["dense_n"], lv1["dense"] = nd, D.ravel(order="F")
    for nrhs in (1, 7, 64):
        _check([lv0, lv1], nrhs, exact_expected=False, tol=1e-12, seed=seed)


def test_degenerate_shapes():
    rng = np.random.default_rng(5)
    # (a) m == n: last level without dense block, single LDU
    m = 257
    a = _level(m, m, _rand_tri(m, 0.05, True, rng), _rand_tri(m, 0.05, False, rng), sp.csr_matrix((0, m)),
               sp.csr_matrix((m, 0)), rng)
    _check([a], 5, exact_expected=False)
    # (b) F absent although n > m (prec_solve.hpp:400-403), dense tail
    n, m = 300, 200
    b = _level(m, n, _rand_tri(m, 0.05, True, rng), _rand_tri(m, 0.05, False, rng),
               sp.random(n - m, m, density=0.1, random_state=np.random.RandomState(3), format="csr"),
               sp.csr_matrix((m, 0)), rng, with_F=False)
    b["dense_n"], b["dense"] = n - m, (rng.normal(size=(n - m, n - m)) + 5 * np.eye(n - m)).ravel(order="F")
    _check([b], 9, exact_expected=False)
    # (c) m == 0: everything deferred to the dense block
    n = 64
    c = _level(0, n, sp.csr_matrix((0, 0)), sp.csr_matrix((0, 0)), sp.csr_matrix((n, 0)), sp.csr_matrix((0, n)), rng)
    c["dense_n"], c["dense"] = n, (rng.normal(size=(n, n)) + 6 * np.eye(n)).ravel(order="F")
    _check([c], 3, exact_expected=False)
    # (d) empty triangles (diagonal leading block): one wavefront
    m = 1000
    d = _level(m, m, sp.csr_matrix((m, m)), sp.csr_matrix((m, m)), sp.csr_matrix((0, m)), sp.csr_matrix((m, 0)), rng)
    _check([d], 64, exact_expected=True)


def test_rank_deficient_dense_block_all_operators():
    # a dense tail with 9 exactly dependent columns: QRCP truncates through ?laic1 (QRCP.hpp:333-364) and every
    # operator uses the numerical rank (solve) or the rank it is given
    rng = np.random.default_rng(12)
    n, m = 400, 260
    nd = n - m
    lv = _level(m, n, _rand_tri(m, 0.05, True, rng), _rand_tri(m, 0.05, False, rng),
                sp.random(nd, m, density=0.1, random_state=np.random.RandomState(3), format="csr"),
                sp.random(m, nd, density=0.1, random_state=np.random.RandomState(4), format="csr"), rng)
    D = rng.normal(size=(nd, nd)) + 3 * np.eye(nd)
    D[:, nd - 9:] = D[:, :9] @ rng.normal(size=(9, 9))
    lv["dense_n"], lv["dense"] = nd, D.ravel(order="F")
    M = hifir_amd.HIF.from_levels([lv], max_nrhs=8)
    O = orc.Oracle([lv])
    assert M.schur_rank() == O.dense_rank == nd - 9
    B = rng.uniform(-1, 1, size=(n, 4))
    for c in range(4):
        b = B[:, c].copy()
        assert relerr(M.solve(b), O.solve(b)) <= 1e-9
        assert relerr(M.solve(b, trans=True), O.solve(b, trans=True)) <= 1e-9
        assert relerr(M.solve(b, rank=50), O.solve(b, rank=50)) <= 1e-9
        assert relerr(M.mmultiply(b, rank=0), O.mmultiply(b, rank=0)) <= 1e-9
        assert relerr(M.mmultiply(b, rank=0, trans=True), O.mmultiply(b, rank=0, trans=True)) <= 1e-9


def _two_level_with_dense(seed, D, n0=3000, m0=2300, m1=520):
    rng = np.random.default_rng(seed)
    n1 = n0 - m0
    nd = n1 - m1
    assert D.shape == (nd, nd)
    lv0 = _level(m0, n0, _rand_tri(m0, 0.004, True, rng), _rand_tri(m0, 0.004, False, rng),
                 sp.random(n0 - m0, m0, density=0.01, random_state=np.random.RandomState(seed), format="csr"),
                 sp.random(m0, n0 - m0, density=0.01, random_state=np.random.RandomState(seed + 7), format="csr"), rng)
    lv1 = _level(m1, n1, _rand_tri(m1, 0.02, True, rng), _rand_tri(m1, 0.02, False, rng),
                 sp.random(nd, m1, density=0.05, random_state=np.random.RandomState(seed + 1), format="csr"),
                 sp.random(m1, nd, density=0.05, random_state=np.random.RandomState(seed + 2), format="csr"), rng)
    lv1["dense_n"], lv1["dense"] = nd, D.ravel(order="F")
    return [lv0, lv1]


def test_tail_operator_with_rank_deficient_block(monkeypatch):
    # The tail of the hierarchy (level 1 and the dense block behind it) is applied as ONE operator formed at finalize
    # (engine.hip build_tail_operator) -- here with a rank-deficient QRCP block inside it (9 exactly dependent columns,
    # QRCP.hpp:371-411 truncates): the numerical rank goes through the operator, any other rank through the recursion;
    # both must agree with the oracle as well as the recursion alone does.
    rng = np.random.default_rng(21)
    nd = 180
    D = rng.normal(size=(nd, nd)) + 3 * np.eye(nd)
    D[:, nd - 9:] = D[:, :9] @ rng.normal(size=(9, 9))
    levels = _two_level_with_dense(21, D)
    O = orc.Oracle(levels)
    B = rng.uniform(-1, 1, size=(3000, 64))
    monkeypatch.setenv("HIFIR_AMD_TAIL_ROWS", "0")
    M0 = hifir_amd.HIF.from_levels(levels, max_nrhs=64)  # the recursion alone
    monkeypatch.delenv("HIFIR_AMD_TAIL_ROWS")
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    se = M.stats_ext()
    assert M0.stats_ext()["tail_rows"] == 0
    assert se["tail_rows"] == 700 or se["tail_rejected"] in (2.0, 3.0), se  # formed, or refused by a guard (never silently)
    assert M.schur_rank() == O.dense_rank == nd - 9
    # (rank = -1, "full", divides by the block's rounding-level pivots: no two implementations agree on that noise)
    for rank in (0, 50, nd - 9, 120):
        X, X0 = M.solve_mrhs(B, rank=rank), M0.solve_mrhs(B, rank=rank)
        for c in (0, 17, 63):
            xo = O.solve(B[:, c].copy(), rank=rank)
            e0 = relerr(X0[:, c], xo)
            assert e0 <= 1e-8, (rank, c, e0)
            assert relerr(X[:, c], xo) <= max(10 * e0, 1e-12), (rank, c, relerr(X[:, c], xo), e0, se)


def test_tail_operator_with_ill_conditioned_coarse_level(monkeypatch):
    # kappa(dense block) ~ 1e10: product and recursion differ by kappa * eps -- the finalize-time probe must notice and
    # keep the recursion (or the operator must be as good as the recursion against the oracle)
    rng = np.random.default_rng(22)
    nd = 180
    Q1, _ = np.linalg.qr(rng.normal(size=(nd, nd)))
    Q2, _ = np.linalg.qr(rng.normal(size=(nd, nd)))
    D = (Q1 * np.logspace(0, -10, nd)) @ Q2.T
    levels = _two_level_with_dense(22, D)
    O = orc.Oracle(levels)
    B = rng.uniform(-1, 1, size=(3000, 64))
    monkeypatch.setenv("HIFIR_AMD_TAIL_ROWS", "0")
    M0 = hifir_amd.HIF.from_levels(levels, max_nrhs=64, rrqr_cond=1e14)
    monkeypatch.delenv("HIFIR_AMD_TAIL_ROWS")
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64, rrqr_cond=1e14)
    se = M.stats_ext()
    X, X0 = M.solve_mrhs(B, rank=-1), M0.solve_mrhs(B, rank=-1)
    for c in (0, 31, 63):
        xo = O.solve(B[:, c].copy(), rank=-1)
        e0 = relerr(X0[:, c], xo)
        assert relerr(X[:, c], xo) <= max(10 * e0, 1e-12), (c, relerr(X[:, c], xo), e0, se)
    # the guard's verdict is visible: either the probe passed at its limit, or the operator was refused
    assert (se["tail_rows"] > 0 and se["tail_probe_relerr"] <= se["tail_probe_tol"]) or se["tail_rejected"] in (1.0, 2.0, 3.0), se


def test_non_finite_column_does_not_poison_later_solves():
    # level 0 with fewer than 32 leading rows (the tail product's right-hand side then sits right in front of rows that
    # the same solve writes) -- a solve with a NaN column, then a finite solve on the same handle: exact same result as
    # on a fresh handle (round 2's product read up to 31 rows behind its panel through zero columns: 0 x NaN)
    rng = np.random.default_rng(23)
    n0, m0 = 620, 20
    n1 = n0 - m0
    m1 = 420
    nd = n1 - m1
    lv0 = _level(m0, n0, _rand_tri(m0, 0.2, True, rng), _rand_tri(m0, 0.2, False, rng),
                 sp.random(n1, m0, density=0.05, random_state=np.random.RandomState(1), format="csr"),
                 sp.random(m0, n1, density=0.05, random_state=np.random.RandomState(2), format="csr"), rng)
    lv1 = _level(m1, n1, _rand_tri(m1, 0.02, True, rng), _rand_tri(m1, 0.02, False, rng),
                 sp.random(nd, m1, density=0.05, random_state=np.random.RandomState(3), format="csr"),
                 sp.random(m1, nd, density=0.05, random_state=np.random.RandomState(4), format="csr"), rng)
    lv1["dense_n"], lv1["dense"] = nd, (rng.normal(size=(nd, nd)) + 4 * np.eye(nd)).ravel(order="F")
    levels = [lv0, lv1]
    O = orc.Oracle(levels)
    B = rng.uniform(-1, 1, size=(n0, 64))
    Mf = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    Xf = Mf.solve_mrhs(B)
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    Bn = B.copy()
    Bn[:, 5] = np.nan
    Bn[3, 9] = np.inf
    Xn = M.solve_mrhs(Bn)
    assert not np.isfinite(Xn[:, 5]).any() or np.isnan(Xn[:, 5]).any()
    good = [c for c in range(64) if c not in (5, 9)]
    assert np.array_equal(Xn[:, good], Xf[:, good])  # columns are independent of each other
    X2 = M.solve_mrhs(B)
    assert np.isfinite(X2).all()
    assert np.array_equal(X2, Xf)
    for c in (0, 5, 9, 63):
        assert relerr(X2[:, c], O.solve(B[:, c].copy())) <= 1e-12


def test_one_based_matrix_input():
    # the outer matrix may come 1-based like the reference accepts it (builder.hpp:311-329)
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    m = 500
    lv = _level(m, m, _rand_tri(m, 0.02, True, rng), _rand_tri(m, 0.02, False, rng), sp.csr_matrix((0, m)),
                sp.csr_matrix((m, 0)), rng)
    A = (sp.random(m, m, density=0.02, random_state=np.random.RandomState(1), format="csr") + 4 * sp.identity(m)).tocsr()
    A.sort_indices()
    X = rng.uniform(-1, 1, size=(m, 5))
    Y = []
    for base in (0, 1):
        M = hifir_amd.HIF.from_levels([lv], max_nrhs=8)
        M.set_matrix(A.indptr.astype(np.int64) + base, A.indices.astype(np.int32) + base, A.data)
        Yd = M.spmv(torch.from_numpy(X).cuda())
        M.sync()
        Y.append(Yd.cpu().numpy())
    assert np.array_equal(Y[0], Y[1])
    assert relerr(Y[0], A @ X) <= 1e-14
