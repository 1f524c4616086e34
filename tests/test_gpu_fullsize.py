"""GPU (-m gpu), BASELINE.json's full sizes: the 1M-row 2-D Poisson hierarchy is factorized on the
box's host by the compiled reference (oracle/_ref, test infrastructure) and applied by the HIP path.
Checked through size-independent properties plus a few columns against the oracle:
  * batch columns are bit-identical to single-RHS solves (column separability),
  * linearity  M^-1(a b1 + c b2) = a M^-1 b1 + c M^-1 b2  to rounding,
  * round trip M (M^-1 b) = b with the oracle's prec_prod (libhifir/tests/test_real.c:110-146, 1e-10),
  * iterative refinement (both variants) and the outer SpMV against the real reference,
  * the conjugate-transpose apply (LHF_SH) against the real reference and the oracle,
  * the batched GMRES driver against the reference's own example driver (iteration counts, flags, x),
  * 3-D 7-pt Poisson and a complex (Helmholtz-like) system at moderate size vs the oracle."""
import numpy as np
import pytest
import scipy.sparse as sp

import hifir_amd
from oracle import orc, ref
from util import poisson2d, relerr, stokes_kkt

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref.available(), reason="compiled reference not present")]


@pytest.fixture(scope="module", params=["fast", "exact"])
def big(request):
    import os

    # "fast" = the default configuration (block-dense thin bands); "exact" keeps the reference's
    # summation order everywhere (HIFIR_AMD_DENSE_BLOCK=0) and must reproduce the oracle bit for bit
    os.environ["HIFIR_AMD_DENSE_BLOCK"] = "0" if request.param == "exact" else "2048"
    A = poisson2d(1000)
    R = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0))
    levels = R.levels()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    M.set_matrix(A.indptr, A.indices, A.data)
    os.environ.pop("HIFIR_AMD_DENSE_BLOCK", None)
    M.exact = request.param == "exact"
    return A, R, levels, M, orc.Oracle(levels)


def test_1m_columns_vs_oracle_and_reference(big):
    A, R, levels, M, O = big
    n = A.shape[0]
    rng = np.random.default_rng(1)
    B = rng.uniform(-1, 1, size=(n, 64))
    X = M.solve_mrhs(B)
    for k in (0, 31, 63):
        xo = O.solve(B[:, k].copy())
        if M.exact:
            assert np.array_equal(X[:, k], xo)  # sparse-only hierarchy: bit-exact at full size
        assert relerr(X[:, k], xo) <= 1e-12
    assert relerr(X[:, 5], R.solve(B[:, 5].copy())) <= 1e-12  # the real reference
    # nrhs = 1 (BASELINE config 2) and an odd batch width take different lane mappings: same bits
    assert np.array_equal(M.solve(B[:, 7].copy()), X[:, 7])
    X3 = M.solve_mrhs(np.ascontiguousarray(B[:, :3]))
    assert np.array_equal(X3, X[:, :3])


@pytest.fixture(scope="module", params=["fast", "exact"])
def big_default(request):
    """THE headline workload (BASELINE configs[1,2], what bench.py times): 1M-row Poisson factorized with
    DEFAULT_PARAMS -- 6 levels, 582^2 dense QRCP level, ~2,900 wavefronts per triangle set."""
    import os

    os.environ["HIFIR_AMD_DENSE_BLOCK"] = "0" if request.param == "exact" else "2048"
    A = poisson2d(1000)
    R = ref.RefHIF(A.indptr, A.indices, A.data)
    levels = R.levels()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    os.environ.pop("HIFIR_AMD_DENSE_BLOCK", None)
    M.exact = request.param == "exact"
    return A, R, levels, M, orc.Oracle(levels)


def test_1m_default_hierarchy_columns(big_default):
    A, R, levels, M, O = big_default
    assert len(levels) >= 5 and int(levels[-1]["dense_n"]) > 0  # multilevel + dense last level
    n = A.shape[0]
    rng = np.random.default_rng(12)
    B = rng.uniform(-1, 1, size=(n, 64))
    X = M.solve_mrhs(B)
    for k in (0, 31, 63):
        assert relerr(X[:, k], O.solve(B[:, k].copy())) <= 1e-12  # (the dense level rules out bit-exactness)
    assert relerr(X[:, 17], R.solve(B[:, 17].copy())) <= 1e-12     # the compiled reference itself
    # column separability: nrhs = 1 (BASELINE configs[1]) and nrhs = 3 give the bits of the 64-column batch
    assert np.array_equal(M.solve(B[:, 7].copy()), X[:, 7])
    assert np.array_equal(M.solve_mrhs(np.ascontiguousarray(B[:, :3])), X[:, :3])
    # conjugate-transpose apply and the product round trip on the same hierarchy
    XT = M.solve_mrhs(np.ascontiguousarray(B[:, :4]), trans=True)
    assert relerr(XT[:, 1], R.solve(B[:, 1].copy(), trans=True)) <= 1e-12
    Y = M.mmultiply(np.ascontiguousarray(X[:, :4]))
    assert (np.linalg.norm(Y - B[:, :4], axis=0) / np.linalg.norm(B[:, :4], axis=0)).max() <= 1e-10


def test_1m_default_fusions_and_kernel_choices_agree(big_default):
    # the fused stages (S1 / S5 / S7 inside the component bands), the tiled Schur products and the re-tiled top product
    # are choices of HOW the same operator is applied: with each of them switched off the columns agree to rounding,
    # and the launch count shows that the switches did change the graph
    import os

    A, R, levels, M, O = big_default
    if M.exact:
        pytest.skip("exact mode plans none of these")
    n = A.shape[0]
    rng = np.random.default_rng(21)
    B = rng.uniform(-1, 1, size=(n, 64))
    X = M.solve_mrhs(B)
    base_launches = M.stats()["launches"]
    seen = set()
    for env in ({"HIFIR_AMD_FUSE_S1": "0"}, {"HIFIR_AMD_FUSE_F": "0"}, {"HIFIR_AMD_FUSE_S7": "0"},
                {"HIFIR_AMD_SPMM_TILES": "0"}, {"HIFIR_AMD_SPMM_SPLIT": "0", "HIFIR_AMD_TOP_GEMM": "1"},
                {"HIFIR_AMD_CD_SPARSE_ROWS": "0"}, {"HIFIR_AMD_TOP_ROWS": "0", "HIFIR_AMD_CD_NNZ": "0"},
                {"HIFIR_AMD_TAIL_ROWS": "0"}, {"HIFIR_AMD_TAIL_ROWS": "12000"}):
        os.environ.update(env)
        try:
            M2 = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
        finally:
            for k in env:
                os.environ.pop(k, None)
        X2 = M2.solve_mrhs(B)
        seen.add(M2.stats()["launches"])
        M2.close()
        assert relerr(X2, X) <= 1e-12, env
        assert relerr(X2[:, 5], O.solve(B[:, 5].copy())) <= 1e-12, env
    assert len(seen - {base_launches}) >= 3


def test_1m_transposed_apply(big):
    # LHF_SH at full size: M^{-H} through the adjoint hierarchy vs the real reference's
    # HIF::solve(b, x, true) and the oracle's prec_solve_tran
    A, R, levels, M, O = big
    n = A.shape[0]
    rng = np.random.default_rng(6)
    B = rng.uniform(-1, 1, size=(n, 64))
    X = M.solve_mrhs(B, trans=True)
    for k in (0, 63):
        assert relerr(X[:, k], O.solve(B[:, k].copy(), trans=True)) <= 1e-12
    assert relerr(X[:, 9], R.solve(B[:, 9].copy(), trans=True)) <= 1e-12
    assert np.array_equal(M.solve(B[:, 7].copy(), trans=True), X[:, 7])


def test_1m_gmres_matches_reference_driver(big):
    # the caller of the hot path at full size: batched GMRES(30) vs the reference's own driver
    # (examples/advanced/gmres.hpp:19-123) on one column; every column must converge
    A, R, levels, M, O = big
    n = A.shape[0]
    rng = np.random.default_rng(8)
    B = rng.uniform(-1, 1, size=(n, 8))
    # (the tuned hierarchy is a weak preconditioner at 1M rows: GMRES(30) needs ~100 steps per decade)
    X, fl, it = M.gmres(B, restart=30, rtol=1e-2, maxit=150)
    assert not fl.any()
    assert (np.linalg.norm(A @ X - B, axis=0) / np.linalg.norm(B, axis=0)).max() <= 2e-2
    xr, fr, ir = R.gmres(B[:, 2].copy(), restart=30, rtol=1e-2, maxit=150)
    assert (int(fl[2]), int(it[2])) == (fr, ir)
    assert relerr(X[:, 2], xr) <= 1e-7
    # the iteration cap is reported per column like the driver does (flag 2, maxit iterations)
    X2, fl2, it2 = M.gmres(np.ascontiguousarray(B[:, :2]), restart=30, rtol=1e-9, maxit=12)
    assert list(fl2) == [2, 2] and list(it2) == [12, 12]


def test_1m_linearity_and_roundtrip(big):
    A, R, levels, M, O = big
    n = A.shape[0]
    rng = np.random.default_rng(2)
    b1, b2 = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    B = np.stack([b1, b2, 2.0 * b1 - 0.5 * b2], axis=1)
    X = M.solve_mrhs(B)
    lin = 2.0 * X[:, 0] - 0.5 * X[:, 1]
    assert relerr(X[:, 2], lin) <= 1e-12
    b_back = O.mmultiply(X[:, 0].copy())
    assert np.linalg.norm(b_back - b1) / np.linalg.norm(b1) <= 1e-10
    # the same round trip entirely on the device (LHF_M / LHF_MH), both directions, vs oracle and reference
    Y = M.mmultiply(X)
    assert (np.linalg.norm(Y - B, axis=0) / np.linalg.norm(B, axis=0)).max() <= 1e-10
    assert relerr(Y[:, 0], b_back) <= 1e-10
    assert relerr(Y[:, 1], R.mmultiply(X[:, 1].copy())) <= 1e-10
    XH = M.solve_mrhs(B, trans=True)
    YH = M.mmultiply(XH, trans=True)
    assert (np.linalg.norm(YH - B, axis=0) / np.linalg.norm(B, axis=0)).max() <= 1e-10
    assert relerr(YH[:, 2], R.mmultiply(XH[:, 2].copy(), trans=True)) <= 1e-10


def test_1m_iterative_refinement_matches_reference(big):
    A, R, levels, M, O = big
    n = A.shape[0]
    b = np.sin(0.001 * np.arange(n)) + 1.0
    x4 = M.hifir(b, 4)
    x4r, _ = R.hifir(b, 4)  # HIF::hifir of the real reference on the same factors
    assert relerr(x4, x4r) <= 1e-11
    xb, it, fl = M.hifir(b, 6, betas=(1e-8, 1e6))
    xbr, (itr, flr) = R.hifir(b, 6, [1e-8, 1e6])
    assert (it, fl) == (itr, flr) and relerr(xb, xbr) <= 1e-11
    # the outer SpMV alone, bitwise
    import torch

    yd = M.spmv(torch.from_numpy(b).cuda())
    M.sync()  # device-pointer entry points enqueue on the handle's stream and return
    y = yd.cpu().numpy()
    assert np.array_equal(y, orc.crs_mv(A.indptr, A.indices, A.data, b))


def _poisson3d(nx):
    I = sp.identity(nx, format="csr")
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    A = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
    A.sort_indices()
    return A


def test_3d_poisson_vs_oracle():
    A = _poisson3d(40)  # 64,000 rows; BASELINE config 4's stencil at a size the oracle finishes in seconds
    R = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0))
    levels = R.levels()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=16)
    O = orc.Oracle(levels)
    rng = np.random.default_rng(3)
    B = rng.uniform(-1, 1, size=(A.shape[0], 16))
    X = M.solve_mrhs(B)
    assert relerr(X, O.solve_batch(B, threads=4)) <= 1e-12
    assert relerr(X[:, 0], R.solve(B[:, 0].copy())) <= 1e-12


def test_3d_poisson_128_config4_on_one_gpu():
    # BASELINE configs[3]'s matrix family at a size that means something and still fits the suite: 3-D 7-pt Poisson 128^3
    # (2,097,152 rows, PDE-tuned parameters: 2 levels + a dense block, ~17 s of host factorization), nrhs = 64 on ONE GPU --
    # what each of the 8 RHS-sharded ranks of config 4 runs with its own columns.  256^3 (226 s of factorization) stays a
    # stamped profile (profiles/); the 8-GPU split itself is unmeasured on hardware.
    A = _poisson3d(128)
    n = A.shape[0]
    R = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0))
    levels = R.levels()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    O = orc.Oracle(levels)
    rng = np.random.default_rng(44)
    B = rng.uniform(-1, 1, size=(n, 64))
    X = M.solve_mrhs(B)
    assert np.isfinite(X).all()
    for k in (0, 63):  # the first and the last column against the oracle
        assert relerr(X[:, k], O.solve(B[:, k].copy())) <= 1e-12
    assert relerr(X[:, 31], R.solve(B[:, 31].copy())) <= 1e-12  # one against the compiled reference itself
    # separability: a column's bits do not depend on the batch it travels in (8 columns = one rank's share of 64)
    X8 = M.solve_mrhs(np.ascontiguousarray(B[:, 24:32]))
    assert np.array_equal(X8, X[:, 24:32])
    assert np.array_equal(M.solve(B[:, 63].copy()), X[:, 63])
    # round trip through the oracle's prec_prod (libhifir/tests/test_real.c:110-146): M (M^-1 b) = b
    b2 = O.mmultiply(X[:, 0].copy(), rank=-1)
    assert relerr(b2, B[:, 0]) <= 1e-10
    # ... and on the device, all 64 columns at once
    Y = M.mmultiply(X)
    assert relerr(Y, B) <= 1e-10
    M.close()


def test_complex_helmholtz_like_vs_oracle():
    # complex fp64 (BASELINE config 5 stand-in at moderate size): shifted Laplacian with absorption
    A = (poisson2d(120) - (0.3 + 0.2j) * sp.identity(120 * 120)).tocsr()
    A.sort_indices()
    R = ref.RefHIF(A.indptr, A.indices, A.data)
    levels = R.levels()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=16)
    O = orc.Oracle(levels)
    rng = np.random.default_rng(4)
    B = rng.uniform(-1, 1, size=(A.shape[0], 16)) + 1j * rng.uniform(-1, 1, size=(A.shape[0], 16))
    X = M.solve_mrhs(B)
    assert relerr(X, O.solve_batch(B, threads=4)) <= 1e-12
    assert relerr(X[:, 3], R.solve(B[:, 3].copy())) <= 1e-12
    XH = M.solve_mrhs(B, trans=True)  # conjugation matters here: A is complex symmetric, not Hermitian
    assert relerr(XH, O.solve_batch(B, threads=4, trans=True)) <= 1e-12
    assert relerr(XH[:, 3], R.solve(B[:, 3].copy(), trans=True)) <= 1e-12
    assert relerr(XH, X) > 1e-3


def test_constant_null_space_filter_matches_reference():
    # HIF::nsp / HIF::nsp_tran in constant mode (NspFilter.hpp:118-125, applied inside solve: builder.hpp:419-422)
    A = poisson2d(60)
    n = A.shape[0]
    R = ref.RefHIF(A.indptr, A.indices, A.data)
    M = hifir_amd.HIF.from_levels(R.levels(), max_nrhs=8)
    M.set_matrix(A.indptr, A.indices, A.data)
    rng = np.random.default_rng(9)
    b = rng.uniform(-1, 1, n)
    B = rng.uniform(-1, 1, size=(n, 70))
    plain = M.solve(b)
    for start, end in ((0, -1), (100, 1300)):
        R.set_nsp_const(start, end)
        M.set_nsp_const(start, end)
        x = M.solve(b)
        assert relerr(x, R.solve(b)) <= 1e-12
        e = n if end < 0 else end
        assert abs(x[start:e].mean()) <= 1e-14 * np.abs(x).max() + 1e-16
        assert relerr(M.solve(b, trans=True), R.solve(b, trans=True)) <= 1e-12  # nsp does not touch the transposed solve
        X = M.solve_mrhs(B)  # batched applies are filtered too, column by column
        assert relerr(X[:, 69], R.solve(B[:, 69].copy())) <= 1e-12
        x4, _ = R.hifir(b, 4)
        assert relerr(M.hifir(b, 4), x4) <= 1e-11  # the filter sits inside every refinement sweep
    R.set_nsp_const(5, 2)  # start > end: remove
    M.set_nsp_const(5, 2)
    assert np.array_equal(M.solve(b), plain)
    R.set_nsp_const(0, -1, trans=True)
    M.set_nsp_const(0, -1, trans=True)
    assert relerr(M.solve(b, trans=True), R.solve(b, trans=True)) <= 1e-12
    assert np.array_equal(M.solve(b), plain)


def test_2m_complex_saddle_point_config5():
    """BASELINE config 5: complex fp64 saddle point, ~2M rows (1,997,568), nrhs = 16 -- SURVEY 8(d) C5's generator
    (tests/util.py stokes_kkt: [[K + i w M, B^T], [B, -eps I]], w = 0.1, eps = 1e-8), factorized on the box by the
    compiled reference with tau = 1e-2, alpha = 3 and the default kappa = 3 (4 levels + a 1,364^2 dense QRCP last
    level, nnz(M) = 74 M; DEFAULT_PARAMS take ~10 min of host factorization at this size -- the 2,028-row fixture
    kkt_26 covers them), applied on the device: columns vs the oracle and the reference, forward and
    conjugate-transpose, plus the product round trip."""
    A = stokes_kkt(816)
    n = A.shape[0]
    assert n == 1997568
    R = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(tau=1e-2, kappa=3.0, alpha=3.0))
    levels = R.levels()
    assert len(levels) >= 2 and int(levels[-1]["dense_n"]) > 0  # dense QRCP last level (on the f64 matrix cores)
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=16)
    O = orc.Oracle(levels)
    rng = np.random.default_rng(13)
    B = rng.uniform(-1, 1, size=(n, 16)) + 1j * rng.uniform(-1, 1, size=(n, 16))
    X = M.solve_mrhs(B)
    assert relerr(X[:, 0], O.solve(B[:, 0].copy())) <= 1e-12
    assert relerr(X[:, 15], R.solve(B[:, 15].copy())) <= 1e-12
    XH = M.solve_mrhs(B, trans=True)
    assert relerr(XH[:, 7], O.solve(B[:, 7].copy(), trans=True)) <= 1e-12
    assert relerr(XH[:, 3], R.solve(B[:, 3].copy(), trans=True)) <= 1e-12
    assert relerr(XH[:, 3], X[:, 3]) > 1e-6  # complex symmetric, not Hermitian: conjugation matters
    Y = M.mmultiply(X)
    assert (np.linalg.norm(Y - B, axis=0) / np.linalg.norm(B, axis=0)).max() <= 1e-10


def test_1m_symmetric_factorization():
    """Full size, Options::is_symm: the 1M-row Poisson matrix factorized symmetrically by the compiled reference
    (its last level is SYEIG) and applied on the device: columns against the real reference, the conjugate-transpose
    apply (same operator for a symmetric hierarchy), the product round trip, a truncated rank."""
    A = poisson2d(1000)
    R = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(is_symm=1))
    levels = R.levels()
    assert int(levels[-1].get("dense_symm", 0)) == 1
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=8)
    assert M.schur_rank() == levels[-1]["dense_rank"]
    n = A.shape[0]
    rng = np.random.default_rng(11)
    B = rng.uniform(-1, 1, size=(n, 8))
    X = M.solve_mrhs(B)
    for k in (0, 7):
        assert relerr(X[:, k], R.solve(B[:, k].copy())) <= 1e-12
    xt = M.solve(B[:, 3].copy(), trans=True)
    assert relerr(xt, R.solve(B[:, 3].copy(), trans=True)) <= 1e-12
    y = M.mmultiply(X[:, 0].copy())
    assert relerr(y, B[:, 0]) <= 1e-10                    # M (M^-1 b) = b
    assert relerr(y, R.mmultiply(X[:, 0].copy(), rank=-1)) <= 1e-10
    assert relerr(M.solve(B[:, 1].copy(), rank=50), R.solve(B[:, 1].copy(), rank=50)) <= 1e-11
