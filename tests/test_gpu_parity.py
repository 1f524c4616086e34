"""GPU (-m gpu): the HIP apply path, called through the C ABI, against the oracle (oracle/liborc.so,
the C restatement pinned to the real reference) and against the committed golden vectors.

Bars: sparse stages bit-exact (real, no dense level); dense level and complex within 1e-12
(relative, infinity norm) -- BASELINE.json north_star "residual within 1e-12 of CPU reference"."""
import os

import numpy as np
import pytest

import hifir_amd
from oracle import orc
from util import HIER_NAMES, load_hier, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-12


MODES = {
    # name: (HIFIR_AMD_MIN_LOGR, HIFIR_AMD_DENSE_BLOCK)
    "R64-exact": (6, 0),    # default 64-wide arena, thin bands solved sequentially: reference summation order
    "narrow-exact": (0, 0), # narrow lane mappings (R = 1, 2, 4, ...): must give the same bits
    "R64-fast": (6, 2048),   # the DEFAULT configuration: thin bands through explicit block inverses (1e-12)
    # ... with sparse-own components (k_band_cd / k_band_cs / k_band_cs_z <sparse>) on every shallow thin triangle, whatever
    # its size: what level 0 of the 1M-row (real) and 2M-row (complex) hierarchies runs, here on the small fixtures
    "R64-fast-sparse-own": (6, 2048, {"HIFIR_AMD_CD_SPARSE_MIN_ROWS": "0"}),
    # ... with the coefficient tiles off (real data: the entry walk of rounds 2-3, k_band_cd / k_band_cs, which still ships
    # behind HIFIR_AMD_CT=0) and ON for complex data (k_band_ct_z, off by default: measured slower)
    "R64-fast-walk-real-tiles-complex": (6, 2048, {"HIFIR_AMD_CT": "1", "HIFIR_AMD_CT_Z": "1", "HIFIR_AMD_CT_REAL": "0"}),
    # ... with S7 of the child's rows on a side stream beside the second solve (off by default: measured slower), sparse-own
    # components everywhere so that the first solve's row flags are in force as well
    # (+ the K splits of the operator products summed by the last split to arrive, also off by default)
    # (+ sparse-own U bands with every row of a component in LDS -- k_band_cd<false, sparse>, the form before k_band_us)
    "R64-fast-early-list": (6, 2048, {"HIFIR_AMD_LIST_EARLY": "1", "HIFIR_AMD_CD_SPARSE_MIN_ROWS": "0", "HIFIR_AMD_TOP_LAST": "1",
                                      "HIFIR_AMD_US": "0"}),
}


@pytest.fixture(scope="module", params=list(MODES), ids=list(MODES))
def cache(request):
    import os

    logr, blk = MODES[request.param][:2]
    extra = MODES[request.param][2] if len(MODES[request.param]) > 2 else {}
    os.environ["HIFIR_AMD_MIN_LOGR"] = str(logr)
    os.environ["HIFIR_AMD_DENSE_BLOCK"] = str(blk)
    os.environ.update(extra)
    yield {"exact": blk == 0}
    os.environ.pop("HIFIR_AMD_MIN_LOGR", None)
    os.environ.pop("HIFIR_AMD_DENSE_BLOCK", None)
    for k in extra:
        os.environ.pop(k, None)


def _get(cache, name):
    if name not in cache:
        levels, d = load_hier(name)
        M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
        M.set_matrix(d["A_indptr"], d["A_indices"], d["A_vals"])
        cache[name] = (levels, d, M, orc.Oracle(levels))
    return cache[name]


def _exact(cache, levels, d):
    return cache["exact"] and int(levels[-1].get("dense_n", 0)) == 0 and not np.iscomplexobj(d["b"])


@pytest.mark.parametrize("name", HIER_NAMES)
def test_solve_single_rhs(cache, name):
    levels, d, M, O = _get(cache, name)
    assert M.levels() == len(levels) + (1 if levels[-1].get("dense_n", 0) else 0)
    x = M.solve(d["b"])
    xo = O.solve(d["b"])
    if _exact(cache, levels, d):
        assert np.array_equal(x, xo), f"sparse-only hierarchy must be bit-identical, relerr={relerr(x, xo):.3e}"
    assert relerr(x, xo) <= TOL
    assert relerr(x, d["x"]) <= TOL  # the real reference's HIF::solve output


@pytest.mark.parametrize("name", HIER_NAMES)
@pytest.mark.parametrize("nrhs", [1, 2, 3, 8, 17, 33, 48, 49, 64, 100])
def test_solve_batch(cache, name, nrhs):
    levels, d, M, O = _get(cache, name)
    n = len(d["b"])
    rng = np.random.default_rng(nrhs)
    B = rng.uniform(-1, 1, size=(n, nrhs)).astype(d["b"].dtype)
    if np.iscomplexobj(B):
        B = B + 1j * rng.uniform(-1, 1, size=(n, nrhs))
    B[:, 0] = d["b"]
    X = M.solve_mrhs(B)
    Xo = O.solve_batch(B, threads=4)
    if _exact(cache, levels, d):
        assert np.array_equal(X, Xo), f"relerr={relerr(X, Xo):.3e}"
    assert relerr(X, Xo) <= TOL
    assert relerr(X[:, 0], d["x"]) <= TOL


@pytest.mark.parametrize("name", ["p2d_100_tuned", "p2d_64_deep", "young1c"])
def test_device_pointer_api_and_strides(cache, name):
    torch = pytest.importorskip("torch")
    levels, d, M, O = _get(cache, name)
    n = len(d["b"])
    nrhs, ld = 5, 9  # padded row stride: B and X are views into wider blocks
    rng = np.random.default_rng(3)
    Bh = rng.uniform(-1, 1, size=(n, nrhs)).astype(d["b"].dtype)
    Bw = torch.zeros((n, ld), dtype=torch.from_numpy(Bh).dtype, device="cuda")
    Bw[:, :nrhs] = torch.from_numpy(Bh).cuda()
    Xw = torch.full((n, ld), 7.0, dtype=Bw.dtype, device="cuda")
    M.solve_mrhs(Bw[:, :nrhs], Xw[:, :nrhs])
    M.sync()
    Xh = Xw.cpu().numpy()
    assert np.all(Xh[:, nrhs:] == 7.0), "columns beyond nrhs must not be touched"
    assert relerr(Xh[:, :nrhs], O.solve_batch(Bh)) <= TOL
    # replays of the cached graph give identical bits
    X2 = torch.empty_like(Xw)
    M.solve_mrhs(Bw[:, :nrhs], X2[:, :nrhs])
    M.solve_mrhs(Bw[:, :nrhs], X2[:, :nrhs])
    M.sync()
    assert torch.equal(X2[:, :nrhs], Xw[:, :nrhs])


@pytest.mark.parametrize("name", HIER_NAMES)
def test_iterative_refinement(cache, name):
    levels, d, M, O = _get(cache, name)
    x4 = M.hifir(d["b"], 4)
    assert relerr(x4, d["x_ir4"]) <= 1e-11
    xb, it, fl = M.hifir(d["b"], 16, betas=(1e-10, 1e3))
    assert (it, fl) == tuple(int(v) for v in d["irb_status"])
    assert relerr(xb, d["x_irb"]) <= 1e-11
    # batched IR: every column behaves like its own call
    B = d["B4"]
    X, its, fls = M.hifir(B, 16, betas=(1e-10, 1e3))
    for k in range(B.shape[1]):
        xo, (ito, flo) = O.hifir(d["A_indptr"], d["A_indices"], d["A_vals"], B[:, k].copy(), 16, [1e-10, 1e3])
        assert (int(its[k]), int(fls[k])) == (ito, flo)
        assert relerr(X[:, k], xo) <= 1e-11


@pytest.mark.parametrize("name", HIER_NAMES)
def test_transposed_solve(cache, name):
    # LHF_SH: x = M^{-H} b through the adjoint hierarchy; bar 1e-12 vs the oracle's prec_solve_tran
    # restatement (tolerance-level by construction: the reference accumulates dot products) and vs
    # the real reference's golden xt / XT4
    torch = pytest.importorskip("torch")
    levels, d, M, O = _get(cache, name)
    xt = M.solve(d["b"], trans=True)
    assert relerr(xt, O.solve(d["b"], trans=True)) <= TOL
    assert relerr(xt, d["xt"]) <= TOL
    XT = M.solve_mrhs(d["B4"], trans=True)
    assert relerr(XT, d["XT4"]) <= TOL
    n = len(d["b"])
    rng = np.random.default_rng(17)
    for nrhs in (3, 64, 70):
        B = rng.uniform(-1, 1, size=(n, nrhs)).astype(d["b"].dtype)
        if np.iscomplexobj(B):
            B = B + 1j * rng.uniform(-1, 1, size=(n, nrhs))
        Xo = O.solve_batch(B, threads=4, trans=True)
        assert relerr(M.solve_mrhs(B, trans=True), Xo) <= TOL
        Xd = M.solve_mrhs(torch.from_numpy(B).cuda(), trans=True)  # device pointers, enqueued
        M.sync()
        torch.cuda.synchronize()
        assert relerr(Xd.cpu().numpy(), Xo) <= TOL
    # the forward operator is untouched by the adjoint engine
    assert relerr(M.solve(d["b"]), d["x"]) <= TOL
    # refinement with A^H and M^{-H} (IterRefine.hpp:77-105 with tran=true): same recurrence on the host
    # with the oracle's transposed solve and scipy's A^H
    import scipy.sparse as sp

    A = sp.csr_matrix((d["A_vals"], d["A_indices"], d["A_indptr"]), shape=(n, n))
    AH = A.conj().T.tocsr()
    assert relerr(M.hifir(d["b"], 1, trans=True), d["xt"]) <= TOL
    x = np.zeros_like(d["b"])
    for i in range(4):
        r = d["b"] - AH @ x if i else d["b"].copy()
        x = O.solve(r, trans=True) + x
    assert relerr(M.hifir(d["b"], 4, trans=True), x) <= 1e-10


@pytest.mark.parametrize("name", HIER_NAMES)
def test_multilevel_product(cache, name):
    # LHF_M / LHF_MH: y = M x and M^H x (prec_prod / prec_prod_tran) vs the oracle restatements, the real
    # reference's golden b2, and the round trips M (M^{-1} b) = b of libhifir/tests/test_real.c:110-146
    torch = pytest.importorskip("torch")
    levels, d, M, O = _get(cache, name)
    nrm = np.linalg.norm(d["b"])
    b2 = M.mmultiply(d["x"])
    assert relerr(b2, O.mmultiply(d["x"])) <= 1e-10
    assert relerr(b2, d["b2"]) <= 1e-10
    assert np.linalg.norm(b2 - d["b"]) / nrm <= 1e-10
    bt = M.mmultiply(d["xt"], trans=True)
    assert relerr(bt, O.mmultiply(d["xt"], trans=True)) <= 1e-10
    assert np.linalg.norm(bt - d["b"]) / nrm <= 1e-10
    # batched, wider than one tile, entirely on the device: M (M^{-1} B) = B and M^H (M^{-H} B) = B
    n = len(d["b"])
    rng = np.random.default_rng(29)
    B = rng.uniform(-1, 1, size=(n, 70)).astype(d["b"].dtype)
    if np.iscomplexobj(B):
        B = B + 1j * rng.uniform(-1, 1, size=(n, 70))
    Bd = torch.from_numpy(B).cuda()
    for tr in (False, True):
        Xd = M.solve_mrhs(Bd, trans=tr, rank=-1)
        Yd = M.mmultiply(Xd, trans=tr)
        M.sync()
        torch.cuda.synchronize()
        Y = Yd.cpu().numpy()
        assert (np.linalg.norm(Y - B, axis=0) / np.linalg.norm(B, axis=0)).max() <= 1e-9
        k = 33
        assert relerr(Y[:, k], O.mmultiply(Xd[:, k].cpu().numpy().copy(), trans=tr, rank=-1)) <= 1e-10
    # an explicit rank goes to the dense block's product like in the reference (QRCP.hpp:466-467)
    if levels[-1].get("dense_n", 0) > 8:
        assert relerr(M.mmultiply(d["x"], rank=5), O.mmultiply(d["x"], rank=5)) <= 1e-10


def test_rotating_buffers_share_one_graph(cache):
    # a Krylov solver hands over a different (B, X) pair every call: the captured graph is keyed by shape
    # and reads the pointers from a device slot, so nothing is re-captured and every pair gets its own result
    torch = pytest.importorskip("torch")
    levels, d, M, O = _get(cache, "p2d_64_deep")
    n = len(d["b"])
    rng = np.random.default_rng(41)
    Bs = [torch.from_numpy(rng.uniform(-1, 1, size=(n, 16))).cuda() for _ in range(12)]
    Xs = [torch.empty_like(b) for b in Bs]
    M.solve_mrhs(Bs[0], Xs[0])
    M.sync()

    for b, x in zip(Bs, Xs):
        M.solve_mrhs(b, x)  # enqueued back to back, no synchronisation in between
    M.sync()
    torch.cuda.synchronize()
    assert M.stats()["launches"] > 0
    # (the wall-clock side of this -- rotating pairs cost what a fixed pair costs -- is tests/test_gpu_perf.py, -m perf)
    for k in (0, 5, 11):
        Xo = O.solve_batch(Bs[k].cpu().numpy(), threads=4)
        assert relerr(Xs[k].cpu().numpy(), Xo) <= TOL


@pytest.mark.parametrize("name", HIER_NAMES)
def test_column_bits_do_not_depend_on_the_batch_width(cache, name):
    # Batches of fewer than 49 columns run the component bands in 16-column slices (k_band_cs / k_band_cs_z), the Schur
    # products with 64 / R rows per wave (k_spmm_epi_narrow) and the operator products on the column tiles they have;
    # 49 ... 64 columns run the 64-column kernels: every width must produce the SAME bits for a column (the arithmetic
    # per row and column is the same sequence of operations in every kernel), forwards and conjugate-transposed
    levels, d, M, O = _get(cache, name)
    n = int(levels[0]["n"])
    rng = np.random.default_rng(99)
    B = rng.uniform(-1, 1, size=(n, 64)).astype(d["b"].dtype)
    if np.iscomplexobj(d["b"]):
        B = B + 1j * rng.uniform(-1, 1, size=(n, 64))
    for tr in (False, True):
        X = M.solve_mrhs(B, trans=tr)
        for k in (1, 5, 16, 17, 32, 33, 48, 49):
            Xk = M.solve_mrhs(np.ascontiguousarray(B[:, :k]), trans=tr)
            assert np.array_equal(Xk, X[:, :k]), (tr, k, relerr(Xk, X[:, :k]))
        # ... and columns in the middle of the batch, as a shard of an RHS-sharded job sees them
        Xm = M.solve_mrhs(np.ascontiguousarray(B[:, 40:48]), trans=tr)
        assert np.array_equal(Xm, X[:, 40:48])


def test_rows_the_first_solve_leaves_out_do_not_change_a_bit():
    # Round 4: the FIRST solve of a sparse-own level (prec_solve.hpp:364; its result only feeds b_2 - E y_1, :366-368) does
    # not store L rows without entries that only their own component reads, nor U rows no column of E refers to
    # (engine.hip build_row_flags).  Same arithmetic on every row that is used: same bits as with every row stored, at
    # the 64-column kernels (k_band_cd) and the 16-column slices (k_band_cs), forwards and transposed.
    import os

    keep = {k: os.environ.get(k) for k in ("HIFIR_AMD_CD_SPARSE_MIN_ROWS", "HIFIR_AMD_SKIP_ROWS", "HIFIR_AMD_DENSE_BLOCK",
                                           "HIFIR_AMD_MIN_LOGR")}
    skipped = 0
    try:
        os.environ["HIFIR_AMD_CD_SPARSE_MIN_ROWS"] = "0"
        os.environ["HIFIR_AMD_DENSE_BLOCK"] = "2048"
        os.environ["HIFIR_AMD_MIN_LOGR"] = "6"
        for name in HIER_NAMES:
            levels, d = load_hier(name)
            if np.iscomplexobj(d["b"]):
                continue
            n = int(levels[0]["n"])
            rng = np.random.default_rng(7)
            B = rng.uniform(-1, 1, size=(n, 64))
            out = []
            for flag in ("3", "0"):
                os.environ["HIFIR_AMD_SKIP_ROWS"] = flag
                M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
                se = M.stats_ext()
                if flag == "3":
                    skipped += int(se["rows_not_stored_L"]) + int(se["rows_not_stored_U"])
                else:
                    assert se["rows_not_stored_L"] == 0 and se["rows_not_stored_U"] == 0
                out.append([M.solve_mrhs(B, trans=tr) for tr in (False, True)] +
                           [M.solve_mrhs(np.ascontiguousarray(B[:, :9]), trans=tr) for tr in (False, True)])
                # a second apply on the same handle: the rows left out hold what an earlier batch left there
                B2 = np.full_like(B, np.nan)
                M.solve_mrhs(B2)
                again = M.solve_mrhs(B)
                assert np.array_equal(again, out[-1][0]), name
            for a, b in zip(*out):
                assert np.array_equal(a, b), name
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert skipped > 0  # (the fixtures do have such rows: the test is not vacuous)


@pytest.mark.parametrize("name", ["cd2d_48", "young1c"])
def test_saved_hierarchy_applies_identically(cache, name, tmp_path):
    # hifamd_save -> hifamd_load -> finalize on a "GPU node without the reference": same bits as the original
    levels, d, M, O = _get(cache, name)
    path = str(tmp_path / "h.hifamd")
    M.save(path)
    M2 = hifir_amd.HIF.load(path, max_nrhs=64)
    # (the per-array device checksums told an intermittent mismatch apart: a second OpenMP runtime in the
    #  process corrupted host memory during the dense factorization -- the library is OpenMP-free since)
    ck = lambda H: (lambda o: o[:hifir_amd.lib().hifamd_debug_checksums(H._h, o.ctypes.data, 256)])(np.zeros(256, np.uint64))
    M3 = hifir_amd.HIF.load(path, max_nrhs=64)
    assert np.array_equal(ck(M2), ck(M3))
    # ... including the operators that finalize FORMS (combined tops on the host, the tail operator by the device's own
    # applies on the identity): built twice from one file, and once from the caller's arrays, they are the same bytes
    assert np.array_equal(ck(M), ck(M2))
    assert M.stats_ext()["tail_rows"] == M2.stats_ext()["tail_rows"]
    for tr in (False, True):
        X2, X1 = M2.solve_mrhs(d["B4"], trans=tr), M.solve_mrhs(d["B4"], trans=tr)
        assert np.array_equal(X2, X1), (tr, relerr(X2, d["XT4"] if tr else d["X4"]))
    assert np.array_equal(M2.mmultiply(d["x"]), M.mmultiply(d["x"]))


@pytest.mark.parametrize("name", ["cd2d_48", "young1c", "p2d_64_deep"])
def test_saved_analysis_is_adopted_and_changes_nothing(cache, name, tmp_path):
    # hifamd_save_ex(HIFAMD_SAVE_ANALYSIS): a load adopts the schedules / band plans / slot-ordered triangles of the
    # trailer instead of analyzing again -- every device-resident array and every result keeps its bits
    levels, d, M, O = _get(cache, name)
    path = str(tmp_path / "h_ana.hifamd")
    M.save(path, analysis=True)
    M2 = hifir_amd.HIF.load(path, max_nrhs=64)
    assert M2.stats_ext()["analysis_cached_levels"] == len(levels)
    ck = lambda H: (lambda o: o[:hifir_amd.lib().hifamd_debug_checksums(H._h, o.ctypes.data, 256)])(np.zeros(256, np.uint64))
    assert np.array_equal(ck(M), ck(M2))
    for tr in (False, True):
        assert np.array_equal(M2.solve_mrhs(d["B4"], trans=tr), M.solve_mrhs(d["B4"], trans=tr))
    assert np.array_equal(M2.mmultiply(d["x"]), M.mmultiply(d["x"]))
    # a handle loaded from a file with a trailer writes a trailer again (the adopted analysis is complete)
    path3 = str(tmp_path / "h_ana3.hifamd")
    M2.save(path3, analysis=True)
    assert open(path3, "rb").read() == open(path, "rb").read()


@pytest.mark.parametrize("name", ["cd2d_48", "young1c", "p2d_64_deep", "p2d_100_tuned"])
def test_block_inverses_formed_on_the_device_are_the_hosts(cache, name, monkeypatch):
    # hifamd_finalize forms the explicit inverses of the diagonal blocks on the device (k_block_inverse); the host version
    # (host.hpp build_dense_block, HIFIR_AMD_DEVICE_INVERSES=0) is the same arithmetic in the same order: every resident
    # array -- the operators included -- and every result has the same bits
    levels, d, M, O = _get(cache, name)
    monkeypatch.setenv("HIFIR_AMD_DEVICE_INVERSES", "0")
    Mh = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    monkeypatch.delenv("HIFIR_AMD_DEVICE_INVERSES")
    ck = lambda H: (lambda o: o[:hifir_amd.lib().hifamd_debug_checksums(H._h, o.ctypes.data, 256)])(np.zeros(256, np.uint64))
    assert np.array_equal(ck(M), ck(Mh))
    for tr in (False, True):
        assert np.array_equal(Mh.solve_mrhs(d["B4"], trans=tr), M.solve_mrhs(d["B4"], trans=tr))
    Mh.close()


def test_two_handles_from_two_threads(cache):
    # "distinct handles may be used from distinct threads" (hifir_amd.h conventions, like the reference):
    # two hierarchies applied concurrently from two host threads (ctypes releases the GIL) keep their results
    import threading

    la, da, Ma, Oa = _get(cache, "cd2d_48")
    lb, db, Mb, Ob = _get(cache, "p2d_64_deep")
    out = {}

    def work(tag, M, d):
        res = []
        for k in range(12):
            res.append(M.solve_mrhs(d["B4"] * (1.0 + k)))
            res.append(M.solve(d["b"], trans=True))
        out[tag] = res

    ta = threading.Thread(target=work, args=("a", Ma, da))
    tb = threading.Thread(target=work, args=("b", Mb, db))
    ta.start()
    tb.start()
    ta.join()
    tb.join()
    for tag, d in (("a", da), ("b", db)):
        for k in range(12):
            assert relerr(out[tag][2 * k], d["X4"] * (1.0 + k)) <= TOL
            assert relerr(out[tag][2 * k + 1], d["xt"]) <= TOL


def test_spmv_bitwise(cache):
    torch = pytest.importorskip("torch")
    levels, d, M, O = _get(cache, "cd2d_48")
    n = len(d["b"])
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, size=(n, 6))
    Yd = M.spmv(torch.from_numpy(X).cuda())
    M.sync()  # device-pointer entry points enqueue on the handle's stream and return
    Y = Yd.cpu().numpy()
    for k in range(6):
        assert np.array_equal(Y[:, k], orc.crs_mv(d["A_indptr"], d["A_indices"], d["A_vals"], X[:, k].copy()))


def test_rank_argument(cache):
    levels, d, M, O = _get(cache, "p2d_30")
    for rank in (0, -1, 100, 215, 10 ** 6):
        assert relerr(M.solve(d["b"], rank=rank), O.solve(d["b"], rank=rank)) <= TOL


@pytest.mark.parametrize("name", ["p2d_32_symm", "herm_24_symm"])
def test_symmetric_last_level(cache, name):
    """is_symm hierarchies: the last level is the reference's SYEIG (eigendecomposition with truncation order), for
    the solve, the conjugate-transpose solve (same operator) and the product, incl. the run-time rank (SYEIG.hpp:187,262)."""
    levels, d, M, O = _get(cache, name)
    assert int(levels[-1]["dense_symm"]) == 1 and M.schur_rank() == levels[-1]["dense_rank"] == O.dense_rank
    b = d["b"]
    assert relerr(M.solve(b), d["x"]) <= TOL          # the real reference's own x
    assert relerr(M.solve(b, trans=True), d["xt"]) <= TOL
    for rank in (0, -1, 5, 40, 10 ** 6):
        assert relerr(M.solve(b, rank=rank), O.solve(b, rank=rank)) <= TOL
        assert relerr(M.solve(b, rank=rank, trans=True), O.solve(b, rank=rank, trans=True)) <= TOL
        assert relerr(M.mmultiply(d["x"], rank=rank), O.mmultiply(d["x"], rank=rank)) <= 1e-10
        assert relerr(M.mmultiply(d["x"], rank=rank, trans=True), O.mmultiply(d["x"], rank=rank, trans=True)) <= 1e-10
    # a definite block with the truncation rule of a definite factorization (Options::spd = -1 on the negated block)
    lv2 = [dict(l) for l in levels]
    lv2[-1]["dense"] = -np.asarray(lv2[-1]["dense"])
    lv2[-1]["spd"] = -1
    M2, O2 = hifir_amd.HIF.from_levels(lv2, max_nrhs=8), orc.Oracle(lv2)
    assert M2.schur_rank() == O2.dense_rank
    assert relerr(M2.solve(b), O2.solve(b)) <= TOL


def test_lup_last_level(cache):
    """Hierarchies of a reference built with HIF_DENSE_MODE=0: the last level is LU with partial pivoting (LUP.hpp).
    Real fixture from that build; complex: the young1c hierarchy with its block handed over as an LUP block (the
    conjugate-transpose apply then uses A^-T for the solve and A^H for the product, LUP.hpp:150,187)."""
    levels, d, M, O = _get(cache, "p2d_30_lup")
    assert int(levels[-1]["dense_lup"]) == 1
    for rank in (0, 7):  # ignored
        assert relerr(M.solve(d["b"], rank=rank), d["x"]) <= TOL
    assert relerr(M.solve(d["b"], trans=True), d["xt"]) <= TOL
    lz, dz = load_hier("young1c")
    lz = [dict(l) for l in lz]
    lz[-1]["dense_lup"] = 1
    Mz, Oz = hifir_amd.HIF.from_levels(lz, max_nrhs=8), orc.Oracle(lz)
    b = dz["b"]
    x = Mz.solve(b)
    assert relerr(x, Oz.solve(b)) <= TOL
    assert relerr(Mz.solve(b, trans=True), Oz.solve(b, trans=True)) <= TOL
    assert relerr(Mz.mmultiply(x), Oz.mmultiply(x)) <= 1e-10
    assert relerr(Mz.mmultiply(x, trans=True), Oz.mmultiply(x, trans=True)) <= 1e-10
    assert relerr(Mz.mmultiply(x), b) <= 1e-10
    # an exactly singular block is refused (the reference only warns and would divide by zero)
    ls = [dict(l) for l in levels]
    sing = np.array(ls[-1]["dense"], dtype=np.float64).reshape(ls[-1]["dense_n"], -1).copy()
    sing[:, 0] = 0.0
    ls[-1]["dense"] = sing.ravel()
    with pytest.raises(hifir_amd.HifAmdError):
        hifir_amd.HIF.from_levels(ls, max_nrhs=8)


def test_error_paths(cache):
    levels, d, M, O = _get(cache, "p2d_5")
    with pytest.raises(hifir_amd.HifAmdError) as e:
        M.solve(np.zeros(7))
    assert e.value.code == 2
    import ctypes as C
    x = np.empty_like(d["b"])
    bb = np.ascontiguousarray(d["b"])
    rc = hifir_amd.lib().hifamd_apply_batch(M._h, 7, bb.ctypes.data_as(C.c_void_p), 1, x.ctypes.data_as(C.c_void_p),
                                            1, 1, 1, None, 0, None)
    assert rc == 2 and b"operator" in hifir_amd.lib().hifamd_last_error()
    b = np.ascontiguousarray(d["b"])
    rc = hifir_amd.lib().hifamd_solve(M._h, b.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), 0)
    assert rc == 3  # aliasing b and x is refused (libhifir Ownership: b and x must not alias)
    assert b"alias" in hifir_amd.lib().hifamd_last_error()


@pytest.mark.parametrize("name", ["p2d_100_tuned", "cd2d_48", "young1c"])
def test_solve_without_the_optional_permutations(cache, name):
    # q and p_inv are optional arguments of hifamd_add_level (only the transposed solve and the products need them);
    # with q the plain solve writes its output from inside the last triangular kernel, without it through the separate
    # scatter: the same columns either way, and the operators that need them are refused
    levels, d, M, O = _get(cache, name)
    ls = [dict(lv, q=None, p_inv=None) for lv in levels]
    M2 = hifir_amd.HIF.from_levels(ls, max_nrhs=8)
    rng = np.random.default_rng(8)
    B = rng.uniform(-1, 1, size=(len(d["b"]), 5)).astype(M.dtype)
    if np.iscomplexobj(B):
        B = B + 1j * rng.uniform(-1, 1, size=B.shape)
    assert relerr(M2.solve_mrhs(B), M.solve_mrhs(B)) <= 1e-13
    with pytest.raises(hifir_amd.HifAmdError):
        M2.solve_mrhs(B, trans=True)
    M2.close()
