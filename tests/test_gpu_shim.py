"""GPU: a C user of the reference's libhifir, relinked against shim/libhifir.so.

Restates the reference's own end-to-end tests through the SAME symbols --
  libhifir/tests/test_real.c:88-146     A.mm:      lhfdCreate, lhfdSolve, lhfdApply(LHF_M), ||M(M^-1 b) - b|| / ||b|| <= 1e-10
  libhifir/tests/test_complex.c:90-140  young1c:   the same with the z family, b = A * 1
-- with the matrices taken from the committed fixtures (the .mm files do not travel), and then checks what those
tests do not: the results against the golden vectors of the compiled reference (solve, transpose solve, product,
iterative refinement incl. the per-call status pair), the rank rule of lhf?Apply (libhifir.cpp:453-461), the one-shot
error message, Update / Refactorize, and the additive entry points (ApplyBatch, SetDevices, Save/LoadHierarchy).
Factorization runs on the host inside the shim (the reference's header-only path), the applies on the GPU."""
import ctypes as C

import numpy as np
import pytest

import shim_util as su
from util import load_hier, relerr

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not su.available(), reason="shim/libhifir.so not built")]
TOL = 1e-12


def _nrm_err(b2, b):
    return float(np.linalg.norm(b2 - b) / np.linalg.norm(b))


@pytest.fixture(scope="module")
def real_case():
    levels, d = load_hier("demo_A")  # examples/demo_inputs/A.mm + b.mm, factorized with the default parameters
    A = su.Matrix("d", d["A_indptr"], d["A_indices"], d["A_vals"])
    M = su.Hif("d", A, None, su.default_params())
    assert M.h, su.errmsg()
    yield levels, d, A, M
    M.close()
    A.close()


@pytest.fixture(scope="module")
def complex_case():
    levels, d = load_hier("young1c")
    A = su.Matrix("z", d["A_indptr"], d["A_indices"], d["A_vals"])
    M = su.Hif("z", A, None, su.default_params())
    assert M.h, su.errmsg()
    yield levels, d, A, M
    M.close()
    A.close()


def test_real_c_restated(real_case):
    levels, d, A, M = real_case
    st, x = M.solve(d["b"])                      # test_real.c:110
    assert st == su.LHF_SUCCESS
    st, b2 = M.apply(su.LHF_M, x)                # :121  lhfdApply(M, LHF_M, x, 1, NULL, LHF_DEFAULT_RANK, b2, NULL)
    assert st == su.LHF_SUCCESS
    assert _nrm_err(b2, d["b"]) <= 1e-10         # :146
    # ... and against the compiled reference's own numbers
    assert relerr(x, d["x"]) <= TOL and relerr(b2, d["b2"]) <= 1e-10


def test_complex_c_restated(complex_case):
    levels, d, A, M = complex_case
    n = A.n
    b = np.array([d["A_vals"][d["A_indptr"][i]:d["A_indptr"][i + 1]].sum() for i in range(n)])  # b = A * 1, test_complex.c:100
    st, x = M.solve(b)
    assert st == su.LHF_SUCCESS
    st, b2 = M.apply(su.LHF_M, x)
    assert st == su.LHF_SUCCESS and _nrm_err(b2, b) <= 1e-10
    st, xg = M.solve(d["b"])
    assert relerr(xg, d["x"]) <= TOL


@pytest.mark.parametrize("case", ["real_case", "complex_case"])
def test_queries_match_the_hierarchy(case, request):
    levels, d, A, M = request.getfixturevalue(case)
    f = M._f
    nd = int(levels[-1].get("dense_n", 0))
    assert f("GetLevels")(M.h) == len(levels) + (1 if nd else 0)  # the dense block counts (builder.hpp:141-147)
    assert f("GetSchurSize")(M.h) == nd and f("GetSchurRank")(M.h) == int(levels[-1].get("dense_rank", 0))
    s = M.stats()
    assert s[0] == f("GetNnz")(M.h) > 0 and s[5] == f("GetLevels")(M.h) and s[8] == nd
    assert s[6] == A.n - (s[8] - s[7])  # HIF::rank


@pytest.mark.parametrize("case", ["real_case", "complex_case"])
def test_apply_operators_and_refinement(case, request):
    levels, d, A, M = request.getfixturevalue(case)
    b = d["b"]
    st, xt = M.apply(su.LHF_SH, b)               # x = M^{-H} b
    assert st == 0 and relerr(xt, d["xt"]) <= TOL
    st, x = M.apply(su.LHF_S, b)
    assert st == 0 and relerr(x, d["x"]) <= TOL
    st, y = M.apply(su.LHF_MH, d["xt"])
    assert st == 0 and _nrm_err(y, b) <= 1e-10
    # nirs <= 1 leaves ir_status alone (libhifir.cpp:457-461)
    st, x1, irs = M.apply(su.LHF_S, b, nirs=1, betas=(1e-10, 1e3), want_status=True)
    assert st == 0 and irs == (-7, -7) and np.array_equal(x1, x)
    # fixed number of sweeps: HIF::hifir(A, b, 4, x) -- no status either (libhifir.cpp:176-187)
    st, x4, irs = M.apply(su.LHF_S, b, nirs=4, want_status=True)
    assert st == 0 and irs == (-7, -7) and relerr(x4, d["x_ir4"]) <= 1e-10
    # bounded variant: (iterations, flag) like the reference's golden run (make_golden.py: betas 1e-10 / 1e3, N = 16)
    st, xb, irs = M.apply(su.LHF_S, b, nirs=16, betas=(1e-10, 1e3), want_status=True)
    assert st == 0 and list(irs) == [int(v) for v in d["irb_status"]] and relerr(xb, d["x_irb"]) <= 1e-10
    # products ignore nirs and never refine (libhifir.cpp:457-460)
    st, y2 = M.apply(su.LHF_M, d["x"], nirs=5)
    assert st == 0 and relerr(y2, d["b2"]) <= 1e-10


def test_update_refactorize_and_errors(real_case):
    levels, d, A, M = real_case
    L = su.lib()
    # a matrix of another size cannot be attached (libhifir.cpp:423-424)
    l5, d5 = load_hier("p2d_5")
    A5 = su.Matrix("d", d5["A_indptr"], d5["A_indices"], d5["A_vals"])
    assert L.lhfdUpdate(M.h, A5.h) == su.LHF_MISMATCHED_SIZES
    assert L.lhfdUpdate(M.h, None) == su.LHF_SUCCESS  # detach: direct solves still work, refinement reports NULL
    st, x = M.solve(d["b"])
    assert st == 0 and relerr(x, d["x"]) <= TOL
    st, _ = M.apply(su.LHF_S, d["b"], nirs=3)
    assert st == su.LHF_NULL_OBJ
    assert L.lhfdUpdate(M.h, A.h) == su.LHF_SUCCESS
    st, x4 = M.apply(su.LHF_S, d["b"], nirs=4)
    assert st == 0 and relerr(x4, d["x_ir4"]) <= 1e-10
    # a second handle: Setup on an empty one, then Refactorize with the PDE-tuned parameters
    M2 = su.Hif("d", A5, None, su.default_params())
    assert M2.h, su.errmsg()
    st, x5 = M2.solve(d5["b"])
    assert st == 0 and relerr(x5, d5["x"]) <= TOL
    p = su.default_params()
    L.lhfSetDroptol(1e-2, p), L.lhfSetAlpha(3.0, p), L.lhfSetKappa(5.0, p)
    assert L.lhfdRefactorize(M2.h, A5.h, p) == su.LHF_SUCCESS
    st, x5b = M2.solve(d5["b"])
    st, b5 = M2.apply(su.LHF_M, x5b)
    assert st == 0 and _nrm_err(b5, d5["b"]) <= 1e-10
    # an error inside the library surfaces as LHF_HIFIR_ERROR + a message that is handed out once
    bad = su.Matrix("d", [0, 1, 2], [0, 5], [1.0, 1.0])  # column index out of range
    h = L.lhfdCreate(bad.h, None, su.default_params())
    if h is None:
        assert su.errmsg() and su.errmsg() is None
    else:
        L.lhfdDestroy(h)
    M2.close(), A5.close(), bad.close()


@pytest.mark.parametrize("case", ["real_case", "complex_case"])
def test_apply_batch_devices_and_hierarchy_files(case, request, tmp_path):
    levels, d, A, M = request.getfixturevalue(case)
    L = su.lib()
    t = M.t
    st, X, _ = M.apply_batch(su.LHF_S, d["B4"])
    assert st == 0 and relerr(X, d["X4"]) <= TOL
    st, XT, _ = M.apply_batch(su.LHF_SH, d["B4"])
    assert st == 0 and relerr(XT, d["XT4"]) <= TOL
    # column c of a batch == the single-RHS entry point, bit for bit
    st, x0 = M.solve(np.ascontiguousarray(d["B4"][:, 0]))
    assert np.array_equal(X[:, 0], x0)
    # bounded refinement for a whole block: one status pair per column
    Bb = np.stack([d["b"], 2.0 * d["b"], -d["b"]], axis=1)
    st, Xb, irs = M.apply_batch(su.LHF_S, Bb, nirs=16, betas=(1e-10, 1e3))
    assert st == 0 and [list(r) for r in irs] == [[int(v) for v in d["irb_status"]]] * 3
    assert relerr(Xb[:, 0], d["x_irb"]) <= 1e-10
    # hierarchy file: save, load into a fresh handle (no host factorization behind it), same bits
    path = str(tmp_path / "h.hifamd").encode()
    assert getattr(L, f"lhf{t}SaveHierarchy")(M.h, path) == su.LHF_SUCCESS
    h2 = getattr(L, f"lhf{t}LoadHierarchy")(path)
    assert h2, su.errmsg()
    M2 = su.Hif(t, handle=h2)
    st, X2, _ = M2.apply_batch(su.LHF_S, d["B4"])
    assert st == 0 and np.array_equal(X2, X)
    assert M2.stats()[5] == M.stats()[5] and M2.stats()[8] == M.stats()[8] and M2.stats()[0] == M.stats()[0]
    other = "z" if t == "d" else "d"
    assert getattr(L, f"lhf{other}LoadHierarchy")(path) is None and "value type" in su.errmsg()
    M2.close()
    # lhfSetDevices: two resident replicas (both on device 0 on a 1-GPU box); ApplyBatch shards the columns over them
    ids = (C.c_int * 2)(0, 0)
    assert L.lhfSetDevices(ids, 2) == su.LHF_SUCCESS
    try:
        M3 = su.Hif(t, A, None, su.default_params())
        assert M3.h, su.errmsg()
        st, X3, _ = M3.apply_batch(su.LHF_S, d["B4"])
        assert st == 0 and np.array_equal(X3, X)
        B7 = np.concatenate([d["B4"], d["B4"][:, :3] * 0.5], axis=1)  # 7 columns: blocks of 4 + 3
        st, X7, _ = M3.apply_batch(su.LHF_SH, B7)
        assert st == 0 and relerr(X7[:, :4], d["XT4"]) <= TOL and relerr(X7[:, 4:], 0.5 * d["XT4"][:, :3]) <= TOL
        st, x = M3.solve(d["b"])
        assert st == 0 and relerr(x, d["x"]) <= TOL
        M3.close()
    finally:
        assert L.lhfSetDevices(None, 0) == su.LHF_SUCCESS
    bad = (C.c_int * 1)(99)
    assert L.lhfSetDevices(bad, 1) == su.LHF_MISMATCHED_SIZES


@pytest.mark.parametrize("case", ["real_case", "complex_case"])
def test_apply_batch_dev_blocks_stay_in_hbm(case, request):
    # lhf?ApplyBatchDev / GatherBatchDev / SyncDevices (include/libhifir_amd_ext.h): a batch that is already sharded over
    # the devices of lhfSetDevices -- here two replicas on device 0 -- is applied where it lies, block d by replica d,
    # enqueue-and-return; the gather writes the blocks side by side into one device array.  Same bits as lhf?ApplyBatch.
    torch = pytest.importorskip("torch")
    levels, d, A, M = request.getfixturevalue(case)
    L = su.lib()
    t = M.t
    st, X, _ = M.apply_batch(su.LHF_S, d["B4"])
    assert st == 0
    st, XM, _ = M.apply_batch(su.LHF_M, d["B4"])
    assert st == 0
    ids = (C.c_int * 2)(0, 0)
    assert L.lhfSetDevices(ids, 2) == su.LHF_SUCCESS
    try:
        M3 = su.Hif(t, A, None, su.default_params())
        assert M3.h, su.errmsg()
        B = np.ascontiguousarray(d["B4"])
        n = B.shape[0]
        blocks = [np.ascontiguousarray(B[:, :3]), np.ascontiguousarray(B[:, 3:])]  # 3 + 1 columns
        Bd = [torch.from_numpy(b).cuda() for b in blocks]
        Xd = [torch.empty_like(b) for b in Bd]
        ncols = (C.c_size_t * 2)(3, 1)
        Bp = (C.c_void_p * 2)(*[b.data_ptr() for b in Bd])
        Xp = (C.c_void_p * 2)(*[x.data_ptr() for x in Xd])
        for op, want in ((su.LHF_S, X), (su.LHF_M, XM)):
            assert getattr(L, f"lhf{t}ApplyBatchDev")(M3.h, op, 2, Bp, ncols, ncols, Xp, ncols) == su.LHF_SUCCESS, su.errmsg()
            G = torch.zeros((n, 6), dtype=Bd[0].dtype, device="cuda")  # gathered with a padded row stride
            torch.cuda.synchronize()  # (torch fills G on ITS stream; the replicas' streams know nothing of it)
            assert getattr(L, f"lhf{t}GatherBatchDev")(M3.h, 2, Xp, ncols, ncols, C.c_void_p(G.data_ptr()), 6) == su.LHF_SUCCESS
            assert getattr(L, f"lhf{t}SyncDevices")(M3.h) == su.LHF_SUCCESS
            got = G.cpu().numpy()
            assert np.array_equal(got[:, :4], want) and not got[:, 4:].any()
            assert np.array_equal(Xd[0].cpu().numpy(), want[:, :3]) and np.array_equal(Xd[1].cpu().numpy(), want[:, 3:])
        # errors: more blocks than replicas, a stride smaller than the block, NULL arrays
        three = (C.c_size_t * 3)(1, 1, 1)
        Bp3 = (C.c_void_p * 3)(Bd[0].data_ptr(), Bd[0].data_ptr(), Bd[0].data_ptr())
        assert getattr(L, f"lhf{t}ApplyBatchDev")(M3.h, su.LHF_S, 3, Bp3, three, three, Bp3, three) == su.LHF_MISMATCHED_SIZES
        small = (C.c_size_t * 2)(2, 1)
        assert getattr(L, f"lhf{t}ApplyBatchDev")(M3.h, su.LHF_S, 2, Bp, ncols, small, Xp, ncols) == su.LHF_MISMATCHED_SIZES
        assert getattr(L, f"lhf{t}ApplyBatchDev")(M3.h, su.LHF_S, 2, None, ncols, ncols, Xp, ncols) == su.LHF_NULL_OBJ
        assert getattr(L, f"lhf{t}ApplyBatchDev")(None, su.LHF_S, 2, Bp, ncols, ncols, Xp, ncols) == su.LHF_NULL_OBJ
        M3.close()
    finally:
        assert L.lhfSetDevices(None, 0) == su.LHF_SUCCESS


def test_apply_batch_dev_across_two_devices(real_case):
    # The same flow with the replicas on TWO devices: the blocks live in each device's own memory, the gather crosses
    # devices (do_copy_columns checks / enables peer access).  Skipped on a one-GPU box -- which is every box of this
    # pool so far: the cross-device copy has never executed.
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices")
    levels, d, A, M = real_case
    L = su.lib()
    st, X, _ = M.apply_batch(su.LHF_S, d["B4"])
    assert st == 0
    ids = (C.c_int * 2)(0, 1)
    assert L.lhfSetDevices(ids, 2) == su.LHF_SUCCESS
    try:
        M3 = su.Hif("d", A, None, su.default_params())
        assert M3.h, su.errmsg()
        B = np.ascontiguousarray(d["B4"])
        n = B.shape[0]
        blocks = [np.ascontiguousarray(B[:, :3]), np.ascontiguousarray(B[:, 3:])]
        Bd = [torch.from_numpy(b).to(f"cuda:{k}") for k, b in enumerate(blocks)]
        Xd = [torch.empty_like(b) for b in Bd]
        ncols = (C.c_size_t * 2)(3, 1)
        Bp = (C.c_void_p * 2)(*[b.data_ptr() for b in Bd])
        Xp = (C.c_void_p * 2)(*[x.data_ptr() for x in Xd])
        assert L.lhfdApplyBatchDev(M3.h, su.LHF_S, 2, Bp, ncols, ncols, Xp, ncols) == su.LHF_SUCCESS, su.errmsg()
        G = torch.zeros((n, 4), dtype=Bd[0].dtype, device="cuda:0")
        for k in range(2):
            torch.cuda.synchronize(k)
        assert L.lhfdGatherBatchDev(M3.h, 2, Xp, ncols, ncols, C.c_void_p(G.data_ptr()), 4) == su.LHF_SUCCESS, su.errmsg()
        assert L.lhfdSyncDevices(M3.h) == su.LHF_SUCCESS
        assert np.array_equal(G.cpu().numpy(), X)
        st, X3, _ = M3.apply_batch(su.LHF_S, d["B4"])  # host batch split over the two devices
        assert st == 0 and np.array_equal(X3, X)
        M3.close()
    finally:
        assert L.lhfSetDevices(None, 0) == su.LHF_SUCCESS


def test_handle_finalized_for_a_narrow_batch_gives_the_same_bits(real_case, monkeypatch):
    """HIFIR_AMD_MAX_NRHS (what a single-vector lhf?Solve user sets): a handle finalized for 16 columns solves to the very
    bits of the default (64-column) handle, a wider lhf?ApplyBatch on it is tiled, and lhf?GetResidentBytes reports what
    one replica keeps in HBM (libhifir.h:685-716 has no counterpart: additive, include/libhifir_amd_ext.h)."""
    levels, d, A, M = real_case
    st, x64 = M.solve(d["b"])
    assert st == su.LHF_SUCCESS
    res64 = (su._sz * 6)()
    assert M._f("GetResidentBytes")(M.h, res64) == su.LHF_SUCCESS
    assert res64[0] > 0 and res64[3] > 0 and res64[4] == 64 and res64[5] == 64
    monkeypatch.setenv("HIFIR_AMD_MAX_NRHS", "16")
    M16 = su.Hif("d", A, None, su.default_params())
    assert M16.h, su.errmsg()
    try:
        res16 = (su._sz * 6)()
        assert M16._f("GetResidentBytes")(M16.h, res16) == su.LHF_SUCCESS
        assert res16[5] == 16 and res16[0] == res64[0] and res16[1] == res64[1]
        st, x16 = M16.solve(d["b"])
        assert st == su.LHF_SUCCESS and np.array_equal(x16, x64)
        B = np.stack([d["b"] * (1.0 + 0.25 * k) for k in range(40)], axis=1)  # wider than the handle was finalized for
        st, X16, _ = M16.apply_batch(su.LHF_S, B)
        st2, X64, _ = M.apply_batch(su.LHF_S, B)
        assert st == su.LHF_SUCCESS and st2 == su.LHF_SUCCESS and np.array_equal(X16, X64)
    finally:
        M16.close()
