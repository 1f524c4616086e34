"""CPU: the C restatement (oracle/) against the committed golden vectors produced by the real
reference (tests/golden/make_golden.py) and against the reference's own known-answer tests."""
import json
import os

import numpy as np
import pytest

from oracle import orc
from util import GOLDEN, HIER_NAMES, load_hier, relerr


@pytest.mark.parametrize("kat", json.load(open(os.path.join(GOLDEN, "kat_dense.json"))), ids=lambda k: k["name"])
def test_dense_known_answer(kat):
    # the reference's MATLAB vectors; tolerance as in tests/test_sss_qrcp.cpp:16,195 (1e-10 absolute)
    n = kat["n"]
    a = (np.array(kat["a_rowmajor_re"]) + 1j * np.array(kat["a_rowmajor_im"])).reshape(n, n)
    b = np.array(kat["b_re"]) + 1j * np.array(kat["b_im"])
    xr = np.array(kat["x_re"]) + 1j * np.array(kat["x_im"])
    if kat["layout"] == "col":
        a = a.T
    if not kat["complex"]:
        a, b, xr = a.real, b.real, xr.real
    x, rk = orc.qrcp(np.asfortranarray(a).ravel(order="F"), b)
    assert rk == n
    assert np.abs(x - xr).max() <= kat["tol"]
    # multiply is the inverse map: A x = b
    bb, _ = orc.qrcp(np.asfortranarray(a).ravel(order="F"), x, op=1)
    assert np.abs(bb - b).max() <= 1e-10 * max(1.0, np.abs(xr).max())
    # test_syev.cpp / test_heev.cpp drive hif::SYEIG itself (eig.solve): the same vectors pin the restatement of
    # the symmetric last level (Jacobi eigensolver + truncation + Q f(w) Q^H)
    if np.abs(a - a.conj().T).max() <= 1e-13:
        xs, rks, w = orc.syeig(np.asfortranarray(a).ravel(order="F"), b)
        assert rks == n
        assert np.abs(xs - xr).max() <= kat["tol"]
        assert np.abs(np.sort(w) - np.linalg.eigvalsh(a)).max() <= 1e-12 * np.abs(w).max()
        bs, _, _ = orc.syeig(np.asfortranarray(a).ravel(order="F"), xs, op=1)
        assert np.abs(bs - b).max() <= 1e-10 * max(1.0, np.abs(xr).max())
    # test_sss_lup.cpp / test_lup_cmplx.cpp drive hif::LUP itself; every vector is a linear system, so all of them pin
    # the ?getrf / ?getrs restatement of the LUP last level
    xl, info = orc.lup(np.asfortranarray(a).ravel(order="F"), b)
    assert info == 0 and np.abs(xl - xr).max() <= kat["tol"]
    bl, _ = orc.lup(np.asfortranarray(a).ravel(order="F"), xl, op=1)
    assert np.abs(bl - b).max() <= 1e-10 * max(1.0, np.abs(xr).max())


@pytest.mark.parametrize("name", HIER_NAMES)
def test_solve_matches_reference(name):
    levels, d = load_hier(name)
    O = orc.Oracle(levels)
    x = O.solve(d["b"])
    has_dense = levels[-1].get("dense_n", 0) > 0
    if not has_dense and not np.iscomplexobj(d["b"]):
        assert np.array_equal(x, d["x"]), "sparse-only hierarchy must be bit-identical"
    assert relerr(x, d["x"]) <= 1e-12
    if has_dense:
        assert O.dense_rank == levels[-1]["dense_rank"]


@pytest.mark.parametrize("name", HIER_NAMES)
def test_transposed_solve_matches_reference(name):
    # x = M^{-H} b: HIF::solve(b, x, trans=true) -> prec_solve_tran (alg/prec_solve.hpp:542-612)
    levels, d = load_hier(name)
    O = orc.Oracle(levels)
    xt = O.solve(d["b"], trans=True)
    has_dense = levels[-1].get("dense_n", 0) > 0
    if not has_dense and not np.iscomplexobj(d["b"]):
        assert np.array_equal(xt, d["xt"]), "sparse-only hierarchy must be bit-identical"
    assert relerr(xt, d["xt"]) <= 1e-12
    XT = O.solve_batch(d["B4"], threads=2, trans=True)
    assert relerr(XT, d["XT4"]) <= 1e-12
    for k in range(4):
        assert np.array_equal(XT[:, k], O.solve(d["B4"][:, k].copy(), trans=True))


@pytest.mark.parametrize("name", HIER_NAMES)
def test_batch_is_columnwise(name):
    levels, d = load_hier(name)
    O = orc.Oracle(levels)
    X = O.solve_batch(d["B4"], threads=2)
    assert relerr(X, d["X4"]) <= 1e-12
    for k in range(4):
        assert np.array_equal(X[:, k], O.solve(d["B4"][:, k].copy()))


@pytest.mark.parametrize("name", HIER_NAMES)
def test_mmultiply_and_roundtrip(name):
    levels, d = load_hier(name)
    O = orc.Oracle(levels)
    b2 = O.mmultiply(d["x"])
    assert relerr(b2, d["b2"]) <= 1e-10
    # libhifir/tests/test_real.c:110-146 invariant
    assert np.linalg.norm(b2 - d["b"]) / np.linalg.norm(d["b"]) <= 1e-10
    # the same round trip with the conjugate transposes: M^H (M^{-H} b) = b (prec_prod_tran, prec_prod.hpp:148-235)
    bt = O.mmultiply(d["xt"], trans=True)
    assert np.linalg.norm(bt - d["b"]) / np.linalg.norm(d["b"]) <= 1e-10


@pytest.mark.parametrize("name", HIER_NAMES)
def test_iterative_refinement(name):
    levels, d = load_hier(name)
    O = orc.Oracle(levels)
    x4, st4 = O.hifir(d["A_indptr"], d["A_indices"], d["A_vals"], d["b"], 4)
    assert st4 == (4, -1)
    assert relerr(x4, d["x_ir4"]) <= 1e-11
    xb, stb = O.hifir(d["A_indptr"], d["A_indices"], d["A_vals"], d["b"], 16, [1e-10, 1e3])
    assert stb == tuple(int(v) for v in d["irb_status"])
    assert relerr(xb, d["x_irb"]) <= 1e-11


def test_mrhs_kernels_equal_columnwise():
    # the reference's own component tests (tests/test_mrhs_trsv.cpp, test_mv_mrhs.cpp) use Nrhs=2
    levels, _ = load_hier("p2d_64_deep")
    rng = np.random.default_rng(7)
    for lv in levels:
        m, nm = lv["m"], lv["n"] - lv["m"]
        for op, nm_, (nr, nc) in [(0, "L", (m, m)), (1, "U", (m, m)), (2, "E", (nm, m)), (2, "F", (m, nm))]:
            cp, ri, v = lv[nm_ + "_colptr"], lv[nm_ + "_rowind"], lv[nm_ + "_vals"]
            X = rng.uniform(-1, 1, size=(nc if op == 2 else nr, 3))
            Y = orc.ccs_kernel(op, nr, nc, cp, ri, v, X, nrhs=3)
            for k in range(3):
                yk = orc.ccs_kernel(op, nr, nc, cp, ri, v, X[:, k].copy())
                assert np.array_equal(Y[:, k], yk)
