#!/usr/bin/env python3
"""Regenerates the golden fixtures in tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference and oracle/_ref/libhifref.so built by
`make -C oracle ref`).  The outputs are committed; the GPU box and CI only read them.

Fixtures are DATA (inputs + expected outputs), never reference source text:
  kat_dense.json   the MATLAB known-answer vectors embedded in the reference's own dense-solver
                   tests (tests/test_sss_qrcp.cpp:22-196, test_qrcp_cmplx.cpp, test_sss_lup.cpp,
                   test_lup_cmplx.cpp, test_syev.cpp, test_heev.cpp): matrix, rhs, x_ref, tol 1e-10
  hier_<name>.npz  a factored hierarchy exported field by field from hif::HIF<> (Prec.hpp:309-323)
                   + the matrix, rhs b, x = HIF::solve(b) (builder.hpp:410), b2 = HIF::mmultiply(x)
                   (:503), x_ir4 = HIF::hifir(A,b,4) (:459), (x_irb, status) = hifir with betas (:482),
                   a 4-RHS batch solved column by column, and xt / XT4 = the same with
                   trans=true (x = M^{-H} b, prec_solve_tran, alg/prec_solve.hpp:542-612).
"""
import json
import os
import re
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref  # noqa: E402

REF = os.environ.get("HIFIR_REFERENCE", "/root/reference")


def _arrays(path):
    """name -> list of floats, for every `name[N] = {...}` initialiser in a reference test file."""
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"(\w+)\[(\d+)\]\s*=\s*\{([^}]*)\}", txt):
        name, cnt, body = m.group(1), int(m.group(2)), m.group(3)
        vals = [float(v) for v in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", body)]
        if name == "inds":
            continue
        assert len(vals) == cnt, (path, name, len(vals), cnt)
        out[name] = vals
    return out


def make_kats():
    kats = []
    for fname, cplx in [("test_sss_qrcp.cpp", False), ("test_qrcp_cmplx.cpp", True), ("test_sss_lup.cpp", False),
                        ("test_lup_cmplx.cpp", True), ("test_syev.cpp", False)]:
        a = _arrays(os.path.join(REF, "tests", fname))
        n = len(a["b"])
        kats.append(dict(name=fname[5:-4], source=f"tests/{fname}", n=n, complex=cplx, tol=1e-10, layout="row",
                         a_rowmajor_re=a["a"], a_rowmajor_im=[0.0] * (n * n), b_re=a["b"], b_im=[0.0] * n,
                         x_re=a["x_ref"], x_im=[0.0] * n))
    a = _arrays(os.path.join(REF, "tests", "test_heev.cpp"))
    n = len(a["b_real"])
    # test_heev.cpp:69-84 assembles by COLUMNS (push_back_col): the data is column-major
    kats.append(dict(name="heev", source="tests/test_heev.cpp", n=n, complex=True, tol=1e-10, layout="col",
                     a_rowmajor_re=a["a_real"], a_rowmajor_im=a["a_imag"], b_re=a["b_real"], b_im=a["b_imag"],
                     x_re=a["x_ref_real"], x_im=a["x_ref_imag"]))
    with open(os.path.join(HERE, "kat_dense.json"), "w") as f:
        json.dump(kats, f)
    print("kat_dense.json:", [k["name"] for k in kats])


def poisson2d(nx):
    I = sp.identity(nx, format="csr")
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    A = (sp.kron(I, T) + sp.kron(T, I)).tocsr()
    A.sort_indices()
    return A


def poisson3d(nx):
    I = sp.identity(nx, format="csr")
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    A = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
    A.sort_indices()
    return A


def convdiff2d(nx, eps=0.05):
    """nonsymmetric: -eps*Lap + (1, 0.5).grad, upwind; exercises L != U^T and t != s."""
    h = 1.0 / (nx + 1)
    I = sp.identity(nx, format="csr")
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr") * (eps / h / h)
    D = sp.diags([-1.0, 1.0], [-1, 0], shape=(nx, nx), format="csr") / h
    A = (sp.kron(I, T + 1.0 * D) + sp.kron(T + 0.5 * D, I)).tocsr()
    A.sort_indices()
    return A


def hermitian2d(nx, gamma=0.3):
    """complex Hermitian: 2-D Laplacian + i*gamma*(skew-symmetric first difference in x)."""
    I = sp.identity(nx, format="csr")
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    S = sp.diags([-1.0, 1.0], [-1, 1], shape=(nx, nx), format="csr")
    A = (sp.kron(I, T) + sp.kron(T, I)).astype(np.complex128) + 1j * gamma * sp.kron(I, S)
    A = A.tocsr()
    A.sort_indices()
    return A


def rhs(n, cplx=False):
    b = np.sin(0.001 * np.arange(n)) + 1.0
    if cplx:
        b = b + 1j * np.cos(0.002 * np.arange(n))
    return b


def save_hier(name, A, params=None, cplx=False, lup=False):
    A = A.tocsr()
    A.sort_indices()
    n = A.shape[0]
    vals = A.data.astype(np.complex128 if cplx else np.float64)
    M = ref.RefHIF(A.indptr, A.indices, vals, params, lup=lup)
    b = rhs(n, cplx)
    x = M.solve(b)
    # (a reference built with HIF_DENSE_MODE=0 cannot instantiate HIF::mmultiply, LUP.hpp:181 vs prec_prod.hpp:85:
    #  the fixture then stores b itself, the value the round trip M (M^-1 b) has to return)
    b2 = b.copy() if lup else M.mmultiply(x)
    x_ir4, _ = M.hifir(b, 4)
    x_irb, st = M.hifir(b, 16, [1e-10, 1e3])
    B = np.stack([b + 0.01 * k for k in range(4)], axis=1)  # (n, 4) row-interleaved
    X = np.stack([M.solve(B[:, k].copy()) for k in range(4)], axis=1)
    xt = M.solve(b, trans=True)  # x = M^{-H} b: HIF::solve(b, x, true) -> prec_solve_tran (prec_solve.hpp:542)
    XT = np.stack([M.solve(B[:, k].copy(), trans=True) for k in range(4)], axis=1)
    d = dict(nlevels=M.nlevels, A_indptr=A.indptr.astype(np.int64), A_indices=A.indices.astype(np.int32), A_vals=vals,
             b=b, x=x, b2=b2, x_ir4=x_ir4, x_irb=x_irb, irb_status=np.array(st, dtype=np.int32), B4=B, X4=X, xt=xt, XT4=XT,
             params=np.zeros(7) if params is None else params)
    for l, lv in enumerate(M.levels()):
        for k, v in lv.items():
            d[f"L{l}_{k}"] = np.asarray(v)
    path = os.path.join(HERE, f"hier_{name}.npz")
    np.savez_compressed(path, **d)
    lv = M.levels()
    print(f"hier_{name}.npz: n={n} levels={M.nlevels} dense={lv[-1]['dense_n']} rank={lv[-1]['dense_rank']} "
          f"nnz(M)={M.nnz} ir={st} size={os.path.getsize(path) / 1e6:.2f} MB "
          f"roundtrip={np.linalg.norm(b2 - b) / np.linalg.norm(b):.2e}")


def main():
    only = set(sys.argv[1:])  # names given on the command line: regenerate just those fixtures
    if only:
        global save_hier
        _save = save_hier

        def save_hier(name, *a, **k):  # noqa: F811
            if name in only:
                _save(name, *a, **k)
    else:
        make_kats()
    tuned = ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0)
    save_hier("p2d_5", poisson2d(5))
    save_hier("p2d_30", poisson2d(30))
    # multi-level, no huge dense block: lower dense_thres so that the recursion goes deeper
    save_hier("p2d_64_deep", poisson2d(64), ref.make_params(dense_thres=60))
    save_hier("p2d_100_tuned", poisson2d(100), ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0, dense_thres=100))
    save_hier("p3d_12", poisson3d(12), ref.make_params(dense_thres=100))
    save_hier("cd2d_48", convdiff2d(48), ref.make_params(dense_thres=80))
    # symmetric factorizations (Options::is_symm -> symm_level_factorize; the last level is SYEIG, not QRCP):
    # a real symmetric multilevel hierarchy and a complex Hermitian one
    save_hier("p2d_32_symm", poisson2d(32), ref.make_params(dense_thres=60, is_symm=1))
    save_hier("herm_24_symm", hermitian2d(24), ref.make_params(dense_thres=60, is_symm=1), cplx=True)
    # the reference compiled with HIF_DENSE_MODE=0 (oracle/_ref/libhifref_lup.so): the last level is LUP, not QRCP
    save_hier("p2d_30_lup", poisson2d(30), lup=True)
    A = scipy.io.mmread(os.path.join(REF, "examples", "demo_inputs", "A.mm")).tocsr()
    save_hier("demo_A", A)  # libhifir/tests/test_real.c:88-146 input
    Z = scipy.io.mmread(os.path.join(REF, "examples", "demo_inputs", "young1c.mtx")).tocsr()
    save_hier("young1c", Z, cplx=True)  # libhifir/tests/test_complex.c:90-140 input
    # BASELINE config 5's generator (tests/util.py stokes_kkt: complex Stokes-like KKT, omega = 0.1, eps = 1e-8) at
    # 2,028 rows, default parameters: indefinite, deferred pressure rows, dense QRCP last level
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import stokes_kkt

    save_hier("kkt_26", stokes_kkt(26), cplx=True)


if __name__ == "__main__":
    main()
