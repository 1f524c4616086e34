"""Development probe (tests-side, may use the compiled reference in oracle/_ref): factorizes a 2-D
Poisson matrix with the REAL reference on the host, imports the hierarchy into the HIP path, checks
parity and times the batched apply.  Usage: python tests/perf_probe.py NX default|tuned NRHS [reps]"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import torch  # noqa: E402

import hifir_amd  # noqa: E402
from oracle import orc, ref  # noqa: E402
from util import poisson2d, relerr  # noqa: E402

nx = int(sys.argv[1])
mode = sys.argv[2]
nrhs = int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
if mode.endswith("-3d"):  # e.g. "tuned-3d": 3-D 7-pt Poisson nx^3 (BASELINE config 4's stencil)
    import scipy.sparse as sp

    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    I = sp.identity(nx, format="csr")
    A = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
    A.sort_indices()
    mode = mode[:-3]
elif mode == "kkt":  # BASELINE config 5: complex saddle point (tests/util.py stokes_kkt), nx = 816 -> 1,997,568 rows
    from util import stokes_kkt

    A = stokes_kkt(nx)
else:
    A = poisson2d(nx)
cplx = mode.endswith("-z") or mode == "kkt"  # e.g. "default-z": complex shifted Laplacian (BASELINE config 5 stand-in)
if cplx and mode != "kkt":
    import scipy.sparse as sp

    A = (A - (0.3 + 0.2j) * sp.identity(A.shape[0])).tocsr()
    A.sort_indices()
    mode = mode[:-2]
n = A.shape[0]
P = None if mode == "default" else ref.make_params(tau=1e-2, kappa=3.0 if mode == "kkt" else 5.0, alpha=3.0)
t0 = time.time()
R = ref.RefHIF(A.indptr, A.indices, A.data, P)
t1 = time.time()
levels = R.levels()
print(f"reference factorize {t1 - t0:.1f}s levels={R.nlevels} nnz={R.nnz}", flush=True)
t0 = time.time()
M = hifir_amd.HIF.from_levels(levels, max_nrhs=min(nrhs, 64))
print(f"import+upload {time.time() - t0:.1f}s stats={M.stats()}", flush=True)
rng = np.random.default_rng(20260101)
B = rng.uniform(-1, 1, size=(n, nrhs))
if cplx:
    B = B + 1j * rng.uniform(-1, 1, size=(n, nrhs))
B[:, 0] = np.sin(0.001 * np.arange(n)) + 1
Bd = torch.from_numpy(B).cuda()
Xd = torch.empty_like(Bd)
t0 = time.time()
M.solve_mrhs(Bd, Xd)
M.sync()
print(f"first apply (graph capture) {time.time() - t0:.2f}s launches={M.stats()['launches']}", flush=True)
X = Xd.cpu().numpy()
t0 = time.time()
x0 = R.solve(B[:, 0].copy())
tref = time.time() - t0
print(f"reference solve 1 rhs: {tref * 1e3:.1f} ms; GPU col0 relerr vs reference {relerr(X[:, 0], x0):.3e}", flush=True)
O = orc.Oracle(levels)
t0 = time.time()
xo = O.solve(B[:, nrhs - 1].copy())
print(f"oracle solve 1 rhs: {(time.time() - t0) * 1e3:.1f} ms; GPU last col relerr vs oracle {relerr(X[:, nrhs - 1], xo):.3e} "
      f"bit-exact={np.array_equal(X[:, nrhs - 1], xo)}", flush=True)
ms = M.time_apply(Bd, Xd, warmup=2, reps=reps)
balg = M.algorithmic_bytes(nrhs)
print(f"RESULT nx={nx} mode={mode} nrhs={nrhs}: {ms:.3f} ms/batch  {nrhs / ms * 1e3:.0f} RHS-applies/s  "
      f"B_alg={balg / 1e9:.3f} GB  {balg / ms / 1e6:.1f} GB/s  frac_of_8TB/s={balg / ms / 1e6 / 8000:.4f}", flush=True)
import json  # noqa: E402
import os  # noqa: E402

# for tests/profile_config.sh: which level / stage every launch of the apply belongs to, and B_alg level by level
print("LAUNCHMAP " + json.dumps({"map": [16 * l + s_ for (l, s_) in M.launch_map()],
                                 "level_bytes": {str(l): {str(k): v for k, v in d_.items()} for l, d_ in M.level_bytes(nrhs).items()},
                                 "level_rows": [int(M.level_stats(l)["n"]) for l in range(int(M.stats()["sparse_levels"]))]}), flush=True)

if os.environ.get("GMRES"):  # GMRES(30) on all columns: time per inner step (apply + SpMM + Gram-Schmidt)
    M.set_matrix(A.indptr, A.indices, A.data)
    for maxit in (10, 30):
        M.gmres(Bd, restart=30, rtol=1e-300, maxit=maxit)  # warm-up (buffers)
        M.sync()
        t0 = time.time()
        Xg, fl, it = M.gmres(Bd, restart=30, rtol=1e-300, maxit=maxit)
        M.sync()
        el = time.time() - t0
        print(f"GMRES(30) nrhs={nrhs} maxit={maxit}: {el * 1e3:.1f} ms total, iters={int(it.max())}, "
              f"{el * 1e3 / max(1, int(it.max())):.2f} ms per inner step for all columns", flush=True)
M.close()
