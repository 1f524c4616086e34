"""Development aid (GPU box): what the analysis trailer of a hierarchy file saves between hifamd_load and the first apply.
  python tests/dev_load_time.py [nx] [2d|3d] [params]
Factorizes with the compiled reference (oracle/_ref), saves the hierarchy with and without the trailer, loads each twice."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import hifir_amd
    from oracle import ref
    import scipy.sparse as sp
    from util import poisson2d

    nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    dim = sys.argv[2] if len(sys.argv) > 2 else "2d"
    params = sys.argv[3] if len(sys.argv) > 3 else "default"
    if dim == "2d":
        A = poisson2d(nx)
    else:  # 3-D 7-pt Poisson nx^3 (BASELINE config 4's stencil)
        T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
        I = sp.identity(nx, format="csr")
        A = (sp.kron(sp.kron(I, I), T) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(T, I), I)).tocsr()
        A.sort_indices()
    t0 = time.time()
    R = ref.RefHIF(A.indptr, A.indices, A.data, ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0) if params == "tuned" else None)
    print("factorize %.1f s, n = %d" % (time.time() - t0, A.shape[0]), flush=True)
    levels = R.levels()
    t0 = time.time()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    print("from_levels %.2f s  analysis %.2f s  finalize %.2f s" % (time.time() - t0, M.stats_ext()["analysis_s"], M.stats_ext()["finalize_s"]), flush=True)
    paths = {"plain": "/tmp/h_plain.hifamd", "with analysis": "/tmp/h_ana.hifamd"}
    for k, p in paths.items():
        t0 = time.time()
        M.save(p, analysis=(k != "plain"))
        print("save %-14s %.2f s  %.0f MB" % (k, time.time() - t0, os.path.getsize(p) / 1e6), flush=True)
    n = A.shape[0]
    B = np.random.default_rng(0).uniform(-1, 1, (n, 8))
    X0 = M.solve_mrhs(B)
    for rep in range(2):
        for k, p in paths.items():
            t0 = time.time()
            M2 = hifir_amd.HIF.load(p, max_nrhs=64)
            t1 = time.time()
            X = M2.solve_mrhs(B)
            s = M2.stats_ext()
            print("load %-14s %.2f s (analysis %.2f s, %d levels from the trailer, finalize %.2f s)  first apply %.2f s  same bits: %s"
                  % (k, t1 - t0, s["analysis_s"], int(s["analysis_cached_levels"]), s["finalize_s"], time.time() - t1, np.array_equal(X, X0)), flush=True)
            M2.close()


if __name__ == "__main__":
    main()
