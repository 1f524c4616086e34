"""Dev probe: do two independent applies (two handles, two streams) overlap on the GPU?"""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
sys.path.insert(0, __file__.rsplit("/", 1)[0])
import torch  # noqa

import hifir_amd  # noqa
from oracle import ref  # noqa
from util import poisson2d  # noqa

mode = sys.argv[1]
nh = int(sys.argv[2])
A = poisson2d(1000)
P = None if mode == "default" else ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0)
levels = ref.RefHIF(A.indptr, A.indices, A.data, P).levels()
n = A.shape[0]
Ms = [hifir_amd.HIF.from_levels(levels, max_nrhs=64) for _ in range(nh)]
Bs = [torch.rand((n, 64), dtype=torch.float64, device="cuda") for _ in range(nh)]
Xs = [torch.empty_like(b) for b in Bs]
for M, B, X in zip(Ms, Bs, Xs):
    M.solve_mrhs(B, X)
    M.sync()
for trial in range(2):
    t0 = time.perf_counter()
    for _ in range(10):
        Ms[0].solve_mrhs(Bs[0], Xs[0])
    Ms[0].sync()
    t1 = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        for M, B, X in zip(Ms, Bs, Xs):
            M.solve_mrhs(B, X)
    for M in Ms:
        M.sync()
    tn = (time.perf_counter() - t0) / 10
    print(f"{mode}: one handle {t1 * 1e3:.2f} ms per 64 RHS; {nh} handles concurrently {tn * 1e3:.2f} ms per {64 * nh} RHS "
          f"-> {tn / nh * 1e3:.2f} ms per 64 RHS")
