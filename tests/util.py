"""Shared helpers for the tests: fixture loading and synthetic matrices (no reference access)."""
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
HIER_NAMES = ["p2d_5", "p2d_30", "p2d_64_deep", "p2d_100_tuned", "p3d_12", "cd2d_48", "demo_A", "young1c",
              "p2d_32_symm", "herm_24_symm",  # is_symm factorizations (last level = SYEIG)
              "p2d_30_lup",  # the reference built with HIF_DENSE_MODE=0 (last level = LUP)
              "kkt_26"]  # complex saddle point (BASELINE config 5's generator at 2,028 rows)
LEVEL_KEYS = ["m", "n", "dense_n", "dense_rank", "dense_symm", "spd", "dense_lup", "d", "s", "t", "p", "p_inv", "q", "q_inv", "dense"] + [
    f"{a}_{b}" for a in "LUEF" for b in ("colptr", "rowind", "vals")]


def load_hier(name):
    """-> (levels, data): levels in the oracle.orc format, data = the other fixture arrays."""
    z = np.load(os.path.join(GOLDEN, f"hier_{name}.npz"))
    nl = int(z["nlevels"])
    levels = []
    for l in range(nl):
        lv = {}
        for k in LEVEL_KEYS:
            key = f"L{l}_{k}"
            if key in z.files:
                v = z[key]
                lv[k] = int(v) if v.ndim == 0 else v
        levels.append(lv)
    data = {k: z[k] for k in z.files if not k.startswith("L") or not k[1].isdigit()}
    return levels, data


def poisson2d(nx, ny=None):
    ny = ny or nx
    Tx = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    Ty = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(ny, ny), format="csr")
    A = (sp.kron(sp.identity(ny), Tx) + sp.kron(Ty, sp.identity(nx))).tocsr()
    A.sort_indices()
    return A


def stokes_kkt(nx, omega=0.1, eps=1e-8):
    """BASELINE config 5: the SuiteSparse saddle point cannot be fetched offline; its stand-in is the
    complex Stokes-like KKT system

        [ K + i*omega*M    B^T   ]      K = vector Laplacian (5-pt, per velocity component) on an nx x nx grid,
        [ B               -eps*I ]      M = lumped mass (identity), B = discrete divergence (forward differences),

    omega = 0.1, eps = 1e-8: 3 nx^2 rows, complex symmetric, indefinite, with a (nearly) zero (2,2) block -- the
    factorization defers the pressure rows into the Schur complements.  nx = 816 gives 1,997,568 rows."""
    n1 = nx * nx
    I = sp.identity(nx, format="csr")
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    lap = sp.kron(I, T) + sp.kron(T, I)
    K = sp.block_diag([lap, lap]).astype(np.complex128) + 1j * omega * sp.identity(2 * n1)
    D = sp.diags([-1.0, 1.0], [0, 1], shape=(nx, nx), format="csr")
    B = sp.hstack([sp.kron(I, D), sp.kron(D, I)]).tocsr()
    A = sp.bmat([[K, B.T], [B, -eps * sp.identity(n1)]], format="csr").astype(np.complex128)
    A.sort_indices()
    return A


def relerr(x, ref):
    return float(np.abs(x - ref).max() / max(np.abs(ref).max(), 1e-300))
