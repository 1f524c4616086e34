"""Development aid (GPU box): time the batched apply of the 1M-row default hierarchy under several environment
settings, one child process per setting (the options are read when a handle is created).
  python tests/dev_sweep.py [--nx 1000] [--params default|tuned] [--nrhs 64,1] "" "HIFIR_AMD_CS=0" "A=1,B=2" ...
The hierarchy is factorized once by the compiled reference (oracle/_ref) and handed to the children through a pickle."""
import os
import pickle
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(path, nrhs_list, check):
    import torch
    import hifir_amd
    levels = pickle.load(open(path, "rb"))
    n = int(levels[0]["n"])
    t0 = time.time()
    M = hifir_amd.HIF.from_levels(levels, max_nrhs=64)
    fin = time.time() - t0
    out = []
    rng = np.random.default_rng(20260101)
    for nrhs in nrhs_list:
        B = rng.uniform(-1, 1, size=(n, nrhs))
        Bd = torch.from_numpy(B).cuda()
        Xd = torch.empty_like(Bd)
        M.solve_mrhs(Bd, Xd)
        M.sync()
        ms = M.time_apply(Bd, Xd, warmup=3, reps=20)
        err = ""
        if check:
            from oracle import orc
            O = orc.Oracle(levels)
            xo = O.solve(B[:, nrhs - 1].copy())
            x = Xd[:, nrhs - 1].cpu().numpy()
            err = " relerr=%.2e" % (np.abs(x - xo).max() / np.abs(xo).max())
        out.append("nrhs=%d %.3f ms launches=%d%s" % (nrhs, ms, M.stats()["launches"], err))
    print("  finalize %.2fs | " % fin + " | ".join(out), flush=True)
    print("  stats_ext", {k: (round(v, 3) if abs(v) > 1e-3 else v) for k, v in M.stats_ext().items()}, flush=True)
    M.close()


def main():
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child(args[1], [int(x) for x in args[2].split(",")], args[3] == "1")
        return
    nx, params, nrhs, check = 1000, "default", "64", "0"
    while args and args[0].startswith("--"):
        k = args.pop(0)
        v = args.pop(0)
        if k == "--nx":
            nx = int(v)
        elif k == "--params":
            params = v
        elif k == "--nrhs":
            nrhs = v
        elif k == "--check":
            check = v
    from oracle import ref
    from util import poisson2d
    A = poisson2d(nx)
    P = None if params == "default" else ref.make_params(tau=1e-2, kappa=5.0, alpha=3.0)
    t0 = time.time()
    R = ref.RefHIF(A.indptr, A.indices, A.data, P)
    levels = R.levels()
    print("factorized in %.1fs" % (time.time() - t0), flush=True)
    path = "/tmp/dev_sweep_levels.pkl"
    pickle.dump(levels, open(path, "wb"), protocol=4)
    for setting in (args or [""]):
        env = dict(os.environ)
        for kv in setting.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        print("[%s]" % setting, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path, nrhs, check], env=env, check=False)


if __name__ == "__main__":
    main()
