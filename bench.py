#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native HIFIR apply path.

Metric (BASELINE.json): preconditioner applies/sec + achieved HBM GB/s on the 1M-row 2-D 5-pt
Laplacian with 64 right-hand sides.  One "step" = one batched apply X = M^{-1} B of the whole
multilevel hierarchy (prec_solve, reference src/hif/alg/prec_solve.hpp:332-412) over one [n][64]
block that is already resident in HBM; `value` counts RHS-applies per second over all ranks.

  python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts the N ranks itself (python -m
torch.distributed.run --nproc-per-node N ... bench.py, as a CHILD, before anything here touches torch or the GPU),
relays rank 0's JSON line and exits with the child's code.  Launched by torch.distributed.run (the driver's way)
it is one of the ranks; --gpus must then equal WORLD_SIZE.

Multi-GPU: the path shards over right-hand sides only.  Every rank holds the whole
hierarchy and applies it to its OWN 64-column block: weak scaling, no collective in the data path;
one RCCL all_gather of the solution blocks after the timed region (reported as gather_ms).  The STRONG
split of one 64-column batch (64/N columns per GPU) is timed as well and reported as `strong_scaling`: shards of
fewer than 49 columns run the 16-column-slice kernels (about 2.7 ms for 8 columns against 3.9 ms for 64).

Hierarchy: factorization is not part of the measured path and stays on the host (north_star).
When the compiled reference is present (oracle/_ref, the GPU box gets it as a prebuilt binary) the
cpu_baseline leg factorizes the matrix with the REAL reference and times the reference's own
single-RHS solve; the hierarchy it produced is then handed to the HIP path field by field through
the import ABI -- exactly the deployment contract of INTEGRATION.md (reference factorizes, GPU
applies), so GPU and CPU numbers refer to the same factors.  For N > 1 rank 0 factorizes, writes the
imported hierarchy in the library's on-disk format (hifamd_save) and the other ranks replay that file
(hifamd_load; hifir_amd/dist.py share_hierarchy).  Without the compiled reference the same file format is
the only way to get a workload: bench.py then looks for $TMPDIR/hifir_amd_hier_p2d_<nx>_<params>.hifamd.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def poisson2d(nx):
    """2-D 5-pt Laplacian, natural ordering, Dirichlet, diag 4 / off-diag -1."""
    import scipy.sparse as sp

    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(nx, nx), format="csr")
    A = (sp.kron(sp.identity(nx), T) + sp.kron(T, sp.identity(nx))).tocsr()
    A.sort_indices()
    return A


PARAM_SETS = {"default": None, "tuned": dict(tau=1e-2, kappa=5.0, alpha=3.0)}


def hier_cache_path(nx, pname):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"hifir_amd_hier_p2d_{nx}_{pname}.hifamd")


def spawn_ranks(n):
    """--gpus N without a launcher: start the N ranks as a child process group.  The parent imports neither torch
    nor hifir_amd and makes no GPU call; it relays the child's output and exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL across processes)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def lib_fingerprint():
    """sha256 of the product library: profiles/ summaries are stamped with it (tests/prof_summarize.py) and only a
    summary made with THIS build may contribute counter traffic / per-stage times to the line."""
    import hashlib

    p = os.path.join(ROOT, "hifir_amd", "libhifir_amd.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None


def cpu_baseline_leg(A, pname, budget_s):
    """The ONLY place bench.py touches oracle/: times the reference's CPU path (1 thread, as
    prec_solve is serial) on a bounded sample of single-RHS solves of the SAME workload, and returns
    the factors it built so that the GPU leg applies the identical hierarchy."""
    from oracle import orc, ref

    n = A.shape[0]
    rng = np.random.default_rng(20260101)
    out = {}
    if ref.available():
        P = PARAM_SETS[pname]
        t0 = time.time()
        R = ref.RefHIF(A.indptr, A.indices, A.data, None if P is None else ref.make_params(**P))
        t_fac = time.time() - t0
        levels = R.levels()
        b = rng.uniform(-1, 1, n)
        R.solve(b)  # warm-up (sizes the reference's work buffer, builder.hpp:414-416)
        cnt, t0 = 0, time.time()
        while True:
            R.solve(b)
            cnt += 1
            el = time.time() - t0
            if el >= budget_s or cnt >= 512:
                break
        out = {"value": cnt / el, "unit": "RHS-applies/s", "cores": 1, "kind": "reference",
               "sample": f"{cnt} single-RHS HIF::solve calls of the reference itself (oracle/_ref, g++ -O2, 1 thread) "
                         f"on the same hierarchy in {el:.1f} s; host factorization took {t_fac:.1f} s",
               "ms_per_rhs": 1e3 * el / cnt, "factorize_s": t_fac}
        # the same workload on ALL host cores: the reference's prec_solve is serial and its handle is not
        # thread-safe, so the column-parallel figure comes from the C restatement (parity-locked to the reference),
        # one right-hand side per thread
        try:
            T = max(1, min(os.cpu_count() or 1, 64))
            O = orc.Oracle(levels)
            B = np.ascontiguousarray(rng.uniform(-1, 1, size=(n, T)))
            O.solve_batch(B, threads=T)
            reps, t0 = 0, time.time()
            while True:
                O.solve_batch(B, threads=T)
                reps += 1
                el2 = time.time() - t0
                if el2 >= budget_s / 2 or reps >= 64:
                    break
            out["all_cores"] = {"value": reps * T / el2, "unit": "RHS-applies/s", "cores": T, "kind": "port",
                                "sample": f"{reps} x {T} columns of the C restatement (oracle/liborc.so), one column per "
                                          f"OpenMP thread, in {el2:.1f} s"}
            O.close()
        except Exception as e:  # the headline baseline above stays valid
            out["all_cores"] = {"error": str(e)[:200]}
        return out, levels
    return None, None


def port_baseline(levels, n, budget_s):
    """cpu_baseline fallback / cross-check: the C restatement (oracle/liborc.so) on the same hierarchy."""
    from oracle import orc

    O = orc.Oracle(levels)
    b = np.random.default_rng(1).uniform(-1, 1, n)
    O.solve(b)
    cnt, t0 = 0, time.time()
    while True:
        O.solve(b)
        cnt += 1
        el = time.time() - t0
        if el >= budget_s or cnt >= 512:
            break
    return {"value": cnt / el, "unit": "RHS-applies/s", "cores": 1, "kind": "port",
            "sample": f"{cnt} single-RHS solves of the C restatement (oracle/liborc.so) in {el:.1f} s",
            "ms_per_rhs": 1e3 * el / cnt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nx", type=int, default=1000, help="grid side; n = nx^2 rows (1000 -> the 1M-row headline)")
    ap.add_argument("--nrhs", type=int, default=64)
    ap.add_argument("--params", choices=list(PARAM_SETS), default="default",
                    help="reference factorization parameters: DEFAULT_PARAMS or the PDE-tuned set "
                         "(examples/advanced/demo_gmreshif.cpp:63-65)")
    ap.add_argument("--secondary", type=int, default=1, help="also time the other parameter set (N=1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="budget of each CPU baseline sample")
    ap.add_argument("--extras", type=int, default=1, help="0: skip the side measurements (exact engine mode, 128-column "
                                                          "batch, nrhs = 1) -- counter passes want the timed kernels only")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to "
                                                      "rehearse the N > 1 flow on a box with fewer GPUs than ranks)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with matching values "
                         f"(python bench.py --gpus N starts its own ranks)")
    if world > 1 and "HIFIR_AMD_THREADS" not in os.environ:
        # every rank analyses the (replicated) hierarchy on the host (std::thread pool of the library): share the cores
        os.environ["HIFIR_AMD_THREADS"] = str(max(1, (os.cpu_count() or 8) // world))
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the hifir_amd apply path has no CPU fallback")
    if args.backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but only {ndev} GPU(s) visible")
    local_rank = local_rank % ndev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(local_rank)
    import hifir_amd
    from hifir_amd import dist as hd

    barrier = hd.barrier

    A = poisson2d(args.nx)
    n = A.shape[0]

    def get_hierarchy(pname, want_cpu):
        """-> (M, levels or None, cpu): rank 0 factorizes on the host (compiled reference) and imports the factors;
        the other ranks replay rank 0's hierarchy file (hifamd_save / hifamd_load)."""
        path = hier_cache_path(args.nx, pname)
        cpu, levels, M = None, None, None
        if rank == 0:
            cpu, levels = cpu_baseline_leg(A, pname, args.cpu_seconds if want_cpu else 0.0)
            if levels is not None:
                M = hifir_amd.HIF.from_levels(levels, max_nrhs=args.nrhs, device=local_rank)
            elif os.path.exists(path):
                M = hifir_amd.HIF.load(path, max_nrhs=args.nrhs, device=local_rank)
            else:
                raise SystemExit("no compiled reference (oracle/_ref) and no hierarchy file " + path +
                                 ": cannot build the workload's factors on this machine")
        M = hd.share_hierarchy(M, path, max_nrhs=args.nrhs, device=local_rank)
        return M, levels, cpu

    def timed(M, B, X, steps, warmup):
        for _ in range(warmup):
            M.solve_mrhs(B, X)
        M.sync()
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            M.solve_mrhs(B, X)
        M.sync()
        torch.cuda.synchronize()
        barrier()
        el = time.perf_counter() - t0
        return hd.max_over_ranks(el, device="cuda" if args.backend == "nccl" else "cpu")

    def run(pname, want_cpu, steps, warmup):
        M, levels, cpu = get_hierarchy(pname, want_cpu)
        g = torch.Generator(device="cuda")
        g.manual_seed(20260101 + rank)
        B = (torch.rand((n, args.nrhs), dtype=torch.float64, device="cuda", generator=g) * 2 - 1)
        X = torch.empty_like(B)
        el = timed(M, B, X, steps, warmup)
        # kernel-side duration of one apply: HIP events on the stream the kernels run on
        dev_ms = M.time_apply(B, X, warmup=1, reps=max(5, steps // 2))
        balg = M.algorithmic_bytes(args.nrhs)
        stage_bytes = M.stage_bytes(args.nrhs)
        # which level / stage each launch of the apply belongs to, and B_alg level by level (tests/prof_summarize.py
        # attributes the kernel trace with them); set-up cost and resident explicit operators
        launch_map = [16 * l + s_ for (l, s_) in M.launch_map()]
        level_bytes = {str(l): {str(k): v for k, v in d_.items()} for l, d_ in M.level_bytes(args.nrhs).items()}
        setup = M.stats_ext()
        # (N > 1: how many levels each rank took from the analysis trailer of rank 0's hierarchy file)
        ranks_cached = None
        if world > 1:
            ranks_cached = [None] * world
            dist.all_gather_object(ranks_cached, int(setup.get("analysis_cached_levels", 0) or 0))
        # BASELINE configs[1]: the same hierarchy with ONE right-hand side (latency-bound; reported, not the metric)
        b1 = B[:, :1].contiguous()
        x1 = torch.empty_like(b1)
        nrhs1_ms = M.time_apply(b1, x1, warmup=1, reps=5) if (world == 1 and args.extras) else None
        # narrow batches run the component bands in 16-column slices (k_band_cs) and the operator products on the column
        # tiles they have: 8 columns = what each of 8 RHS-sharded GPUs gets of one 64-column batch; 16 = BASELINE configs[4]
        narrow = None
        if world == 1 and args.extras:
            narrow = {}
            for k in (8, 16, 32):
                bk = B[:, :k].contiguous()
                xk = torch.empty_like(bk)
                narrow[str(k)] = M.time_apply(bk, xk, warmup=1, reps=5)
                del bk, xk
        st = M.stats()
        # one end-of-batch gather of the solution blocks (not in the per-step data path)
        gather_ms = None
        if world > 1:
            torch.cuda.synchronize()
            barrier()
            t1 = time.perf_counter()
            Xall = hd.gather_blocks(X)
            torch.cuda.synchronize()
            gather_ms = 1e3 * (time.perf_counter() - t1)
            del Xall
        # STRONG split of ONE 64-column batch: rank r applies columns column_block(64, r, N) of the same block
        strong = None
        if world > 1:
            c0, c1 = hd.column_block(args.nrhs, rank, world)
            gs = torch.Generator(device="cuda")
            gs.manual_seed(20260101)  # the same block on every rank
            Bs = (torch.rand((n, args.nrhs), dtype=torch.float64, device="cuda", generator=gs) * 2 - 1)[:, c0:c1].contiguous()
            Xs = torch.empty_like(Bs)
            els = timed(M, Bs, Xs, steps, 1)
            strong = {"nrhs_total": args.nrhs, "columns_per_gpu": c1 - c0, "ms_per_batch": 1e3 * els / steps,
                      "rhs_applies_per_s": args.nrhs * steps / els,
                      "note": "one 64-column batch split over the ranks: a shard of fewer than 49 columns runs the "
                              "16-column-slice kernels (narrow_batches_ms at N=1: ~2.7 ms for 8 or 16 columns against ~3.9 ms "
                              "for 64 on the 1M-row default hierarchy), so the split buys at most ~1.45x -- the levels above "
                              "the finest are bound by dependent launches, not by bytes"}
            del Bs, Xs
        # parity spot check of what was timed: column 0 against the oracle restatement (rank 0)
        parity = None
        xo = None
        if rank == 0 and want_cpu and levels is not None:
            from oracle import orc

            xo = orc.Oracle(levels).solve(B[:, 0].cpu().numpy())
            xg = X[:, 0].cpu().numpy()
            parity = float(np.abs(xg - xo).max() / np.abs(xo).max())
        # the EXACT engine mode (HIFIR_AMD_DENSE_BLOCK=0: reference summation order everywhere, thin runs on one
        # workgroup) on the same hierarchy: its time next to the default (fast) mode's, and bit-exactness of column 0
        exact = None
        if rank == 0 and want_cpu and levels is not None and args.extras:
            os.environ["HIFIR_AMD_DENSE_BLOCK"] = "0"
            try:
                Me = hifir_amd.HIF.from_levels(levels, max_nrhs=args.nrhs, device=local_rank)
            finally:
                os.environ.pop("HIFIR_AMD_DENSE_BLOCK", None)
            Xe = torch.empty_like(B)
            ems = Me.time_apply(B, Xe, warmup=1, reps=max(3, steps // 4))
            xe = Xe[:, 0].cpu().numpy()
            exact = {"ms_per_apply": ems, "col0_bit_exact_vs_oracle": bool(np.array_equal(xe, xo)),
                     "col0_relerr_vs_oracle": float(np.abs(xe - xo).max() / np.abs(xo).max()),
                     "roofline_frac": balg / (ems * 1e-3) / 1e9 / HBM_PEAK_GBS}
            Me.close()
            del Xe
        # a 128-column batch through the same handle: its two 64-column tiles run on two lanes (second work
        # arena + stream, shared matrices) and hide each other's latency-bound phases; reported next to the
        # metric, never as the metric
        pipelined = None
        if world == 1 and want_cpu and args.extras:
            B2 = torch.cat([B, B.flip(1)], dim=1).contiguous()
            X2 = torch.empty_like(B2)
            M.solve_mrhs(B2, X2)
            M.sync()
            t1 = time.perf_counter()
            for _ in range(steps):
                M.solve_mrhs(B2, X2)
            M.sync()
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t1
            pipelined = {"nrhs": 2 * args.nrhs, "tiles_in_flight": 2, "rhs_applies_per_s": 2 * args.nrhs * steps / el2,
                         "ms_per_batch": 1e3 * el2 / steps,
                         "roofline_frac": M.algorithmic_bytes(2 * args.nrhs) / (el2 / steps) / 1e9 / HBM_PEAK_GBS,
                         "first_tile_equals_64_column_result": bool(torch.equal(X2[:, :args.nrhs], X))}
            del B2, X2
            # ... and 256 columns (four tiles over the same two lanes; four lanes measured no better)
            B4 = torch.cat([B, B.flip(1), B, B.flip(1)], dim=1).contiguous()
            X4 = torch.empty_like(B4)
            M.solve_mrhs(B4, X4)
            M.sync()
            t1 = time.perf_counter()
            for _ in range(max(3, steps // 2)):
                M.solve_mrhs(B4, X4)
            M.sync()
            torch.cuda.synchronize()
            el4 = (time.perf_counter() - t1) / max(3, steps // 2)
            pipelined["nrhs_256"] = {"nrhs": 4 * args.nrhs, "ms_per_batch": 1e3 * el4, "rhs_applies_per_s": 4 * args.nrhs / el4,
                                     "roofline_frac": M.algorithmic_bytes(4 * args.nrhs) / el4 / 1e9 / HBM_PEAK_GBS,
                                     "last_tile_equals_the_second": bool(torch.equal(X4[:, 3 * args.nrhs:], X4[:, args.nrhs:2 * args.nrhs]))}
            del B4, X4
        res = dict(ms_per_step=1e3 * el / steps, value=world * args.nrhs * steps / el, dev_ms=dev_ms, balg=balg, stage_bytes=stage_bytes, nrhs1_ms=nrhs1_ms, pipelined=pipelined,
                   launch_map=launch_map, level_bytes=level_bytes, setup=setup, narrow=narrow, ranks_cached=ranks_cached,
                   stats=st, cpu=cpu, gather_ms=gather_ms, parity=parity, levels=levels, strong=strong, exact=exact)
        M.close()
        del B, X
        torch.cuda.empty_cache()
        return res

    want_cpu = (world == 1)
    r = run(args.params, want_cpu, args.steps, args.warmup)
    if rank == 0:
        st = r["stats"]
        achieved = r["balg"] / (r["dev_ms"] * 1e-3) / 1e9
        cpu = r["cpu"]
        if want_cpu and r["levels"] is not None:
            port = port_baseline(r["levels"], n, min(args.cpu_seconds, 4.0))
            if cpu is None:
                cpu = port
            else:
                cpu["port_value"] = port["value"]
        # HBM traffic of one apply from the committed PMC profile of this same command (separate
        # rocprofv3 --pmc passes, corrected as MI355X_MICROARCH.md prescribes; tests/prof_summarize.py)
        traffic, traffic_src, traffic_note, stages = None, None, None, None
        if args.params == "default" and args.nx == 1000 and args.nrhs == 64:
            import glob

            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
            if cands:
                summ = json.load(open(cands[-1]))
                if summ.get("lib_sha256") and summ.get("lib_sha256") == lib_fingerprint():
                    traffic = summ.get("hbm_bytes_per_apply_corrected")
                    traffic_src = os.path.relpath(cands[-1], ROOT)
                    # per-stage rooflines: algorithmic bytes of a stage group / its kernels' time in the committed
                    # rocprofv3 kernel trace of this same command and build (tests/prof_summarize.py)
                    sp_ = cands[-1].replace("_pmc_summary.json", "_stage_roofline.json")
                    if os.path.exists(sp_):
                        st_ = json.load(open(sp_))
                        if st_.get("lib_sha256") == lib_fingerprint():
                            stages = st_
                else:
                    traffic_note = (f"{os.path.relpath(cands[-1], ROOT)} was collected with another build of "
                                    f"libhifir_amd.so (sha256 {str(summ.get('lib_sha256'))[:12]} vs {str(lib_fingerprint())[:12]}): dropped")
        line = {
            "metric": "preconditioner applies/sec + achieved HBM GB/s, 1M-row 5-pt Laplacian, nrhs=64",
            "value": r["value"], "unit": "RHS-applies/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"2-D 5-pt Poisson {args.nx}x{args.nx} (n={n}), multilevel HIFIR hierarchy factorized on "
                                   f"the host by the reference with {args.params} parameters, batched apply X=M^-1 B, "
                                   f"nrhs={args.nrhs} per GPU, B/X resident in HBM",
                       "params": args.params, "nrhs_per_gpu": args.nrhs, "levels": int(st["sparse_levels"]),
                       "dense_n": int(st["dense_n"]), "sum_n": int(st["sum_n"]), "sum_m": int(st["sum_m"]),
                       "nnz_LU": int(st["nnz_LU"]), "nnz_EF": int(st["nnz_EF"]),
                       "wavefronts_L": int(st["wavefronts_L"]), "wavefronts_U": int(st["wavefronts_U"]),
                       "launches_per_apply": int(st["launches"]) if st["launches"] else None,
                       "parallelism": f"rhs-sharded x{world} (hierarchy replicated)",
                       # set-up cost (once per hierarchy) and what stays resident beside the factors
                       "analysis_s": r["setup"].get("analysis_s"), "analysis_levels_from_file": r["setup"].get("analysis_cached_levels"),
                       "finalize_s": r["setup"].get("finalize_s"), "graph_capture_ms": r["setup"].get("graph_capture_ms"),
                       "operator_bytes": {"block_inverses": r["setup"].get("bytes_inverses"), "top_operators": r["setup"].get("bytes_top"),
                                          "tail_operator": r["setup"].get("bytes_tail")},
                       "tail_operator": {"rows": r["setup"].get("tail_rows"), "level": r["setup"].get("tail_level"),
                                         "probe_relerr": r["setup"].get("tail_probe_relerr"),
                                         "max_abs": r["setup"].get("tail_max_abs"), "rejected": r["setup"].get("tail_rejected")}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_note": traffic_note,
                         "kernel": "one whole batched apply (hipGraph of k_band_ct (levels >= 1) / k_band_cd (level 0) / k_top_gemm / k_top_reduce / k_trsv_wide / k_spmm_tile(4) / k_spmm_epi / k_scatter_scale_list; S1, S5 (level 0) and S7 fused into the component bands)",
                         "algorithmic_bytes": r["balg"], "apply_ms_hip_events": r["dev_ms"],
                         "algorithmic_bytes_by_stage": r["stage_bytes"], "algorithmic_bytes_by_level": r["level_bytes"],
                         "launch_map": r["launch_map"], "levels_from_profile": stages},
            "cpu_baseline": cpu if world == 1 else None,  # timed on rank 0 at N = 1 only
            "parity_relerr_col0_vs_oracle": r["parity"],
        }
        if r["gather_ms"] is not None:
            line["gather_ms"] = r["gather_ms"]
        if r.get("ranks_cached") is not None:
            line["config"]["analysis_levels_from_file_by_rank"] = r["ranks_cached"]
        if r["strong"] is not None:
            line["strong_scaling"] = r["strong"]
        if r["exact"] is not None:
            line["exact_mode"] = r["exact"]
        if r["pipelined"] is not None:
            line["wide_batch"] = r["pipelined"]
        if r["nrhs1_ms"] is not None:
            line["nrhs1"] = {"config": "same hierarchy, nrhs=1 (BASELINE configs[1])", "ms_per_apply": r["nrhs1_ms"],
                             "applies_per_s": 1e3 / r["nrhs1_ms"]}
        if r["narrow"] is not None:
            line["narrow_batches_ms"] = r["narrow"]
            # what the STRONG split of one 64-column batch over 8 GPUs would cost per batch (each GPU: 8 columns), from
            # this GPU's own 8-column time -- the hierarchy is replicated, the shards are independent
            line["strong_split_8_gpus_estimate"] = {"columns_per_gpu": 8, "ms_per_batch": r["narrow"]["8"],
                                                    "speedup_vs_64_columns_on_one_gpu": r["dev_ms"] / r["narrow"]["8"],
                                                    "note": "measured on ONE GPU with an 8-column batch; not a multi-GPU measurement"}
        if world == 1 and args.secondary:
            other = "tuned" if args.params == "default" else "default"
            r2 = run(other, True, max(5, args.steps // 2), 2)
            a2 = r2["balg"] / (r2["dev_ms"] * 1e-3) / 1e9
            line["secondary"] = {"params": other, "value": r2["value"], "ms_per_step": r2["ms_per_step"],
                                 "roofline_achieved_GBs": a2, "roofline_frac": a2 / HBM_PEAK_GBS,
                                 "algorithmic_bytes": r2["balg"], "cpu_baseline": r2["cpu"],
                                 "parity_relerr_col0_vs_oracle": r2["parity"], "exact_mode": r2["exact"]}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
