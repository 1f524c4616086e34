"""CPU, world_size 2, gloo: the N > 1 plumbing of the RHS-sharded path (hifir_amd/dist.py) --
factor sharing through a node-local file, column ownership, MAX-over-ranks timing and the one
end-of-batch all_gather.  The per-rank apply itself needs a GPU; here the oracle stands in for it,
which is exactly what makes the test able to check the gathered block column by column."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmpdir, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    from hifir_amd import dist as hd
    from oracle import orc
    from util import load_hier

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        levels, d = (load_hier("p2d_64_deep") if rank == 0 else (None, None))
        levels = hd.share_levels(levels, os.path.join(tmpdir, "hier.npz"))
        assert levels is not None and len(levels) == 3
        n = int(levels[0]["n"])
        nrhs = 3
        rng = np.random.default_rng(100 + rank)  # every rank owns its own block (weak scaling)
        B = rng.uniform(-1, 1, size=(n, nrhs))
        X = orc.Oracle(levels).solve_batch(B)
        G = hd.gather_blocks(torch.from_numpy(X)).numpy()
        assert G.shape == (n, world * nrhs)
        # every rank can verify every other rank's block
        for r in range(world):
            Br = np.random.default_rng(100 + r).uniform(-1, 1, size=(n, nrhs))
            assert np.array_equal(G[:, r * nrhs:(r + 1) * nrhs], orc.Oracle(levels).solve_batch(Br))
        # hierarchies whose last level is NOT the QRCP default (symmetric SYEIG, LUP) and a complex one: every rank
        # must rebuild the same solver kind -- rank 1 checks its copy against the golden x of the compiled reference
        for name in ("p2d_32_symm", "p2d_30_lup", "young1c"):
            lv0, d0 = load_hier(name)  # (every rank reads the fixture only to have the expected vectors)
            lv = hd.share_levels(lv0 if rank == 0 else None, os.path.join(tmpdir, f"{name}.npz"))
            assert all(set(a) == set(b) for a, b in zip(lv, lv0)), name
            x = orc.Oracle(lv).solve(d0["b"])
            assert np.abs(x - d0["x"]).max() <= 1e-12 * np.abs(d0["x"]).max(), name
            # the product hand-off: the library's own on-disk format (hifamd_save -> hifamd_load); without a GPU the
            # loaded handle is not finalized, but what it holds can be saved again: same bytes as rank 0 wrote
            import hifir_amd

            path = os.path.join(tmpdir, f"{name}.hifamd")
            M0 = None
            if rank == 0:
                M0 = hifir_amd.HIF(dtype=np.complex128 if name == "young1c" else np.float64)
                for l in lv0:
                    M0.add_level(l)
                last = lv0[-1]
                if int(last.get("dense_lup", 0)):
                    M0.set_dense_lup(last["dense"])
                elif int(last.get("dense_symm", 0)):
                    M0.set_dense_symm(last["dense"], int(last.get("spd", 0)))
                else:
                    M0.set_dense(last["dense"])
            M = hd.share_hierarchy(M0, path, max_nrhs=0)
            assert M.levels() == len(lv0) + 1 and M.schur_rank() == int(lv0[-1]["dense_rank"]), name
            again = path + f".again{rank}"
            M.save(again, analysis=True)  # (the hand-off carries the host analysis: the other rank adopted it)
            assert open(again, "rb").read() == open(path, "rb").read(), name
            if rank != 0:
                assert int(M.stats_ext()["analysis_cached_levels"]) == len(lv0), name
        t = hd.max_over_ranks(1.0 + rank)
        assert t == float(world)
        c = [hd.column_block(64, r, world) for r in range(world)]
        assert c[0][0] == 0 and c[-1][1] == 64 and all(c[i][1] == c[i + 1][0] for i in range(world - 1))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_rhs_sharding_world2_gloo():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=300) for _ in procs]
        for p in procs:
            p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_column_block_partition():
    sys.path.insert(0, ROOT)
    from hifir_amd.dist import column_block

    for total in (1, 7, 64, 100):
        for world in (1, 2, 3, 8):
            blocks = [column_block(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(b[1] >= b[0] for b in blocks)
            assert sum(b[1] - b[0] for b in blocks) == total
