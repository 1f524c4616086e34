"""RHS-sharded multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" on CPU for the tests).  The apply path shards over right-hand-side columns only: every
rank holds the whole hierarchy and applies it to its own column block; there is NO collective in
the data path.  What is shared between ranks:

  share_hierarchy rank 0's imported hierarchy -> every rank, through the library's on-disk format (hifamd_save /
                 hifamd_load) on the node + two barriers; share_levels: the same for the dict form (npz), tests
  max_over_ranks the bench's timing convention (MAX over ranks)
  gather_blocks  ONE all_gather of the per-rank solution blocks at the end of a batch
"""
import os

import numpy as np

def save_levels(path, levels):
    """Every field of every level (the dict layout of tests/util.py / oracle.orc) into one npz."""
    d = {"nlevels": len(levels)}
    for l, lv in enumerate(levels):
        for k, v in lv.items():
            d[f"L{l}_{k}"] = np.asarray(v)
    tmp = path + f".tmp{os.getpid()}.npz"
    np.savez(tmp, **d)
    os.replace(tmp, path)  # atomic: readers never see a partial file


def load_levels(path):
    """Inverse of save_levels.  No key filter: whatever a level carries (dense_symm, spd, dense_lup, ...) comes
    back, so every rank builds the SAME last-level solver."""
    z = np.load(path)
    levels = []
    for l in range(int(z["nlevels"])):
        lv = {}
        pre = f"L{l}_"
        for key in z.files:
            if key.startswith(pre):
                v = z[key]
                lv[key[len(pre):]] = int(v) if (v.ndim == 0 and v.dtype.kind in "iub") else (float(v) if v.ndim == 0 else v)
        levels.append(lv)
    return levels


def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


def barrier():
    d = _dist()
    if d is not None:
        d.barrier()


def share_levels(levels, path):
    """levels: the hierarchy on rank 0 (None elsewhere).  Returns it on every rank (npz hand-off; tests)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return levels
    if d.get_rank() == 0:
        save_levels(path, levels)
    d.barrier()
    if d.get_rank() != 0:
        levels = load_levels(path)
    d.barrier()
    return levels


def share_hierarchy(M, path, max_nrhs=64, device=-1):
    """The product hand-off: rank 0 holds the imported hierarchy `M` (hifir_amd.HIF; None elsewhere), writes it in
    the library's own on-disk format (hifamd_save_ex -- exactly the add_level / set_dense arguments, any last-level
    kind, plus the host analysis of every level so that the other ranks do not repeat it) and every other rank replays
    the file (hifamd_load + finalize on ITS device).  No collective carries matrix data: the file lives on the node,
    ranks only meet at two barriers."""
    from .hif import HIF

    d = _dist()
    if d is None or d.get_world_size() == 1:
        return M
    if d.get_rank() == 0:
        tmp = path + f".tmp{os.getpid()}"
        M.save(tmp, analysis=True)
        os.replace(tmp, path)
    d.barrier()
    if d.get_rank() != 0:
        M = HIF.load(path, max_nrhs=max_nrhs, device=device)
    d.barrier()
    return M


def max_over_ranks(seconds, device="cpu"):
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return float(seconds)
    import torch

    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


def column_block(nrhs_total, rank, world):
    """[c0, c1) of a batch of nrhs_total columns owned by `rank` (strong-scaling split)."""
    base, rem = divmod(nrhs_total, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


def gather_blocks(X):
    """all_gather of equally shaped [n][nrhs] blocks; returns the [n][world*nrhs] block on every rank."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return X
    import torch

    if d.get_backend() != "nccl" and X.is_cuda:  # rehearsal backends: stage through the host
        Xh = X.cpu()
        out = [torch.empty_like(Xh) for _ in range(d.get_world_size())]
        d.all_gather(out, Xh.contiguous())
        return torch.cat(out, dim=1).to(X.device)
    out = [torch.empty_like(X) for _ in range(d.get_world_size())]
    d.all_gather(out, X.contiguous())
    return torch.cat(out, dim=1)
