"""RHS-sharded multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL on
ROCm, "gloo" on CPU for the tests).  The apply path shards over right-hand-side columns only: every
rank holds the whole hierarchy and applies it to its own column block; there is NO collective in
the data path.  What is shared between ranks:

  share_levels   rank 0's host-side factors -> every rank (through a file on the node + barrier)
  max_over_ranks the bench's timing convention (MAX over ranks)
  gather_blocks  ONE all_gather of the per-rank solution blocks at the end of a batch
"""
import os

import numpy as np

LEVEL_KEYS = ["m", "n", "dense_n", "dense_rank", "d", "s", "t", "p", "p_inv", "q", "q_inv", "dense"] + [
    f"{a}_{b}" for a in "LUEF" for b in ("colptr", "rowind", "vals")]


def save_levels(path, levels):
    d = {"nlevels": len(levels)}
    for l, lv in enumerate(levels):
        for k, v in lv.items():
            d[f"L{l}_{k}"] = np.asarray(v)
    tmp = path + f".tmp{os.getpid()}.npz"
    np.savez(tmp, **d)
    os.replace(tmp, path)  # atomic: readers never see a partial file


def load_levels(path):
    z = np.load(path)
    levels = []
    for l in range(int(z["nlevels"])):
        lv = {}
        for k in LEVEL_KEYS:
            if f"L{l}_{k}" in z.files:
                v = z[f"L{l}_{k}"]
                lv[k] = int(v) if v.ndim == 0 else v
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
    """levels: the hierarchy on rank 0 (None elsewhere).  Returns it on every rank."""
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
