"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards over bootstrap replicates (src/mSWEEP.cpp:496-518: B independent solves on
the same likelihood).  Every rank holds the likelihood, solves a contiguous slice of the ONE
sequential replicate stream (so the result does not depend on the number of GPUs) and the
per-replicate abundances are exchanged with a single all-gather at the end -- the only
collective; xGMI bandwidth is irrelevant at (B/P) x G doubles per rank.
"""
import numpy as np


def replicate_slice(n_replicates, rank, world):
    """Contiguous, balanced slice [begin, end) of the replicate stream for `rank`."""
    base, extra = divmod(int(n_replicates), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def all_gather_rows(local_rows, n_total, dist=None, device=None):
    """All-gather the per-rank (n_local x G) blocks into the (n_total x G) matrix in replicate
    order (the layout of BootstrapSample::bootstrap_results, include/Sample.hpp:157).  Ranks may
    hold different numbers of rows; blocks are padded to the largest for the collective."""
    local_rows = np.ascontiguousarray(local_rows, np.float64)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        assert local_rows.shape[0] == n_total
        return local_rows
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    G = local_rows.shape[1]
    slices = [replicate_slice(n_total, r, world) for r in range(world)]
    nmax = max(e - b for b, e in slices)
    pad = np.zeros((nmax, G))
    pad[:local_rows.shape[0]] = local_rows
    t = torch.from_numpy(pad)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    full = np.empty((n_total, G))
    for r, (b, e) in enumerate(slices):
        full[b:e] = out[r][:e - b].cpu().numpy()
    return full


def bootstrap_sharded(solve_slice, n_replicates, n_groups, dist=None, device=None):
    """Runs `solve_slice(begin, end) -> (end-begin) x G` on this rank's slice and all-gathers."""
    if dist is not None and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    b, e = replicate_slice(n_replicates, rank, world)
    local = solve_slice(b, e) if e > b else np.zeros((0, n_groups))
    return all_gather_rows(np.asarray(local).reshape(e - b, n_groups), n_replicates, dist, device)


def shard_ecs(rowptr, n_shards):
    """Contiguous EC blocks balanced by cell count for the EC-sharded single solve: returns
    n_shards + 1 EC boundaries ('contiguous EC blocks balanced by nnz')."""
    rp = np.asarray(rowptr, dtype=np.int64)
    E = len(rp) - 1
    # weight = cells + 1 per EC so that empty ECs are spread too
    w = rp[1:] - rp[:-1] + 1
    cw = np.concatenate([[0], np.cumsum(w)])
    targets = cw[-1] * np.arange(1, n_shards) / n_shards
    cuts = np.searchsorted(cw, targets, side="left")
    b = np.concatenate([[0], np.minimum(cuts, E), [E]]).astype(np.int64)
    return np.maximum.accumulate(b)


def csr_block(prob, e0, e1):
    """The ECs [e0, e1) of a CSR-of-ECs problem dict as a problem of their own."""
    rp = np.asarray(prob["rowptr"], dtype=np.int64)
    k0, k1 = rp[e0], rp[e1]
    out = dict(prob)
    out["rowptr"] = (rp[e0:e1 + 1] - k0).astype(np.uint64)
    out["grp"] = prob["grp"][k0:k1]
    out["cnt"] = prob["cnt"][k0:k1]
    out["ec_counts"] = prob["ec_counts"][e0:e1]
    return out
