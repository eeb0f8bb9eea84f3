"""Cross-rank half of MergeChains (sampler/chain.go:96-148) for one-process-per-GPU
runs: chains are sharded over ranks with no data-path collective; the only
exchange is one all-reduce of the per-variable histograms (RCCL over xGMI when
the backend is "nccl", gloo in the CPU tests).

Works on what gr_merge_local() returns on every rank:
  local_sum      [H]  sum over this rank's chains of (1/Card prior + counts) for
                      variables not collapsed in the chain's group
  coll_variant   [N]  lowest GLOBAL variant id on this rank that collapsed the variable, -1 if none
  coll_marginal  [H]  that variant's exact marginal
"""
import numpy as np
import torch
import torch.distributed as dist

_NONE = np.iinfo(np.int64).max


def allreduce_merge(local_sum, coll_variant, coll_marginal, cards, group=None, device="cpu"):
    """Returns (merged un-normalised marginals as a flat [H] array, collapsed flags [N]) -- identical on all ranks."""
    cards = np.asarray(cards)
    H = int(cards.sum())
    s = torch.as_tensor(np.asarray(local_sum, dtype=np.float64), device=device).clone()
    cv = np.asarray(coll_variant, dtype=np.int64)
    owner = torch.as_tensor(np.where(cv < 0, _NONE, cv), device=device).clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)          # the histogram all-reduce
        dist.all_reduce(owner, op=dist.ReduceOp.MIN, group=group)      # "first collapsed chain wins" (chain.go:113-120)
    owner_np = owner.cpu().numpy()
    # the winning rank contributes its exact marginal, everyone else zeros; one more sum spreads it
    mine = np.zeros(H)
    off = np.concatenate([[0], np.cumsum(cards)])
    for v in range(len(cards)):
        if owner_np[v] != _NONE and cv[v] == owner_np[v]:
            mine[off[v]:off[v + 1]] = np.asarray(coll_marginal)[off[v]:off[v + 1]]
    m = torch.as_tensor(mine, device=device).clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
    out = s.cpu().numpy()
    m_np = m.cpu().numpy()
    collapsed = (owner_np != _NONE).astype(np.int32)
    for v in range(len(cards)):
        if collapsed[v]:
            out[off[v]:off[v + 1]] = m_np[off[v]:off[v + 1]]
    return out, collapsed


def allreduce_counts(counts, group=None, device="cpu"):
    """Sum of raw uint64 count arrays over ranks (sent as int64: the values are < 2^63)."""
    t = torch.as_tensor(np.asarray(counts).astype(np.int64), device=device).clone()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy().astype(np.uint64)
