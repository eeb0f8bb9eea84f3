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


# ---------------------------------------------------------------- Adapt over several devices / processes
def adapt_plan(model, fixed, merged, collapsed, firsts, seconds, cw, actives, capacities, new_chain_count):
    """ConvergenceSampler.Adapt's decisions (sampler/adaptive.go:57-157) for chain groups that live on several
    samplers.  `merged` / `collapsed`: the MergeChains result over all of them; `firsts` / `seconds`: the
    half-window histograms of every running group ([groups, H], any order); `actives` / `capacities`: running
    groups and group slots per sampler.  Returns (list of (variable, sampler index), convergence scores or None):
    every target goes to the sampler with the fewest running groups that still has a free slot (lowest index on
    ties).  Pure host logic: identical inputs give identical plans on every rank."""
    from . import api
    actives, capacities = [int(a) for a in actives], [int(c) for c in capacities]
    if sum(actives) < 2:
        raise api.GrError("At least 2 chains required for adaptation")
    slots = min(int(new_chain_count), sum(capacities) - sum(actives))
    if slots <= 0:
        return [], None  # MaxChains reached (adaptive.go:62-64)
    conv = None
    try:
        targets = api.AdaptSelect(model, collapsed, None, slots)
    except api.GrError:  # more candidates than slots: rank by ChainConvergence over the running groups
        if firsts is None or len(firsts) == 0:
            raise api.GrError("adapt needs a preceding AdvanceChain / AdvanceWindow")
        conv = api.ChainConvergence(model.cards(), fixed, collapsed, firsts, seconds, merged, cw)
        targets = api.AdaptSelect(model, collapsed, conv, slots)
    plan, act = [], list(actives)
    for var in targets:
        free = [i for i in range(len(act)) if act[i] < capacities[i]]
        owner = min(free, key=lambda i: (act[i], i))
        act[owner] += 1
        plan.append((int(var), owner))
    return plan, conv


def adapt_multi(samplers, model, new_chain_count):
    """Adapt over several samplers of ONE process (one per device): merge, rank, activate.  Returns the plan."""
    from . import api
    merged, collapsed = api.MergeChains(samplers)
    merged = np.concatenate(merged)
    firsts, seconds, cw = [], [], 2.0
    for s in samplers:
        try:
            a, b, cw = s.window_halves()
        except api.GrError:
            continue
        n = s.active_groups()
        firsts.append(a[:n]); seconds.append(b[:n])
    firsts = np.concatenate(firsts) if firsts else None
    seconds = np.concatenate(seconds) if seconds else None
    plan, conv = adapt_plan(model, model.fixed(), merged, collapsed, firsts, seconds, cw,
                            [s.active_groups() for s in samplers], [s.n_variants for s in samplers], new_chain_count)
    for var, owner in plan:
        samplers[owner].ActivateGroup(var)
    return plan, conv


def adapt_distributed(sampler, model, new_chain_count, group=None, device="cpu"):
    """Adapt with one sampler per rank (torch.distributed; RCCL for "nccl", gloo in the CPU tests).  Exchanges: the
    MergeChains all-reduces, one all-gather of (running groups, slots) and one of the half-window histograms.
    `sampler` needs merge_local(), window_halves(), active_groups(), n_variants, ActivateGroup(var).
    Every rank computes the same plan and activates its own share; returns (plan, scores)."""
    cards = np.asarray(model.cards())
    H = int(cards.sum())
    s, cv, cm = sampler.merge_local()
    merged, collapsed = allreduce_merge(s, cv, cm, cards, group=group, device=device)
    on = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    try:
        a, b, cw = sampler.window_halves()
        have = 1
    except Exception:
        a = b = np.zeros((sampler.n_variants, H)); cw, have = 2.0, 0
    n_act = int(sampler.active_groups())
    meta = torch.tensor([n_act, int(sampler.n_variants), have, int(round(cw * 4))], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    if on:
        dist.all_gather(metas, meta, group=group)
    else:
        metas = [meta]
    metas = [m.cpu().numpy() for m in metas]
    cap_max = int(max(m[1] for m in metas))
    pack = np.zeros((2, cap_max, H))
    pack[0, :n_act] = a[:n_act]; pack[1, :n_act] = b[:n_act]
    t = torch.as_tensor(pack, device=device)
    packs = [torch.zeros_like(t) for _ in range(world)]
    if on:
        dist.all_gather(packs, t, group=group)
    else:
        packs = [t]
    firsts = np.concatenate([p.cpu().numpy()[0, : int(m[0])] for p, m in zip(packs, metas)])
    seconds = np.concatenate([p.cpu().numpy()[1, : int(m[0])] for p, m in zip(packs, metas)])
    if not all(int(m[2]) for m in metas):
        firsts = seconds = None
    plan, conv = adapt_plan(model, model.fixed(), merged, collapsed, firsts, seconds, max(int(m[3]) for m in metas) / 4.0,
                            [int(m[0]) for m in metas], [int(m[1]) for m in metas], new_chain_count)
    for var, owner in plan:
        if owner == rank:
            sampler.ActivateGroup(var)
    return plan, conv
