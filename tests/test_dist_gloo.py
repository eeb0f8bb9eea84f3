"""N>1 path on CPU: world_size-2 gloo run of the cross-rank merge (the path's only exchange step)
against a single-process restatement of MergeChains (chain.go:96-148).  No GPU needed: each rank
feeds the arrays gr_merge_local() would return for its shard of the chains."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


CARDS = [2, 3, 2, 4]
H = sum(CARDS)


def _rank_inputs(rank):
    """rank r holds variants 2r, 2r+1 with 64 chains each; variant 1 collapsed var 0, variant 2 collapsed vars 0 and 2."""
    rng = np.random.default_rng(100 + rank)
    off = np.concatenate([[0], np.cumsum(CARDS)])
    coll = {0: {}, 1: {0: np.array([0.3, 0.7])}, 2: {0: np.array([0.6, 0.4]), 2: np.array([0.1, 0.9])}, 3: {}}
    local_sum, cv, cm = np.zeros(H), np.full(len(CARDS), -1, dtype=np.int64), np.zeros(H)
    counts = {}
    for g in (2 * rank, 2 * rank + 1):
        counts[g] = rng.integers(0, 1000, H).astype(np.float64)
        for v, c in enumerate(CARDS):
            if v in coll[g]:
                if cv[v] < 0:
                    cv[v] = g
                    cm[off[v]:off[v + 1]] = coll[g][v]
                continue
            local_sum[off[v]:off[v + 1]] += 64.0 / c + counts[g][off[v]:off[v + 1]]
    return local_sum, cv, cm, counts, coll


def _expected():
    off = np.concatenate([[0], np.cumsum(CARDS)])
    allc, allcoll = {}, {}
    for r in range(2):
        _, _, _, counts, coll = _rank_inputs(r)
        allc.update(counts)
        allcoll = coll
    out = np.zeros(H)
    flags = np.zeros(len(CARDS), dtype=np.int32)
    for v, c in enumerate(CARDS):
        first = next((g for g in range(4) if v in allcoll[g]), None)
        if first is not None:  # collapsed in ANY chain: first such chain's marginal, no summation
            out[off[v]:off[v + 1]] = allcoll[first][v]
            flags[v] = 1
        else:
            for g in range(4):
                out[off[v]:off[v + 1]] += 64.0 / c + allc[g][off[v]:off[v + 1]]
    return out, flags


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grample_amd.dist import allreduce_counts, allreduce_merge
    ls, cv, cm, counts, _ = _rank_inputs(rank)
    merged, flags = allreduce_merge(ls, cv, cm, CARDS)
    raw = allreduce_counts(np.arange(H, dtype=np.uint64) * (rank + 1))
    q.put((rank, merged, flags, raw))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_merge_matches_mergechains():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp, flags = _expected()
    for rank, merged, fl, raw in res:
        assert np.allclose(merged, exp, rtol=1e-14), (rank, merged, exp)
        assert fl.tolist() == flags.tolist() == [1, 0, 1, 0]
        assert raw.tolist() == (np.arange(H) * 3).tolist()


def test_single_process_is_identity():
    from grample_amd.dist import allreduce_merge
    ls, cv, cm, _, _ = _rank_inputs(0)
    merged, flags = allreduce_merge(ls, cv, cm, CARDS)
    assert flags.tolist() == [1, 0, 0, 0]
    assert np.allclose(merged[2:], ls[2:]) and np.allclose(merged[:2], [0.3, 0.7])


# ------------------------------------------------------------- Adapt over two ranks (N2 x multi-GPU)
RES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "res")


class _FakeSampler:
    """What adapt_distributed needs from a sampler, fed with synthetic chain histories (no GPU)."""

    def __init__(self, rank, cards, n_variants, n_active):
        self.rank, self.cards, self.n_variants, self.n_active = rank, np.asarray(cards), n_variants, n_active
        self.H = int(self.cards.sum())
        self.activated = []

    def active_groups(self):
        return self.n_active

    def _hist(self, group, half):
        rng = np.random.default_rng(1000 * self.rank + 10 * group + half)
        out = np.zeros(self.H)
        off = 0
        for c in self.cards:
            p = rng.dirichlet(np.full(c, 1.0 + 4.0 * rng.random()))
            out[off:off + c] = np.rint(p * 64 * 20)
            off += c
        return out

    def window_halves(self):
        a, b = np.zeros((self.n_variants, self.H)), np.zeros((self.n_variants, self.H))
        for g in range(self.n_active):
            a[g], b[g] = self._hist(g, 0), self._hist(g, 1)
        return a, b, 40.0

    def merge_local(self):
        s = np.zeros(self.H)
        off = 0
        for c in self.cards:
            s[off:off + c] += self.n_active * 64.0 / c
            off += c
        for g in range(self.n_active):
            s += self._hist(g, 0) + self._hist(g, 1)
        return s, np.full(len(self.cards), -1, dtype=np.int64), np.zeros(self.H)

    def ActivateGroup(self, var):
        self.activated.append(var)
        self.n_active += 1
        return var


def _adapt_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import grample_amd as g
    from grample_amd.dist import adapt_distributed
    m = g.Model.NewModelFromFile(os.path.join(RES, "Segmentation_11.uai"))
    fs = _FakeSampler(rank, m.cards(), 4 + rank, 1 + rank)  # rank 0: 1 of 4 groups running, rank 1: 2 of 5
    plan, conv = adapt_distributed(fs, m, 3)
    q.put((rank, plan, fs.activated, None if conv is None else conv.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_adapt_matches_single_process_plan():
    """Both ranks compute the plan a single process computes from the union of the groups (adaptive.go:57-157),
    targets go to the rank with the fewest running groups, and every rank activates exactly its share."""
    import grample_amd as g
    from grample_amd.dist import adapt_plan
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_adapt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m = g.Model.NewModelFromFile(os.path.join(RES, "Segmentation_11.uai"))
    fakes = [_FakeSampler(0, m.cards(), 4, 1), _FakeSampler(1, m.cards(), 5, 2)]
    merged = sum(f.merge_local()[0] for f in fakes)
    firsts = np.concatenate([f.window_halves()[0][: f.n_active] for f in fakes])
    seconds = np.concatenate([f.window_halves()[1][: f.n_active] for f in fakes])
    exp_plan, exp_conv = adapt_plan(m, m.fixed(), merged, np.zeros(m.n_vars, dtype=np.int32), firsts, seconds, 40.0, [1, 2], [4, 5], 3)
    assert len(exp_plan) == 3 and [o for _, o in exp_plan] == [0, 0, 1]  # 1|2 running -> 2|2 -> 3|2 -> 3|3
    for rank, plan, activated, conv in res:
        assert [tuple(x) for x in plan] == exp_plan
        assert activated == [v for v, o in exp_plan if o == rank]
        assert np.allclose(conv, exp_conv, rtol=1e-12, equal_nan=True)
    # lowest scores are taken (adaptive.go:111-119 takes from the tail of the descending order)
    picked = [v for v, _ in exp_plan]
    elig = [v for v in range(m.n_vars) if not np.isnan(exp_conv[v])]
    assert max(exp_conv[picked]) <= np.sort(exp_conv[elig])[len(picked) - 1] + 1e-15
