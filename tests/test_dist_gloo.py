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
