#!/usr/bin/env python3
"""Cross-form fuzzing on the GPU: random models of several shapes, every kernel form the library can be steered into
(default choice, chain per lane, direct, sequential, state in global memory, strict pairs), identical seeds -- counts and
final states must agree bit for bit.  python scripts/fuzz_forms.py [N_MODELS] [SEED]"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import grample_amd as g  # noqa: E402


def model(rng, kind):
    if kind == "bigcard_pairwise":      # k_sweep_pw4 territory
        n = int(rng.integers(6, 40))
        card = [int(rng.choice([2, 3, 7, 9, 10, 12, 15, 16, 16])) for _ in range(n)]
        card[int(rng.integers(0, n))] = 16 if rng.random() < 0.7 else 9
        arity = 2
    elif kind == "binary_sparse":       # memoised fast forms (par4 / bin), small blankets
        n = int(rng.integers(10, 300))
        card = [2] * n
        arity = int(rng.choice([2, 3]))
    elif kind == "binary_medium":       # blankets around the LDS / L2 table boundary
        n = int(rng.integers(40, 200))
        card = [2] * n
        arity = 4
    elif kind == "tiny_binary":         # 1..6 binary variables: sweeps shorter than a quad / a round of the wide forms
        n = int(rng.integers(1, 7))
        card = [2] * n
        arity = 2
    elif kind == "dense_small":         # k_sweep_dir4 territory: Card 2..4, blankets too large to memoise
        n = int(rng.integers(20, 90))
        card = [int(rng.choice([2, 3, 4, 4])) for _ in range(n)]
        arity = int(rng.choice([2, 3]))
    else:                               # small mixed cards, arity up to 4: generic kernels
        n = int(rng.integers(5, 60))
        card = [int(rng.choice([2, 2, 3, 4, 5])) for _ in range(n)]
        arity = int(rng.choice([2, 3, 4]))
    scopes, tables = [], []
    for v in range(n):
        scopes.append([v]); tables.append(np.exp(rng.uniform(-2, 2, card[v])))
    nf = int(n * (rng.uniform(3.0, 6.0) if kind == "dense_small" else rng.uniform(0.8, 2.5 if kind != "binary_medium" else 1.2)))
    for _ in range(nf):
        k = int(rng.integers(2, arity + 1))
        if n < 2:
            break
        sc = [int(x) for x in rng.choice(n, min(k, n), replace=False)]
        size = int(np.prod([card[v] for v in sc]))
        t = np.exp(rng.uniform(-3, 3, size))
        if rng.random() < 0.5:
            t[rng.random(size) < 0.2] = rng.choice([0.0, 1e-9, 1e-6, 2e-6, 1e-5])
        scopes.append(sc); tables.append(t)
    fixed = [int(rng.integers(0, c)) if rng.random() < 0.1 else -1 for c in card]
    if all(f >= 0 for f in fixed):
        fixed[0] = -1
    return SimpleNamespace(card=card, scopes=scopes, tables=tables, fixed=fixed)


def run(m, chains, seed, off, collapse=None, **kw):
    s = g.HIPChains(m, chains, seed=seed, chain_offset=off, **kw)
    if collapse is not None:  # the last group collapses this variable (-1: the sampler's own pick; the caller passes the pick on to the other forms)
        collapse.append(s.Collapse(kw["n_variants"] - 1, collapse.pop() if collapse else -1)[0])
    s.BurnIn(29); s.Advance(77); s.Advance(50)
    i = s.info()
    out = (i["path"] + (32 if i["tables_in_lds"] == 2 else 0), s.counts().copy(), s.checkpoint()["states"].copy())  # 35 = path 3, split pool
    s.close()
    return out


def main():
    n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    kinds = ["bigcard_pairwise", "binary_sparse", "binary_medium", "mixed", "dense_small", "tiny_binary"]
    seen = {}
    for i in range(n_models):
        kind = kinds[i % len(kinds)]
        pm = model(rng, kind)
        m = g.Model.from_arrays(pm.card, pm.scopes, pm.tables, pm.fixed)
        chains = int(rng.choice([64, 128, 320]))
        seed, off = int(rng.integers(1, 1 << 40)), 64 * int(rng.integers(0, 1000))
        ref = run(m, chains, seed, off)
        forms = {"chain_per_lane": dict(fast_form=1), "direct": dict(flags=g.GR_FLAG_DIRECT), "sequential": dict(flags=g.GR_FLAG_SEQUENTIAL),
                 "state_global": dict(flags=g.GR_FLAG_STATE_GLOBAL), "state_global_direct": dict(flags=g.GR_FLAG_STATE_GLOBAL | g.GR_FLAG_DIRECT),
                 "form2": dict(fast_form=2), "form3_no_wide": dict(fast_form=3), "form6": dict(fast_form=6), "form7_split_pool": dict(fast_form=7),
                 "form8_wide8": dict(fast_form=8), "form9_wide16": dict(fast_form=9), "form11_bit_packed_state": dict(fast_form=11)}
        paths = [ref[0]]
        for name, kw in forms.items():
            try:
                r = run(m, chains, seed, off, **kw)
            except g.GrError as e:
                if "[-3]" in str(e) or "[-1]" in str(e):
                    continue  # form not available for this model
                raise
            paths.append(r[0])
            assert np.array_equal(ref[1], r[1]), (i, kind, name, "counts")
            assert np.array_equal(ref[2], r[2]), (i, kind, name, "states")
        # strict mode: memoised vs direct
        try:
            a = run(m, chains, seed, off, flags=g.GR_FLAG_SCAN_PER_CHAIN)
            b = run(m, chains, seed, off, flags=g.GR_FLAG_SCAN_PER_CHAIN | g.GR_FLAG_DIRECT)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (i, kind, "strict")
            paths += [a[0], b[0]]
        except g.GrError:
            pass
        # systematic scan: chromatic kernel (when the model qualifies) vs the generic kernel's systematic mode
        try:
            a = run(m, chains, seed, off, flags=g.GR_FLAG_SCAN_SYSTEMATIC)
            b = run(m, chains, seed, off, flags=g.GR_FLAG_SCAN_SYSTEMATIC | g.GR_FLAG_DIRECT | g.GR_FLAG_CHAIN_PER_LANE)
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (i, kind, "systematic")
            paths += [a[0] + 100, b[0] + 100]
        except g.GrError:
            pass
        # two chain groups, the second with one variable collapsed (a factor over its whole blanket joins the model): the
        # sampler's own choice of kernel (4-lane forms with chained entries / split pools) vs the generic kernel, random and systematic scan
        if chains % 128 == 0:
            for fl, tag in ((0, 200), (g.GR_FLAG_SCAN_SYSTEMATIC, 300)):
                try:
                    cv = []
                    a = run(m, chains, seed, off, collapse=cv, n_variants=2, flags=fl)
                    b = run(m, chains, seed, off, collapse=cv, n_variants=2, flags=fl | g.GR_FLAG_DIRECT | g.GR_FLAG_CHAIN_PER_LANE)
                    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (i, kind, "collapsed", tag, cv)
                    paths += [a[0] + tag, b[0] + tag]
                except g.GrError as e:
                    if "[-5]" not in str(e) and "[-3]" not in str(e) and "[-1]" not in str(e):
                        raise
        for p in paths:
            seen[p] = seen.get(p, 0) + 1
        print(f"model {i:3d} {kind:18s} n={len(pm.card):3d} chains={chains:3d} default path {ref[0]:2d} forms {sorted(set(paths))} ok", flush=True)
    print("paths exercised:", dict(sorted(seen.items())))


if __name__ == "__main__":
    main()
