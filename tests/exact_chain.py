"""Exact long-run behaviour of the REFERENCE's Markov chain on tiny models.

Builds the full transition matrix of grample's random-scan Gibbs kernel
(uniform variable choice, sampler.go:135-174; floored conditional,
gibbs-simple.go:175-258) from the oracle's conditionals, solves for the
stationary law pi, and returns what Chain.oneSample's histogram converges to:

    E[ Marginal[v][val] share ] = sum_x pi(x) * p_v(val | x)

(the value written is counted only when v is the selected variable,
chain.go:230-236).  This is a deterministic target for the GPU sampler: it
includes the effect of the 1e-6 table epsilon and the 1e-6 probability floor,
which exact model marginals do not.
"""
import ctypes as C
import itertools

import numpy as np


def expected_histogram(oracle, pm, order=None):
    """order = None: the reference's random scan.  order = list of the selectable variables: the systematic scan that
    resamples them one after the other in that order (GR_FLAG_SCAN_SYSTEMATIC): stationary law of a whole sweep, then the law the
    chain has when it reaches each position of the sweep."""
    L = oracle.lib()
    om = oracle.Model.from_parsed(pm)
    gen = L.orc_gen_new(1)
    samp = L.orc_simple_new(gen, om.ptr)
    assert samp, oracle.err()
    card, fixed = list(pm.card), list(pm.fixed)
    n = len(card)
    free = [v for v in range(n) if fixed[v] < 0]
    ranges = [range(c) if f < 0 else [f] for c, f in zip(card, fixed)]
    states = list(itertools.product(*ranges))
    index = {s: i for i, s in enumerate(states)}
    S = len(states)
    assert S <= 4096
    P = np.zeros((S, S))
    cond = {}
    for i, s in enumerate(states):
        st = np.array(s, dtype=np.int32)
        for v in free:
            p = np.zeros(card[v])
            rc = L.orc_simple_conditional(samp, v, st.ctypes.data_as(C.POINTER(C.c_int)), p.ctypes.data_as(C.POINTER(C.c_double)))
            assert rc == 0
            cond[(i, v)] = p
            for val in range(card[v]):
                t = list(s)
                t[v] = val
                P[i, index[tuple(t)]] += p[val] / len(free)
    def stationary(P):  # left eigenvector for eigenvalue 1 (solve (P^T - I) pi = 0, sum pi = 1)
        A = np.vstack([P.T - np.eye(S), np.ones((1, S))])
        b = np.zeros(S + 1)
        b[-1] = 1.0
        pi = np.linalg.lstsq(A, b, rcond=None)[0]
        pi = np.clip(pi, 0, None)
        return pi / pi.sum()
    if order is None:
        pi = stationary(P)
        out = []
        for v in range(n):
            if fixed[v] >= 0:
                out.append(None)
                continue
            f = np.zeros(card[v])
            for i in range(S):
                f += pi[i] * cond[(i, v)]
            out.append(f)
    else:
        assert sorted(order) == free
        K = {}
        for v in free:  # the kernel of one update of v
            Kv = np.zeros((S, S))
            for i, s in enumerate(states):
                for val in range(card[v]):
                    t = list(s)
                    t[v] = val
                    Kv[i, index[tuple(t)]] += cond[(i, v)][val]
            K[v] = Kv
        Psweep = np.eye(S)
        for v in order:
            Psweep = Psweep @ K[v]
        pi = stationary(Psweep)
        out = [None] * n
        for v in order:
            f = np.zeros(card[v])
            for i in range(S):
                f += pi[i] * cond[(i, v)]
            out[v] = f
            pi = pi @ K[v]
    L.orc_simple_free(samp)
    L.orc_gen_free(gen)
    return out


def tiny_models():
    """Synthetic tiny models covering card 2-4, arity 1-3, zeros (epsilon rule), near-deterministic rows (floor)."""
    from types import SimpleNamespace
    rng = np.random.default_rng(7)
    out = {}
    # 3x3 binary grid, moderate couplings + one hard constraint-like table
    card = [2] * 9
    scopes, tables = [], []
    for v in range(9):
        scopes.append([v]); tables.append(np.exp(rng.uniform(-1, 1, 2)))
    for r in range(3):
        for c in range(3):
            v = 3 * r + c
            if c < 2:
                scopes.append([v, v + 1]); tables.append(np.exp(rng.uniform(-1.2, 1.2, 4)))
            if r < 2:
                scopes.append([v + 3, v]); tables.append(np.exp(rng.uniform(-1.2, 1.2, 4)))  # reversed scope order
    tables[-1] = np.array([1.0, 0.0, 0.7, 1.0])  # a zero -> table epsilon rule and floored conditionals; still mixes
    out["grid3x3"] = SimpleNamespace(card=card, scopes=scopes, tables=tables, fixed=[-1] * 9)
    # mixed cards, arity 3, a unary on every var
    card = [2, 3, 4, 2, 3]
    scopes = [[0], [1], [2], [3], [4], [0, 1, 2], [2, 3], [4, 3, 1], [1, 4]]
    tables = [np.exp(rng.uniform(-1, 1, int(np.prod([card[v] for v in s])))) for s in scopes]
    out["mixed"] = SimpleNamespace(card=card, scopes=scopes, tables=tables, fixed=[-1] * 5)
    ev = SimpleNamespace(card=card, scopes=scopes, tables=tables, fixed=[-1, -1, 2, -1, 0])
    out["mixed_evidence"] = ev
    return out
