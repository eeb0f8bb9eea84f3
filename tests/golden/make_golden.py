#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json from the CPU oracle (oracle/grample_oracle.c).

The reference is Go and cannot run here, and it holds no golden sample trajectory, so
these vectors are DERIVED from the restatement + canonical MT19937-64, not captured from Go.  They pin
the oracle against silent changes and record the numbers SURVEY.md 8(c) quotes from an independent
std::mt19937_64 computation (first Float64 draws of seed 42; 255/769 split of 1024 one.uai draws).

    python tests/golden/make_golden.py        # rewrites oracle_vectors.json
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as O  # noqa: E402
from tests import uai_py  # noqa: E402

RES = os.path.join(ROOT, "tests", "golden", "res")


def ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def build():
    L = O.lib()
    out = {"_comment": "derived from oracle/grample_oracle.c + canonical MT19937-64; not captured from Go"}
    g = L.orc_gen_new(42)
    out["mt_seed42_int63"] = [int(L.orc_int63(g)) for _ in range(5)]
    L.orc_gen_free(g)
    g = L.orc_gen_new(42)
    out["mt_seed42_float64"] = [repr(L.orc_float64(g)) for _ in range(3)]
    # 1024 WeightedSample-style draws on one.uai WITHOUT the start-state draw: value 0 iff f <= 0.25
    L.orc_gen_free(g)
    g = L.orc_gen_new(42)
    c0 = sum(1 for _ in range(1024) if L.orc_float64(g) * 1.0 <= 0.25)
    out["one_uai_seed42_1024_draws_no_init"] = [c0, 1024 - c0]
    L.orc_gen_free(g)
    # the real sampler on one.uai, seed 42 (1 Int31n for the start state, then 1 Float64 per Sample)
    m = O.Model.from_parsed(uai_py.load(RES, "one.uai"))
    g = L.orc_gen_new(42)
    s = L.orc_simple_new(g, m.ptr)
    one = np.zeros(1, dtype=np.int32)
    traj = []
    for _ in range(1024):
        L.orc_simple_sample(s, ip(one), 1)
        traj.append(int(one[0]))
    out["one_uai_seed42_gibbs_counts"] = [traj.count(0), traj.count(1)]
    out["one_uai_seed42_gibbs_first32"] = traj[:32]
    L.orc_simple_free(s); L.orc_gen_free(g)
    # multi-chain runs (cmd/root.go flow) on the small fixtures
    for name, nch, burn, upd in (("sample.uai", 4, 100, 1000), ("deterministic.uai", 2, 50, 500), ("Grids_11.uai", 2, 200, 2000)):
        pm = uai_py.load(RES, name)
        marg, _, _, tot = O.run_chains(O.Model.from_parsed(pm), 42, nch, burn, upd)
        out[f"run_{name}"] = {"chains": nch, "burn_in": burn, "updates": upd, "total": tot,
                              "merged_marginal": [[repr(float(x)) for x in v] for v in marg]}
    # Collapse known answers
    for name in ("sample.uai", "deterministic.uai"):
        pm = uai_py.load(RES, name)
        res = {}
        for v in range(len(pm.card)):
            mm = O.Model.from_parsed(pm)
            g = L.orc_gen_new(42)
            c = L.orc_collapsed_new(g, mm.ptr)
            if L.orc_collapsed_collapse(c, v) == v:
                res[str(v)] = {"marginal": [repr(float(x)) for x in mm.marginal(v)],
                               "new_factor_vars": mm.func_vars(mm.nfuncs - 1),
                               "new_factor_logtable": [repr(float(x)) for x in mm.func_table(mm.nfuncs - 1)]}
            L.orc_collapsed_free(c); L.orc_gen_free(g)
        out[f"collapse_{name}"] = res
    return out


if __name__ == "__main__":
    O.build()
    with open(os.path.join(ROOT, "tests", "golden", "oracle_vectors.json"), "w") as f:
        json.dump(build(), f, indent=1)
    print("wrote tests/golden/oracle_vectors.json")
