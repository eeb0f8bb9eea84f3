"""Seed-to-seed noise of the merged marginals, CPU oracle vs GPU forms: python scripts/dev_noise.py MODEL CHAINS BURN_SW SWEEPS"""
import itertools, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
from oracle import pyoracle as O
from tests import uai_py
RES = os.path.join(ROOT, "tests", "golden", "res")
name, chains, bsw, sw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
O.build()
pm = uai_py.load(RES, name)
n = len(pm.card)
sol = uai_py.load_mar(RES, name)
m = g.Model.NewModelFromFile(os.path.join(RES, name))
seeds = [11, 12, 13, 14, 15, 16]
def cpu(seed):
    return O.run_chains(O.Model.from_parsed(pm), seed, chains, bsw * n, sw * n)[0]
def gpu(seed, **kw):
    s = g.HIPChains(m, chains, seed=seed, **kw)
    s.BurnIn(bsw * n); s.Advance(sw * n)
    return g.MergeChains([s])[0]
runs = {"cpu": [cpu(s) for s in seeds], "gpu": [gpu(s) for s in seeds], "gpu_strict": [gpu(s, flags=g.GR_FLAG_SCAN_PER_CHAIN) for s in seeds],
        "gpu_chain_per_lane": [gpu(s, fast_form=1) for s in seeds]}
for k, rs in runs.items():
    hs = [g.NewErrorSuite(pm.card, a, b) for a, b in itertools.combinations(rs, 2)]
    vm = [g.NewErrorSuite(pm.card, sol, a)["MeanHellinger"] for a in rs]
    print(f"{k:20s} pairwise mean Hellinger {np.mean([h['MeanHellinger'] for h in hs]):.4f} max {np.mean([h['MaxHellinger'] for h in hs]):.4f} | vs MAR {np.mean(vm):.4f} +- {np.std(vm):.4f}")
for a, b in (("cpu", "gpu"), ("cpu", "gpu_strict"), ("gpu", "gpu_strict"), ("gpu", "gpu_chain_per_lane")):
    hs = [g.NewErrorSuite(pm.card, x, y) for x in runs[a] for y in runs[b]]
    print(f"{a} x {b}: mean Hellinger {np.mean([h['MeanHellinger'] for h in hs]):.4f} max {np.mean([h['MaxHellinger'] for h in hs]):.4f}")
