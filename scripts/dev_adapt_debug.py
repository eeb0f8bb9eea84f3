import os, sys, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
from oracle import pyoracle as O
from tests import uai_py
from tests.test_gpu_adaptive import _run_gpu, _run_oracle
res = os.path.join(ROOT, "tests", "golden", "res")
name = "Segmentation_11.uai"
pm = uai_py.load(res, name); sol = uai_py.load_mar(res, name)
gm = g.Model.NewModelFromFile(os.path.join(res, name)); base = O.Model.from_parsed(pm)
S, cw = 10, int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = [_run_gpu(gm, pm, sol, 1000 + k, 1, 20, cw, 2000) for k in range(S)]
Or = [_run_oracle(O, base, pm, sol, 2000 + k, 1, 20, cw, 2000) for k in range(S)]
n = len(pm.card)
pg = np.array([[run[0]["merged"][v][1] for v in range(n)] for run in G])
po = np.array([[run[0]["merged"][v][1] for v in range(n)] for run in Or])
se = np.sqrt(pg.var(axis=0, ddof=1) / S + po.var(axis=0, ddof=1) / S)
z = (pg.mean(axis=0) - po.mean(axis=0)) / np.maximum(se, 1e-3)
for v in np.argsort(-np.abs(z))[:12]:
    print(v, "z %.1f" % z[v], "gpu %.4f +- %.4f" % (pg[:, v].mean(), pg[:, v].std()), "oracle %.4f +- %.4f" % (po[:, v].mean(), po[:, v].std()), "MAR %.4f" % (sol[v][1] / sum(sol[v])), "samples", G[0][0]["samples"] / 128, Or[0][0]["samples"] / 2)
