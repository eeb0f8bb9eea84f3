"""Collapsed Segmentation_11 (128 groups): timing of the split threshold pool form (the sweep over resident blocks per CU recorded in DESIGN.md section 3 was made with a development build)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = g.Model.NewModelFromFile(os.path.join(RES, "Segmentation_11.uai"))
s = g.HIPChains(m, nch, n_variants=128, seed=7)
for k in range(128):
    s.Collapse(k, -1)
n = m.n_vars
s.BurnIn(n * 10)
s.Advance(n * 200)
i = s.info()
c = np.asarray(s.counts())
print(f"chains={nch} path={i['path']} lds={i['lds_bytes']}: {i['kernel_ms_last']:.2f} ms -> {nch * n * 200 / (i['kernel_ms_last'] * 1e-3):.3e} updates/s  counts sha {hashlib.sha1(c.tobytes()).hexdigest()[:12]}", flush=True)
