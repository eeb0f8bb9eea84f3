"""k_sweep_dir4 (path 8) on the dense small-Card fixtures: updates/s by chain count."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
import numpy as np
RES = os.path.join(ROOT, "tests", "golden", "res")
for name in ("CSP_13.uai", "CSP_12.uai", "CSP_11.uai", "Alchemy_11.uai", "Pedigree_11.uai"):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    row = []
    for nch in (4096, 32768, 262144):
        s = g.HIPChains(m, nch, seed=7)
        sweeps = max(20, min(2000, int(3e9 / (nch * n))))
        s.BurnIn(n * 10); s.Advance(n * sweeps)
        i = s.info()
        h = int((s.counts()[0].astype(np.uint64) * np.arange(1, s.counts().shape[1] + 1, dtype=np.uint64)).sum() % 1000003)
        row.append(f"{nch}: {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} (path {i['path']}, lds {i['lds_bytes']}, hash {h})")
        s.close()
    print(f"{name}: " + " | ".join(row), flush=True)
