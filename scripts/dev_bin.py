"""Chain-per-lane memoised form (k_sweep_bin) at large chain counts: timing + counts hash."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, nch, sweeps in (("Grids_13.uai", 131072, 500), ("Grids_13.uai", 262144, 400), ("Grids_13.uai", 1 << 20, 200), ("Segmentation_11.uai", 262144, 100), ("Grids_13.uai", 65536, 500)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    s = g.HIPChains(m, nch, seed=7, fast_form=1)
    s.BurnIn(n * 10 + 3)
    s.Advance(n * sweeps)
    i = s.info()
    c = np.asarray(s.counts())
    print(f"{name:22s} chains={nch:8d} path={i['path']} bt={i['block_threads']} lds={i['lds_bytes']}: {i['kernel_ms_last']:.2f} ms -> {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s  sha {hashlib.sha1(c.tobytes()).hexdigest()[:10]}", flush=True)
    s.close()
