"""4-lane form at small chain counts (<= 4 096 chains run as four 64-lane blocks per 64-chain group: FastParams::par_spread)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, nch in (("Grids_11.uai", 4096), ("Grids_13.uai", 8192), ("Grids_13.uai", 16384), ("Grids_13.uai", 32768), ("Segmentation_11.uai", 4096), ("Segmentation_11.uai", 16384), ("Promedus_14.uai", 8192)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    s = g.HIPChains(m, nch, seed=7)
    s.BurnIn(n * 10)
    sweeps = 65536 * 300 // nch
    s.Advance(n * sweeps)
    i = s.info()
    c = np.asarray(s.counts())
    print(f"{name:22s} chains={nch:6d} path={i['path']} lds={i['lds_bytes']}: {i['kernel_ms_last']:.2f} ms -> {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s  sha {hashlib.sha1(c.tobytes()).hexdigest()[:10]}", flush=True)
