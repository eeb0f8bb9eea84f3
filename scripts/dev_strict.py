"""Strict per-chain scan (GR_FLAG_SCAN_PER_CHAIN): timing + counts hash."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
for name, nch in (("Grids_13.uai", 65536), ("Segmentation_11.uai", 32768), ("Grids_13.uai", 262144)):
    m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests", "golden", "res", name))
    s = g.HIPChains(m, nch, seed=7, flags=g.GR_FLAG_SCAN_PER_CHAIN)
    n = m.n_vars
    s.BurnIn(10 * n); s.Advance(300 * n)
    i = s.info()
    rate = nch * n * 300 / (i["kernel_ms_last"] * 1e-3)
    print(f"{name:22s} chains={nch:7d} path={i['path']} {rate:.3e} updates/s  sha {hashlib.sha1(np.asarray(s.counts()).tobytes()).hexdigest()[:10]}", flush=True)
