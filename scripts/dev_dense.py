"""Experiment: 8 lanes per chain with <= 64 registers (8 wavefronts per SIMD, fast_form 10) against the 4-lane form at the headline size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
import numpy as np
RES = os.path.join(ROOT, "tests", "golden", "res")
m = g.Model.NewModelFromFile(os.path.join(RES, "Grids_13.uai"))
ref = None
for nch in (32768, 65536, 131072):
    for form in (3, 10, 8):
        s = g.HIPChains(m, nch, seed=7, fast_form=form)
        s.BurnIn(1000); s.Advance(100 * 1000); s.Advance(100 * 1000)
        i = s.info()
        h = int(s.counts().astype(np.uint64).sum() % 1000003), int((s.counts()[0] * np.arange(1, 201, dtype=np.uint64)).sum() % 1000003)
        print(f"Grids_13 chains={nch} form={form} lanes={i['block_threads']//64}: {nch * 100000 / (i['kernel_ms_last'] * 1e-3):.3e} upd/s hash {h}", flush=True)
