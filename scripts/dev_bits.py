"""fast_form 11 (k_sweep_bits: 4 lanes per chain, chain state bit-packed -- one 16-bit mask per variable and wavefront) against the byte layout's
k_sweep_par4 (fast_form 3: no wide form): same counts (hash), updates/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, nch in (("Grids_13.uai", 65536), ("Grids_13.uai", 131072), ("Grids_11.uai", 32768), ("Segmentation_11.uai", 32768), ("dv-rel_1.uai", 65536)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    row = []
    for form in (3, 11):
        try:
            s = g.HIPChains(m, nch, seed=7, fast_form=form)
        except g.GrError as e:
            row.append(f"form {form}: {e}"); continue
        s.BurnIn(10 * n); s.Advance(1000 * n); s.Advance(1000 * n)
        i = s.info()
        h = int((s.counts()[0].astype(np.uint64) * np.arange(1, s.counts().shape[1] + 1, dtype=np.uint64)).sum() % 1000003)
        row.append(f"form {form} (path {i['path']}, lds {i['lds_bytes']}): {nch * 1000 * n / (i['kernel_ms_last'] * 1e-3):.3e} upd/s hash {h}")
        s.close()
    print(f"{name} chains={nch}: " + " | ".join(row), flush=True)
