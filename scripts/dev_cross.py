"""Crossover between the 4-lane form (fast_form=2) and the chain-per-lane form (fast_form=1) of the memoised binary sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name in ("Grids_13.uai", "Segmentation_11.uai"):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    for nch in (65536, 98304, 131072, 196608, 262144, 524288):
        out = []
        for form in (2, 1, 0):
            s = g.HIPChains(m, nch, seed=7, fast_form=form)
            s.BurnIn(n * 10)
            sweeps = max(50, 65536 * 400 // nch)
            s.Advance(n * sweeps)
            i = s.info()
            out.append(f"form{form}: path {i['path']} {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e}")
            s.close()
        print(f"{name:22s} chains={nch:7d}  " + "   ".join(out), flush=True)
