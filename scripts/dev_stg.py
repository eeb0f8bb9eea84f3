"""LDS-resident vs global-memory chain state on mid-size models: python scripts/dev_stg.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, ev in (("Alchemy_11.uai", False), ("CSP_13.uai", False), ("Pedigree_11.uai", False), ("Promedus_17.uai", True), ("dv-rel_3.uai", True), ("Grids_13.uai", False)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name), ev)
    n = m.n_vars
    for nch in (32768, 131072):
        for flags in (0, g.GR_FLAG_STATE_GLOBAL):
            s = g.HIPChains(m, nch, flags=flags)
            sweeps = max(2, 8000 // n)
            s.BurnIn(n * 2)
            s.Advance(n * sweeps)
            i = s.info()
            print(f"{name:22s} chains={nch:6d} flags={flags:2d} path={i['path']:2d} bt={i['block_threads']:3d} lds={i['lds_bytes']:6d}: {i['kernel_ms_last']:8.2f} ms {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s", flush=True)
            s.close()
