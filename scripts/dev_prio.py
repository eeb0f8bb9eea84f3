"""wave_prio.h: the rotation period per kernel family.  GR_DEV_PRIO_SHIFT=<n> overrides every default (32 = no rotation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
CASES = [("Grids_13.uai", 65536, 0, 1, False), ("Grids_11.uai", 4096, 0, 1, False), ("Grids_13.uai", 65536, g.GR_FLAG_SCAN_SYSTEMATIC, 1, False), ("Grids_13.uai", 65536, g.GR_FLAG_SCAN_PER_CHAIN, 1, False),
         ("Grids_13.uai", 262144, 0, 1, False), ("Segmentation_11.uai", 32768, 0, 128, True), ("ObjectDetection_13.uai", 32768, 0, 1, False), ("CSP_13.uai", 32768, 0, 1, False),
         ("Pedigree_11.uai", 32768, 0, 1, False), ("Promedus_14.uai", 32768, 0, 1, False), ("dv-rel_2.uai", 32768, 0, 1, False)]
row = []
for name, nch, flags, groups, coll in CASES:
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    s = g.HIPChains(m, nch, n_variants=groups, seed=7, flags=flags)
    if coll:
        for k in range(groups): s.Collapse(k, -1)
    sweeps = max(20, min(2000, int(4e9 / (nch * n))))
    s.BurnIn(n * 10); s.Advance(n * sweeps)
    i = s.info()
    row.append(f"{name.split('.')[0]}@{nch}{'/sys' if flags == g.GR_FLAG_SCAN_SYSTEMATIC else '/strict' if flags else ''}{'/coll' if coll else ''} p{i['path']}: {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e}")
    s.close()
print(f"shift={os.environ.get('GR_DEV_PRIO_SHIFT', 'default')}: " + " | ".join(row), flush=True)
