"""Lanes per chain of the step-parallel memoised sweep: 4 with every other choice automatic (fast_form 3), 8 (fast_form 8), 16 (fast_form 9), auto (0),
by model and chain count.  (Forms 8 / 9 switch the LDS-layout heuristics of form 0 off, so compare 3 with 0.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
CASES = [("Grids_11.uai", 4096, 1), ("Grids_13.uai", 8192, 1), ("Grids_13.uai", 16384, 1), ("Grids_13.uai", 32768, 1), ("Grids_13.uai", 65536, 1),
         ("Segmentation_11.uai", 4096, 1), ("Segmentation_11.uai", 32768, 1), ("Segmentation_11.uai", 32768, 128), ("Segmentation_11.uai", 65536, 128),
         ("Promedus_14.uai", 32768, 1), ("Segmentation_11.uai", 512, 2)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0].startswith(sys.argv[1])]
for name, nch, groups in CASES:
    ev = name.startswith("Promedus")
    m = g.Model.NewModelFromFile(os.path.join(RES, name), ev)
    n = m.n_vars
    row = []
    for form in (3, 8, 9, 0):
        try:
            s = g.HIPChains(m, nch, n_variants=groups, seed=7, fast_form=form)
            if groups > 1:
                for k in range(groups):
                    s.Collapse(k, -1)
            sweeps = max(50, min(2000, int(3e9 / (nch * n))))
            s.BurnIn(n * 10)
            s.Advance(n * sweeps)
            i = s.info()
            row.append(f"form{form}: {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} (lanes {i['block_threads'] // 64}, lds {i['lds_bytes']})")
            s.close()
        except g.GrError as e:
            row.append(f"form{form}: {str(e)[:40]}")
    print(f"{name} chains={nch} groups={groups}: " + " | ".join(row), flush=True)
