"""GR_FLAG_SCAN_SYSTEMATIC throughput: chromatic kernel (path 9) by model and chain count, next to the default random scan."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, nch in [("Grids_11.uai", 4096), ("Grids_13.uai", 4096), ("Grids_13.uai", 16384), ("Grids_13.uai", 65536), ("Grids_13.uai", 262144), ("Segmentation_11.uai", 32768), ("ObjectDetection_13.uai", 32768),
                  ("CSP_13.uai", 32768), ("Alchemy_11.uai", 32768), ("Pedigree_11.uai", 32768), ("Promedus_14.uai", 32768)][int(os.environ.get("DEV_SYS_FROM", "0")):]:
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    n = m.n_vars
    row = []
    for flags, tag in ((g.GR_FLAG_SCAN_SYSTEMATIC, "systematic"), (0, "random scan")):
        s = g.HIPChains(m, nch, seed=7, flags=flags)
        sweeps = max(50, min(4000, int(6e9 / (nch * n))))
        if s.info()["path"] == 0:
            sweeps = max(10, sweeps // 20)
        s.BurnIn(n * 10)
        s.Advance(n * sweeps)
        i = s.info()
        mg, _ = g.MergeChains([s])
        sol = g.Solution.NewSolutionFromFile(os.path.join(RES, name + ".MAR")).marginals()
        es = g.NewErrorSuite(m.cards(), sol, mg)
        row.append(f"{tag}: {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} upd/s (path {i['path']}, {i['kernel_ms_last']:.1f} ms, MeanHellinger {es['MeanHellinger']:.4f})")
        s.close()
    print(f"{name} chains={nch}: " + " | ".join(row), flush=True)
