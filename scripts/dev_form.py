import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.dev_time import timing
name = sys.argv[1] if len(sys.argv) > 1 else "Grids_13.uai"
nch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 160
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
for form in [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0").split(",")]:
    timing(name, nch, sweeps, flags, 0, 1, form)
if os.environ.get("DEV_COLLAPSED"):
    import grample_amd as g
    from scripts.dev_time import RES
    m = g.Model.NewModelFromFile(os.path.join(RES, "Segmentation_11.uai"))
    K = 128
    s = g.HIPChains(m, 32768, n_variants=K, seed=42)
    for k in range(K):
        s.Collapse(k, -1)
    n = m.n_vars
    s.BurnIn(n * 10); s.Advance(n * 400)
    i = s.info()
    print(f"SEG collapsed 32768x128 path={i['path']}: {32768 * n * 400 / (i['kernel_ms_last'] * 1e-3):.3e} updates/s", flush=True)
