import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import grample_amd as g
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests/golden/res/Grids_13.uai"))
s = g.HIPChains(m, 65536, seed=7, fast_form=2)
s.BurnIn(1000)
for n in (50000, 50000, 100000, 100000, 50000, 25000, 200000):
    s.Advance(n); i = s.info()
    print(n, "launches", i["n_launches_last"], "kernel ms %.2f" % i["kernel_ms_last"], "rate %.3e" % (65536 * n / (i["kernel_ms_last"] * 1e-3)), flush=True)
