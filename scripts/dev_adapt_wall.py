"""Adaptive run on Segmentation_11 (config 3'): wall time of the Adapt calls vs kernel time of the windows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
m = g.Model.NewModelFromFile(os.path.join(RES, "Segmentation_11.uai"))
n = m.n_vars
t00 = time.time()
s = g.HIPChains(m, 32768, n_variants=128, seed=42)
s.set_active_groups(2)
s.BurnIn(200 * n)
t_create = time.time() - t00
kms, t_adapt, t_win, n_act = 0.0, 0.0, 0.0, 0
for r in range(40):
    t0 = time.time()
    s.AdvanceWindow(max(2, 1000 * n // 40))
    t_win += time.time() - t0
    kms += s.info()["kernel_ms_last"]
    if r < 20:
        t0 = time.time()
        got = s.Adapt(8)
        t_adapt += time.time() - t0
print(f"create+burn-in {t_create*1e3:.0f} ms; windows wall {t_win*1e3:.0f} ms (kernel {kms:.0f} ms); Adapt wall {t_adapt*1e3:.0f} ms for {s.active_groups() - 2} activations "
      f"({t_adapt*1e3/max(1, s.active_groups()-2):.1f} ms each)")
