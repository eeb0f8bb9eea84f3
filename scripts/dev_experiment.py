"""The reference's adaptive experiment (script/experiment: burn-in 2000, 2 start chains, +4 per Adapt, cwin 2000) on the
device: one chain group of 256 chains per reference chain, AdvanceChain + Adapt rounds.  Wall time per phase."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
name = sys.argv[1] if len(sys.argv) > 1 else "Segmentation_11.uai"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cw = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
form = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # fast_form (1: chain per lane = the generic kernel for the direct models)
m = g.Model.NewModelFromFile(os.path.join(RES, name), True)
n = m.n_vars
t0 = time.time()
s = g.HIPChains(m, 128 * 256, n_variants=128, seed=42, fast_form=form)
s.set_active_groups(2)
s.BurnIn(2000)
t_create = time.time() - t0
t_ac = t_adapt = t_merge = 0.0
kms = 0.0
sol = g.Solution.NewSolutionFromFile(os.path.join(RES, name + ".MAR")).marginals()
for r in range(rounds):
    t = time.time(); L = s.AdvanceChain(cw); t_ac += time.time() - t
    kms += s.info()["kernel_ms_last"]
    t = time.time(); merged, coll = g.MergeChains([s]); es = g.NewErrorSuite(m.cards(), sol, merged); t_merge += time.time() - t
    t = time.time(); s.Adapt(4); t_adapt += time.time() - t
    if r % 8 == 7 or r == rounds - 1:
        print(f"round {r + 1:3d}: groups {s.active_groups():3d} window {L} updates/chain, samples {s.TotalSampleCount():.3e}, MeanHellinger {es['MeanHellinger']:.4f} MaxHellinger {es['MaxHellinger']:.4f}", flush=True)
tot = time.time() - t0
print(f"{name} (kernel path {s.info()['path']}): total wall {tot:.2f} s = create+burn-in {t_create:.2f} + AdvanceChain {t_ac:.2f} (sweep kernels {kms / 1e3:.2f}) + merge/score {t_merge:.2f} + Adapt {t_adapt:.2f}; "
      f"{s.TotalSampleCount() / tot:.3e} counted updates/s end to end")
