"""One ObjectDetection_13 run for profiling: python scripts/dev_od_one.py CHAINS FORM SWEEPS"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
nch, form, sweeps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
name = sys.argv[4] if len(sys.argv) > 4 else "ObjectDetection_13.uai"
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests", "golden", "res", name))
s = g.HIPChains(m, nch, fast_form=form)
s.BurnIn(m.n_vars * 2)
s.Advance(m.n_vars * sweeps)
i = s.info()
print(f"{name} chains={nch} form={form} path={i['path']}: {i['kernel_ms_last']:.3f} ms, {nch * m.n_vars * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s, steps={m.n_vars * sweeps}")
