"""Grids_13 / Segmentation_11 timing per kernel form: python scripts/dev_form.py FORM [FORM ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, nch, sweeps in (("Grids_13.uai", 65536, 1000), ("Grids_13.uai", 32768, 1000), ("Segmentation_11.uai", 32768, 300)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    for form in [int(a) for a in sys.argv[1:]]:
        s = g.HIPChains(m, nch, fast_form=form)
        s.BurnIn(m.n_vars * 10)
        s.Advance(m.n_vars * sweeps)
        s.Advance(m.n_vars * sweeps)
        i = s.info()
        print(f"{name} chains={nch} form={form} path={i['path']}: {i['kernel_ms_last']:.3f} ms -> {nch * m.n_vars * sweeps / (i['kernel_ms_last'] * 1e-3):.4e} updates/s", flush=True)
