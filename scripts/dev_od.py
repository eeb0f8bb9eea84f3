"""ObjectDetection_13 timing: value-parallel form (fast_form=2) vs chain per lane (fast_form=1) at several chain counts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
m = g.Model.NewModelFromFile(os.path.join(RES, "ObjectDetection_13.uai"))
for nch in (8192, 32768, 65536, 131072, 262144):
    for form in (2, 1):
        s = g.HIPChains(m, nch, fast_form=form)
        sweeps = max(20, 32768 * 100 // nch)
        s.BurnIn(60 * 10)
        s.Advance(60 * sweeps)
        i = s.info()
        print(f"OD13 chains={nch} form={form} path={i['path']} lds={i['lds_bytes']}: {i['kernel_ms_last']:.2f} ms -> {nch * 60 * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s", flush=True)
