"""Segmentation_11 at 32 768 chains: one group / 128 groups without a collapse / 128 groups each with one Collapse(-1) -- what the collapse costs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
m = g.Model.NewModelFromFile(os.path.join(RES, "Segmentation_11.uai"))
n = m.n_vars
for groups, collapse in ((1, False), (128, False), (128, True)):
    for form in (0, 3):
        s = g.HIPChains(m, 32768, n_variants=groups, seed=42, fast_form=form)
        if collapse:
            for k in range(groups):
                s.Collapse(k, -1)
        s.BurnIn(50 * n); s.Advance(500 * n)
        i = s.info()
        print(f"groups={groups} collapsed={collapse} fast_form={form}: {32768 * 500 * n / (i['kernel_ms_last'] * 1e-3):.3e} upd/s path {i['path']} lanes {i['block_threads'] // 64} lds {i['lds_bytes']} tables_in_lds {i['tables_in_lds']}", flush=True)
        s.close()
