"""Grids_13, the headline kernel (4 lanes per chain) by chain count: does the rate at 65 536 chains (exactly 4 resident blocks per CU) lose to block-time imbalance?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
m = g.Model.NewModelFromFile(os.path.join(RES, "Grids_13.uai"))
for nch in (32768, 49152, 65536, 66560, 81920, 98304, 131072, 196608, 262144):
    for form in (2,):
        s = g.HIPChains(m, nch, seed=7, fast_form=form)
        s.BurnIn(1000); s.Advance(100000); s.Advance(100000)
        i = s.info()
        print(f"chains={nch} blocks={nch // 64} ({nch / 64 / 1024:.2f} x 1024) form={form} path {i['path']} lanes {i['block_threads'] // 64}: {nch * 100000 / (i['kernel_ms_last'] * 1e-3):.3e} upd/s  kernel {i['kernel_ms_last']:.2f} ms in {i['n_launches_last']} launches", flush=True)
        s.close()
