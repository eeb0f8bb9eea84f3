"""Segmentation_11 collapsed in 128 groups: a logged AdvanceChain window vs a plain Advance of the same length."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
m = g.Model.NewModelFromFile(os.path.join(ROOT, "tests", "golden", "res", "Segmentation_11.uai"))
cw = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
s = g.HIPChains(m, 32768, n_variants=128, seed=42)
for k in range(128):
    s.Collapse(k, -1)
s.BurnIn(2000)
for rep in range(2):
    t = time.time(); L = s.AdvanceChain(cw); w1 = time.time() - t; k1 = s.info()["kernel_ms_last"]
    t = time.time(); s.Advance(L); w2 = time.time() - t; k2 = s.info()["kernel_ms_last"]
    print(f"window {L}: AdvanceChain wall {w1*1e3:.1f} ms (sweeps {k1:.1f}) | Advance wall {w2*1e3:.1f} ms (sweeps {k2:.1f}) | {32768*L/w1:.3e} vs {32768*L/w2:.3e} updates/s", flush=True)
