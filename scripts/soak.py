"""Soak: two samplers with the same seed at the headline size, N x 100 000 updates each in different chunkings; counts and states
must be identical at the end (determinism over ~1e13 updates, no device error on the way).  python scripts/soak.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
m = g.Model.NewModelFromFile(os.path.join(RES, "Grids_13.uai"))
a = g.HIPChains(m, 65536, seed=5)
b = g.HIPChains(m, 65536, seed=5)
t0 = time.time()
for i in range(N):
    a.Advance(100000)
    b.Advance(33333); b.Advance(66667)
    if i % 10 == 9:
        print(f"{i + 1} x 100000 updates per chain, {time.time() - t0:.1f} s, rate {2 * 65536 * 100000 * (i + 1) / (time.time() - t0):.3e}", flush=True)
assert np.array_equal(a.counts(), b.counts()) and np.array_equal(a.checkpoint()["states"], b.checkpoint()["states"])
print("soak ok:", int(a.counts().sum()), "counted updates per sampler, identical counts and states")
