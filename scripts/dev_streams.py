"""Hypothesis test: does the headline launch (1 024 equal blocks = exactly 4 per CU) lose time to slow CUs / XCDs?  The same 65 536
chains as K samplers of 65 536 / K chains on K streams (K host threads; the shards are independent), against one sampler."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
m = g.Model.NewModelFromFile(os.path.join(RES, "Grids_13.uai"))
TOTAL, UPD = 65536, 100000
for K in (1, 2, 4, 8, 16):
    per = TOTAL // K
    ss = [g.HIPChains(m, per, seed=7, chain_offset=k * per, fast_form=2) for k in range(K)]
    for s in ss:
        s.BurnIn(1000); s.Advance(UPD)
    best = None
    for rep in range(3):
        bar = threading.Barrier(K + 1)
        def work(s):
            bar.wait(); s.Advance(UPD); s.Advance(UPD)
        th = [threading.Thread(target=work, args=(s,)) for s in ss]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"K={K:2d} samplers x {per} chains: {TOTAL * 2 * UPD / best:.3e} upd/s wall ({best * 1e3:.2f} ms for 2 x {UPD} updates per chain)", flush=True)
    for s in ss: s.close()
