"""Time gr_advance_chain (stop rule + draw log + ring rebuild) against gr_advance of the same length."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import grample_amd as g

res = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "res")
name, chains = (sys.argv[1] if len(sys.argv) > 1 else "Grids_13.uai"), int(sys.argv[2]) if len(sys.argv) > 2 else 65536
for cw in [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["100", "2000", "20000"])]:
    for flags in (0, g.GR_FLAG_CHAIN_PER_LANE):
        s = g.HIPChains(g.Model.NewModelFromFile(os.path.join(res, name)), chains, seed=3, flags=flags)
        s.BurnIn(1000)
        t0 = time.time(); L = s.AdvanceChain(cw); t1 = time.time()
        s.Advance(L); t2 = time.time()
        print(f"{name} chains={chains} path={s.info()['path']} cw={cw}: L={L} AdvanceChain {t1 - t0:.3f}s  Advance(L) {t2 - t1:.3f}s", flush=True)
