"""AdvanceChain (the reference's stopping rule + exact ChainHistory halves) vs plain Advance of the same length: wall time and
kernel time per model.  python scripts/dev_advance_chain.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, nch, cw in (("Grids_13.uai", 65536, 1000), ("Segmentation_11.uai", 32768, 400), ("ObjectDetection_13.uai", 32768, 400),
                      ("CSP_13.uai", 32768, 200), ("dv-rel_3.uai", 32768, 50), ("Alchemy_11.uai", 32768, 50)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    s = g.HIPChains(m, nch, seed=42)
    s.BurnIn(m.n_vars * 10)
    s.AdvanceChain(cw)  # (the first window also allocates the draw log)
    t0 = time.perf_counter(); L = s.AdvanceChain(cw); t1 = time.perf_counter()
    i = s.info()
    t2 = time.perf_counter(); s.Advance(L); t3 = time.perf_counter()
    j = s.info()
    print(f"{name:24s} chains={nch} cw={cw} path={i['path']}: window {L} updates/chain | AdvanceChain wall {1e3*(t1-t0):8.1f} ms (kernels {i['kernel_ms_last']:8.1f} ms) "
          f"{nch*L/(t1-t0):.3e} updates/s | Advance wall {1e3*(t3-t2):8.1f} ms (kernels {j['kernel_ms_last']:8.1f} ms) {nch*L/(t3-t2):.3e} updates/s", flush=True)
