"""Every fixture model at N chains: kernel path and updates/s (kernel time).  python scripts/dev_all_models.py [CHAINS]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for name in sorted(f for f in os.listdir(RES) if f.endswith(".uai")):
    for ev in (False, True):
        try:
            m = g.Model.NewModelFromFile(os.path.join(RES, name), ev)
            if ev and (m.fixed() < 0).all():
                continue
            s = g.HIPChains(m, nch)
            n = m.n_vars
            sweeps = max(20, 40000 // n)
            s.BurnIn(n * 2)
            s.Advance(n * sweeps)
            i = s.info()
            print(f"{name:26s} ev={int(ev)} N={n:5d} path={i['path']} bt={i['block_threads']:4d} lds={i['lds_bytes']:6d} tab={i['tables_in_lds']} {i['kernel_ms_last']:8.2f} ms {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s", flush=True)
            s.close()
        except Exception as e:
            print(f"{name:26s} ev={int(ev)} ERROR {e}", flush=True)
