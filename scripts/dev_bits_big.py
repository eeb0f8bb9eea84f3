import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
for name, ev, nch in (("dv-rel_2.uai", False, 32768), ("dv-rel_2.uai", False, 65536), ("dv-rel_3.uai", False, 32768), ("Promedus_17.uai", False, 32768), ("Promedus_11.uai", False, 32768), ("Promedus_14.uai", False, 32768), ("Pedigree_11.uai", True, 32768)):
    m = g.Model.NewModelFromFile(os.path.join(RES, name), ev)
    n = m.n_vars
    row = []
    for form in (0, 11):
        try:
            s = g.HIPChains(m, nch, seed=7, fast_form=form)
            sw = max(20, int(3e9 / (nch * n)))
            s.BurnIn(5 * n); s.Advance(sw * n)
            i = s.info()
            h = int((s.counts()[0].astype(np.uint64) * np.arange(1, s.counts().shape[1] + 1, dtype=np.uint64)).sum() % 1000003)
            row.append(f"form {form} (path {i['path']}, lanes {i['block_threads']//64}, lds {i['lds_bytes']}): {nch * sw * n / (i['kernel_ms_last'] * 1e-3):.3e} hash {h}")
            s.close()
        except g.GrError as e:
            row.append(f"form {form}: {str(e)[:70]}")
    print(f"{name} ev={int(ev)} chains={nch}: " + " | ".join(row), flush=True)
