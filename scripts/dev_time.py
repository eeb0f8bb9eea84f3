import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
def timing(name, nch, sweeps, flags=0, bt=0, nvar=1, form=0):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    s = g.HIPChains(m, nch, n_variants=nvar, flags=flags, block_threads=bt, fast_form=form)
    n = m.n_vars
    s.BurnIn(n * 10)
    s.Advance(n * sweeps)
    i = s.info()
    ups = nch * n * sweeps / (i['kernel_ms_last'] * 1e-3)
    print(f"time {name} chains={nch} flags={flags} bt={i['block_threads']} path={i['path']} lds={i['lds_bytes']} launches={i['n_launches_last']}: {i['kernel_ms_last']:.2f} ms for {sweeps} sweeps -> {ups:.3e} updates/s", flush=True)
if __name__ == "__main__":
    for form in (2, 1):
        for nch in (32768, 65536, 131072, 262144):
            timing("Grids_13.uai", nch, 65536 * 500 // nch, 0, 0, 1, form)
    timing("Grids_13.uai", 65536, 1000, g.GR_FLAG_SEQUENTIAL)
    timing("Grids_13.uai", 1 << 20, 200, 0)
    timing("Grids_13.uai", 1 << 18, 400, 0)
    timing("Grids_11.uai", 4096, 1000, 0)
    timing("Segmentation_11.uai", 32768, 200)
    timing("Segmentation_11.uai", 32768, 200, g.GR_FLAG_SEQUENTIAL)
    timing("ObjectDetection_13.uai", 32768, 100)
    timing("ObjectDetection_13.uai", 65536, 100)
    timing("ObjectDetection_13.uai", 262144, 50)
    timing("Grids_13.uai", 65536, 200, g.GR_FLAG_DIRECT)
    timing("CSP_12.uai", 32768, 100)
    timing("Pedigree_11.uai", 32768, 50)
    def timing_collapsed(name, nch, K, sweeps):
        m = g.Model.NewModelFromFile(os.path.join(RES, name))
        s = g.HIPChains(m, nch, n_variants=K)
        for k in range(K):
            s.Collapse(k, -1)
        n = m.n_vars
        s.BurnIn(n * 10)
        s.Advance(n * sweeps)
        i = s.info()
        print(f"time {name} COLLAPSED chains={nch} groups={K} path={i['path']} lds={i['lds_bytes']}: {i['kernel_ms_last']:.2f} ms for {sweeps} sweeps -> {nch * n * sweeps / (i['kernel_ms_last'] * 1e-3):.3e} updates/s", flush=True)
    timing_collapsed("Segmentation_11.uai", 32768, 128, 200)
    timing("Grids_13.uai", 65536, 100, g.GR_FLAG_SCAN_PER_CHAIN)
    timing("ObjectDetection_13.uai", 32768, 30, g.GR_FLAG_SCAN_PER_CHAIN)
