"""Developer smoke script (GPU): conditional parity, small statistical parity, first timings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
from oracle import pyoracle as O
from tests import uai_py
import ctypes as C
RES = os.path.join(ROOT, "tests", "golden", "res")

def cond_check(name, ev=False, n=200):
    pm = uai_py.load(RES, name, ev)
    om = O.Model.from_parsed(pm); L = O.lib()
    gen = L.orc_gen_new(1); os_ = L.orc_simple_new(gen, om.ptr)
    m = g.Model.NewModelFromFile(os.path.join(RES, name), ev)
    s = g.HIPChains.NewGibbsSimple(m, 64, flags=g.GR_FLAG_DIRECT)
    rng = np.random.default_rng(0)
    free = [v for v in range(len(pm.card)) if pm.fixed[v] < 0]
    worst = 0
    for _ in range(n):
        st = np.array([rng.integers(0, c) if f < 0 else f for c, f in zip(pm.card, pm.fixed)], dtype=np.int32)
        v = int(rng.choice(free))
        pd = s.conditional(0, v, st)
        po = np.zeros(pm.card[v]); L.orc_simple_conditional(os_, v, st.ctypes.data_as(C.POINTER(C.c_int)), po.ctypes.data_as(C.POINTER(C.c_double)))
        err = np.max(np.abs(pd - po) / (3e-10 + po))
        worst = max(worst, err)
    print(f"cond {name}: worst rel err {worst:.3e}  info={s.info()}")

for name, ev in [("Grids_13.uai", False), ("sample.uai", False), ("ObjectDetection_13.uai", False), ("Promedus_11.uai", True), ("Pedigree_11.uai", True), ("CSP_12.uai", False)]:
    cond_check(name, ev)

def stat(name, nch, burn, upd, flags=0):
    pm = uai_py.load(RES, name); sol = uai_py.load_mar(RES, name)
    om = O.Model.from_parsed(pm)
    t = time.time(); mo, _, _, tot = O.run_chains(om, 42, nch, burn, upd); to = time.time() - t
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    s = g.HIPChains.NewGibbsSimple(m, nch, flags=flags)
    t = time.time(); s.BurnIn(burn); s.Advance(upd); tg = time.time() - t
    mg, _ = g.MergeChains([s])
    eo = O.error_suite(pm.card, sol, mo); eg = O.error_suite(pm.card, sol, mg); d = O.error_suite(pm.card, mo, mg)
    print(f"stat {name} chains={nch}: oracle {to:.2f}s gpu {tg:.3f}s | vsMAR oracle H={eo['MeanHellinger']:.4f}/{eo['MaxHellinger']:.4f} gpu H={eg['MeanHellinger']:.4f}/{eg['MaxHellinger']:.4f} | gpu-vs-oracle meanH={d['MeanHellinger']:.5f} maxH={d['MaxHellinger']:.5f} info={s.info()}")

stat("one.uai", 64, 100, 10000)
stat("Grids_11.uai", 1024, 2000, 20000)
stat("Grids_11.uai", 1024, 2000, 20000, g.GR_FLAG_DIRECT)
stat("ObjectDetection_13.uai", 256, 1200, 12000)

def timing(name, nch, sweeps, flags=0, bt=0):
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    s = g.HIPChains(m, nch, flags=flags, block_threads=bt)
    n = m.n_vars
    s.BurnIn(n * 10)
    s.Advance(n * sweeps)
    i = s.info()
    ups = nch * n * sweeps / (i['kernel_ms_last'] * 1e-3)
    print(f"time {name} chains={nch} flags={flags} bt={i['block_threads']} path={i['path']} lds={i['lds_bytes']}: {i['kernel_ms_last']:.2f} ms for {sweeps} sweeps -> {ups:.3e} updates/s")

for flags in (0, g.GR_FLAG_SEQUENTIAL, g.GR_FLAG_DIRECT, g.GR_FLAG_DIRECT | g.GR_FLAG_SEQUENTIAL):
    timing("Grids_13.uai", 65536, 1000, flags)
timing("Grids_13.uai", 65536, 1000, 0, 64)
timing("Grids_13.uai", 1 << 20, 200, 0)
timing("Segmentation_11.uai", 32768, 200)
timing("ObjectDetection_13.uai", 32768, 100)
