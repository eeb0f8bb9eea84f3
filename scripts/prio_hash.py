"""Counts / state hashes of a few sweeps -- one per kernel family -- for tests/test_gpu_parity.py::test_results_do_not_depend_on_wavefront_priority,
which runs this with GR_DEV_PRIO_SHIFT = 32 (no rotation), 0 (every trip) and unset (the defaults), and with GR_DEV_PLAN_AHEAD = 0
(no plan-ahead on the second stream), and compares the output."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import grample_amd as g
RES = os.path.join(ROOT, "tests", "golden", "res")
SYS = g.GR_FLAG_SCAN_SYSTEMATIC
CASES = [("Grids_13.uai", 65536, 0, 1), ("Grids_13.uai", 8192, 0, 1), ("Grids_13.uai", 2048, 0, 1), ("Grids_13.uai", 8192, SYS, 1), ("Grids_13.uai", 4096, g.GR_FLAG_SCAN_PER_CHAIN, 1), ("Grids_13.uai", 65536, g.GR_FLAG_CHAIN_PER_LANE, 1),
         ("Segmentation_11.uai", 4096, 0, 16), ("ObjectDetection_13.uai", 2048, 0, 1), ("CSP_13.uai", 2048, 0, 1), ("Promedus_14.uai", 2048, SYS, 1), ("ObjectDetection_13.uai", 1024, g.GR_FLAG_DIRECT | g.GR_FLAG_CHAIN_PER_LANE, 1)]
for name, nch, flags, groups in CASES:
    m = g.Model.NewModelFromFile(os.path.join(RES, name))
    s = g.HIPChains(m, nch, n_variants=groups, seed=11, flags=flags)
    if groups > 1:
        for k in range(1, groups): s.Collapse(k, -1)
    s.BurnIn(37); s.Advance(3 * m.n_vars + 5); s.Advance(64); s.Advance(64); s.Advance(640)  # (equal calls in a row: the plan computed ahead is used; then not)
    print(name, nch, flags, groups, s.info()["path"], zlib.crc32(s.counts().tobytes()), zlib.crc32(s.checkpoint()["states"].tobytes()))
    s.close()
