#!/usr/bin/env python3
"""Measures every BASELINE.json config on one MI355X (config 4 as its per-GPU share of 32 768 chains):
updates/s from the library's HIP-event kernel time, algorithmic-bytes roofline fraction, and the
reference's ErrorSuite vs res/<model>.uai.MAR.  Writes one JSON object per config to stdout."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import grample_amd as g  # noqa: E402

RES = os.path.join(ROOT, "tests", "golden", "res")
# (name, model, sampler, chains, groups, burn-in sweeps, counted sweeps, algorithmic bytes per update)
CONFIGS = [
    ("configs[0] one.uai GibbsSimple 1 chain 1k iters", "one.uai", "simple", 1, 1, 0, 1000, 17.0),
    ("configs[1] Grids_11 GibbsSimple 4096 chains", "Grids_11.uai", "simple", 4096, 1, 200, 2000, 53.0),
    ("configs[2] Grids_13 GibbsSimple 65536 chains", "Grids_13.uai", "simple", 65536, 1, 200, 2000, 53.0),
    ("configs[3] Segmentation_11 GibbsCollapsed 32768 chains, 128 groups", "Segmentation_11.uai", "collapsed", 32768, 128, 200, 1000, 65.7),
    ("configs[3'] Segmentation_11 adaptive 32768 chains, up to 128 groups", "Segmentation_11.uai", "adaptive", 32768, 128, 200, 1000, 65.7),
    ("configs[4] ObjectDetection_13 GibbsSimple, per-GPU share 32768 of 262144 chains", "ObjectDetection_13.uai", "simple", 32768, 1, 200, 600, 430.5),
    ("configs[2'] Grids_13 65536 chains, strict per-chain variable selection (GR_FLAG_SCAN_PER_CHAIN)", "Grids_13.uai", "strict", 65536, 1, 200, 2000, 53.0),
]


def main():
    for title, model, samp, chains, groups, burn, sweeps, abytes in CONFIGS:
        m = g.Model.NewModelFromFile(os.path.join(RES, model))
        n = m.n_vars
        s = g.HIPChains(m, chains, n_variants=groups, seed=42, flags=g.GR_FLAG_SCAN_PER_CHAIN if samp == "strict" else 0)
        if samp == "collapsed":
            for k in range(groups):
                s.Collapse(k, -1)
        if samp == "adaptive":
            s.set_active_groups(2)
        s.BurnIn(burn * n)
        ms, updates = 0.0, 0
        if samp == "adaptive":
            rounds = 40
            for r in range(rounds):
                before = s.TotalSampleCount()
                s.AdvanceWindow(max(2, sweeps * n // rounds))
                ms += s.info()["kernel_ms_last"]
                updates += s.TotalSampleCount() - before
                if r < rounds // 2:
                    s.Adapt(8)
        else:
            s.Advance(sweeps * n)
            ms = s.info()["kernel_ms_last"]
            updates = s.TotalSampleCount()
        merged, coll = g.MergeChains([s])
        sol = g.Solution.NewSolutionFromFile(os.path.join(RES, model + ".MAR")).marginals()
        es = g.NewErrorSuite(m.cards(), sol, merged)
        rate = updates / (ms * 1e-3)
        info = s.info()
        print(json.dumps({"config": title, "updates_per_s": rate, "kernel_ms": ms, "updates": int(updates),
                          "path": info["path"], "groups_active": s.active_groups(), "collapsed_vars": int(np.sum(coll)),
                          "mean_hellinger_vs_MAR": es["MeanHellinger"], "max_hellinger_vs_MAR": es["MaxHellinger"],
                          "mean_abs_err_vs_MAR": es["MeanMeanAbsError"],
                          "roofline": {"algorithmic_bytes_per_update": abytes, "achieved_GBs": abytes * rate / 1e9,
                                       "frac_of_8TBs": abytes * rate / 8e12}}), flush=True)


if __name__ == "__main__":
    main()
