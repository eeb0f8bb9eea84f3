#!/usr/bin/env python3
"""Per-config parity table (the north_star's "estimated marginals must match grample's own CPU output ... against res/*.uai.MAR"):
for every BASELINE model the SAME small run -- same chain count, burn-in and counted updates, seed 42 -- on the CPU oracle
(oracle/, MT19937-64, f64: the restatement of grample's CPU path) and on the GPU, each scored with the reference's
ErrorSuite against res/<model>.uai.MAR, plus the distance between the two merged marginals.  The streams differ by
construction (Philox vs MT19937-64), so the comparison is between two estimators of the same law at the same
horizon; "seed42_vs_seed43" is the Hellinger distance two independent runs of that size have from each other (CPU against
CPU, GPU against GPU), and "gpu_strict_scan" is the GPU with the reference's exact per-chain variable selection
(GR_FLAG_SCAN_PER_CHAIN) instead of one selection per 64-chain group.  One JSON object per line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from tests import uai_py  # noqa: E402

RES = os.path.join(ROOT, "tests", "golden", "res")
# model, sampler ("simple" / "collapsed": Collapse(-1) in every chain resp. chain group), chains, burn-in sweeps, counted sweeps
CASES = [("one.uai", "simple", 64, 0, 1000), ("Grids_11.uai", "simple", 256, 20, 200), ("Grids_13.uai", "simple", 256, 20, 200),
         ("Segmentation_11.uai", "simple", 256, 20, 100), ("Segmentation_11.uai", "collapsed", 256, 20, 100),
         ("ObjectDetection_13.uai", "simple", 256, 50, 400)]


def main():
    O.build()
    for name, samp, chains, burn_sw, sw in CASES:
        pm = uai_py.load(RES, name)
        n = len(pm.card)
        sol = uai_py.load_mar(RES, name)
        burn, upd = burn_sw * n, sw * n
        t0 = time.time()
        groups = 4 if samp == "collapsed" else 1
        cvars = [10, 50, 100, 150]  # collapsed: chain group k collapses variable cvars[k] (the same variable in its CPU chains)
        ckw = dict(collapse_mode=2, collapse_vars=[cvars[c * groups // chains] for c in range(chains)]) if samp == "collapsed" else {}
        mo, _, _, _ = O.run_chains(O.Model.from_parsed(pm), 42, chains, burn, upd, **ckw)
        mo2, _, _, _ = O.run_chains(O.Model.from_parsed(pm), 43, chains, burn, upd, **ckw)
        t_cpu = time.time() - t0
        m = g.Model.NewModelFromFile(os.path.join(RES, name))
        s = g.HIPChains(m, chains, n_variants=groups, seed=42)
        if samp == "collapsed":
            for k in range(groups):
                s.Collapse(k, cvars[k])
        s.BurnIn(burn)
        s.Advance(upd)
        mg, _ = g.MergeChains([s])

        def gpu(seed, flags=0):
            t = g.HIPChains(m, chains, n_variants=groups, seed=seed, flags=flags)
            if samp == "collapsed":
                for k in range(groups):
                    t.Collapse(k, cvars[k])
            t.BurnIn(burn)
            t.Advance(upd)
            return g.MergeChains([t])[0]
        mg2, mgs = gpu(43), gpu(42, g.GR_FLAG_SCAN_PER_CHAIN)
        e_cpu, e_gpu = g.NewErrorSuite(pm.card, sol, mo), g.NewErrorSuite(pm.card, sol, mg)
        e_x, e_self = g.NewErrorSuite(pm.card, mo, mg), g.NewErrorSuite(pm.card, mo, mo2)
        e_gself, e_strict = g.NewErrorSuite(pm.card, mg, mg2), g.NewErrorSuite(pm.card, mo, mgs)
        print(json.dumps({"model": name, "sampler": samp, "chains": chains, "burn_in": burn, "updates_per_chain": upd, "gpu_path": s.info()["path"],
                          "cpu_oracle_vs_MAR": {"mean_hellinger": e_cpu["MeanHellinger"], "max_hellinger": e_cpu["MaxHellinger"], "mean_abs_err": e_cpu["MeanMeanAbsError"]},
                          "gpu_vs_MAR": {"mean_hellinger": e_gpu["MeanHellinger"], "max_hellinger": e_gpu["MaxHellinger"], "mean_abs_err": e_gpu["MeanMeanAbsError"]},
                          "gpu_vs_cpu_oracle": {"mean_hellinger": e_x["MeanHellinger"], "max_hellinger": e_x["MaxHellinger"]},
                          "cpu_oracle_seed42_vs_seed43": {"mean_hellinger": e_self["MeanHellinger"], "max_hellinger": e_self["MaxHellinger"]},
                          "gpu_seed42_vs_seed43": {"mean_hellinger": e_gself["MeanHellinger"], "max_hellinger": e_gself["MaxHellinger"]},
                          "gpu_strict_scan_vs_cpu_oracle": {"mean_hellinger": e_strict["MeanHellinger"], "max_hellinger": e_strict["MaxHellinger"]},
                          "cpu_seconds": round(t_cpu, 1)}), flush=True)


if __name__ == "__main__":
    main()
