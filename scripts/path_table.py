#!/usr/bin/env python3
"""The kernel-selection decision table: for every fixture model (with and without its evidence file) and a ladder of chain
counts, what gr_sampler picks -- path, lanes per chain, block, LDS bytes, where the tables live.  The selection code
(sampler.cpp: configure / configure_impl / run) is a set of measured thresholds; this table is their observable outcome.

  python scripts/path_table.py            -> prints the table (one line per model / evidence / chains / groups)
  python scripts/path_table.py --json     -> the same as JSON (tests/golden/path_table.json is compared with it by
                                             tests/test_gpu_parity.py::test_kernel_selection_table: a threshold change must be deliberate)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grample_amd as g  # noqa: E402

RES = os.path.join(ROOT, "tests", "golden", "res")
CHAINS = [512, 4096, 32768, 65536, 262144]
PATH = {0: "k_sweep direct", 1: "k_sweep memoised", 2: "k_sweep_bin", 3: "k_sweep_par4/wide", 5: "k_sweep_strict", 6: "k_sweep_pw4", 8: "k_sweep_dir4",
        9: "k_sweep_chroma", 11: "k_sweep_bits (bit-packed state)", 16: "k_sweep direct, state in global memory", 17: "k_sweep memoised, state in global memory"}


def rows():
    out = []
    for f in sorted(os.listdir(RES)):
        if not f.endswith(".uai"):
            continue
        evs = [False]
        ev_path = os.path.join(RES, f + ".evid")
        if os.path.exists(ev_path) and open(ev_path).read().split()[:1] not in ([], ["0"]):
            evs.append(True)
        for ev in evs:
            try:
                m = g.Model.NewModelFromFile(os.path.join(RES, f), ev)
            except g.GrError as e:
                out.append({"model": f, "evidence": ev, "error": str(e)[:60]})
                continue
            for chains in CHAINS:
                for groups, flags, tag in ((1, 0, ""), (1, g.GR_FLAG_SCAN_SYSTEMATIC, "systematic"), (16, 0, "collapsed")):
                    if tag == "systematic" and chains not in (4096, 65536):
                        continue
                    if tag == "collapsed" and (chains != 32768 or ev):
                        continue
                    try:
                        s = g.HIPChains(m, chains, n_variants=groups, flags=flags)
                        if tag == "collapsed":  # every group but the first collapses one variable (the adaptive sampler's shape); models where no blanket fits keep theirs
                            try:
                                for k in range(1, groups):
                                    s.Collapse(k, -1)
                            except g.GrError:
                                pass
                        s.BurnIn(64)  # the lanes per chain are chosen per launch
                        i = s.info()
                        out.append({"model": f, "evidence": ev, "chains": chains, "scan": tag or "random", "path": i["path"], "kernel": PATH.get(i["path"], "?"),
                                    "lanes_per_chain": i["block_threads"] // 64 if i["path"] == 3 else None, "lds_bytes": i["lds_bytes"],
                                    "tables_in_lds": i["tables_in_lds"], "max_card": i["max_card"]})
                        s.close()
                    except g.GrError as e:
                        out.append({"model": f, "evidence": ev, "chains": chains, "scan": tag or "random", "error": str(e)[:60]})
    return out


if __name__ == "__main__":
    r = rows()
    if "--json" in sys.argv:
        print(json.dumps(r, indent=0))
    else:
        for x in r:
            if "error" in x:
                print(f"{x['model']:26s} ev={int(x['evidence'])} chains={x.get('chains', '-'):>7} {x.get('scan', ''):10s} ERROR {x['error']}")
            else:
                print(f"{x['model']:26s} ev={int(x['evidence'])} chains={x['chains']:>7} {x['scan']:10s} path {x['path']:2d} {x['kernel']:38s} lanes {str(x['lanes_per_chain']):4s} "
                      f"lds {x['lds_bytes']:6d} tables_in_lds {x['tables_in_lds']} card {x['max_card']}")
