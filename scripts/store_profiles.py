#!/usr/bin/env python3
"""Copies the measurement files of the last `scripts/collect_profiles.sh` pass (gpurun_out/<tag>/) into profiles/ under the
round tag and recomputes profiles/hbm_traffic.json: per-launch HBM bytes of the dominant kernel, corrected as
MI355X_MICROARCH.md says (KiB units, FETCH_SIZE doubled on gfx950), and its VALU instructions per 64 updates.
bench.py reads `bytes_per_launch` and `valu_inst_per_64_updates` from that file, so run bench.py again afterwards."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
G, P = os.path.join(ROOT, "gpurun_out", TAG), os.path.join(ROOT, "profiles")
KERN = "k_sweep_par4<4, true, 1, 2, 0>"
UPD_PER_LAUNCH = 50000 * 65536  # a bench step (100 000 updates per chain) = 2 launches


def kernel_rows(path, keep="gr::"):
    rows = list(csv.reader(open(path)))
    ki = rows[0].index("Kernel_Name")
    return [rows[0]] + [r for r in rows[1:] if keep in r[ki]]


def main():
    for src, dst in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("configs.jsonl", "configs.jsonl"),
                     ("stats/run_kernel_stats.csv", "kernel_stats.csv"), ("dev_time.txt", "dev_time.txt"), ("dev_od.txt", "dev_od.txt"), ("dev_wide.txt", "dev_wide.txt"), ("dev_systematic.txt", "dev_systematic.txt"), ("issue_rate_ubench.txt", "issue_rate_ubench.txt"),
                     ("all_models_32768.txt", "all_models_32768.txt"), ("parity_configs.jsonl", "parity_configs.jsonl"),
                     ("seed_noise.txt", "seed_noise.txt"), ("advance_chain.txt", "advance_chain.txt"), ("dev_chains.txt", "dev_chains_run.txt"), ("dev_bits.txt", "dev_bits_run.txt"),
                     ("adaptive_rounds.txt", "adaptive_rounds_run.txt"), ("dev_prio.txt", "dev_prio.txt"), ("state_form_ubench_run.txt", "state_form_ubench_run.txt")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, f"{TAG}_{dst}"))
    for tag in ("fetch", "write", "sqA", "sqB", "gui"):
        src = os.path.join(G, f"pmc_{tag}", "run_counter_collection.csv")
        if os.path.exists(src):
            csv.writer(open(os.path.join(P, f"{TAG}_pmc_{tag}.csv"), "w", newline="")).writerows(kernel_rows(src))

    def rows(tag, ctr, kern):
        return [r for r in csv.DictReader(open(os.path.join(P, f"{TAG}_pmc_{tag}.csv"))) if r["Counter_Name"] == ctr and kern in r["Kernel_Name"]]

    def vals(tag, ctr, kern):
        return [float(r["Counter_Value"]) for r in rows(tag, ctr, kern)]

    f, w, pw = vals("fetch", "FETCH_SIZE", KERN), vals("write", "WRITE_SIZE", KERN), vals("write", "WRITE_SIZE", "k_scan_plan")
    ks = {r["Name"]: r for r in csv.DictReader(open(os.path.join(P, f"{TAG}_kernel_stats.csv")))}
    sw = [r for k, r in ks.items() if KERN in k][0]
    sp = [r for k, r in ks.items() if "k_scan_plan" in k][0]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    # VALU instructions per 64 updates: one bench step = two sweep launches = 100 000 updates on each of 65 536 chains
    valu = vals("sqA", "SQ_INSTS_VALU", KERN)
    per64 = sum(valu[-2:]) / (100000.0 * 65536.0 / 64.0)
    # shader clock the chip sustained inside the sweep launches: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / launch duration
    sclk = None
    if os.path.exists(os.path.join(P, f"{TAG}_pmc_gui.csv")):
        gr_ = rows("gui", "GRBM_GUI_ACTIVE", KERN)
        if gr_:
            sclk = sum(float(r["Counter_Value"]) / 8.0 / (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in gr_) / len(gr_)
    # ---- profiles/roofline_inputs.json: what bench.py's roofline object is computed from
    import hashlib
    import subprocess
    UPD2 = 100000.0 * 65536.0  # one bench step = two sweep launches
    def per64_of(tag, ctr):
        v = vals(tag, ctr, KERN)
        return sum(v[-2:]) / (UPD2 / 64.0)
    valu64, lds64, salu64 = per64_of("sqA", "SQ_INSTS_VALU"), per64_of("sqA", "SQ_INSTS_LDS"), per64_of("sqA", "SQ_INSTS_SALU")
    vmem64, br64 = per64_of("sqA", "SQ_INSTS_VMEM_RD"), per64_of("sqB", "SQ_INSTS_BRANCH")
    mix_path = os.path.join(G, "isa_mix.json")
    mix = json.load(open(mix_path)) if os.path.exists(mix_path) and os.path.getsize(mix_path) else \
        json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_mix.py"), "--json"], check=True, capture_output=True, text=True).stdout)
    json.dump(mix, open(os.path.join(P, f"{TAG}_isa_mix.json"), "w"), indent=1)

    def priced(cls, dyn_per64):
        """dynamic instruction count x static price: the loop's shortest path (a wavefront iteration = 256 updates without dependency
        levels) at its own mean cost, the instructions beyond it (level passes) at the mean cost of the rest of the loop"""
        full, short = mix["classes"][cls], mix["classes_short_path"][cls]
        unroll = 2.0  # the loop body holds two wavefront iterations (register sets A / B)
        n_short = short["instructions"] / unroll / 4.0   # per 64 updates
        c_short = short["port_cycles"] / unroll / 4.0
        rest_n = full["instructions"] - short["instructions"]
        rest_mean = (full["port_cycles"] - short["port_cycles"]) / rest_n if rest_n else short["mean_cycles"]
        extra = max(dyn_per64 - n_short, 0.0)
        cyc = (c_short if dyn_per64 >= n_short else dyn_per64 * short["mean_cycles"]) + extra * rest_mean
        return cyc, cyc / dyn_per64
    vc, vmean = priced("valu", valu64)
    lc, lmean = priced("lds", lds64)
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "grample_amd", "csrc")
    for f_ in sorted(os.listdir(cs)):
        if f_.endswith((".hip", ".h", ".hpp", ".cpp")):
            h.update(f_.encode()); h.update(open(os.path.join(cs, f_), "rb").read())
    ri = {"round": int(TAG[1:]), "sources_sha": h.hexdigest()[:16],
          "kernel": f"gr::{KERN}", "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline under rocprofv3 --pmc (scripts/collect_profiles.sh)",
          "updates_per_launch": UPD_PER_LAUNCH, "hbm_bytes_per_launch": int(2 * fm * 1024 + wm * 1024),
          "sclk_ghz_measured": round(sclk, 3) if sclk else None,
          "valu_inst_per_64_updates": round(valu64, 3), "lds_inst_per_64_updates": round(lds64, 3), "salu_inst_per_64_updates": round(salu64, 3),
          "vmem_inst_per_64_updates": round(vmem64, 3), "branch_inst_per_64_updates": round(br64, 3),
          "valu_mean_cycles": round(vmean, 3), "lds_mean_cycles": round(lmean, 3),
          "valu_port_cycles_per_update": round(vc / 64.0, 4), "lds_port_cycles_per_update": round(lc / 64.0, 4),
          "avg_launch_ns": int(float(sw["AverageNs"])),
          "pricing": "dynamic SQ counts (profiles/%s_pmc_sq*.csv) x static mix of the loop (profiles/%s_isa_mix.json, scripts/isa_mix.py) x measured issue costs "
                     "(profiles/r03_issue_rate_ubench.txt, 4 wavefronts per SIMD)" % (TAG, TAG)}
    json.dump(ri, open(os.path.join(P, "roofline_inputs.json"), "w"), indent=2)
    print("roofline_inputs:", ri["valu_inst_per_64_updates"], ri["valu_mean_cycles"], ri["valu_port_cycles_per_update"], ri["lds_port_cycles_per_update"], ri["sclk_ghz_measured"])
    d = {
        "round": int(TAG[1:]),
        "sclk_ghz_measured": round(sclk, 3) if sclk else None,
        "sclk_note": "GRBM_GUI_ACTIVE / 8 XCDs / launch duration of the sweep launches (rocprofv3 --pmc GRBM_GUI_ACTIVE pass)",
        "kernel": f"gr::{KERN} (binary memoised path, 4 lanes per chain, counting) -- the dominant kernel of bench.py",
        "command": "rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE; SQ_* in two more passes) --kernel-trace --output-format csv -- python3 bench.py "
                   "--steps 5 --warmup 1 --no-cpu-baseline ; kernel durations from rocprofv3 --kernel-trace --stats of the same command (scripts/collect_profiles.sh)",
        "launch_shape": "a bench step (100000 updates per chain) = 1 launch of 65536 updates + 1 of 34464: mean 50000 updates x 65536 chains = 3.28e9 updates per launch",
        "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1),
        "correction": "MI355X_MICROARCH.md HBM section: counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of wide coalesced read bytes -> doubled; WRITE_SIZE as is",
        "bytes_per_launch": int(2 * fm * 1024 + wm * 1024),
        "algorithmic_bytes_per_launch": int(53.0 * UPD_PER_LAUNCH),
        "scan_plan_kernel": {"write_size_kb_mean": round(sum(pw) / len(pw), 1), "avg_ns": int(float(sp["AverageNs"]))},
        "avg_launch_ns": {KERN: int(float(sw["AverageNs"])), "k_scan_plan": int(float(sp["AverageNs"]))},
        "valu_inst_per_64_updates": round(per64, 2),
        "valu_note": "SQ_INSTS_VALU of one bench step's two sweep launches / (6.5536e9 updates / 64): VALU wave-instructions per 64 updates",
        "files": sorted(x for x in os.listdir(P) if x.startswith(TAG + "_")),
    }
    d["note"] = ("real HBM traffic of the sweep = scan plan read + chain state in/out + counters: %.3f B per update against 53 B algorithmic; "
                 "tables, records, state and histograms stay in LDS for the whole launch" % (d["bytes_per_launch"] / UPD_PER_LAUNCH))
    json.dump(d, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=2)
    print(d["bytes_per_launch"], d["avg_launch_ns"], d["valu_inst_per_64_updates"])


if __name__ == "__main__":
    main()
