#!/usr/bin/env python3
"""bench.py -- headline metric of the Gibbs hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Metric (BASELINE.json): variable-updates/s summed over all chains on
res/Grids_13.uai, GibbsSimple, 65 536 chains per GPU (configs[2], the
configuration the metric is quoted on), plus Hellinger vs Grids_13.uai.MAR.
One "step" = one pass of the hot path over the whole batch of chains:
SWEEPS_PER_STEP sweeps = SWEEPS_PER_STEP * n_vars single-variable updates on
every chain (one gr_advance call, chain state resident in HBM/LDS).
Chains shard over ranks with no data-path collective (weak scaling); a single
RCCL all-reduce of the 200 marginal counters closes the timed region.

Prints ONE JSON line on rank 0.  At N = 1 the line also carries
  roofline      the resource that binds the dominant kernel (VALU issue), priced with the committed
                micro-benchmark profiles/r03_issue_rate_ubench.txt; the algorithmic-bytes figure of
                SURVEY 8(d) and the measured HBM traffic ride along as sub-objects
  cpu_baseline  the oracle (C restatement of grample's CPU path) timed on this host
  configs       the other four BASELINE.json configurations, each timed (wall), scored against its
                .MAR and probed against the CPU oracle at 256 chains
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "Grids_13.uai"
CHAINS_PER_GPU = 65536
SWEEPS_PER_STEP = 1000
BURNIN_SWEEPS = 200
# SURVEY.md 8(d) / BASELINE.md: algorithmic bytes per single-variable update
# (u8 state gathers + f32 table reads + state write 1 + counter RMW 8)
ALGO_BYTES = {"one.uai": 17.0, "Grids_11.uai": 53.0, "Grids_13.uai": 53.0, "Segmentation_11.uai": 65.7, "ObjectDetection_13.uai": 430.5}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4       # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
NOMINAL_GHZ = 2.4
ROOFLINE_INPUTS = os.path.join(ROOT, "profiles", "roofline_inputs.json")


def sources_sha():
    """Identity of the kernel sources the stored SQ counters were taken on: a counter file from another build is reported as stale."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "grample_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(res_dir):
    """The oracle (C restatement of grample's CPU path, oracle/) timed on this host: a bounded
    sample of the SAME workload (Grids_13, GibbsSimple, burn-in then counted updates), one chain
    set per hardware thread.  Reported baseline, not a target."""
    from oracle import pyoracle as O
    from tests import uai_py
    O.build()
    pm = uai_py.load(res_dir, MODEL)
    om = O.Model.from_parsed(pm)
    threads = max(1, min(len(os.sched_getaffinity(0)), 64))
    chains_per_thread, burn, upd = 4, 2000, 4_000_000
    t0 = time.time()
    probe = O.lib().orc_bench_simple(om.ptr, 42, 1, 1, 0, 200_000, None)  # calibrate: 200k updates on one thread
    rate1 = probe / max(time.time() - t0, 1e-6)
    upd = int(max(100_000, min(upd, rate1 * 12.0 / chains_per_thread)))  # aim at ~12 s of CPU work per thread
    t0 = time.time()
    total = O.lib().orc_bench_simple(om.ptr, 42, threads, chains_per_thread, burn, upd, None)
    dt = time.time() - t0
    if total < 0:
        raise RuntimeError(O.err())
    return {"value": (total + threads * chains_per_thread * burn) / dt, "unit": "variable-updates/s", "cores": threads,
            "cpu_model": cpu_model(), "host_cores_online": os.cpu_count(), "single_thread_value": rate1,
            "kind": "port",
            "sample": f"{MODEL} GibbsSimple, {threads} threads x {chains_per_thread} chains, burn-in {burn} + {upd} counted "
                      f"updates per chain, f64 + MT19937-64 per thread ({dt:.1f} s wall); single_thread_value = a 200 000-update calibration run on one thread"}


def oracle_probe(res_dir, g, name, sampler="simple", chains=256, burn_sweeps=20, sweeps=200, cvars=(10, 50, 100, 150)):
    """Matched small run: the SAME chain count, burn-in and update count on the oracle (MT19937-64, f64) and on the
    GPU (Philox, f32 tables), each scored against <model>.MAR with the reference's ErrorSuite, plus the Hellinger
    distance between the two merged marginals.  (The seed-for-seed streams differ by construction -- DESIGN.md D5 --
    so the comparison is between estimators of the same law at the same horizon.)  sampler "collapsed": four chain
    groups, group k with variable cvars[k] collapsed, on both sides."""
    from oracle import pyoracle as O
    from tests import uai_py
    pm = uai_py.load(res_dir, name)
    sol = uai_py.load_mar(res_dir, name)
    n = len(pm.card)
    burn, upd = burn_sweeps * n, sweeps * n
    groups = len(cvars) if sampler == "collapsed" else 1
    ckw = dict(collapse_mode=2, collapse_vars=[cvars[c * groups // chains] for c in range(chains)]) if sampler == "collapsed" else {}
    t0 = time.time()
    mo, _, _, _ = O.run_chains(O.Model.from_parsed(pm), 42, chains, burn, upd, **ckw)
    t_cpu = time.time() - t0
    s = g.HIPChains(g.Model.NewModelFromFile(os.path.join(res_dir, name)), chains, n_variants=groups, seed=42)
    if sampler == "collapsed":
        for k in range(groups):
            s.Collapse(k, cvars[k])
    s.BurnIn(burn)
    s.Advance(upd)
    mg, _ = g.MergeChains([s])
    e_cpu, e_gpu, e_x = g.NewErrorSuite(pm.card, sol, mo), g.NewErrorSuite(pm.card, sol, mg), g.NewErrorSuite(pm.card, mo, mg)
    return {"chains": chains, "burn_in": burn, "updates_per_chain": upd, "cpu_seconds": round(t_cpu, 2),
            "cpu_oracle_vs_MAR": {"mean_hellinger": e_cpu["MeanHellinger"], "max_hellinger": e_cpu["MaxHellinger"], "mean_abs_err": e_cpu["MeanMeanAbsError"]},
            "gpu_vs_MAR": {"mean_hellinger": e_gpu["MeanHellinger"], "max_hellinger": e_gpu["MaxHellinger"], "mean_abs_err": e_gpu["MeanMeanAbsError"]},
            "gpu_vs_cpu_oracle": {"mean_hellinger": e_x["MeanHellinger"], "max_hellinger": e_x["MaxHellinger"], "mean_abs_err": e_x["MeanMeanAbsError"]}}


PATHS = {0: "direct (k_sweep)", 1: "memoised (k_sweep)", 2: "memoised+plan, chain per lane (k_sweep_bin)", 3: "memoised+plan, step-parallel (k_sweep_par4 / k_sweep_wide)",
         5: "memoised, strict per-chain scan (k_sweep_strict)", 6: "direct, value-parallel (k_sweep_pw4)", 8: "direct, step-parallel (k_sweep_dir4)",
         9: "memoised, systematic scan, one wavefront per chain (k_sweep_chroma)", 11: "memoised+plan, step-parallel, bit-packed state (k_sweep_bits)"}


def other_configs(res_dir, g, device):
    """BASELINE.json configs 0, 1, 3, 4 (config 2 is the headline): wall-clock updates/s of the counted phase, kernel path,
    ErrorSuite vs .MAR, and a 256-chain probe against the CPU oracle.  Config 0 is the reference's CPU plumbing case (1 chain,
    1 000 updates): on the GPU it is one launch-bound call; config 4 is its per-GPU share (32 768 of 262 144 chains)."""
    import numpy as np
    specs = [  # title, model, sampler, chains, groups, burn-in sweeps, counted sweeps, probe kwargs
        ("configs[0] one.uai GibbsSimple, 1 chain, 1k iters", "one.uai", "simple", 1, 1, 0, 1000, dict(chains=64, burn_sweeps=0, sweeps=1000)),
        ("configs[1] Grids_11 GibbsSimple, 4096 chains", "Grids_11.uai", "simple", 4096, 1, 200, 4000, dict()),
        ("configs[1'] the same with GR_FLAG_SCAN_SYSTEMATIC (chromatic scan; not the reference's chain law)", "Grids_11.uai", "systematic", 4096, 1, 200, 4000, None),
        ("configs[3] Segmentation_11 GibbsCollapsed, 32768 chains in 128 groups (one Collapse(-1) per group)", "Segmentation_11.uai", "collapsed", 32768, 128, 200, 1000,
         dict(sampler="collapsed", burn_sweeps=20, sweeps=100)),
        ("configs[4] ObjectDetection_13 GibbsSimple, per-GPU share: 32768 of 262144 chains", "ObjectDetection_13.uai", "simple", 32768, 1, 200, 600,
         dict(burn_sweeps=50, sweeps=400)),
    ]
    out = []
    for title, model, samp, chains, groups, burn, sweeps, pk in specs:
        m = g.Model.NewModelFromFile(os.path.join(res_dir, model))
        n = m.n_vars
        s = g.HIPChains(m, chains, n_variants=groups, device=device, seed=42, flags=g.GR_FLAG_SCAN_SYSTEMATIC if samp == "systematic" else 0)
        if samp == "collapsed":
            for k in range(groups):
                s.Collapse(k, -1)
        s.BurnIn(burn * n)
        s.Advance(min(sweeps, 50) * n)  # warm: tables built, kernels loaded
        t0 = time.perf_counter()
        s.Advance(sweeps * n)           # synchronous
        dt = time.perf_counter() - t0
        info = s.info()
        merged, coll = g.MergeChains([s])
        sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, model + ".MAR")).marginals()
        es = g.NewErrorSuite(m.cards(), sol, merged)
        rate = chains * sweeps * n / dt
        out.append({"config": title, "value": rate, "unit": "variable-updates/s", "timing": "wall clock around one gr_advance call of the counted phase",
                    "ms": dt * 1e3, "kernel_ms": info["kernel_ms_last"], "updates": chains * sweeps * n,
                    "kernel": PATHS.get(info["path"] & 15, str(info["path"])) + (f": {info['block_threads'] // 64} lanes per chain" if info["path"] == 3 else ""),
                    "collapsed_vars": int(np.sum(coll)),
                    "hellinger_vs_MAR": {"mean": es["MeanHellinger"], "max": es["MaxHellinger"], "mean_abs_err": es["MeanMeanAbsError"]},
                    "effective_hbm_frac": ALGO_BYTES[model] * rate / 1e9 / HBM_PEAK_GBS,
                    "oracle_probe": oracle_probe(res_dir, g, model, **pk) if pk is not None else None})
        s.close()
    out.append(adaptive_config(res_dir, g, device))
    return out


def adaptive_config(res_dir, g, device, rounds=34, cw=2000, start=2, step=4):
    """BASELINE.json configs[3] as the reference RUNS it (script/experiment + sampler/adaptive.go): start with 2 chain groups, each round
    AdvanceChain(cw) -> merge + score -> Adapt(+4 groups, each collapsing the variable AdaptSelect picks), until all 128 groups
    of 256 chains are live.  End-to-end wall clock, everything included (table builds for the new groups, merges, selection)."""
    import numpy as np
    model = "Segmentation_11.uai"
    m = g.Model.NewModelFromFile(os.path.join(res_dir, model), True)
    sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, model + ".MAR")).marginals()
    t0 = time.perf_counter()
    s = g.HIPChains(m, 128 * 256, n_variants=128, device=device, seed=42)
    s.set_active_groups(start)
    s.BurnIn(2000)
    t_ac = t_adapt = t_merge = kms = 0.0
    for _ in range(rounds):
        t = time.perf_counter(); s.AdvanceChain(cw); t_ac += time.perf_counter() - t
        kms += s.info()["kernel_ms_last"]
        t = time.perf_counter(); merged, coll = g.MergeChains([s]); es = g.NewErrorSuite(m.cards(), sol, merged); t_merge += time.perf_counter() - t
        t = time.perf_counter(); s.Adapt(step); t_adapt += time.perf_counter() - t
    tot = time.perf_counter() - t0
    counted = float(s.TotalSampleCount())
    out = {"config": f"configs[3'] Segmentation_11 adaptive GibbsCollapsed as script/experiment runs it: {start} groups -> {s.active_groups()} of 256 chains, "
                     f"{rounds} rounds of AdvanceChain({cw}) + merge + Adapt(+{step})",
           "value": counted / tot, "unit": "counted variable-updates/s, end to end", "timing": "wall clock from sampler creation to the last Adapt",
           "ms": tot * 1e3, "phases_ms": {"AdvanceChain": t_ac * 1e3, "sweep_kernels": kms, "merge_and_score": t_merge * 1e3, "Adapt": t_adapt * 1e3},
           "updates": counted, "groups": s.active_groups(), "collapsed_vars": int(np.sum(coll)),
           "hellinger_vs_MAR": {"mean": es["MeanHellinger"], "max": es["MaxHellinger"], "mean_abs_err": es["MeanMeanAbsError"]}}
    s.close()
    return out


def roofline(kernel_rate, wall_rate, per_launch_updates, per_launch_s):
    """kernel_rate: updates/s inside the sweep launches (HIP events on the sampler's stream, measured in this run)."""
    eff = ALGO_BYTES[MODEL] * kernel_rate / 1e9
    effective = {"achieved": eff, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_of_hbm_peak": eff / HBM_PEAK_GBS,
                 "note": f"SURVEY 8(d)'s ALGORITHMIC bytes ({ALGO_BYTES[MODEL]} B/update, no reuse assumed) x updates/s.  Not traffic: the kernel keeps chain "
                         "state, memoised thresholds, records and histograms in LDS for a whole launch, so a value above the HBM peak only says that those "
                         "bytes are never streamed (north_star's '>= 40 % of HBM roofline' is met in this effective sense only)"}
    out = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G VALU-port cycles/s", "frac": None, "traffic": None,
           "effective_hbm": effective,
           "launch": {"updates": per_launch_updates, "ms": per_launch_s * 1e3, "note": "mean over the timed steps' launches, HIP events on the sampler's stream: k_sweep_par4 (+ its k_scan_plan when that was not computed ahead, beside the previous sweep, on the sampler's second stream)"}}
    if not os.path.exists(ROOFLINE_INPUTS):
        out["note"] = "profiles/roofline_inputs.json missing: run scripts/collect_profiles.sh + scripts/store_profiles.py on the GPU box"
        return out
    inp = json.load(open(ROOFLINE_INPUTS))
    sha = sources_sha()
    sclk = inp.get("sclk_ghz_measured") or NOMINAL_GHZ
    vcyc, lcyc = inp["valu_port_cycles_per_update"], inp["lds_port_cycles_per_update"]
    achieved = vcyc * kernel_rate / 1e9
    peak = N_SIMD * NOMINAL_GHZ
    out.update({
        "achieved": achieved, "peak": peak, "frac": achieved / peak,
        "frac_at_measured_clock": achieved / (N_SIMD * sclk), "sclk_ghz_measured": sclk,
        "traffic": inp.get("hbm_bytes_per_launch"),
        "definition": "achieved = VALU-port cycles the dominant kernel's instruction stream needs per update (dynamic SQ_INSTS_VALU of the rocprofv3 pass x the "
                      "measured issue cost of the kernel's instruction mix: 2.4 cycles for the simple 32-bit ops, 4.25 for multiplies / multiply-adds / compares / "
                      "SDWA, 8.1 for the lane swaps; scripts/isa_mix.py + profiles/r03_issue_rate_ubench.txt) x this run's updates/s inside the launches; "
                      "peak = 1024 SIMDs x 2.4 GHz: the cycles the VALU ports have.  frac = the share of the chip's VALU issue capacity the kernel uses",
        "valu": {"port_cycles_per_update": vcyc, "instructions_per_64_updates": inp["valu_inst_per_64_updates"], "mean_cycles_per_instruction": inp["valu_mean_cycles"]},
        "lds": {"bound": "lds_pipe", "port_cycles_per_update": lcyc, "achieved": lcyc * kernel_rate / 1e9, "peak": peak, "frac": lcyc * kernel_rate / 1e9 / peak,
                "instructions_per_64_updates": inp["lds_inst_per_64_updates"], "mean_cycles_per_instruction": inp["lds_mean_cycles"],
                "note": "LDS instructions priced per SIMD with all four SIMDs of a CU issuing (ds_read_u8/b32 8.3 cycles, ds_write_b8 16.3, ds_add_u64 24.3): the second most loaded resource"},
        "frac_if_every_instruction_were_full_rate": inp["valu_inst_per_64_updates"] / 64.0 * 2.4 * kernel_rate / 1e9 / peak,
        "inputs": {"file": "profiles/roofline_inputs.json", "round": inp.get("round"), "sources_sha": inp.get("sources_sha"), "this_build_sources_sha": sha,
                   "matches_build": inp.get("sources_sha") == sha,
                   "note": "instruction counts and clock are STORED counters of a rocprofv3 --pmc pass over this same command (they cannot be read from inside "
                           "bench.py); the update rate they are multiplied with is measured live.  matches_build = false means the kernel sources changed since"}})
    if inp.get("hbm_bytes_per_launch") and inp.get("updates_per_launch"):
        bpu = inp["hbm_bytes_per_launch"] / inp["updates_per_launch"]
        gbs = bpu * kernel_rate / 1e9
        out["hbm_measured"] = {"GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "bytes_per_update": bpu,
                               "note": "rocprofv3 FETCH_SIZE / WRITE_SIZE passes, KiB units, FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes: scan plan in/out, "
                                       "chain state in/out, counters.  traffic / algorithmic bytes = %.4f: nothing is re-read" % (bpu / ALGO_BYTES[MODEL])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline, parity_probe, strict_scan and configs (profiling runs)")
    ap.add_argument("--no-configs", action="store_true", help="skip the configs array only")
    args = ap.parse_args()

    import numpy as np
    import torch
    import grample_amd as g

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the Gibbs path has no CPU fallback")
    # BENCH_REHEARSE_ONE_GPU=1: every rank uses GPU 0 and the collective runs over gloo -- lets the
    # N > 1 code path (sharding, offsets, all-reduce, max-over-ranks timing) be rehearsed on a 1-GPU box.
    rehearse = os.environ.get("BENCH_REHEARSE_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist, backend = None, None
    # BENCH_FORCE_DIST=1 initialises the process group even for one rank (exercises RCCL on a 1-GPU box)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = "gloo" if rehearse else "nccl"  # "nccl" is RCCL on ROCm
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    res_dir = os.path.join(ROOT, "tests", "golden", "res")
    model = g.Model.NewModelFromFile(os.path.join(res_dir, MODEL))
    n_vars = model.n_vars
    cards = model.cards()
    sampler = g.HIPChains.NewGibbsSimple(model, CHAINS_PER_GPU, device=local_rank, seed=42,
                                         chain_offset=rank * CHAINS_PER_GPU)
    upd_per_step = SWEEPS_PER_STEP * n_vars
    sampler.BurnIn(BURNIN_SWEEPS * n_vars)  # NewChain burn-in (chain.go:166-172), not timed

    from grample_amd.dist import allreduce_counts
    n_allreduce = [0]

    def reduce_counts():
        """the path's only exchange: all-reduce(sum) of the per-variable histograms (RCCL when a process group is up)"""
        n_allreduce[0] += 1 if dist is not None else 0
        return allreduce_counts(sampler.counts()[0], device="cpu" if rehearse else "cuda")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sampler.Advance(upd_per_step)
    reduce_counts()  # warms the communicator up
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches = 0.0, 0
    for _ in range(args.steps):
        sampler.Advance(upd_per_step)  # synchronous: the sampler's stream is drained before it returns
        info = sampler.info()
        kernel_ms += info["kernel_ms_last"]
        launches += info["n_launches_last"]
    total_counts = reduce_counts()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        dev = "cpu" if rehearse else "cuda"
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        k = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        kernel_ms = float(k.item())

    if rank == 0:
        total_updates = world * CHAINS_PER_GPU * upd_per_step * args.steps
        value = total_updates / dt
        # merged marginals = per-chain 1/Card prior + counts (chain.go:96-148), all ranks, warm-up included
        prior = np.concatenate([np.full(c, world * CHAINS_PER_GPU / c) for c in cards])
        merged = g.split(total_counts.astype(np.float64) + prior, cards)
        sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, MODEL + ".MAR")).marginals()
        es = g.NewErrorSuite(cards, sol, merged)
        info = sampler.info()
        per_launch_updates = CHAINS_PER_GPU * upd_per_step * args.steps / max(launches, 1)
        per_launch_s = kernel_ms * 1e-3 / max(launches, 1)
        kernel_rate = per_launch_updates / per_launch_s  # one GPU's updates/s inside its launches
        out = {
            "metric": "variable-updates/sec (all chains), Grids_13.uai GibbsSimple",
            "value": value, "unit": "variable-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "real model file res/Grids_13.uai (vendored fixture), random start states, seed 42",
            "config": {"workload": f"{MODEL} GibbsSimple, {CHAINS_PER_GPU} chains per GPU, {SWEEPS_PER_STEP} sweeps "
                                   f"({upd_per_step} single-variable updates per chain) per step",
                       "chains_total": world * CHAINS_PER_GPU, "burnin_sweeps": BURNIN_SWEEPS,
                       "parallelism": f"chains sharded over {world} GPU(s), one RCCL all-reduce of 200 counters at the end",
                       "kernel": {"path": PATHS.get(info["path"], str(info["path"])), "lanes_per_chain": info["block_threads"] // 64 if info["path"] == 3 else None}},
            "hellinger_vs_MAR": {"mean": es["MeanHellinger"], "max": es["MaxHellinger"], "mean_abs_err": es["MeanMeanAbsError"]},
            "collective": {"backend": backend, "world": world if dist is not None else None, "allreduce_calls": n_allreduce[0],
                           "reduced_total": int(total_counts.sum()),
                           "note": "torch.distributed all_reduce(sum) of the 200 histogram counters: once after warm-up, once closing the timed region"
                                   if dist is not None else "single process, no process group: nothing to exchange"},
            "roofline": roofline(kernel_rate, value, per_launch_updates, per_launch_s),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res_dir)
            out["parity_probe"] = oracle_probe(res_dir, g, MODEL, chains=512, burn_sweeps=20, sweeps=200)
            # the same workload with the reference's exact selection rule (every chain draws its own variable,
            # sampler.go:135-174) instead of one draw per 64-chain group: a reported side figure
            strict = g.HIPChains(model, CHAINS_PER_GPU, device=local_rank, seed=42, flags=g.GR_FLAG_SCAN_PER_CHAIN)
            strict.BurnIn(BURNIN_SWEEPS * n_vars)
            strict.Advance(upd_per_step)
            out["strict_scan"] = {"value": CHAINS_PER_GPU * upd_per_step / (strict.info()["kernel_ms_last"] * 1e-3),
                                  "unit": "variable-updates/s", "note": "GR_FLAG_SCAN_PER_CHAIN, kernel time of one step"}
            strict.close()
            # ... and with the systematic (chromatic) scan, GR_FLAG_SCAN_SYSTEMATIC: NOT the reference's chain law (a fixed update order
            # instead of VarSample's random choice; same conditionals, same stationary law) -- the option SURVEY 8(f) N4 names
            systematic = g.HIPChains(model, CHAINS_PER_GPU, device=local_rank, seed=42, flags=g.GR_FLAG_SCAN_SYSTEMATIC)
            systematic.BurnIn(BURNIN_SWEEPS * n_vars)
            systematic.Advance(upd_per_step)
            t1 = time.perf_counter()
            systematic.Advance(upd_per_step)
            dts = time.perf_counter() - t1
            ms, _ = g.MergeChains([systematic])
            ess = g.NewErrorSuite(cards, sol, ms)
            out["systematic_scan"] = {"value": CHAINS_PER_GPU * upd_per_step / dts, "unit": "variable-updates/s",
                                      "kernel_value": CHAINS_PER_GPU * upd_per_step / (systematic.info()["kernel_ms_last"] * 1e-3),
                                      "hellinger_vs_MAR": {"mean": ess["MeanHellinger"], "max": ess["MaxHellinger"]},
                                      "note": "GR_FLAG_SCAN_SYSTEMATIC (k_sweep_chroma), wall clock of one step; off by default: the headline value above is the "
                                              "reference's random scan"}
            systematic.close()
            if not args.no_configs:
                sampler.close()
                out["configs"] = other_configs(res_dir, g, local_rank)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
