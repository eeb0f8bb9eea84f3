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

Prints ONE JSON line on rank 0.
"""
import argparse
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
# SURVEY.md 8(d) / BASELINE.md: algorithmic bytes per single-variable update on Grids_13
# (u8 state gathers 4 + f32 table reads 40 + state write 1 + counter RMW 8)
ALGO_BYTES_PER_UPDATE = 53.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# The physical limiter of the dominant kernel is VALU issue, not HBM: every SIMD issues one
# wave-wide VALU instruction per 4 cycles.  256 CUs x 4 SIMDs x 2.4 GHz / 4 = 6.14e11 wave-instructions/s.
VALU_PEAK_GINST = 256 * 4 * 2.4 / 4.0  # G wave-instructions/s


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
    chains_per_thread, burn, upd = 4, 2000, 4_000_000  # ~4 * 4e6 / 3.2e6/s ~ 5 s per thread... scaled below
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
            "kind": "port",
            "sample": f"{MODEL} GibbsSimple, {threads} threads x {chains_per_thread} chains, burn-in {burn} + {upd} counted "
                      f"updates per chain, f64 + MT19937-64 per thread ({dt:.1f} s wall)"}


def parity_probe(res_dir, g):
    """Matched small run next to the headline number: the SAME chain count, burn-in and update count
    on the oracle (MT19937-64, f64) and on the GPU (Philox, f32 tables), each scored against
    Grids_13.uai.MAR with the reference's ErrorSuite, plus the Hellinger distance between the two
    merged marginals.  (The seed-for-seed streams differ by construction -- DESIGN.md D5 -- so the
    comparison is between estimators of the same law at the same horizon.)"""
    import numpy as np
    from oracle import pyoracle as O
    from tests import uai_py
    pm = uai_py.load(res_dir, MODEL)
    sol = uai_py.load_mar(res_dir, MODEL)
    chains, burn, upd = 512, 2000, 20000
    mo, _, _, _ = O.run_chains(O.Model.from_parsed(pm), 42, chains, burn, upd)
    s = g.HIPChains.NewGibbsSimple(g.Model.NewModelFromFile(os.path.join(res_dir, MODEL)), chains, seed=42)
    s.BurnIn(burn)
    s.Advance(upd)
    mg, _ = g.MergeChains([s])
    e_cpu, e_gpu, e_x = g.NewErrorSuite(pm.card, sol, mo), g.NewErrorSuite(pm.card, sol, mg), g.NewErrorSuite(pm.card, mo, mg)
    return {"chains": chains, "burn_in": burn, "updates_per_chain": upd,
            "cpu_oracle_vs_MAR": {"mean_hellinger": e_cpu["MeanHellinger"], "max_hellinger": e_cpu["MaxHellinger"], "mean_abs_err": e_cpu["MeanMeanAbsError"]},
            "gpu_vs_MAR": {"mean_hellinger": e_gpu["MeanHellinger"], "max_hellinger": e_gpu["MaxHellinger"], "mean_abs_err": e_gpu["MeanMeanAbsError"]},
            "gpu_vs_cpu_oracle": {"mean_hellinger": e_x["MeanHellinger"], "max_hellinger": e_x["MaxHellinger"], "mean_abs_err": e_x["MeanMeanAbsError"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    dist = None
    # BENCH_FORCE_DIST=1 initialises the process group even for one rank (exercises RCCL on a 1-GPU box)
    if world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))  # "nccl" is RCCL on ROCm

    res_dir = os.path.join(ROOT, "tests", "golden", "res")
    model = g.Model.NewModelFromFile(os.path.join(res_dir, MODEL))
    n_vars = model.n_vars
    cards = model.cards()
    sampler = g.HIPChains.NewGibbsSimple(model, CHAINS_PER_GPU, device=local_rank, seed=42,
                                         chain_offset=rank * CHAINS_PER_GPU)
    upd_per_step = SWEEPS_PER_STEP * n_vars
    sampler.BurnIn(BURNIN_SWEEPS * n_vars)  # NewChain burn-in (chain.go:166-172), not timed

    from grample_amd.dist import allreduce_counts

    def reduce_counts():
        """the path's only exchange: all-reduce(sum) of the per-variable histograms (RCCL when world > 1)"""
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
        achieved = ALGO_BYTES_PER_UPDATE * per_launch_updates / per_launch_s / 1e9
        traffic, valu_per_update, sclk = None, None, None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath))
                traffic = prof.get("bytes_per_launch")
                valu_per_update = prof.get("valu_inst_per_64_updates")  # SQ_INSTS_VALU per wave iteration (rocprofv3 --pmc pass)
                sclk = prof.get("sclk_ghz_measured")
            except Exception:
                traffic = None
        # what actually bounds the kernel, next to the contract's algorithmic-bytes figure: measured HBM traffic as a
        # fraction of the HBM peak, and the VALU issue rate as a fraction of the chip's issue peak
        limiter = {"kind": "valu_issue", "unit": "G wave-instructions/s", "peak": VALU_PEAK_GINST,
                   "note": "the working set (chain state, threshold tables, records, histograms) is LDS-resident for a whole launch; "
                           "the most loaded resource is the SIMDs' VALU issue port (4 cycles per wave instruction; busy 82 % of the wave "
                           "cycles inside the sweep by the SQ counters), see DESIGN.md sections 3 and 6"}
        if valu_per_update:
            a = valu_per_update / 64.0 * per_launch_updates / per_launch_s / 1e9
            limiter.update({"achieved": a, "frac": a / VALU_PEAK_GINST, "valu_inst_per_64_updates": valu_per_update})
            if sclk:  # the same fraction against the clock the chip actually sustained in this kernel (GRBM_GUI_ACTIVE pass)
                limiter.update({"sclk_ghz_measured": sclk, "frac_at_measured_clock": a / (VALU_PEAK_GINST * sclk / 2.4)})
        hbm_measured = None
        if traffic:
            gbs = traffic / per_launch_s / 1e9
            hbm_measured = {"GBps": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS, "bytes_per_update": traffic / per_launch_updates}
        out = {
            "metric": "variable-updates/sec (all chains), Grids_13.uai GibbsSimple",
            "value": value, "unit": "variable-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "real model file res/Grids_13.uai (vendored fixture), random start states, seed 42",
            "config": {"workload": f"{MODEL} GibbsSimple, {CHAINS_PER_GPU} chains per GPU, {SWEEPS_PER_STEP} sweeps "
                                   f"({upd_per_step} single-variable updates per chain) per step",
                       "chains_total": world * CHAINS_PER_GPU, "burnin_sweeps": BURNIN_SWEEPS,
                       "parallelism": f"chains sharded over {world} GPU(s), one RCCL all-reduce of 200 counters at the end",
                       "kernel": {"path": {0: "direct", 1: "memoised", 2: "memoised+plan, chain per lane",
                                           3: "memoised+plan, 4 lanes per chain"}[info["path"]]}},
            "hellinger_vs_MAR": {"mean": es["MeanHellinger"], "max": es["MaxHellinger"], "mean_abs_err": es["MeanMeanAbsError"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "hbm_measured": hbm_measured, "limiter": limiter,
                         "note": f"achieved = ALGORITHMIC bytes (SURVEY 8d: {ALGO_BYTES_PER_UPDATE} B/update, no reuse assumed) x "
                                 f"{per_launch_updates:.0f} updates/launch / {per_launch_s * 1e3:.3f} ms mean launch (k_scan_plan + sweep "
                                 "kernel, HIP events on the sampler stream).  frac > 1 does NOT mean HBM is saturated: it means the "
                                 "kernel never streams those bytes -- state and tables stay in LDS for the whole launch.  The honest "
                                 "pair is hbm_measured (rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command, profiles/hbm_traffic.json, "
                                 "gfx950 corrections applied: what HBM really carries) and limiter (the resource that is actually "
                                 "near its peak: VALU issue slots)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res_dir)
            out["parity_probe"] = parity_probe(res_dir, g)
            # the same workload with the reference's exact selection rule (every chain draws its own variable,
            # sampler.go:135-174) instead of one draw per 64-chain group: a reported side figure
            strict = g.HIPChains(model, CHAINS_PER_GPU, device=local_rank, seed=42, flags=g.GR_FLAG_SCAN_PER_CHAIN)
            strict.BurnIn(BURNIN_SWEEPS * n_vars)
            strict.Advance(upd_per_step)
            out["strict_scan"] = {"value": CHAINS_PER_GPU * upd_per_step / (strict.info()["kernel_ms_last"] * 1e-3),
                                  "unit": "variable-updates/s", "note": "GR_FLAG_SCAN_PER_CHAIN, kernel time of one step"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
