"""The multi-GPU half of the path, executed on ONE MI355X (run with -m gpu):
  * the path's only collective -- ncclAllReduce(sum, uint64) of the marginal histograms inside
    gr_allreduce_counts -- forced through RCCL with a one-device communicator;
  * bench.py with its process group on backend "nccl" (= RCCL), world size 1;
  * BASELINE config 4 at its stated shape: ObjectDetection_13, 262 144 chains in 8 shards of 32 768
    (chain_offset = k * 32 768, the split bench.py makes over 8 GPUs), merged like MergeChains
    (chain.go:96-148) == one 262 144-chain sampler, bit for bit.
The RCCL checks run in fresh subprocesses: GR_ALLREDUCE_FORCE_RCCL is read once per process, and the
library binds /opt/rocm's librccl.so itself, which must not meet the copy PyTorch brings."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import grample_amd as g

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RCCL_SCRIPT = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
import grample_amd as g
res = os.path.join(sys.argv[1], "tests", "golden", "res")
out = {}
for name, chains, groups in (("Grids_13.uai", 4096, 1), ("Segmentation_11.uai", 512, 2), ("ObjectDetection_13.uai", 256, 1)):
    m = g.Model.NewModelFromFile(os.path.join(res, name))
    def build():
        s = g.HIPChains(m, chains, n_variants=groups, seed=11)
        if groups > 1:
            s.Collapse(0, 3); s.Collapse(1, 9)
        s.BurnIn(50); s.Advance(700)
        return s
    s, twin = build(), build()
    before = g.rccl_calls()
    cnt = s.counts()
    red = g.allreduce_counts([s])
    ran = g.rccl_calls() - before
    merged, coll = g.MergeChains([s])
    merged_t, coll_t = g.MergeChains([twin])
    s.Advance(300); twin.Advance(300)
    out[name] = {"rccl_calls": ran, "reduced_equals_counts": bool(np.array_equal(red, cnt)),
                 "counts_unchanged": bool(np.array_equal(s.counts() - 0, twin.counts())),
                 "merge_unchanged": bool(all(np.array_equal(a, b) for a, b in zip(merged, merged_t)) and np.array_equal(coll, coll_t)),
                 "sum": int(red.sum()), "expect": int(chains * 700 - (chains // groups * 700 // 1 if False else 0))}
print("RESULT " + json.dumps(out))
"""


def test_allreduce_counts_runs_rccl_on_one_device():
    """gr_allreduce_counts with GR_ALLREDUCE_FORCE_RCCL=1: dlopen(librccl) -> ncclCommInitAll(1 device) -> ncclAllReduce
    (uint64, sum, out of place) -> the total equals gr_counts; the sampler's own counts, its merge and its later
    updates are what an untouched twin sampler gives."""
    env = dict(os.environ, GR_ALLREDUCE_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[7:])
    assert set(res) == {"Grids_13.uai", "Segmentation_11.uai", "ObjectDetection_13.uai"}
    for name, r in res.items():
        assert r["rccl_calls"] == 1, (name, r)          # the collective really ran (not the n == 1 shortcut)
        assert r["reduced_equals_counts"] and r["counts_unchanged"] and r["merge_unchanged"], (name, r)
        assert r["sum"] > 0


def test_allreduce_counts_default_is_the_shortcut(res_dir):
    """without the switch one sampler takes no RCCL call (nothing to exchange)"""
    if os.environ.get("GR_ALLREDUCE_FORCE_RCCL") == "1":
        pytest.skip("switch set for the whole test process")
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "Grids_11.uai"))
    s = g.HIPChains.NewGibbsSimple(m, 256, seed=3)
    s.Advance(100)
    before = g.rccl_calls()
    assert np.array_equal(g.allreduce_counts([s]), s.counts())
    assert g.rccl_calls() == before
    t = g.HIPChains.NewGibbsSimple(m, 256, seed=4)
    with pytest.raises(g.GrError):  # two samplers on one device: refused before any communicator is made
        g.allreduce_counts([s, t])


def test_bench_world1_over_rccl():
    """bench.py --gpus 1 with BENCH_FORCE_DIST=1: process group "nccl" (RCCL) of world size 1 -- barrier, the count all-reduce and
    the max-over-ranks timing all go through RCCL; the JSON line must be the headline's."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-4000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["chains_total"] == 65536 and d["value"] > 1e10
    assert d["collective"]["backend"] == "nccl" and d["collective"]["world"] == 1 and d["collective"]["allreduce_calls"] >= 2
    # 2 timed steps + 1 warm-up step of 100 000 updates on 65 536 chains: every update is in the reduced histogram
    assert d["collective"]["reduced_total"] == 65536 * 100000 * 3


def test_objectdetection13_eight_shards_equal_one_sampler(res_dir):
    """BASELINE config 4 as stated: 262 144 ObjectDetection_13 chains as 8 shards of 32 768 (chain_offset = k * 32 768: exactly the
    samplers 8 ranks build), run one after another on this GPU, then MergeChains over the 8 == the single 262 144-chain
    sampler, bit for bit; the per-shard counts sum to the single sampler's counts."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "ObjectDetection_13.uai"))
    n, per, K = m.n_vars, 32768, 8
    burn, upd = 3 * n + 1, 20 * n + 2  # not multiples of the kernels' quad / iteration sizes: heads and tails included
    whole = g.HIPChains.NewGibbsSimple(m, per * K, seed=42)
    assert whole.info()["path"] == 6  # k_sweep_pw4, the kernel config 4 is measured on
    whole.BurnIn(burn); whole.Advance(upd)
    shards, total = [], np.zeros_like(whole.counts())
    for k in range(K):
        s = g.HIPChains.NewGibbsSimple(m, per, seed=42, chain_offset=k * per)
        assert s.info()["path"] == 6
        s.BurnIn(burn); s.Advance(upd)
        total += s.counts()
        shards.append(s)
    assert int(total.sum()) == per * K * upd
    assert np.array_equal(total, whole.counts())
    merged_s, coll_s = g.MergeChains(shards)
    merged_w, coll_w = g.MergeChains([whole])
    assert not coll_s.any() and not coll_w.any()
    for a, b in zip(merged_s, merged_w):
        assert np.array_equal(a, b)
    # last states of shard k == chains [k * per, (k + 1) * per) of the whole run
    for k in (0, 5, 7):
        for c in (0, 77, per - 1):
            assert np.array_equal(shards[k].LastSample(c), whole.LastSample(k * per + c))
    sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, "ObjectDetection_13.uai.MAR")).marginals()
    es = g.NewErrorSuite(m.cards(), sol, merged_s)
    assert es["MeanHellinger"] < 0.1  # 262 144 chains x 20 sweeps (it mixes: 0.02 after long runs)
