"""GR_FLAG_SCAN_SYSTEMATIC ("chromatic/systematic scan option"), run with -m gpu.

The systematic scan replaces the reference's variable selection (UniformSampler.VarSample, sampler/sampler.go:135-174) by a
fixed order -- colour class by colour class of a greedy colouring of the factor graph -- so its chain law is NOT the
reference's; what it shares with it is every conditional (same thresholds as the random-scan kernels, checked against the oracle
in test_gpu_parity.py) and therefore the stationary law.  Evidence here:
  1. bit-for-bit replay by the numpy emulator (order, draw stream, counting), chromatic kernel and generic kernel alike;
  2. the chromatic kernel (path 9) == the generic kernel's systematic mode, at sizes the emulator cannot reach;
  3. exact long-run histogram of the systematic chain built from ORACLE conditionals (tests/exact_chain.py) on tiny models;
  4. stationary marginals vs the ORACLE's random-scan runs (two-sample z) on models that mix.
"""
import os

import numpy as np
import pytest

import grample_amd as g
from tests import exact_chain, uai_py
from tests.emulator import Emulator, systematic_order

pytestmark = pytest.mark.gpu
SYS = g.GR_FLAG_SCAN_SYSTEMATIC


def _gm(pm):
    return g.Model.from_arrays(pm.card, pm.scopes, pm.tables, pm.fixed)


@pytest.mark.parametrize("name,ev,chains,burn,steps,flags,path", [
    ("Grids_11.uai", False, 128, 13, 457, 0, 9),            # chromatic kernel; the launches start and end inside a sweep
    ("Grids_11.uai", False, 64, 250, 230, g.GR_FLAG_DIRECT, 0),  # generic kernel, systematic mode
    ("ObjectDetection_13.uai", False, 64, 7, 150, 0, 6),    # Card 16: k_scan_vars + k_sweep_pw4 on the sweep order, one Philox call per step
    ("ObjectDetection_13.uai", False, 64, 7, 150, g.GR_FLAG_CHAIN_PER_LANE, 0),  # ... and the generic kernel
    ("Pedigree_11.uai", True, 64, 5, 700, 0, 9),            # with its evidence every selectable variable is binary: records in LDS
    ("Pedigree_11.uai", False, 64, 5, 300, 0, 8),           # Card 3, arity 4: k_scan_plan + k_sweep_dir4 on the sweep order
    ("Pedigree_11.uai", False, 64, 5, 300, g.GR_FLAG_CHAIN_PER_LANE, 0),  # ... and the generic kernel
    ("CSP_12.uai", False, 64, 3, 180, 0, 8),                # Card 4, dense
    ("grid3x3", False, 192, 3, 100, 0, 9),
    ("Segmentation_11.uai", False, 64, 9, 500, 0, 9),      # 228 variables, more lane slots than registers hold: records staged in LDS (k_sweep_chroma_l)
    ("Promedus_11.uai", True, 64, 5, 900, 0, 3),           # evidence, arity 3: more than 8 lane slots -> k_scan_plan_wide + k_sweep_wide on the sweep order
    ("Promedus_11.uai", True, 64, 5, 900, g.GR_FLAG_DIRECT, 0),  # ... and the generic kernel
    ("mixed_evidence", False, 64, 2, 41, 0, 0),
])
def test_systematic_sweep_bitwise_vs_emulator(res_dir, name, ev, chains, burn, steps, flags, path):
    from tests.test_gpu_parity import _thresholds_from_device
    pm = uai_py.load(res_dir, name, ev) if name.endswith(".uai") else exact_chain.tiny_models()[name]
    s = g.HIPChains(_gm(pm), chains, seed=321, chain_offset=64 * 7, flags=SYS | flags)
    assert s.info()["path"] == path
    order, color = systematic_order(pm.card, pm.fixed, pm.scopes)
    for sc in pm.scopes:  # a proper colouring: no factor holds two selectable variables of one colour
        cs = [color[v] for v in sc if v in color]
        assert len(cs) == len(set(cs))
    free = [v for v in range(len(pm.card)) if pm.fixed[v] < 0]
    em = Emulator(pm.card, pm.fixed, free, _thresholds_from_device(s, pm.card), chains, 321, chain_offset=64 * 7, systematic=order)
    s.BurnIn(burn); em.run(burn, False)
    s.Advance(steps); em.run(steps, True)
    s.Advance(len(order) + 1); em.run(len(order) + 1, True)
    assert np.array_equal(s.counts()[0], em.counts)
    for c in (0, 63, chains - 1):
        assert np.array_equal(s.LastSample(c), em.state[:, c])
    assert s.TotalSampleCount() == chains * (steps + len(order) + 1)


@pytest.mark.parametrize("name,chains", [("Grids_13.uai", 4096), ("Grids_11.uai", 8192), ("Segmentation_11.uai", 2048)])
def test_chromatic_kernel_equals_generic_systematic(res_dir, name, chains):
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    if name.startswith("Seg"):
        out = []
        for flags in (SYS, SYS | g.GR_FLAG_DIRECT):
            s = g.HIPChains(m, chains, seed=5, flags=flags)
            s.BurnIn(1234); s.Advance(228 * 9 + 11); s.Advance(228)
            out.append((s.info()["path"], s.counts(), s.checkpoint()["states"]))
        assert out[0][0] == 9 and out[1][0] == 0
        assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
        return
    out = []
    for flags in (SYS, SYS | g.GR_FLAG_DIRECT):
        s = g.HIPChains(m, chains, seed=5, flags=flags)
        s.BurnIn(1234); s.Advance(100 * 37 + 11); s.Advance(100)
        out.append((s.info()["path"], s.counts(), s.checkpoint()["states"]))
    assert out[0][0] == 9 and out[1][0] == 0
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert int(out[0][1].sum()) == chains * (3711 + 100)
    # every selectable variable is updated once per sweep: its counts sum to chains * sweeps (+ the partial sweeps' share)
    per_var = out[0][1][0].reshape(100, 2).sum(axis=1)
    assert per_var.min() >= chains * 38 and per_var.max() <= chains * 39


@pytest.mark.parametrize("model", ["grid3x3", "mixed", "mixed_evidence"])
def test_systematic_long_run_histogram_matches_exact_chain(oracle, model):
    """the exact stationary behaviour of the systematic-scan chain whose single-variable kernels are the ORACLE's conditionals"""
    pm = exact_chain.tiny_models()[model]
    order, _ = systematic_order(pm.card, pm.fixed, pm.scopes)
    exp = exact_chain.expected_histogram(oracle, pm, order=order)
    s = g.HIPChains(_gm(pm), 4096, seed=99, flags=SYS)
    s.BurnIn(200 * len(order)); s.Advance(600 * len(order))
    cnt = s.counts()[0]
    off = 0
    for v, e in enumerate(exp):
        c = cnt[off:off + pm.card[v]].astype(float)
        off += pm.card[v]
        if e is None:
            assert c.sum() == 0
            continue
        n = c.sum()
        assert n == 4096 * 600
        tol = 6.0 * np.sqrt(np.maximum(e * (1 - e), 1e-6) / (n / 30.0)) + 1e-4
        assert np.all(np.abs(c / n - e) <= tol), (model, v, c / n, e, tol)


@pytest.mark.parametrize("name,burn_sw,upd_sw,per", [("ObjectDetection_13.uai", 20, 60, 8), ("CSP_13.uai", 20, 60, 8), ("one.uai", 0, 200, 8)])
def test_systematic_marginals_match_oracle_random_scan(oracle, res_dir, name, burn_sw, upd_sw, per):
    """Models that mix: the systematic chain's marginal estimates agree with the oracle's random-scan runs (same burn-in and update
    counts; two-sample z over G groups as in test_marginals_match_oracle_runs), and with .MAR like they do."""
    from tests.test_gpu_parity import _group_props
    pm = uai_py.load(res_dir, name)
    n = len(pm.card)
    burn, upd = burn_sw * n, upd_sw * n
    G = 32
    s = g.HIPChains(g.Model.NewModelFromFile(os.path.join(res_dir, name)), G * 64, n_variants=G, seed=42, flags=SYS)
    s.BurnIn(burn); s.Advance(upd)
    pg = _group_props(s.counts(), pm.card)
    om = oracle.Model.from_parsed(pm)
    prior = np.concatenate([np.full(c, per / c) for c in pm.card])
    co = np.zeros((G, sum(pm.card)))
    for grp in range(G):
        mo, _, _, tot = oracle.run_chains(om, 3000 + grp, per, burn, upd)
        co[grp] = np.concatenate(mo) - prior
    po = _group_props(np.rint(co), pm.card)
    se = np.sqrt(po.var(axis=0, ddof=1) / G + pg.var(axis=0, ddof=1) / G)
    z = (po.mean(axis=0) - pg.mean(axis=0)) / np.maximum(se, 2e-4)
    assert np.max(np.abs(z)) <= 5.5 and np.mean(z ** 2) <= 2.0, (name, np.max(np.abs(z)), np.mean(z ** 2))
    sol = uai_py.load_mar(res_dir, name)
    split = lambda flat: g.split(flat, pm.card)  # noqa: E731
    eo = g.NewErrorSuite(pm.card, sol, split(po.mean(axis=0) + 1e-12))
    eg = g.NewErrorSuite(pm.card, sol, split(pg.mean(axis=0) + 1e-12))
    assert abs(eo["MeanHellinger"] - eg["MeanHellinger"]) <= 0.02, (eo, eg)


@pytest.mark.parametrize("name,ev,chains,path", [("ObjectDetection_13.uai", False, 2048, 6), ("ObjectDetection_11.uai", False, 1024, 6), ("CSP_13.uai", False, 2048, 8),
                                                  ("Alchemy_11.uai", False, 1024, 8), ("Pedigree_11.uai", False, 1024, 8),
                                                  ("Promedus_14.uai", False, 1024, 3), ("Promedus_14.uai", False, 40960, 3), ("Promedus_11.uai", True, 2048, 3),
                                                  ("Promedus_17.uai", True, 512, 3), ("dv-rel_2.uai", False, 1024, 11), ("dv-rel_2.uai", True, 512, 11)])  # (11: k_sweep_bits, the state-dominated model)
def test_plan_based_kernels_equal_generic_systematic(res_dir, name, ev, chains, path):
    """The systematic scan on the plan-based kernels (k_sweep_pw4: path 6, k_sweep_dir4: path 8, k_sweep_wide with 8 or 16 lanes
    per chain: path 3 -- the plan follows the sweep order, draws are keyed by (sweep, position)) is the generic kernel's
    systematic mode bit for bit: launches that start and end inside a quad and inside a sweep, 64-bit chain ids, several sweeps."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name), ev)
    n = m.n_vars
    out = []
    for flags in (SYS, SYS | (g.GR_FLAG_DIRECT if path in (3, 11) else g.GR_FLAG_CHAIN_PER_LANE)):
        s = g.HIPChains(m, chains, seed=9, chain_offset=(1 << 35) + 64 * 3, flags=flags)
        s.BurnIn(45); s.Advance(3 * n + 7); s.Advance(66); s.Advance(2 * n)
        out.append((s.info()["path"], s.counts(), s.checkpoint()["states"]))
        s.close()
    assert out[0][0] == path and out[1][0] == 0
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert int(out[0][1].sum()) == chains * (5 * n + 73)


@pytest.mark.parametrize("name,path", [("CSP_13.uai", 8), ("Alchemy_11.uai", 8), ("Promedus_14.uai", 3), ("Segmentation_11.uai", 9)])  # (9: the chromatic kernel takes groups of different sweep lengths)
def test_systematic_plan_kernels_with_collapsed_groups(res_dir, name, path):
    """Systematic scan + collapsed chain groups (each group its own order: the collapsed variable leaves it, the COLLAPSE factor
    re-colours its neighbours; groups of different sweep lengths in one launch): k_sweep_dir4 with chained entries and
    k_sweep_wide equal the generic kernel bit for bit -- counts, states, the two halves of gr_advance_window, resume from a
    checkpoint taken inside a sweep."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    out = []
    for flags in (SYS, SYS | g.GR_FLAG_DIRECT | g.GR_FLAG_CHAIN_PER_LANE):
        s = g.HIPChains(m, 768, n_variants=3, seed=4, flags=flags)
        cv = [s.Collapse(k, -1)[0] for k in (1, 2)]
        s.BurnIn(29); s.Advance(2 * m.n_vars + 5)
        path_seen = s.info()["path"]
        ck = s.checkpoint()
        s.AdvanceWindow(150)
        a, b, cw = s.window_halves()
        t = g.HIPChains(m, 768, n_variants=3, seed=4, flags=flags)
        for k, v in zip((1, 2), cv):
            t.Collapse(k, v)
        t.restore(ck); t.AdvanceWindow(150)
        assert np.array_equal(t.counts(), s.counts())
        out.append((path_seen, cv, s.counts(), s.checkpoint()["states"], a, b))
        s.close(); t.close()
    assert out[0][0] == path and out[1][0] == 0 and out[0][1] == out[1][1]
    for i in (2, 3, 4, 5):
        assert np.array_equal(out[0][i], out[1][i])
    assert out[0][4].sum() + out[0][5].sum() == 768 * 150  # every update of the window lands in one half


@pytest.mark.parametrize("name,chains,path", [("Grids_13.uai", 65536, 9), ("ObjectDetection_13.uai", 32768, 6), ("CSP_13.uai", 32768, 8),
                                               ("Promedus_14.uai", 32768, 3), ("Segmentation_11.uai", 32768, 3), ("Segmentation_11.uai", 8192, 9)])  # (Segmentation_11: records in LDS -- chromatic kernel up to 8 192 chains, the 8-lane form beyond)
def test_systematic_full_size_properties(res_dir, name, chains, path):
    """The systematic scan at the BASELINE sizes, size-independent properties: conservation, every selectable variable updated
    once per sweep, determinism, independence of the launch chunking (launches that start and end inside a sweep) and of the
    split of the chains over devices (two shards with chain_offset == one sampler)."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    n = m.n_vars
    upd = 6 * n

    def run(chunks, nch=chains, off=0):
        s = g.HIPChains(m, nch, seed=7, chain_offset=off, flags=SYS)
        s.BurnIn(n + 3)
        for c in chunks:
            s.Advance(c)
        return s

    a = run([upd])
    assert a.info()["path"] == path
    ca = a.counts()[0]
    assert int(ca.sum()) == chains * upd == a.TotalSampleCount()
    cards = m.cards()
    hoff = np.concatenate([[0], np.cumsum(cards)])
    per_var = np.array([ca[hoff[v]:hoff[v + 1]].sum() for v in range(n)])
    assert set(per_var.tolist()) == {chains * 6}  # 6 whole sweeps counted (no evidence in these models): every variable 6 times per chain
    assert np.array_equal(ca, run([upd]).counts()[0])
    assert np.array_equal(ca, run([7, upd - 7 - 2 * n - 1, 2 * n + 1]).counts()[0])
    h = chains // 2
    parts = [run([upd], h, k * h) for k in range(2)]
    assert np.array_equal(ca, parts[0].counts()[0] + parts[1].counts()[0])
    assert np.array_equal(a.LastSample(h + 5), parts[1].LastSample(5))


@pytest.mark.parametrize("name,chains,groups", [("Grids_13.uai", 512, 1), ("Segmentation_11.uai", 1024, 4), ("CSP_13.uai", 512, 2), ("ObjectDetection_13.uai", 256, 1)])
def test_advance_chain_under_the_systematic_scan(res_dir, name, chains, groups):
    """gr_advance_chain with GR_FLAG_SCAN_SYSTEMATIC: cw + 1 sweeps of the longest order (every variable gets its cw + 1 new samples,
    chain.go:184-215), ring halves = the last 2 * (cw / 2) sweeps split in time -- exactly cw / 2 selections of every variable per chain
    in each half for the groups that sweep n_max variables (circular.go:16-28), and gr_adapt works on them."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    s = g.HIPChains(m, chains, n_variants=groups, seed=3, flags=SYS)
    cv = [s.Collapse(k, -1)[0] for k in range(1, groups)]
    s.BurnIn(17)
    cw = 21
    c0 = s.counts().copy()
    L = s.AdvanceChain(cw)
    n_max = m.n_vars  # group 0 keeps the base model
    assert L == (cw + 1) * n_max
    c1 = s.counts()
    assert int((c1 - c0).sum()) == chains * L
    a, b, w = s.window_halves()
    assert w == cw
    cards = m.cards()
    hoff = np.concatenate([[0], np.cumsum(cards)])
    cpv = chains // groups
    h = cw // 2
    for v in range(m.n_vars):  # group 0: exactly h selections of every variable per chain and half
        assert a[0][hoff[v]:hoff[v + 1]].sum() == cpv * h and b[0][hoff[v]:hoff[v + 1]].sum() == cpv * h
    for k in range(1, groups):  # a group with a collapsed variable sweeps n_max - 1 variables: h or h + 1 selections per half, none of the collapsed one
        for v in range(m.n_vars):
            tot = a[k][hoff[v]:hoff[v + 1]].sum()
            assert (tot == 0) if v == cv[k - 1] else (cpv * h <= tot <= cpv * (h + 1))
    assert np.all((a + b) <= (c1 - c0))  # the halves are part of the window's counts
    s.close()
    if groups > 1:  # the adaptive flow on these windows: 2 live groups, Adapt adds collapsed ones, the next window covers them
        t = g.HIPChains(m, 6 * 128, n_variants=6, seed=3, flags=SYS)
        t.set_active_groups(2); t.BurnIn(50)
        t.AdvanceChain(cw)
        added, conv = t.Adapt(3)
        assert len(added) == 3 and t.active_groups() == 5 and np.isfinite(conv[[v for v in range(m.n_vars) if v not in added]]).any()
        L2 = t.AdvanceChain(cw)
        assert L2 == (cw + 1) * m.n_vars
        merged, coll = g.MergeChains([t])
        assert int(np.sum(coll)) == len(set(added))
        t.close()


def test_systematic_flag_rules(res_dir):
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "Grids_11.uai"))
    with pytest.raises(g.GrError):
        g.HIPChains(m, 64, flags=SYS | g.GR_FLAG_SCAN_PER_CHAIN)
    s = g.HIPChains(m, 256, flags=SYS, seed=1)
    s.AdvanceWindow(400)  # the halves of the window in time
    a, b, _ = s.window_halves()
    assert a.sum() + b.sum() == 256 * 400
    # the default (random scan) is untouched by the flag's existence: same counts as ever for the same seed
    d1 = g.HIPChains(m, 256, seed=1); d1.Advance(400)
    d2 = g.HIPChains(m, 256, seed=1, flags=SYS | g.GR_FLAG_DIRECT); d2.Advance(400)
    assert not np.array_equal(d1.counts(), d2.counts())
    # checkpoint / resume inside a sweep
    t = g.HIPChains(m, 256, flags=SYS, seed=1)
    t.BurnIn(37); t.Advance(255)
    ck = t.checkpoint()
    t.Advance(333)
    u = g.HIPChains(m, 256, flags=SYS, seed=1)
    u.restore(ck); u.Advance(333)
    assert np.array_equal(t.counts(), u.counts())


def test_systematic_scan_after_collapse_vs_emulator(res_dir):
    """GibbsCollapsed + systematic scan: the collapsed variable leaves the order, the COLLAPSE factor makes its former neighbours
    mutually adjacent (they may no longer share a colour); replayed bit for bit, chromatic and generic kernels."""
    from tests.test_gpu_parity import _thresholds_from_device
    pm = exact_chain.tiny_models()["grid3x3"]
    for flags in (SYS, SYS | g.GR_FLAG_DIRECT):
        s = g.HIPChains.NewGibbsCollapsed(_gm(pm), 128, 1, seed=77, flags=flags)
        v, marg = s.Collapse(0, 4)  # the centre of the 3 x 3 grid: neighbours 1, 3, 5, 7
        assert v == 4 and s.info()["path"] in (0, 9)  # (5 colour classes after the collapse: more lane slots than the chromatic kernel takes)
        scopes = [sc for sc in pm.scopes if 4 not in sc] + [[1, 3, 5, 7]]
        order, color = systematic_order(pm.card, pm.fixed, scopes, collapsed={4})
        assert 4 not in order and len({color[u] for u in (1, 3, 5, 7)}) == 4
        em = Emulator(pm.card, pm.fixed, [u for u in range(9) if u != 4], _thresholds_from_device(s, pm.card), 128, 77, systematic=order)
        from tests.emulator import init_state
        em.state = init_state(pm.card, pm.fixed, 128, 77, 0, epoch=1)  # Collapse re-draws the start state (gibbs-simple.go:132-142)
        s.BurnIn(11); em.run(11, False)
        s.Advance(203); em.run(203, True)
        assert np.array_equal(s.counts()[0], em.counts)
        assert s.counts()[0][8:10].sum() == 0  # the collapsed variable is never sampled
        for c in (0, 127):
            got, want = s.LastSample(c), em.state[:, c]
            assert np.array_equal(np.delete(got, 4), np.delete(want, 4))
