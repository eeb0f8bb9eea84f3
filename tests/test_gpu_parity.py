"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the
C ABI (include/grample_hip.h); the oracle (oracle/) is only the checker.

Layers of evidence, strongest first:
  1. conditional law: device conditional of (variable, state) == oracle's
     GibbsSimple.SampleVar weights, entry by entry (deterministic);
  2. process replay: an independent numpy emulator of the device process
     (Philox streams, scan, threshold draws, counting) reproduces states and
     counts BIT FOR BIT;
  3. exact long-run law of the reference's chain on tiny models (transition
     matrix built from oracle conditionals) vs GPU histograms;
  4. statistical agreement with the oracle's own multi-chain runs and with
     res/*.uai.MAR on the BASELINE models;
  5. size-independent properties at full BASELINE sizes (determinism, chunk /
     split independence, conservation, path equivalence).
"""
import ctypes as C
import os

import numpy as np
import pytest

import grample_amd as g
from tests import exact_chain, uai_py
from tests.emulator import Emulator

pytestmark = pytest.mark.gpu


def _gm(pm):
    return g.Model.from_arrays(pm.card, pm.scopes, pm.tables, pm.fixed)


def _norm(ms):
    return [np.asarray(m, dtype=float) / np.sum(m) for m in ms]


def _oracle_cond(oracle, samp, v, st, card):
    p = np.zeros(card)
    st = np.ascontiguousarray(st, dtype=np.int32)
    rc = oracle.lib().orc_simple_conditional(samp, v, st.ctypes.data_as(C.POINTER(C.c_int)), p.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return p


# tolerance of a device conditional: thresholds are multiples of 2^-32 (abs 2.4e-10), log potentials and exp are f32
COND_ABS, COND_REL = 4e-10, 5e-5


# ------------------------------------------------------------ 1. conditionals
@pytest.mark.parametrize("name,ev", [("one.uai", False), ("sample.uai", False), ("deterministic.uai", False),
                                     ("Grids_11.uai", False), ("Grids_13.uai", False), ("Segmentation_11.uai", False),
                                     ("ObjectDetection_13.uai", False), ("Promedus_11.uai", True),
                                     ("Pedigree_11.uai", True), ("CSP_12.uai", False),
                                     ("ObjectDetection_11.uai", False),  # Card 11 < the kernel's 16-wide rows
                                     ("Alchemy_11.uai", False),          # blankets up to 60 variables, arity 3
                                     ("dv-rel_2.uai", True),             # 1200 variables, arity 4, 307 evidence variables
                                     ("dv-rel_3.uai", True),             # arity 10, 1024-entry tables (400 KB of tables: not LDS-resident)
                                     ("CSP_13.uai", False),              # Card 4, 31 factors on one variable
                                     ("Promedus_17.uai", True),          # 916 variables
                                     ("Promedus_14.uai", True),          # up to 12 neighbours: the widest 64-byte packed records
                                     ("CSP_11.uai", False)])             # pairwise, Card 4, 29 neighbours on one variable
def test_conditional_matches_oracle(oracle, res_dir, name, ev):
    """a4-a8: gather + log-sum + stabilise/exp/floor + normalise (gibbs-simple.go:175-258)."""
    pm = uai_py.load(res_dir, name, ev)
    L = oracle.lib()
    om = oracle.Model.from_parsed(pm)
    gen = L.orc_gen_new(1)
    osamp = L.orc_simple_new(gen, om.ptr)
    s = g.HIPChains.NewGibbsSimple(g.Model.NewModelFromFile(os.path.join(res_dir, name), ev), 64)
    rng = np.random.default_rng(0)
    free = [v for v in range(len(pm.card)) if pm.fixed[v] < 0]
    hit_floor = knife = 0
    for _ in range(300):
        st = np.array([rng.integers(0, c) if f < 0 else f for c, f in zip(pm.card, pm.fixed)], dtype=np.int32)
        v = int(rng.choice(free))
        pd = s.conditional(0, v, st)
        po = _oracle_cond(oracle, osamp, v, st, pm.card[v])
        assert abs(pd.sum() - 1.0) < 1e-12
        bad = np.abs(pd - po) > COND_ABS + COND_REL * po
        if bad.any():
            # Knife edge of the floor rule (gibbs-simple.go:247-255): a value whose probability is 1e-6 up to
            # rounding (tables written with literal 1e-06 entries, e.g. Promedus) is floored, i.e. doubled to
            # 2e-6, or not, depending on the last bits of exp/log -- in f64 as much as in f32.
            # Either outcome is accepted there and nowhere else.
            edge = lambda p: np.minimum(np.abs(p / 1e-6 - 1.0), np.abs(p / 2e-6 - 1.0)) < 2e-4
            assert np.all(edge(pd[bad]) & edge(po[bad])), (name, v, pd, po)
            knife += 1
        hit_floor += int(po.min() < 2e-6)
    assert knife <= 15, (name, knife)
    if name == "Grids_13.uai":
        assert hit_floor > 10  # the 1e-6 floor is exercised (tables span 5.5e-5 .. 1.8e4)
    fx = [v for v in range(len(pm.card)) if pm.fixed[v] >= 0]
    if fx:  # SampleVar on evidence is an error (gibbs-simple.go:167-169)
        st = np.array([0 if f < 0 else f for f in pm.fixed], dtype=np.int32)
        with pytest.raises(g.GrError):
            s.conditional(0, fx[0], st)
    L.orc_simple_free(osamp); L.orc_gen_free(gen)


def test_conditional_exhaustive_tiny(oracle):
    """Every (variable, state) of the tiny synthetic models, incl. table zeros and floored rows."""
    import itertools
    for name, pm in exact_chain.tiny_models().items():
        L = oracle.lib()
        om = oracle.Model.from_parsed(pm)
        gen = L.orc_gen_new(1)
        osamp = L.orc_simple_new(gen, om.ptr)
        s = g.HIPChains.NewGibbsSimple(_gm(pm), 64)
        ranges = [range(c) if f < 0 else [f] for c, f in zip(pm.card, pm.fixed)]
        for st in itertools.islice(itertools.product(*ranges), 0, 600):
            for v in range(len(pm.card)):
                if pm.fixed[v] >= 0:
                    continue
                pd = s.conditional(0, v, list(st))
                po = _oracle_cond(oracle, osamp, v, st, pm.card[v])
                assert np.all(np.abs(pd - po) <= COND_ABS + COND_REL * po), (name, v, st, pd, po)
        L.orc_simple_free(osamp); L.orc_gen_free(gen)


def _random_model(rng):
    """Random factor graph: cards 1..6, arity 1..5, zeros / 1e-6 / 1e-9 table entries (table epsilon rule, floored
    conditionals), 0-30 % evidence.  Variable n-1 only ever gets a unary factor."""
    from types import SimpleNamespace
    n = int(rng.integers(3, 40))
    card = [int(c) for c in rng.choice([1, 2, 2, 2, 3, 4, 5, 6], size=n)]
    scopes, tables = [], []
    for _ in range(int(rng.integers(n // 2, 3 * n))):
        k = int(min(n, rng.integers(1, 6)))
        sc = [int(v) for v in rng.choice(n - 1, size=min(k, n - 1), replace=False)]  # variable n-1 never gets a factor
        size = int(np.prod([card[v] for v in sc]))
        if size > 4096:
            continue
        t = np.exp(rng.uniform(-3, 3, size))
        u = rng.random(size)
        t[u < 0.10] = 0.0
        t[(u >= 0.10) & (u < 0.14)] = 1e-6
        t[(u >= 0.14) & (u < 0.17)] = 1e-9
        scopes.append(sc); tables.append(t)
    covered = {v for sc in scopes for v in sc}
    for v in range(n):  # a variable in no factor is an error in the reference (gibbs-simple.go:96-99), tested separately
        if v not in covered:
            scopes.append([v]); tables.append(np.exp(rng.uniform(-1, 1, card[v])))
    fixed = [int(rng.integers(0, c)) if rng.random() < rng.choice([0.0, 0.1, 0.3]) else -1 for c in card]
    if all(f >= 0 for f in fixed):
        fixed[0] = -1
    return SimpleNamespace(card=card, scopes=scopes, tables=tables, fixed=fixed)


def test_random_models_conditional_and_replay_vs_oracle(oracle):
    """N4 generality: 30 random models through whatever kernel path the library picks -- device conditionals vs the
    oracle's (same tolerance as the fixtures), and for the first 10 a bitwise replay by the numpy emulator."""
    rng = np.random.default_rng(2024)
    L = oracle.lib()
    paths = set()
    from types import SimpleNamespace
    lonely = SimpleNamespace(card=[2, 2, 3], scopes=[[0, 1]], tables=[np.ones(4)], fixed=[-1, -1, -1])
    gen = L.orc_gen_new(1)
    assert not L.orc_simple_new(gen, oracle.Model.from_parsed(lonely).ptr)  # "There are no functions for var" on both sides
    L.orc_gen_free(gen)
    with pytest.raises(g.GrError, match="no functions for var"):
        g.HIPChains(_gm(lonely), 64)
    for mi in range(30):
        pm = _random_model(rng)
        om = oracle.Model.from_parsed(pm)
        gen = L.orc_gen_new(1)
        osamp = L.orc_simple_new(gen, om.ptr)
        assert osamp, oracle.err()
        s = g.HIPChains(_gm(pm), 64, seed=mi)
        paths.add(s.info()["path"])
        free = [v for v in range(len(pm.card)) if pm.fixed[v] < 0]
        knife = 0
        for _ in range(40):
            st = np.array([rng.integers(0, c) if f < 0 else f for c, f in zip(pm.card, pm.fixed)], dtype=np.int32)
            v = int(rng.choice(free))
            pd = s.conditional(0, v, st)
            po = _oracle_cond(oracle, osamp, v, st, pm.card[v])
            bad = np.abs(pd - po) > COND_ABS + COND_REL * po
            if bad.any():  # floor knife edge, see test_conditional_matches_oracle
                edge = lambda p: np.minimum(np.abs(p / 1e-6 - 1.0), np.abs(p / 2e-6 - 1.0)) < 2e-4
                assert np.all(edge(pd[bad]) & edge(po[bad])), (mi, v, pd, po)
                knife += 1
        assert knife <= 4, (mi, knife)
        if mi < 10:
            em = Emulator(pm.card, pm.fixed, free, _thresholds_from_device(s, pm.card), 64, mi)
            assert np.array_equal(s.LastSample(63), em.state[:, 63])
            s.BurnIn(5); em.run(5, False)
            s.Advance(38); em.run(38, True)
            assert np.array_equal(s.counts()[0], em.counts), mi
            for c in (0, 31, 63):
                assert np.array_equal(s.LastSample(c), em.state[:, c]), mi
        L.orc_simple_free(osamp); L.orc_gen_free(gen)
    assert len(paths) >= 2, paths  # the draw exercises more than one kernel path


# --------------------------------------------------------- 2. process replay
def _thresholds_from_device(s, cards):
    cache = {}

    def thr(v, col):
        key = (v, tuple(int(x) for x in col))
        if key not in cache:
            p = s.conditional(0, v, np.asarray(col, dtype=np.int32))
            cache[key] = np.rint(np.cumsum(p)[:-1] * 4294967296.0).astype(np.uint64)
        return cache[key]

    return thr


@pytest.mark.parametrize("name,ev,chains,steps,flags", [
    ("Grids_11.uai", False, 128, 203, 0),
    ("Grids_11.uai", False, 128, 203, g.GR_FLAG_DIRECT),
    ("Grids_11.uai", False, 70, 57, g.GR_FLAG_SEQUENTIAL),  # ragged last wavefront
    ("ObjectDetection_13.uai", False, 64, 41, 0),
    ("Pedigree_11.uai", True, 64, 61, 0),
    ("one.uai", False, 1, 64, 0),
    ("dv-rel_2.uai", True, 64, 61, 0),   # 1 200 variables: the 4-lane form's lean LDS layout (records from global memory)
    ("Grids_11.uai", False, 128, 75, g.GR_FLAG_STATE_GLOBAL),           # chain state left in global memory, memoised tables
    ("ObjectDetection_13.uai", False, 70, 29, g.GR_FLAG_STATE_GLOBAL),  # ... direct pairwise path, ragged wavefront
    ("Pedigree_11.uai", True, 64, 37, g.GR_FLAG_STATE_GLOBAL | g.GR_FLAG_DIRECT),  # ... direct generic path, evidence
    ("Promedus_14.uai", True, 64, 53, 0),  # 12 neighbours: packed records at their widest
    ("CSP_11.uai", False, 64, 37, 0),      # pairwise Card-4 programs on the generic direct kernel
    ("dv-rel_3.uai", False, 64, 45, 0),  # 512 K threshold words: the 4-lane form with tables read from global memory
    ("Pedigree_11.uai", False, 64, 33, 0),  # Card 3, arity 4, memoisable: the step-parallel direct kernel (path 8)
    ("Grids_11.uai", False, 128, 97, g.GR_FLAG_SCAN_PER_CHAIN),       # strict mode: per-chain variable draws
    ("ObjectDetection_13.uai", False, 64, 33, g.GR_FLAG_SCAN_PER_CHAIN),
    ("Pedigree_11.uai", True, 70, 45, g.GR_FLAG_SCAN_PER_CHAIN),
])
def test_sweep_bitwise_vs_numpy_emulator(res_dir, name, ev, chains, steps, flags):
    pm = uai_py.load(res_dir, name, ev)
    off = 64 * 5 if chains % 64 == 0 else 0
    s = g.HIPChains(g.Model.NewModelFromFile(os.path.join(res_dir, name), ev), chains, seed=1234, chain_offset=off, flags=flags)
    free = [v for v in range(len(pm.card)) if pm.fixed[v] < 0]
    em = Emulator(pm.card, pm.fixed, free, _thresholds_from_device(s, pm.card), chains, 1234, chain_offset=off,
                  per_chain=bool(flags & g.GR_FLAG_SCAN_PER_CHAIN))
    for c in (0, chains - 1):  # start state (gibbs-simple.go:84-112)
        assert np.array_equal(s.LastSample(c), em.state[:, c])
    burn = 13
    s.BurnIn(burn); em.run(burn, False)
    a = steps // 2
    s.Advance(a); s.Advance(steps - a)  # two launches, quad-unaligned boundary
    em.run(steps, True)
    assert np.array_equal(s.counts()[0], em.counts)
    for c in range(chains):
        assert np.array_equal(s.LastSample(c), em.state[:, c])
    assert s.TotalSampleCount() == chains * steps
    assert s.info()["step"] == burn + steps


@pytest.mark.parametrize("kw", [dict(), dict(fast_form=1), dict(flags=g.GR_FLAG_DIRECT), dict(flags=g.GR_FLAG_SCAN_PER_CHAIN)])
def test_bitwise_with_64bit_chain_offset_and_seed(res_dir, kw):
    """Global chain ids and seeds above 2^32 reach the Philox counter / key words intact in every kernel form
    (a sharded job numbers its chains globally: chain_offset = rank * chains_per_rank)."""
    name, chains, off, seed = "Grids_11.uai", 128, (1 << 40) + 64 * 7, (0xDEADBEEF << 32) | 0x12345678
    pm = uai_py.load(res_dir, name)
    s = g.HIPChains(g.Model.NewModelFromFile(os.path.join(res_dir, name)), chains, seed=seed, chain_offset=off, **kw)
    em = Emulator(pm.card, pm.fixed, list(range(len(pm.card))), _thresholds_from_device(s, pm.card), chains, seed, chain_offset=off,
                  per_chain=bool(kw.get("flags", 0) & g.GR_FLAG_SCAN_PER_CHAIN))
    assert np.array_equal(s.LastSample(chains - 1), em.state[:, chains - 1])
    s.BurnIn(7); em.run(7, False)
    s.Advance(90); em.run(90, True)
    assert np.array_equal(s.counts()[0], em.counts)
    for c in (0, 64, chains - 1):
        assert np.array_equal(s.LastSample(c), em.state[:, c])


def test_all_kernel_forms_agree_bit_for_bit(res_dir):
    """Grids_13: direct (0), memoised (1), chain per lane (2) and 4 lanes per chain (3) are the same process:
    identical histograms and states after quad-unaligned burn-in and advance."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "Grids_13.uai"))
    runs = {}
    for path, kw in {0: dict(flags=g.GR_FLAG_DIRECT), 1: dict(block_threads=64), 2: dict(fast_form=1), 3: dict(fast_form=2),
                     34: dict(fast_form=4), 35: dict(fast_form=5), 36: dict(fast_form=6), 37: dict(fast_form=7),
                     38: dict(fast_form=8), 39: dict(fast_form=9), 41: dict(fast_form=11)}.items():
        s = g.HIPChains(m, 2048, seed=9, chain_offset=64 * 3, **kw)
        s.BurnIn(333); s.Advance(1001); s.Advance(2 * 4096 + 3)
        runs[path] = (s.info()["path"], s.counts(), np.stack([s.LastSample(c) for c in (0, 63, 64, 1000, 2047)]), s.info()["block_threads"])
    assert [runs[k][0] for k in sorted(runs)] in ([0, 1, 2, 3, 3, 3, 3, 3, 3, 3, 11], [0, 2, 2, 3, 3, 3, 3, 3, 3, 3, 11])
    assert [runs[k][3] for k in (36, 38, 39)] == [256, 512, 1024]  # 4, 8 and 16 lanes per chain
    for k in (1, 2, 3, 34, 35, 36, 37, 38, 39, 41):  # 34..36: the 4-lane form's three record sources; 37: split threshold pool; 38, 39: wide forms; 41: bit-packed state (k_sweep_bits)
        assert np.array_equal(runs[0][1], runs[k][1]), k
        assert np.array_equal(runs[0][2], runs[k][2]), k


@pytest.mark.parametrize("name,ev,path", [("dv-rel_3.uai", False, 3), ("dv-rel_3.uai", True, 3), ("Pedigree_11.uai", False, 1)])
def test_global_threshold_tables_equal_direct(res_dir, name, ev, path):
    """Memoised threshold pools too large for LDS are read from global memory (4-lane form for binary models, generic
    memoised kernel otherwise): same process as the direct path, bit for bit."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name), ev)
    out = []
    for flags in (0, g.GR_FLAG_DIRECT):
        # (path 1 = the chain-per-lane memoised kernel: asked for explicitly, the default for a non-binary model is path 8)
        s = g.HIPChains(m, 1024, seed=3, flags=flags, chain_offset=64, fast_form=1 if path == 1 else 0)
        s.BurnIn(50); s.Advance(203); s.Advance(64)
        i = s.info()
        out.append((i["path"], i["tables_in_lds"], s.counts(), s.checkpoint()["states"]))
    assert (out[0][0], out[0][1]) == (path, 0) and out[1][0] == 0
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])


def test_model_larger_than_lds_runs_from_global_state():
    """A 60 x 60 binary grid (3 600 variables: 225 KiB of state per 64-chain wavefront, more than the 160 KiB of LDS) is not
    refused: the sampler leaves the chain state in global memory (path 17 = memoised + global state) and the numpy
    emulator replays it bit for bit."""
    from types import SimpleNamespace
    rng = np.random.default_rng(5)
    L = 60
    n = L * L
    scopes, tables = [], []
    for v in range(n):
        scopes.append([v]); tables.append(np.exp(rng.uniform(-1, 1, 2)))
        r, c = divmod(v, L)
        if c + 1 < L:
            scopes.append([v, v + 1]); tables.append(np.exp(rng.uniform(-1.5, 1.5, 4)))
        if r + 1 < L:
            scopes.append([v, v + L]); tables.append(np.exp(rng.uniform(-1.5, 1.5, 4)))
    pm = SimpleNamespace(card=[2] * n, scopes=scopes, tables=tables, fixed=[-1] * n)
    s = g.HIPChains(_gm(pm), 128, seed=8)
    assert s.info()["path"] == 17
    em = Emulator(pm.card, pm.fixed, list(range(n)), _thresholds_from_device(s, pm.card), 128, 8)
    s.BurnIn(9); em.run(9, False)
    s.Advance(150); em.run(150, True)
    assert np.array_equal(s.counts()[0], em.counts)
    assert np.array_equal(s.checkpoint()["states"].astype(np.int64), em.state)
    # AdvanceChain on a global-state model (round 3; the small cases are replayed by the emulator in test_gpu_collapse.py): whole
    # batches of 2 N, and every variable's ring halves hold cw / 2 selections of each of the 128 chains
    Lw = s.AdvanceChain(4)
    assert Lw % (2 * n) == 0 and Lw >= 5 * n
    a, b, cw = s.window_halves()
    assert cw == 4 and np.all(a[0].reshape(n, 2).sum(axis=1) == 128 * 2) and np.all(b[0].reshape(n, 2).sum(axis=1) == 128 * 2)


def _pairwise_big_card_model(rng, n=24):
    """random pairwise model with Card up to 16, evidence, and many near-zero entries (so that most updates floor several
    values and some sit close to the 1e-6 edge): the shape the value-parallel direct kernel exists for"""
    from types import SimpleNamespace
    card = [int(rng.choice([2, 3, 5, 8, 9, 11, 12, 13, 16, 16, 16])) for _ in range(n)]
    card[0] = 16
    scopes, tables = [], []
    for v in range(n):
        scopes.append([v]); tables.append(np.exp(rng.uniform(-2, 2, card[v])))
    for _ in range(3 * n):
        a, b = (int(x) for x in rng.choice(n, 2, replace=False))
        t = np.exp(rng.uniform(-3, 3, card[a] * card[b]))
        t[rng.random(t.size) < 0.25] = rng.choice([0.0, 1e-9, 1e-6, 3e-6, 1e-5])
        scopes.append([a, b]); tables.append(t)
    fixed = [int(rng.integers(0, c)) if (v % 7 == 3) else -1 for v, c in enumerate(card)]
    return SimpleNamespace(card=card, scopes=scopes, tables=tables, fixed=fixed)


def test_results_do_not_depend_on_wavefront_priority():
    """wave_prio.h rotates the issue priority of every sweep kernel's wavefronts (the blocks of a CU then finish together: +11 % on
    the headline).  Priorities only reorder the issue of independent wavefronts, so nothing may depend on them: counts and states of
    one run per kernel family (4-lane, wide, chromatic, strict, chain per lane, collapsed groups, value-parallel, step-parallel,
    generic) are the same without rotation (GR_DEV_PRIO_SHIFT=32), rotating every loop trip (0) and with the defaults."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for shift in ("32", "0", None, "no-plan-ahead"):
        env = dict(os.environ)
        env.pop("GR_DEV_PRIO_SHIFT", None)
        env.pop("GR_DEV_PLAN_AHEAD", None)
        if shift == "no-plan-ahead":  # ... nor on whether the next launch's scan plan is computed beside the running sweep (sampler.cpp: plan-ahead)
            env["GR_DEV_PLAN_AHEAD"] = "0"
        elif shift is not None:
            env["GR_DEV_PRIO_SHIFT"] = shift
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "prio_hash.py")], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] == outs[2] == outs[3] and len(outs[0].strip().splitlines()) == 11
    assert {int(line.split()[4]) for line in outs[0].strip().splitlines()} >= {0, 2, 3, 5, 6, 8, 9}


@pytest.mark.parametrize("name,ev,chains,groups", [("Grids_13.uai", False, 65536, 1), ("Segmentation_11.uai", False, 4096, 4), ("Promedus_11.uai", True, 1024, 1), ("dv-rel_1.uai", True, 2048, 1)])
def test_bit_packed_state_form_equals_byte_form(res_dir, name, ev, chains, groups):
    """fast_form 11 (k_sweep_bits: one 16-bit mask per variable and wavefront instead of one byte per chain; SURVEY 8(f) N4 "bit-packed
    binary state") is the same process as the byte layout's kernels, bit for bit: counts and states after quad-unaligned launches, with
    evidence, with collapsed chain groups; logged AdvanceChain windows (which keep the byte layout) interleave freely."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name), ev)
    out = []
    for form in (11, 3):
        s = g.HIPChains(m, chains, n_variants=groups, seed=19, chain_offset=(1 << 33) + 128, fast_form=form)
        cv = [s.Collapse(k, -1)[0] for k in range(1, groups)]
        s.BurnIn(77); s.Advance(3 * m.n_vars + 5); s.Advance(64)
        path = s.info()["path"]
        L = s.AdvanceChain(10) if chains <= 4096 else 0
        s.Advance(129)
        out.append((path, cv, L, s.counts(), s.checkpoint()["states"]))
        s.close()
    assert out[0][0] == 11 and out[1][0] == 3 and out[0][1] == out[1][1] and out[0][2] == out[1][2]
    assert np.array_equal(out[0][3], out[1][3]) and np.array_equal(out[0][4], out[1][4])


@pytest.mark.parametrize("ev", [False, True])
def test_state_dominated_model_takes_the_bit_packed_form(res_dir, ev):
    """dv-rel_2 (1 200 variables: 96 KB of state bytes per 64-chain block) is the model whose STATE fills LDS: the sampler picks the
    bit-packed form by itself (path 11: two blocks per CU instead of one, 3.0e11 -> 4.6e11 updates/s at 32 768 chains) -- the same
    process as the generic direct kernel, bit for bit."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "dv-rel_2.uai"), ev)
    out = []
    for flags in (0, g.GR_FLAG_DIRECT):
        s = g.HIPChains(m, 512, seed=31, chain_offset=64 * 5, flags=flags)
        s.BurnIn(33); s.Advance(2 * m.n_vars + 7); s.Advance(64)
        out.append((s.info()["path"], s.counts(), s.checkpoint()["states"]))
        s.close()
    assert out[0][0] == 11 and out[1][0] == 0
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


def test_cross_form_fuzz():
    """scripts/fuzz_forms.py: 32 random models of four shapes (large-Card pairwise, sparse binary, medium-blanket binary, small
    mixed cards with arity up to 4, evidence, near-zero entries) through every kernel form the library can be steered into --
    default, chain per lane, direct, sequential, state in global memory, split threshold pool, strict memoised / direct: all bit-identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_forms.py"), "32", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("paths exercised:")
    seen = eval(last.split(":", 1)[1])
    assert {0, 1, 3, 6, 8, 16, 17, 35} <= set(seen), seen  # (35 = path 3 with the split threshold pool, fast_form 7)


@pytest.mark.parametrize("name", ["CSP_12.uai", "CSP_13.uai", "CSP_11.uai", "Alchemy_11.uai"])
def test_step_parallel_direct_form_equals_generic(res_dir, name):
    """k_sweep_dir4 (path 8: 4 lanes per chain, the conditional evaluated from the factor tables by each lane) on the dense
    small-Card fixtures -- blankets of up to 59 variables, up to 31 factors on a variable, arity 3, Card 2..4 -- is the
    same process as the generic direct kernel (GR_FLAG_DIRECT), bit for bit; quad-unaligned launches, 64-bit chain ids."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    out = []
    for flags in (0, g.GR_FLAG_DIRECT):
        s = g.HIPChains(m, 1024, seed=13, chain_offset=(1 << 34) + 128, flags=flags)
        s.BurnIn(45); s.Advance(211); s.Advance(64)
        out.append((s.info()["path"], s.counts(), s.checkpoint()["states"]))
    assert out[0][0] == 8 and out[1][0] == 0
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    assert int(out[0][1].sum()) == 1024 * 275


@pytest.mark.parametrize("name,groups", [("CSP_13.uai", 4), ("CSP_12.uai", 2), ("Alchemy_11.uai", 4), ("Pedigree_11.uai", 2)])
def test_step_parallel_direct_form_with_collapse_factors(res_dir, name, groups):
    """Collapsed chain groups of the dense models stay on k_sweep_dir4 (path 8): a collapse factor (up to 12 scope variables,
    32-bit view strides) is walked as CHAINED factor entries, two scope variables per entry, and still contributes one view
    row at its place in Model.Funcs order -- the same process as the generic direct kernel, bit for bit, plain windows and
    AdvanceChain's logged ones."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    out = []
    for flags in (0, g.GR_FLAG_DIRECT):
        s = g.HIPChains(m, 256 * groups, n_variants=groups, seed=29, chain_offset=(1 << 33) + 64, flags=flags)
        cv = [s.Collapse(k, -1)[0] for k in range(1, groups)]  # group 0 keeps the base model
        s.BurnIn(45); s.Advance(211); s.Advance(64)
        path = s.info()["path"]
        L = s.AdvanceChain(40)
        a, b, cw = s.window_halves()
        out.append((path, cv, s.counts(), s.checkpoint()["states"], L, a, b))
        s.close()
    assert out[0][0] == 8 and out[1][0] == 0 and out[0][1] == out[1][1] and len(set(out[0][1])) >= 1
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    assert out[0][4] == out[1][4] and np.array_equal(out[0][5], out[1][5]) and np.array_equal(out[0][6], out[1][6])


def test_value_parallel_form_equals_chain_per_lane(res_dir):
    """k_sweep_pw4 (4 lanes per chain, one per block of 4 values; path 6) and the chain-per-lane direct kernel (path 0) are
    the same process bit for bit: ObjectDetection_13 and random pairwise models with Card 2..16, evidence and floor-heavy
    tables; quad-unaligned launches; 64-bit chain ids."""
    rng = np.random.default_rng(77)
    cases = [(g.Model.NewModelFromFile(os.path.join(res_dir, "ObjectDetection_13.uai")), 320)]
    cases += [(_gm(_pairwise_big_card_model(rng)), 128) for _ in range(3)]
    for m, chains in cases:
        runs = []
        for form in (2, 1):
            s = g.HIPChains(m, chains, seed=31, chain_offset=(1 << 36) + 64 * 9, fast_form=form)
            s.BurnIn(37); s.Advance(101); s.Advance(66)
            runs.append((s.info()["path"], s.counts(), s.checkpoint()["states"]))
        assert runs[0][0] == 6 and runs[1][0] == 0
        assert np.array_equal(runs[0][1], runs[1][1])
        assert np.array_equal(runs[0][2], runs[1][2])
        assert int(runs[0][1].sum()) == chains * 167


@pytest.mark.parametrize("name,chains,groups", [("Grids_13.uai", 2048, 1), ("Grids_13.uai", 66560, 1), ("Segmentation_11.uai", 1024, 4)])
def test_strict_mode_memoised_equals_direct(res_dir, name, chains, groups):
    """GR_FLAG_SCAN_PER_CHAIN (every chain draws its own variable, sampler.go:135-174): the memoised strict kernel
    (path 5; one and four wavefronts per block; collapsed groups) and the generic kernel's strict mode (path 0)
    are the same process bit for bit."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    out = []
    for flags in (g.GR_FLAG_SCAN_PER_CHAIN, g.GR_FLAG_SCAN_PER_CHAIN | g.GR_FLAG_DIRECT):
        s = g.HIPChains(m, chains, n_variants=groups, seed=17, flags=flags)
        cv = [s.Collapse(k, -1)[0] for k in range(groups)] if groups > 1 else []
        s.BurnIn(41); s.Advance(50); s.Advance(83)
        out.append((s.info()["path"], cv, s.counts(), np.stack([s.LastSample(c) for c in (0, 63, chains // 2 + 5, chains - 1)])))
    assert out[0][0] == 5 and out[1][0] == 0 and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3])
    assert int(out[0][2].sum()) == chains * 133


# ------------------------------------------- 3. exact long-run law, tiny models
@pytest.mark.parametrize("model", ["sample.uai", "grid3x3", "mixed", "mixed_evidence"])
@pytest.mark.parametrize("flags", [0, g.GR_FLAG_DIRECT])
def test_histogram_converges_to_reference_chain_law(oracle, res_dir, model, flags):
    pm = uai_py.load(res_dir, model) if model.endswith(".uai") else exact_chain.tiny_models()[model]
    exp = exact_chain.expected_histogram(oracle, pm)
    chains, burn, upd = 16384, 400, 6000
    s = g.HIPChains(_gm(pm), chains, seed=99, flags=flags)
    s.BurnIn(burn); s.Advance(upd)
    cnt = s.counts()[0]
    off = 0
    for v, e in enumerate(exp):
        c = cnt[off:off + pm.card[v]].astype(float)
        off += pm.card[v]
        if e is None:
            assert c.sum() == 0  # evidence variables are never sampled
            continue
        # chains are independent; within a chain samples are correlated -> generous 6 sigma on an
        # effective sample size of n/30, plus 1e-4 absolute.  Stated tolerance: |p_gpu - p_ref| <= tol.
        n = c.sum()
        p = c / n
        tol = 6.0 * np.sqrt(np.maximum(e * (1 - e), 1e-6) / (n / 30.0)) + 1e-4
        assert np.all(np.abs(p - e) <= tol), (model, v, p, e, tol)


def test_one_uai_reference_config(oracle, res_dir):
    """BASELINE config 0 / gibbs-simple_test.go:13-38: one.uai, 1 chain, 1k updates: both values occur;
    many chains: histogram -> (0.25, 0.75) = one.uai.MAR."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "one.uai"))
    s = g.HIPChains.NewGibbsSimple(m, 1, seed=42)
    seen = set()
    for _ in range(64):
        s.Advance(16)
        seen.add(int(s.LastSample(0)[0]))
    c = s.counts()[0]
    assert c.sum() == 1024 and c[0] > 0 and c[1] > 0 and seen == {0, 1}
    s2 = g.HIPChains.NewGibbsSimple(m, 4096, seed=42)
    s2.Advance(1000)
    merged, coll = g.MergeChains([s2])
    sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, "one.uai.MAR")).marginals()
    es = g.NewErrorSuite([2], sol, merged)
    assert es["MaxHellinger"] < 1e-3 and es["MaxMeanAbsError"] < 1e-3
    # Marginal = per-chain prior 1/Card + counts (variable.go:45, chain.go:230-236)
    assert merged[0].sum() == pytest.approx(4096 * (1.0 + 1000), rel=1e-12)


# ----------------------------------------- 4. statistical parity with the oracle
def _group_props(counts, cards):
    """counts [G, H] -> per-group value shares p[G, H] (each variable's block normalised)."""
    G = counts.shape[0]
    p = np.zeros(counts.shape)
    off = 0
    for c in cards:
        blk = counts[:, off:off + c].astype(float)
        p[:, off:off + c] = blk / np.maximum(blk.sum(axis=1, keepdims=True), 1.0)
        off += c
    return p


@pytest.mark.parametrize("name,burn_sw,upd_sw,per,ev", [("Grids_11.uai", 10, 30, 16, False), ("Segmentation_11.uai", 10, 30, 8, False),
                                                        ("ObjectDetection_13.uai", 10, 40, 8, False),
                                                        ("Pedigree_11.uai", 5, 15, 4, True),   # evidence, Card 3, arity 4
                                                        ("CSP_12.uai", 10, 30, 8, False),      # Card 4, arity 3
                                                        ("Promedus_11.uai", 5, 15, 4, True),   # evidence, binary, arity 3
                                                        ("ObjectDetection_11.uai", 10, 30, 8, False),  # Card 11
                                                        ("Alchemy_11.uai", 5, 10, 4, False),   # blankets up to 60
                                                        ("dv-rel_3.uai", 4, 8, 2, True),       # arity 10, tables in HBM/L2
                                                        ("CSP_13.uai", 10, 30, 8, False)])     # Card 4, degree 31
def test_marginals_match_oracle_runs(oracle, res_dir, name, burn_sw, upd_sw, per, ev):
    """Two-sample test, oracle (MT19937-64, f64, Go semantics) vs GPU (Philox, f32 tables): same
    burn-in and update counts per chain.  Both sides are run as G independent groups; a group's
    histogram shares are i.i.d. across groups, so for every (variable, value)
        z = (mean_oracle - mean_gpu) / sqrt(se_oracle^2 + se_gpu^2)
    is ~N(0,1) if and only if the two samplers have the same law at that horizon (mixed or not).
    Stated tolerance: max |z| <= 5.5 and mean z^2 <= 2 over all (variable, value) pairs."""
    pm = uai_py.load(res_dir, name, ev)
    n = len(pm.card)
    burn, upd = burn_sw * n, upd_sw * n
    G = 32
    s = g.HIPChains(g.Model.NewModelFromFile(os.path.join(res_dir, name), ev), G * 64, n_variants=G, seed=42)
    s.BurnIn(burn); s.Advance(upd)
    pg = _group_props(s.counts(), pm.card)
    om = oracle.Model.from_parsed(pm)
    prior = np.concatenate([np.full(c, per / c) for c in pm.card])
    co = np.zeros((G, sum(pm.card)))
    for grp in range(G):
        mo, _, _, tot = oracle.run_chains(om, 1000 + grp, per, burn, upd)
        assert tot == per * upd
        co[grp] = np.concatenate(mo) - prior  # Marginal = 1/Card prior + counts (variable.go:45, chain.go:230-236)
    assert np.all(co > -1e-6)
    po = _group_props(np.rint(co), pm.card)
    se = np.sqrt(po.var(axis=0, ddof=1) / G + pg.var(axis=0, ddof=1) / G)
    z = (po.mean(axis=0) - pg.mean(axis=0)) / np.maximum(se, 2e-4)
    free = np.concatenate([np.full(c, f < 0) for c, f in zip(pm.card, pm.fixed)])
    assert np.all(pg[:, ~free] == 0) and np.all(po[:, ~free] == 0)  # evidence variables are never sampled on either side
    z = z[free]
    assert np.max(np.abs(z)) <= 5.5, (name, np.max(np.abs(z)), np.argmax(np.abs(z)))
    assert np.mean(z ** 2) <= 2.0, (name, np.mean(z ** 2))
    # and the headline metric agrees: Hellinger vs res/<model>.uai.MAR is the same for both
    sol = uai_py.load_mar(res_dir, name)
    split = lambda flat: g.split(flat, pm.card)
    eo = g.NewErrorSuite(pm.card, sol, split(po.mean(axis=0) + 1e-12), fixed1=pm.fixed)
    eg = g.NewErrorSuite(pm.card, sol, split(pg.mean(axis=0) + 1e-12), fixed1=pm.fixed)
    assert abs(eo["MeanHellinger"] - eg["MeanHellinger"]) <= 0.02, (eo, eg)


def test_objectdetection_reaches_mar(res_dir):
    """ObjectDetection_13 mixes (slowly): with long chains the GPU estimate approaches the exact .MAR."""
    name = "ObjectDetection_13.uai"
    pm = uai_py.load(res_dir, name)
    sol = uai_py.load_mar(res_dir, name)
    s = g.HIPChains.NewGibbsSimple(g.Model.NewModelFromFile(os.path.join(res_dir, name)), 1024, seed=42)
    s.BurnIn(60 * 1000); s.Advance(60 * 3000)
    es = g.NewErrorSuite(pm.card, sol, g.MergeChains([s])[0])
    assert es["MeanHellinger"] < 0.02 and es["MaxHellinger"] < 0.08, es


# ----------------------------------------------- 5. properties at BASELINE sizes
@pytest.mark.parametrize("name,chains", [("Grids_13.uai", 65536), ("Grids_11.uai", 4096), ("ObjectDetection_13.uai", 32768)])
def test_full_size_properties(res_dir, name, chains):
    path = os.path.join(res_dir, name)
    m = g.Model.NewModelFromFile(path)
    n = m.n_vars
    upd = 8 * n + 3

    def run(chunks, flags=0, **kw):
        s = g.HIPChains.NewGibbsSimple(m, chains, seed=7, flags=flags, **kw)
        s.BurnIn(2 * n + 1)
        for c in chunks:
            s.Advance(c)
        return s

    a = run([upd])
    ca = a.counts()[0]
    assert int(ca.sum()) == chains * upd == a.TotalSampleCount()  # conservation
    assert np.array_equal(ca, run([upd]).counts()[0])  # determinism
    assert np.array_equal(ca, run([5, upd - 5 - 2 * n, 2 * n]).counts()[0])  # launch chunking
    assert np.array_equal(ca, run([upd], g.GR_FLAG_SEQUENTIAL).counts()[0])  # co-issue == sequential
    assert np.array_equal(ca, run([upd], g.GR_FLAG_DIRECT).counts()[0])  # memoised == direct
    assert np.array_equal(ca, run([upd], block_threads=64).counts()[0])  # block shape
    # the split of the chains over devices does not change the result
    h = chains // 2
    parts = []
    for k in range(2):
        s = g.HIPChains.NewGibbsSimple(m, h, seed=7, chain_offset=k * h)
        s.BurnIn(2 * n + 1); s.Advance(upd)
        parts.append(s)
    assert np.array_equal(ca, parts[0].counts()[0] + parts[1].counts()[0])
    assert np.array_equal(a.LastSample(h + 5), parts[1].LastSample(5))
    merged, _ = g.MergeChains(parts)
    whole, _ = g.MergeChains([a])
    assert all(np.allclose(x, y, rtol=1e-15) for x, y in zip(merged, whole))
    # every free variable keeps both/all values reachable (floor) and counts are per-variable consistent
    cards = m.cards()
    off = 0
    for v in range(n):
        assert ca[off:off + cards[v]].sum() % 1 == 0
        off += cards[v]


def test_evidence_variables_never_change(res_dir):
    """a2/a13: evidence variables start at FixedVal, are never selected, never counted."""
    name = "Promedus_11.uai"
    pm = uai_py.load(res_dir, name, True)
    s = g.HIPChains.NewGibbsSimple(g.Model.NewModelFromFile(os.path.join(res_dir, name), True), 256, seed=3)
    s.BurnIn(1000); s.Advance(5000)
    cnt = s.counts()[0]
    st = s.LastSample(17)
    off = 0
    for v, (c, f) in enumerate(zip(pm.card, pm.fixed)):
        if f >= 0:
            assert st[v] == f and cnt[off:off + c].sum() == 0
        off += c
    sol = uai_py.load_mar(res_dir, name)
    merged, _ = g.MergeChains([s])
    es = g.NewErrorSuite(pm.card, sol, merged, fixed1=pm.fixed)
    assert es["MeanHellinger"] < 0.3  # sanity only: Promedus mixes slowly


def test_state_roundtrip_and_errors(res_dir):
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "sample.uai"))
    s = g.HIPChains.NewGibbsSimple(m, 64)
    s.set_state(3, [1, 0, 2])
    assert s.LastSample(3).tolist() == [1, 0, 2]
    with pytest.raises(g.GrError):
        s.set_state(3, [1, 0, 3])
    with pytest.raises(g.GrError):
        s.LastSample(64)
    with pytest.raises(g.GrError):
        g.HIPChains(m, 100, n_variants=3)
    with pytest.raises(g.GrError):
        g.HIPChains(m, 64, chain_offset=5)


def test_checkpoint_resume_is_bit_exact(res_dir):
    """gr_state_download/upload + gr_counts/gr_counts_upload + gr_get/set_position: a sampler rebuilt from a checkpoint
    continues every Philox stream where the first run stood -- counts, states and TotalSampleCount of the resumed run equal
    the uninterrupted run's bit for bit (fast 4-lane form, generic direct form, and a collapsed two-group sampler)."""
    for name, chains, groups, kw in [("Grids_13.uai", 2048, 1, dict()), ("ObjectDetection_13.uai", 192, 1, dict()),
                                     ("Segmentation_11.uai", 256, 2, dict())]:
        m = g.Model.NewModelFromFile(os.path.join(res_dir, name))

        def build():
            s = g.HIPChains(m, chains, n_variants=groups, seed=77, chain_offset=128, **kw)
            if groups > 1:
                for k in range(groups):
                    s.Collapse(k, 3 + 5 * k)
            return s
        a = build()
        a.BurnIn(101); a.Advance(999)
        ck = a.checkpoint()
        a.Advance(1203)
        b = build()
        b.restore(ck)
        assert b.TotalSampleCount() == chains * 999 and b.info()["step"] == 1100
        b.Advance(1203)
        assert np.array_equal(a.counts(), b.counts()), name
        assert np.array_equal(a.checkpoint()["states"], b.checkpoint()["states"]), name
        assert a.TotalSampleCount() == b.TotalSampleCount() == chains * (999 + 1203)
        # a refused restore: wrong group count / bad state value
        with pytest.raises(g.GrError):
            bad = dict(ck); bad["states"] = ck["states"].copy(); bad["states"][0, 0] = 200
            b.restore(bad)


def test_cpp_host_driver(res_dir):
    """tools/grample_hip_sample (C++ host above the C ABI, cmd/root.go's sample flow) runs Grids_11 and
    reports the same Hellinger as the Python view of the same run."""
    import subprocess
    root = os.path.dirname(res_dir.rstrip("/").rsplit("/tests/", 1)[0] + "/x")
    exe = os.path.join(root, "tools", "grample_hip_sample")
    if not os.path.exists(exe):
        pytest.skip("tools/grample_hip_sample not built")
    out = subprocess.run([exe, "-m", os.path.join(res_dir, "Grids_11.uai"), "-o", "-c", "4096", "-b", "2000", "-i", str(8000 * 4096),  # --maxiters = total samples over all chains
                          "-r", "2", "-e", "42"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "DONE" in out.stdout
    final = [ln for ln in out.stdout.splitlines() if ln.startswith("FINAL MeanHellinger=")][0]
    mh = float(final.split("MeanHellinger=")[1].split()[0])
    s = g.HIPChains.NewGibbsSimple(g.Model.NewModelFromFile(os.path.join(res_dir, "Grids_11.uai")), 4096, seed=42)
    s.BurnIn(2000); s.Advance(4000); s.Advance(4000)
    pm = uai_py.load(res_dir, "Grids_11.uai")
    es = g.NewErrorSuite(pm.card, uai_py.load_mar(res_dir, "Grids_11.uai"), g.MergeChains([s])[0])
    assert mh == pytest.approx(es["MeanHellinger"], abs=1e-7)


def test_bench_two_ranks_rehearsal():
    """bench.py's N > 1 path (sharding by chain_offset, all-reduce, max-over-ranks timing), rehearsed with two
    ranks on GPU 0 over gloo: the JSON line must report both ranks' chains."""
    import json
    import subprocess
    import sys
    import socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_REHEARSE_ONE_GPU="1")
    with socket.socket() as sk:  # a port nobody else on the box is using
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["chains_total"] == 2 * 65536 and d["value"] > 1e9
    assert d["scaling"] == "weak" and "cpu_baseline" not in d


@pytest.mark.parametrize("name,sweeps", [("Grids_11.uai", 40), ("ObjectDetection_13.uai", 40)])
def test_group_scan_equals_per_chain_scan_in_law(res_dir, name, sweeps):
    """sharing the scan variable inside a 64-chain group (default) must estimate the same
    marginals as the strict mode where every chain draws its own variable like the reference does.
    Two-sample z-test over G groups, as in test_marginals_match_oracle_runs."""
    pm = uai_py.load(res_dir, name)
    n, G = len(pm.card), 32
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name))
    ps = []
    for flags, seed in ((0, 5), (g.GR_FLAG_SCAN_PER_CHAIN, 6)):
        s = g.HIPChains(m, G * 64, n_variants=G, seed=seed, flags=flags)
        s.BurnIn(10 * n); s.Advance(sweeps * n)
        ps.append(_group_props(s.counts(), pm.card))
    se = np.sqrt(ps[0].var(axis=0, ddof=1) / G + ps[1].var(axis=0, ddof=1) / G)
    z = (ps[0].mean(axis=0) - ps[1].mean(axis=0)) / np.maximum(se, 2e-4)
    assert np.max(np.abs(z)) <= 5.5 and np.mean(z ** 2) <= 2.0, (np.max(np.abs(z)), np.mean(z ** 2))


def test_kernel_selection_table():
    """The kernel / layout choices of gr_sampler (sampler.cpp: configure, configure_impl, run -- thresholds measured on the 19
    fixture models) as one observable table: every fixture model, with and without evidence, 512 .. 262 144 chains, random and
    systematic scan -> path, lanes per chain, tables in LDS.  tests/golden/path_table.json is the recorded table
    (scripts/path_table.py --json); a change of a threshold shows up here and must be re-recorded deliberately."""
    import json
    from scripts import path_table
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = json.load(open(os.path.join(root, "tests", "golden", "path_table.json")))
    got = path_table.rows()
    key = lambda x: (x["model"], x["evidence"], x.get("chains"), x.get("scan"))  # noqa: E731
    gw, gg = {key(x): x for x in want}, {key(x): x for x in got}
    assert gw.keys() == gg.keys()
    diff = [(k, {f: (gw[k].get(f), gg[k].get(f)) for f in ("path", "lanes_per_chain", "tables_in_lds", "error") if gw[k].get(f) != gg[k].get(f)}) for k in gw]
    diff = [d for d in diff if d[1]]
    assert not diff, diff[:10]
    assert len({x["path"] for x in got if "path" in x}) >= 6  # the fixtures exercise direct, memoised, 4-lane, value-parallel, step-parallel, chromatic
