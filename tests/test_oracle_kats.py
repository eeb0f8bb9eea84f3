"""Pins the CPU oracle (oracle/grample_oracle.c) against every golden vector and
known-answer test the reference holds for the Gibbs hot path.

Each test names the reference test it restates.  CPU only.
"""
import ctypes as C
import math

import numpy as np
import pytest

from tests import uai_py


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


# ---------------------------------------------------------------- rand_test.go
def test_mt_canonical_seed(oracle):
    """rand/rand_test.go:17-39 TestMTCanonicalSeed."""
    L = oracle.lib()
    seed = (C.c_uint64 * 4)(0x12345, 0x23456, 0x34567, 0x45678)
    g = L.orc_gen_new_slice(seed, 4)
    assert g
    for v in [7266447313870364031, 4946485549665804864, 16945909448695747420,
              16394063075524226720, 4873882236456199058]:
        assert L.orc_int63(g) == v & 0x7FFFFFFFFFFFFFFF
    L.orc_gen_free(g)


def test_mt_bad_seed(oracle):
    """rand/rand_test.go:9-15 TestMTBadSeed."""
    L = oracle.lib()
    assert not L.orc_gen_new_slice((C.c_uint64 * 1)(0), 0)


def test_mt_single_seed_is_canonical_init_genrand64(oracle):
    """Seed(int64) path (rand.go:28) is not pinned by the reference; this pins the
    oracle to canonical init_genrand64: seed 5489 -> 14514284786278117030 (the
    value std::mt19937_64's specification requires for its 1st output)."""
    L = oracle.lib()
    g = L.orc_gen_new(5489)
    assert L.orc_int63(g) == 14514284786278117030 & 0x7FFFFFFFFFFFFFFF
    L.orc_gen_free(g)
    # 10000th output of default-seeded mt19937_64 is 9981545732273789042 (C++11 [rand.predef])
    g = L.orc_gen_new(5489)
    for _ in range(9999):
        L.orc_int63(g)
    assert L.orc_int63(g) == 9981545732273789042 & 0x7FFFFFFFFFFFFFFF
    L.orc_gen_free(g)


def test_generator_bit_slicing(oracle):
    """rand/rand.go:57-105: Int31 = top 31 bits of Int63, Float64 = low 53 bits / 2^53."""
    L = oracle.lib()
    g1, g2, g3 = L.orc_gen_new(42), L.orc_gen_new(42), L.orc_gen_new(42)
    for _ in range(100):
        v = L.orc_int63(g1)
        assert L.orc_int31(g2) == v >> 32
        assert L.orc_float64(g3) == (v & ((1 << 53) - 1)) / float(1 << 53)
    # Int31n: power of two masks, otherwise rejection + modulo
    g4, g5 = L.orc_gen_new(7), L.orc_gen_new(7)
    for _ in range(100):
        assert L.orc_int31n(g4, 8) == (L.orc_int63(g5) >> 32) & 7
    g6, g7 = L.orc_gen_new(9), L.orc_gen_new(9)
    mx = (1 << 31) - 1 - (1 << 31) % 100
    for _ in range(200):
        v = L.orc_int63(g7) >> 32
        while v > mx:
            v = L.orc_int63(g7) >> 32
        assert L.orc_int31n(g6, 100) == v % 100
    for g in (g1, g2, g3, g4, g5, g6, g7):
        L.orc_gen_free(g)


# ------------------------------------------------------------ function_test.go
def _model23(oracle, table):
    return oracle.Model([2, 3], [[0, 1]], [table])


def test_func_eval_order_and_logspace(oracle):
    """model/function_test.go:81-157 TestFuncTestEval."""
    m = _model23(oracle, [0.01, 1.02, 2.03, 3.04, 4.05, 5.06])
    cases = {(0, 0): 0.01, (0, 1): 1.02, (0, 2): 2.03, (1, 0): 3.04, (1, 1): 4.05, (1, 2): 5.06}
    prod = 1.0
    for vals, exp in cases.items():
        rc, v = m.func_eval(0, vals)
        assert rc == 0 and v == pytest.approx(exp, rel=1e-14)
        prod *= v
    for bad in ([], [0], [0, 0, 0], [2, 0], [0, 3]):
        rc, v = m.func_eval(0, bad)
        assert rc != 0 and math.isnan(v)
    L = oracle.lib()
    assert L.orc_func_is_log(m.ptr, 0) == 0
    assert L.orc_func_use_log_space(m.ptr, 0) == 0
    assert L.orc_func_is_log(m.ptr, 0) == 1
    assert L.orc_func_use_log_space(m.ptr, 0) != 0  # double call is an error
    logsum = sum(m.func_eval(0, vals)[1] for vals in cases)
    assert math.exp(logsum) == pytest.approx(prod, rel=1e-14)


def test_use_log_space_eps_rule(oracle):
    """model/function.go:126-142: v < 1e-6 => v += 1e-6 before log."""
    m = oracle.Model([2, 2], [[0, 1]], [[0.0, 5e-7, 1e-6, 2.0]])
    oracle.lib().orc_func_use_log_space(m.ptr, 0)
    t = m.func_table(0)
    assert t[0] == math.log(1e-6)
    assert t[1] == math.log(5e-7 + 1e-6)
    assert t[2] == math.log(1e-6)  # not < eps: untouched
    assert t[3] == math.log(2.0)


def test_func_buildup(oracle):
    """model/function_test.go:192-251 TestFuncBuildup."""
    L = oracle.lib()
    m = oracle.Model([2, 3], [[0]], [[1.0, 1.0]])
    sc = np.array([0, 1], dtype=np.int32)
    f = L.orc_model_new_function(m.ptr, _ip(sc), 2)
    assert f == 1
    for inc, exp in ((1.0, 1.0), (2.42, 3.42)):
        it = L.orc_viter_new(m.ptr, _ip(sc), 2, 0)
        vals = np.zeros(2, dtype=np.int32)
        while True:
            L.orc_viter_val(it, _ip(vals))
            assert L.orc_func_add_value(m.ptr, f, _ip(vals), 2, inc) == 0
            if not L.orc_viter_next(it):
                break
        L.orc_viter_free(it)
        t = m.func_table(f)
        assert len(t) == 6 and np.allclose(t, exp, rtol=1e-6)
    assert L.orc_func_use_log_space(m.ptr, f) == 0
    z = np.zeros(2, dtype=np.int32)
    assert L.orc_func_add_value(m.ptr, f, _ip(z), 2, 123.45) != 0
    assert m.func_table(f)[0] == pytest.approx(math.log(3.42), rel=1e-6)


def test_new_function_limits(oracle):
    """model/function.go:59,77-79 maxTabSize = 1<<23."""
    L = oracle.lib()
    m = oracle.Model([16] * 7, [[0]], [[1.0] * 16])
    ok = np.arange(5, dtype=np.int32)
    assert L.orc_model_new_function(m.ptr, _ip(ok), 5) >= 0  # 16^5 = 2^20
    bad = np.arange(6, dtype=np.int32)
    assert L.orc_model_new_function(m.ptr, _ip(bad), 6) < 0  # 16^6 = 2^24 > 2^23
    assert L.orc_model_new_function(m.ptr, _ip(bad), 0) < 0


# -------------------------------------------------------- variable_iter_test.go
def _iter_seq(oracle, cards, fixed, order, honor):
    L = oracle.lib()
    m = oracle.Model(cards, [[0]], [[1.0] * cards[0]], fixed=fixed)
    sc = np.array(order, dtype=np.int32)
    it = L.orc_viter_new(m.ptr, _ip(sc), len(order), honor)
    out = []
    vals = np.zeros(len(order), dtype=np.int32)
    while True:
        L.orc_viter_val(it, _ip(vals))
        out.append(vals.tolist())
        if not L.orc_viter_next(it):
            break
    L.orc_viter_free(it)
    return out


def test_var_iter(oracle):
    """model/variable_iter_test.go:40-65 TestVarIter."""
    seq = _iter_seq(oracle, [2, 3, 2], None, [0, 1, 2], 0)
    assert seq == [[a, b, c] for a in range(2) for b in range(3) for c in range(2)]


def test_var_iter_corners(oracle):
    """variable_iter_test.go:67-93 TestVarIterCorners."""
    L = oracle.lib()
    m = oracle.Model([2], [[0]], [[1.0, 1.0]])
    assert not L.orc_viter_new(m.ptr, None, 0, 0)
    assert _iter_seq(oracle, [2], None, [0], 0) == [[0], [1]]


def test_var_iter_fixed(oracle):
    """variable_iter_test.go:95-138 TestVarIterFixedVals."""
    assert _iter_seq(oracle, [2, 2], [1, 0], [0, 1], 1) == [[1, 0]]
    # vars: 0=v1, 1=v2 free; 2=vFix fixed to 1
    assert _iter_seq(oracle, [2, 2, 2], [-1, -1, 1], [2, 0, 1], 1) == [[1, 0, 0], [1, 0, 1], [1, 1, 0], [1, 1, 1]]
    assert _iter_seq(oracle, [2, 2, 2], [-1, -1, 1], [0, 2, 1], 1) == [[0, 1, 0], [0, 1, 1], [1, 1, 0], [1, 1, 1]]
    assert _iter_seq(oracle, [2, 2, 2], [-1, -1, 1], [0, 1, 2], 1) == [[0, 0, 1], [0, 1, 1], [1, 0, 1], [1, 1, 1]]


# --------------------------------------------------------------- error_test.go
def test_error_suite_normed(oracle):
    """model/error_test.go:11-87 TestErrorSuiteNormed."""
    hell = math.sqrt((math.sqrt(0.75) - math.sqrt(0.5)) ** 2 + (math.sqrt(0.25) - math.sqrt(0.5)) ** 2) / math.sqrt(2)
    assert hell == pytest.approx(0.18459191128251448, rel=1e-12)
    js = 0.0487949406953985
    v1 = [np.array([250.0, 750.0]), np.array([25.1, 75.3])]
    v2 = [np.array([42.0, 42.0]), np.array([3.1, 3.1])]

    def check(a, b):
        es = oracle.error_suite([2, 2], a, b)
        for k in ("MeanMeanAbsError", "MaxMeanAbsError", "MeanMaxAbsError", "MaxMaxAbsError"):
            assert es[k] == pytest.approx(0.25, rel=1e-8)
        assert es["MeanHellinger"] == pytest.approx(hell, rel=1e-8)
        assert es["MaxHellinger"] == pytest.approx(hell, rel=1e-8)
        assert es["MeanJSDiverge"] == pytest.approx(js, rel=1e-8)
        assert es["MaxJSDiverge"] == pytest.approx(js, rel=1e-8)

    check(v1, v2)
    n = lambda a: a / a.sum()
    check([n(v1[0]), v1[1]], [v2[0], n(v2[1])])
    check([n(x) for x in v1], [n(x) for x in v2])


def test_error_suite_max_mean(oracle):
    """model/error_test.go:90-119 TestErrorSuiteMaxMean."""
    v1 = [np.array([30.0, 40.0, 30.0]), np.array([30.0, 40.0, 30.0])]
    v2 = [np.array([90.0, 5.0, 5.0]), np.array([60.0, 30.0, 10.0])]
    es = oracle.error_suite([3, 3], v1, v2)
    exp = dict(MeanMeanAbsError=.30000000, MaxMeanAbsError=.39999999, MeanMaxAbsError=.45000000,
               MaxMaxAbsError=.60000000, MeanHellinger=.35109087, MaxHellinger=.46528369,
               MeanJSDiverge=.18806933, MaxJSDiverge=.29645726)
    for k, v in exp.items():
        assert es[k] == pytest.approx(v, rel=1e-7)


def test_error_suite_fixed_and_errors(oracle):
    """error.go:28-45: card mismatch and 'no un-fixed vars' are errors; fixed vars score 0."""
    a = [np.array([0.5, 0.5]), np.array([0.1, 0.9])]
    b = [np.array([0.9, 0.1]), np.array([0.1, 0.9])]
    with pytest.raises(ValueError):
        oracle.error_suite([2, 2], a, b, fixed1=[0, 1])
    es = oracle.error_suite([2, 2], a, b, fixed1=[1, -1])
    assert es["MaxHellinger"] == 0.0


def test_one_uai_mar_start_score(oracle, res_dir):
    """model/uai_test.go:141-170 TestUAIMarSolFile: uniform start vs one.uai.MAR."""
    pm = uai_py.load(res_dir, "one.uai")
    sol = uai_py.load_mar(res_dir, "one.uai")
    m = oracle.Model.from_parsed(pm)
    es = oracle.error_suite(pm.card, sol, [m.marginal(0)])
    for k in ("MeanMeanAbsError", "MeanMaxAbsError", "MaxMeanAbsError", "MaxMaxAbsError"):
        assert es[k] == pytest.approx(0.25, rel=1e-8)
    assert es["MeanHellinger"] == pytest.approx(0.18459191128251448, rel=1e-8)
    assert es["MaxHellinger"] == pytest.approx(0.18459191128251448, rel=1e-8)


# -------------------------------------------------------------- sampler_test.go
def test_uniform_sampler(oracle):
    """sampler/sampler_test.go:69-131 TestUniformSampler."""
    L = oracle.lib()
    g = L.orc_gen_new(42)
    out = C.c_int()
    assert L.orc_uni_sample(g, 0, C.byref(out)) != 0
    assert L.orc_uni_sample(g, (1 << 30) + 1, C.byref(out)) != 0
    d0 = L.orc_gen_draws(g)
    assert L.orc_uni_sample(g, 1, C.byref(out)) == 0 and out.value == 0
    assert L.orc_gen_draws(g) == d0  # card 1 consumes no draw
    m = oracle.Model([2], [[0]], [[1.0, 1.0]])
    L.orc_var_set_collapsed(m.ptr, 0, 1)
    assert L.orc_var_sample(g, m.ptr, 1, C.byref(out)) != 0
    assert L.orc_var_sample(g, m.ptr, 0, C.byref(out)) == 0 and out.value == 0
    m2 = oracle.Model([2, 2], [[0]], [[1.0, 1.0]])
    seen = set()
    for _ in range(2500):
        assert L.orc_var_sample(g, m2.ptr, 0, C.byref(out)) == 0
        assert out.value in (0, 1)
        seen.add(out.value)
    assert seen == {0, 1}
    L.orc_gen_free(g)


def test_uniform_sampler_fixed(oracle):
    """sampler/sampler_test.go:133-167 TestUniformSamplerFixed."""
    L = oracle.lib()
    g = L.orc_gen_new(42)
    out = C.c_int()
    m = oracle.Model([2, 2], [[0]], [[1.0, 1.0]], fixed=[0, -1])
    d0 = L.orc_gen_draws(g)
    assert L.orc_var_sample(g, m.ptr, 0, C.byref(out)) == 0 and out.value == 1
    assert L.orc_gen_draws(g) == d0  # single candidate: no draw (sampler.go:162-165)
    L.orc_var_set_fixed(m.ptr, 1, 1)
    assert L.orc_var_sample(g, m.ptr, 0, C.byref(out)) != 0
    L.orc_var_set_fixed(m.ptr, 0, -1)
    L.orc_var_set_fixed(m.ptr, 1, -1)
    L.orc_var_set_collapsed(m.ptr, 0, 1)
    assert L.orc_var_sample(g, m.ptr, 1, C.byref(out)) == 0 and out.value == 1
    L.orc_var_set_collapsed(m.ptr, 1, 1)
    assert L.orc_var_sample(g, m.ptr, 1, C.byref(out)) != 0
    L.orc_gen_free(g)


def test_weighted_sampler(oracle):
    """sampler/sampler_test.go:169-220 TestWeightedSampler."""
    L = oracle.lib()
    g = L.orc_gen_new(42)
    out = C.c_int()

    def ws(card, w):
        a = np.array(w, dtype=np.float64)
        return L.orc_weighted_sample(g, card, _dp(a), len(a), C.byref(out))

    assert ws(0, []) != 0
    assert ws((1 << 30) + 1, []) != 0
    assert ws(1, []) != 0
    assert ws(1, [1.0, 1.0]) != 0
    assert ws(2, [1.0, -1.0]) != 0
    assert ws(1, [1.0]) == 0 and out.value == 0
    heads = tails = flips = 0
    while heads < 100 or tails < 100:
        assert ws(2, [100.1, 200.2]) == 0
        if out.value == 0:
            heads += 1
        else:
            tails += 1
        flips += 1
        assert flips <= 5000
    assert heads / tails == pytest.approx(0.5, rel=0.1)
    L.orc_gen_free(g)


# ---------------------------------------------------------------- chain_test.go
def test_merge_chains(oracle):
    """sampler/chain_test.go:11-80 TestMergeChains."""
    L = oracle.lib()

    def mk(coll):
        m = oracle.Model([2, 2, 3], [[0]], [[1.0, 1.0]])
        m.set_marginal(0, [0.5, 0.5]); m.set_marginal(1, [5.1, 5.1]); m.set_marginal(2, [1.1, 2.2, 3.3])
        for v, c in enumerate(coll):
            L.orc_var_set_collapsed(m.ptr, v, c)
        return m, L.orc_chain_new_nosampler(m.ptr)

    def merge(chs):
        arr = (C.c_void_p * len(chs))(*chs)
        out = np.zeros(7)
        rc = L.orc_merge_chains(arr, len(chs), _dp(out), None)
        return rc, out

    assert merge([])[0] != 0
    m1, ch1 = mk([0, 0, 0])
    base = [0.5, 0.5, 5.1, 5.1, 1.1, 2.2, 3.3]
    rc, out = merge([ch1, ch1])
    assert rc == 0 and np.allclose(out, [1.0, 1.0, 10.2, 10.2, 2.2, 4.4, 6.6], atol=1e-8)
    assert np.allclose(merge([ch1])[1], base, atol=1e-8)
    m2, ch2 = mk([1, 0, 0])
    for order in ([ch1, ch2], [ch2, ch1]):
        rc, out = merge(order)
        assert rc == 0 and np.allclose(out, [0.5, 0.5, 10.2, 10.2, 2.2, 4.4, 6.6], atol=1e-8)
    assert np.allclose(merge([ch1])[1], base, atol=1e-8)
    assert np.allclose(merge([ch2])[1], base, atol=1e-8)
    m3, ch3 = mk([1, 1, 1])
    assert np.allclose(merge([ch1, ch2, ch3])[1], base, atol=1e-8)
    assert np.allclose(merge([ch1])[1], base, atol=1e-8)
    for ch in (ch1, ch2, ch3):
        L.orc_chain_free(ch)


# --------------------------------------------------------- gibbs-simple_test.go
def test_working_gibbs_simple(oracle, res_dir):
    """sampler/gibbs-simple_test.go:13-38 TestWorkingGibbsSimple (one.uai, seed 42)."""
    L = oracle.lib()
    m = oracle.Model.from_parsed(uai_py.load(res_dir, "one.uai"))
    g = L.orc_gen_new(42)
    s = L.orc_simple_new(g, m.ptr)
    assert s
    one = np.zeros(1, dtype=np.int32)
    counts = [0, 0]
    for _ in range(1024):
        assert L.orc_simple_sample(s, _ip(one), 1) == 0
        counts[one[0]] += 1
    assert counts[0] > 0 and counts[1] > 0
    # draw accounting: 1 Int31n for the start state, then exactly one Float64 per
    # Sample (VarSample's single-candidate shortcut consumes nothing).
    assert L.orc_gen_draws(g) == 1 + 1024
    assert L.orc_simple_sample(s, _ip(np.zeros(2, dtype=np.int32)), 2) == -2  # length check (148-151)
    L.orc_simple_free(s); L.orc_gen_free(g)


def test_gibbs_simple_fixed_var_rejected(oracle, res_dir):
    """gibbs-simple.go:167-169: SampleVar on an evidence variable is an error."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "sample.uai")
    pm.fixed[1] = 1
    m = oracle.Model.from_parsed(pm)
    g = L.orc_gen_new(42)
    s = L.orc_simple_new(g, m.ptr)
    st = np.zeros(3, dtype=np.int32)
    L.orc_simple_get_last(s, _ip(st))
    assert st[1] == 1  # start state honours evidence (101-103)
    assert L.orc_simple_sample_var(s, 1, _ip(st), None) == -2
    for _ in range(200):
        idx = L.orc_simple_sample(s, _ip(st), 3)
        assert idx in (0, 2) and st[1] == 1
    L.orc_simple_free(s); L.orc_gen_free(g)


# ------------------------------------------------------ gibbs-collapsed_test.go
def test_working_gibbs_collapsed(oracle, res_dir):
    """sampler/gibbs-collapsed_test.go:14-48 TestWorkingGibbsCollapsed (deterministic.uai)."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "deterministic.uai")
    g = L.orc_gen_new(42)
    for i in range(3):
        m = oracle.Model.from_parsed(pm)
        c = L.orc_collapsed_new(g, m.ptr)
        assert c
        assert L.orc_collapsed_collapse(c, i) == i
        for j in range(3):
            assert m.collapsed(j) == (j == i)
        mg = m.marginal(i)
        assert mg[0] == pytest.approx(0.5, rel=1e-5) and mg[1] == pytest.approx(0.5, rel=1e-5)
        L.orc_collapsed_free(c)
    L.orc_gen_free(g)


def test_full_gibbs_collapsed(oracle, res_dir):
    """sampler/gibbs-collapsed_test.go:51-111 TestFullGibbsCollapsed (sample.uai)."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "sample.uai")
    g = L.orc_gen_new(42)
    m = oracle.Model.from_parsed(pm)
    c = L.orc_collapsed_new(g, m.ptr)
    assert L.orc_collapsed_collapse(c, 0) == 0
    assert m.collapsed(0) and not m.collapsed(1)
    assert L.orc_collapsed_collapse(c, 1) == 1
    assert m.collapsed(0) and m.collapsed(1)
    L.orc_collapsed_free(c)
    m = oracle.Model.from_parsed(pm)
    c = L.orc_collapsed_new(g, m.ptr)
    ncoll = lambda: sum(m.collapsed(v) for v in range(3))
    assert ncoll() == 0
    assert L.orc_collapsed_collapse(c, -1) >= 0 and ncoll() == 1
    assert L.orc_collapsed_collapse(c, -1) >= 0 and ncoll() == 2
    assert L.orc_collapsed_collapse(c, -1) < 0 and ncoll() == 2  # third must error
    L.orc_collapsed_free(c); L.orc_gen_free(g)


def test_collapse_matches_brute_force(oracle, res_dir):
    """Collapse (gibbs-collapsed.go:206-272) == brute-force sum over the variable,
    computed here independently with numpy on sample.uai (var 1, card 2, blanket {0,1,2})."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "sample.uai")
    m = oracle.Model.from_parsed(pm)
    g = L.orc_gen_new(42)
    c = L.orc_collapsed_new(g, m.ptr)
    assert L.orc_collapsed_collapse(c, 1) == 1
    eps = lambda t: np.where(t < 1e-6, t + 1e-6, t)
    f1 = eps(pm.tables[1]).reshape(2, 2)       # over (0,1)
    f2 = eps(pm.tables[2]).reshape(2, 3)       # over (1,2)
    joint = f1[:, :, None] * f2[None, :, :]    # [x0, x1, x2]; factor on var 0 alone is not in var 1's blanket
    new = joint.sum(axis=1)                    # over (0,2)
    marg = joint.sum(axis=(0, 2)) + 1e-12
    assert np.allclose(m.marginal(1), marg / marg.sum(), rtol=1e-12)
    assert m.nfuncs == 2
    assert m.func_vars(1) == [0, 2]
    assert np.allclose(m.func_table(1), np.log(eps(new.reshape(-1))), rtol=1e-12)
    L.orc_collapsed_free(c); L.orc_gen_free(g)


def test_collapse_errors(oracle, res_dir):
    """gibbs-collapsed.go:122-137,175-177: bad index, fixed var, isolated var."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "sample.uai")
    pm.fixed[0] = 0
    m = oracle.Model.from_parsed(pm)
    g = L.orc_gen_new(42)
    c = L.orc_collapsed_new(g, m.ptr)
    assert L.orc_collapsed_collapse(c, 3) < 0
    assert L.orc_collapsed_collapse(c, 0) < 0
    L.orc_collapsed_free(c)
    m1 = oracle.Model.from_parsed(uai_py.load(res_dir, "one.uai"))
    c = L.orc_collapsed_new(g, m1.ptr)
    assert L.orc_collapsed_collapse(c, 0) < 0  # "New function would have 0 variables"
    assert "0 variables" in oracle.err()
    L.orc_collapsed_free(c); L.orc_gen_free(g)


# ------------------------------------------------------------ derived goldens
def test_oracle_reproduces_committed_vectors(oracle):
    """tests/golden/oracle_vectors.json (made by tests/golden/make_golden.py): the oracle is pinned
    against silent change, and against the numbers SURVEY.md 8(c) obtained independently from
    std::mt19937_64: first Float64 draws of seed 42 and the 255/769 split of 1024 one.uai draws."""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(here, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = json.load(open(os.path.join(here, "golden", "oracle_vectors.json")))
    got = json.loads(json.dumps(mg.build()))
    assert got == want
    assert [float(x) for x in want["mt_seed42_float64"]] == [0.55853149089588316, 0.73629461442040611, 0.39337113195856088]
    assert want["one_uai_seed42_1024_draws_no_init"] == [255, 769]


# ------------------------------------------------- buffer / convergence / adapt
def test_circular_int(oracle):
    """buffer/circular_test.go:9-60 TestCircularInt."""
    L = oracle.lib()
    r = L.orc_ring_new(6)
    assert L.orc_ring_buf_size(r) == 6 and L.orc_ring_count(r) == 0
    buf = np.zeros(3, dtype=np.int32)
    for v in (1, 2, 3, 4, 5):
        L.orc_ring_add(r, v)
    assert L.orc_ring_count(r) == 5
    assert L.orc_ring_half(r, 0, _ip(buf)) < 0 and L.orc_ring_half(r, 1, _ip(buf)) < 0  # nil iterators until full
    L.orc_ring_add(r, 6)
    assert L.orc_ring_count(r) == 6
    assert L.orc_ring_half(r, 0, _ip(buf)) == 3 and buf.tolist() == [1, 2, 3]
    assert L.orc_ring_half(r, 1, _ip(buf)) == 3 and buf.tolist() == [4, 5, 6]
    L.orc_ring_add(r, 8); L.orc_ring_add(r, 8)
    assert L.orc_ring_half(r, 0, _ip(buf)) == 3 and buf.tolist() == [3, 4, 5]
    assert L.orc_ring_half(r, 1, _ip(buf)) == 3 and buf.tolist() == [6, 8, 8]
    L.orc_ring_free(r)
    r = L.orc_ring_new(7)  # odd sizes are rounded down (circular.go:18-20)
    assert L.orc_ring_buf_size(r) == 6
    L.orc_ring_free(r)


def _conv_by_hand(first, second, merged, fixed, collapsed, cw, hell):
    """chain.go:32-92 + 253-290 re-derived with numpy for a cross-check of the C restatement."""
    m, nv = len(first), len(merged)
    out = []
    for i in range(nv):
        if collapsed[i] or fixed[i] >= 0:
            out.append(1.0)
            continue
        W = B = 1e-8
        for c in range(m):
            v1, v2 = first[c][i] + 1e-8, second[c][i] + 1e-8
            W += hell(v1, v2)
            B += hell(merged[i], v1 + v2)
        W /= m
        B *= cw / (m - 1)
        vhat = (cw - 1) / cw * W + (m + 1) / (m * cw) * B
        out.append(math.sqrt(4 * vhat / (2 * W)))
    return out


def test_chain_convergence_formula(oracle, res_dir):
    """sampler/chain.go:32-92 ChainConvergence on real chains == an independent numpy evaluation of the
    same formula from the chains' ring halves; fewer than 2 chains is an error; collapsed/fixed -> 1.0."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "Grids_11.uai")
    base = oracle.Model.from_parsed(pm)
    cw = 40
    e = L.orc_ensemble_new(base.ptr, 42, 3, cw, 500)
    assert e and L.orc_ensemble_n(e) == 3
    assert L.orc_ensemble_advance(e) == 0  # AdvanceChain: every variable seen cw+1 more times
    nv, tot = 100, 200
    conv = np.zeros(nv)
    assert L.orc_ensemble_convergence(e, 0, _dp(conv)) == 0
    merged = np.zeros(tot); coll = np.zeros(nv, dtype=np.int32)
    assert L.orc_ensemble_merge(e, _dp(merged), _ip(coll)) == 0
    # rebuild the halves from the exported rings
    first, second = [], []
    half = cw // 2
    tmp = np.zeros(half, dtype=np.int32)
    import ctypes as Cc
    class Chain(Cc.Structure):
        _fields_ = [("target", Cc.c_void_p), ("simple", Cc.c_void_p), ("coll", Cc.c_void_p), ("cw", Cc.c_int),
                    ("history", Cc.POINTER(Cc.c_void_p))]
    for c in range(3):
        ch = Cc.cast(L.orc_ensemble_chain(e, c), Cc.POINTER(Chain)).contents
        f, s2 = [], []
        for v in range(nv):
            ring = ch.history[v]
            assert L.orc_ring_half(ring, 0, _ip(tmp)) == half
            f.append(np.bincount(tmp, minlength=2).astype(float))
            assert L.orc_ring_half(ring, 1, _ip(tmp)) == half
            s2.append(np.bincount(tmp, minlength=2).astype(float))
        first.append(f); second.append(s2)
    hell = lambda a, b: float(L.orc_hellinger(_dp(np.ascontiguousarray(a)), _dp(np.ascontiguousarray(b)), 2))
    exp = _conv_by_hand(first, second, [merged[2 * v:2 * v + 2] for v in range(nv)], pm.fixed, coll, cw, hell)
    assert np.allclose(conv, exp, rtol=1e-12)
    assert np.all(conv > 0) and np.all(np.isfinite(conv))
    arr = (Cc.c_void_p * 1)(L.orc_ensemble_chain(e, 0))
    assert L.orc_chain_convergence(arr, 1, 0, None, None, _dp(conv)) != 0  # needs >= 2 chains
    L.orc_ensemble_free(e)


def test_adapt_policy(oracle, res_dir):
    """sampler/adaptive.go:57-157: eligibility (free, not collapsed, blanket 2..12), ranking by ChainConvergence
    and picking from the tail of the descending order, one new chain per target with that variable collapsed,
    nothing once MaxChains (128) is reached, error below 2 chains."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "Grids_11.uai")
    base = oracle.Model.from_parsed(pm)
    e = L.orc_ensemble_new(base.ptr, 42, 2, 40, 500)
    assert L.orc_ensemble_advance(e) == 0
    out_vars = np.zeros(8, dtype=np.int32); n_out = C.c_int(); conv = np.zeros(100)
    assert L.orc_ensemble_adapt(e, 4, _ip(out_vars), C.byref(n_out), _dp(conv)) == 0
    assert n_out.value == 4 and L.orc_ensemble_n(e) == 6
    picked = out_vars[:4].tolist()
    order = sorted(range(100), key=lambda v: (-conv[v], v))  # descending score, stable
    assert picked == order[::-1][:4]  # taken from the tail: the LOWEST scores (adaptive.go:111-119)
    merged = np.zeros(200); coll = np.zeros(100, dtype=np.int32)
    assert L.orc_ensemble_merge(e, _dp(merged), _ip(coll)) == 0
    assert sorted(np.where(coll)[0].tolist()) == sorted(picked)
    # second round: already collapsed variables are no longer eligible
    assert L.orc_ensemble_advance(e) == 0
    assert L.orc_ensemble_adapt(e, 4, _ip(out_vars), C.byref(n_out), _dp(conv)) == 0
    assert n_out.value == 4 and not set(out_vars[:4].tolist()) & set(picked)
    L.orc_ensemble_free(e)
    e1 = L.orc_ensemble_new(base.ptr, 42, 1, 40, 100)
    assert L.orc_ensemble_adapt(e1, 1, _ip(out_vars), C.byref(n_out), None) != 0  # at least 2 chains
    L.orc_ensemble_free(e1)
    # fewer candidates than slots: all of them, no ranking (sample.uai: vars 0,1,2 with blankets 2,3,2)
    small = oracle.Model.from_parsed(uai_py.load(res_dir, "sample.uai"))
    e2 = L.orc_ensemble_new(small.ptr, 42, 2, 20, 50)
    assert L.orc_ensemble_advance(e2) == 0
    assert L.orc_ensemble_adapt(e2, 5, _ip(out_vars), C.byref(n_out), _dp(conv)) == 0
    assert n_out.value == 3 and sorted(out_vars[:3].tolist()) == [0, 1, 2] and L.orc_ensemble_n(e2) == 5
    assert np.all(np.isnan(conv[:3]))  # no ranking was needed
    L.orc_ensemble_free(e2)
