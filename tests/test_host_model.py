"""CPU tests of the product's host code behind the C ABI: UAI / evidence / MAR
readers, error metrics, symbol export, and the "no GPU => loud failure" rule.
Restates model/uai_test.go and model/error_test.go against libgrample_hip.so.
No kernel is launched here.
"""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import grample_amd as g
from grample_amd import _lib
from tests import uai_py

PASCAL = """MARKOV
3
2 2 3
3
1 0
2 0 1
2 1 2

2
 0.436 0.564

4
 0.128 0.872
 0.920 0.080

6
 0.210 0.333 0.457
 0.811 0.000 0.189
"""


def test_library_exports_every_declared_symbol():
    """Every function declared in include/grample_hip.h is exported and bound."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "grample_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in grample_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)


def test_uai_doc_example():
    """model/uai_test.go:76-118 TestUAIDoc."""
    m = g.Model.NewModelFromBuffer(PASCAL)
    assert m.n_vars == 3 and m.cards().tolist() == [2, 2, 3] and m.n_funcs == 3
    cases = [([0], [0.436, 0.564]), ([0, 1], [0.128, 0.872, 0.920, 0.080]),
             ([1, 2], [0.210, 0.333, 0.457, 0.811, 0.000, 0.189])]
    for f, (scope, table) in enumerate(cases):
        assert m.func_scope(f) == scope
        assert m.func_table(f).tolist() == table
    rc, v = m.Eval(2, [1, 2])
    assert rc == 0 and abs(v - 0.189) < 1e-12


def test_func_eval_errors():
    """model/function_test.go:118-132: bad arity / out-of-range => error + NaN."""
    m = g.Model.NewModelFromBuffer(PASCAL)
    for bad in ([], [0], [0, 0, 0], [2, 0], [0, 3]):
        rc, v = m.Eval(2, bad)
        assert rc != 0 and math.isnan(v)


def test_uai_large_file(res_dir):
    """model/uai_test.go:120-139 TestUAILargeFile (dv-rel_1.uai)."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "dv-rel_1.uai"))
    assert m.n_vars == 120 and m.n_funcs == 40
    rc, v = m.Eval(39, [1, 1, 1])
    assert rc == 0 and v == pytest.approx(2.038, rel=1e-12)


def test_reader_errors():
    """uai.go:56-79,103-105: short buffers, unknown type, bad counts."""
    for bad in ("", "MARKOV\n1\n2", "FOO\n1\n2\n1\n1 0\n2\n 0.5 0.5\n", "MARKOV\n0\n\n1\n1 0\n2\n 0.5 0.5\n",
                "MARKOV\n1\n2\n1\n1 0\n3\n 0.5 0.5 0.1\n", "MARKOV\n1\n2\n1\n1 4\n2\n 0.5 0.5\n"):
        with pytest.raises(g.GrError):
            g.Model.NewModelFromBuffer(bad)
    with pytest.raises(g.GrError):
        g.Model.NewModelFromFile("/nonexistent/file.uai")
    # absurd counts are malformed input, not allocation requests (no std::bad_alloc, no 800 MB marginals)
    for bad, what in (("MARKOV 99999999999 2 1 1 0 2 0.5 0.5", "invalid integer"), ("MARKOV 2000000000 2 1 1 0 2 0.5 0.5", "exceeds the remaining"),
                      ("MARKOV 1 2 2000000000 1 0 2 0.5 0.5", "exceeds the remaining"), ("MARKOV 1 99999999 1 1 0 2 0.5 0.5", "Invalid card"),
                      ("MARKOV 1 2 1 2000000000 0 2 0.5 0.5", "exceeds the remaining")):
        with pytest.raises(g.GrError, match=what):
            g.Model.NewModelFromBuffer(bad)
    with pytest.raises(g.GrError, match="exceeds the remaining"):
        g.Solution.NewSolutionFromBuffer("MAR 2000000000 2 0.5 0.5") if hasattr(g.Solution, "NewSolutionFromBuffer") else g.Model.NewModelFromBuffer("MARKOV 2000000000 2 1 1 0 2 0.5 0.5")


@pytest.mark.parametrize("name,ev", [("one.uai", False), ("sample.uai", False), ("deterministic.uai", False),
                                     ("Grids_11.uai", False), ("Grids_13.uai", False), ("Segmentation_11.uai", False),
                                     ("ObjectDetection_13.uai", False), ("Promedus_11.uai", True),
                                     ("Pedigree_11.uai", True), ("CSP_12.uai", False), ("dv-rel_1.uai", True),
                                     ("ObjectDetection_11.uai", False), ("Alchemy_11.uai", False), ("dv-rel_2.uai", True),
                                     ("dv-rel_3.uai", True), ("CSP_13.uai", False), ("Promedus_17.uai", True),
                                     ("Promedus_14.uai", True), ("CSP_11.uai", False)])
def test_reader_matches_independent_twin(res_dir, name, ev):
    """The C++ reader and the independent Python reader agree on every fixture."""
    pm = uai_py.load(res_dir, name, ev)
    m = g.Model.NewModelFromFile(os.path.join(res_dir, name), ev)
    assert m.cards().tolist() == pm.card
    assert m.fixed().tolist() == pm.fixed
    assert m.n_funcs == len(pm.scopes)
    for f in range(m.n_funcs):
        assert m.func_scope(f) == pm.scopes[f]
        assert np.array_equal(m.func_table(f), pm.tables[f])
    if os.path.exists(os.path.join(res_dir, name + ".MAR")):  # sample.uai ships without a solution
        sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, name + ".MAR")).marginals()
        ref = uai_py.load_mar(res_dir, name)
        assert len(sol) == len(ref) and all(np.allclose(a, b, rtol=0, atol=1e-15) for a, b in zip(sol, ref))


def test_merlin_mar_skips_pr_preamble(res_dir):
    """uai.go:258-262: everything before the first MAR line is dropped."""
    sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, "Grids_11.uai.merlin.MAR")).marginals()
    ref = uai_py.load_mar(res_dir, "Grids_11.uai", ".merlin.MAR")
    assert len(sol) == 100 and all(np.allclose(a, b) for a, b in zip(sol, ref))


def test_mar_sol_file_start_score(res_dir):
    """model/uai_test.go:141-170 TestUAIMarSolFile."""
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "one.uai"))
    assert m.fixed().tolist() == [-1]
    sol = g.Solution.NewSolutionFromFile(os.path.join(res_dir, "one.uai.MAR")).marginals()
    es = g.NewErrorSuite(m.cards(), sol, [np.array([0.5, 0.5])])  # NewVariable's uniform start marginal
    for k in ("MeanMeanAbsError", "MeanMaxAbsError", "MaxMeanAbsError", "MaxMaxAbsError"):
        assert es[k] == pytest.approx(0.25, rel=1e-8)
    assert es["MeanHellinger"] == pytest.approx(0.18459191128251448, rel=1e-8)


def test_evidence_formats(res_dir):
    """model/uai_test.go:172-208 TestUAIMariEvidFile."""
    p = os.path.join(res_dir, "one.uai")
    m = g.Model.NewModelFromFile(p, True)
    assert m.fixed().tolist() == [-1]
    with pytest.raises(g.GrError):
        m.ApplyEvidence("2\n1 0 0\n1 0 1")
    assert m.fixed().tolist() == [-1]
    for evid, exp in (("1 0 0", 0), ("1\n1 0 1", 1)):
        m = g.Model.NewModelFromFile(p)
        m.ApplyEvidence(evid)
        assert m.fixed().tolist() == [exp]
        with pytest.raises(g.GrError):
            m.ApplyEvidence(evid)  # re-applying is an error


def test_error_suite_goldens(oracle):
    """model/error_test.go:11-119 on the product's implementation, and equality with the oracle's."""
    v1 = [np.array([250.0, 750.0]), np.array([25.1, 75.3])]
    v2 = [np.array([42.0, 42.0]), np.array([3.1, 3.1])]
    es = g.NewErrorSuite([2, 2], v1, v2)
    assert es["MeanHellinger"] == pytest.approx(0.18459191128251448, rel=1e-8)
    assert es["MaxJSDiverge"] == pytest.approx(0.0487949406953985, rel=1e-8)
    assert es["MeanMeanAbsError"] == pytest.approx(0.25, rel=1e-8)
    a = [np.array([30.0, 40.0, 30.0]), np.array([30.0, 40.0, 30.0])]
    b = [np.array([90.0, 5.0, 5.0]), np.array([60.0, 30.0, 10.0])]
    es = g.NewErrorSuite([3, 3], a, b)
    exp = dict(MeanMeanAbsError=.30000000, MaxMeanAbsError=.39999999, MeanMaxAbsError=.45000000,
               MaxMaxAbsError=.60000000, MeanHellinger=.35109087, MaxHellinger=.46528369,
               MeanJSDiverge=.18806933, MaxJSDiverge=.29645726)
    for k, v in exp.items():
        assert es[k] == pytest.approx(v, rel=1e-7)
    eo = oracle.error_suite([3, 3], a, b)
    for k in exp:
        assert es[k] == pytest.approx(eo[k], rel=1e-13)
    with pytest.raises(g.GrError):
        g.NewErrorSuite([2], [np.array([.5, .5])], [np.array([.5, .5])], fixed1=[0])


def test_model_create_checks():
    """gr_model_create runs Model.Check (model.go:115-157): all-fixed models are rejected."""
    with pytest.raises(g.GrError):
        g.Model.from_arrays([2], [[0]], [[0.5, 0.5]], fixed=[1])
    with pytest.raises(g.GrError):
        g.Model.from_arrays([2, 2], [[0, 1]], [[0.5, 0.5]])  # table size mismatch
    m = g.Model.from_arrays([2, 3], [[0, 1]], [[0.01, 1.02, 2.03, 3.04, 4.05, 5.06]])
    assert m.Eval(0, [1, 2]) == (0, 5.06)


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 (Salmon et al. SC'11)."""
    kats = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
            ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
            ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
             [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, exp in kats:
        assert g.philox4x32_10(ctr, key).tolist() == exp
    from tests.emulator import philox4x32_10 as np_philox
    rng = np.random.default_rng(1)
    for _ in range(20):
        ctr = rng.integers(0, 2**32, 4, dtype=np.uint64).astype(np.uint32)
        key = rng.integers(0, 2**32, 2, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(g.philox4x32_10(ctr, key), np.array([int(x) for x in np_philox(*[np.uint32(c) for c in ctr], *[np.uint32(k) for k in key])], dtype=np.uint32))


def test_no_gpu_fails_loudly(res_dir):
    """Without a HIP device the sampling entry points must error (no CPU fallback)."""
    if g.device_count() > 0:
        pytest.skip("a GPU is present")
    m = g.Model.NewModelFromFile(os.path.join(res_dir, "one.uai"))
    with pytest.raises(g.GrError) as ei:
        g.HIPChains.NewGibbsSimple(m, 64)
    assert "no HIP device" in str(ei.value)


def test_product_never_imports_oracle():
    """Nothing under grample_amd/ may reference the oracle."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "grample_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                txt = open(os.path.join(dp, fn)).read()
                assert "oracle" not in txt.lower().replace("the oracle's", ""), f"{fn} mentions the oracle"


def test_host_code_under_sanitizers(res_dir, tmp_path):
    """The product's host code (UAI reader, evidence, solutions, device-image compiler) under
    AddressSanitizer + UBSan on every fixture and on malformed inputs (GPU sanitizers are not
    available on this pool; the device side is covered by the bitwise replay tests instead)."""
    import glob
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_asan")
    src = [os.path.join(root, "tests", "host_asan_main.cpp"), os.path.join(root, "grample_amd", "csrc", "host_model.cpp"),
           os.path.join(root, "grample_amd", "csrc", "compiled_model.cpp")]
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-o", exe] + src, check=True)
    files = sorted(glob.glob(os.path.join(res_dir, "*.uai")))
    out = subprocess.run([exe] + files, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ERROR" not in out.stdout and out.stdout.count("image_words") == 2 * len(files)


def _oracle_ensemble_halves(oracle, L, e, n_chains, nv, cw):
    import ctypes as Cc

    class Chain(Cc.Structure):
        _fields_ = [("target", Cc.c_void_p), ("simple", Cc.c_void_p), ("coll", Cc.c_void_p), ("cw", Cc.c_int),
                    ("history", Cc.POINTER(Cc.c_void_p))]
    half = cw // 2
    tmp = np.zeros(half, dtype=np.int32)
    first = np.zeros((n_chains, 2 * nv)); second = np.zeros((n_chains, 2 * nv))
    for c in range(n_chains):
        ch = Cc.cast(L.orc_ensemble_chain(e, c), Cc.POINTER(Chain)).contents
        for v in range(nv):
            if L.orc_ring_half(ch.history[v], 0, tmp.ctypes.data_as(Cc.POINTER(Cc.c_int))) == half:
                first[c, 2 * v:2 * v + 2] = np.bincount(tmp, minlength=2)
            if L.orc_ring_half(ch.history[v], 1, tmp.ctypes.data_as(Cc.POINTER(Cc.c_int))) == half:
                second[c, 2 * v:2 * v + 2] = np.bincount(tmp, minlength=2)
    return first, second


def test_chain_convergence_and_adapt_select_match_oracle(oracle, res_dir):
    """N2 host logic: gr_chain_convergence == the oracle's ChainConvergence (chain.go:32-92) on the same
    half-window histograms, for all four measures; gr_adapt_select == the oracle's Adapt picks
    (adaptive.go:78-119) given the same scores, incl. the second round where collapsed variables drop out."""
    L = oracle.lib()
    pm = uai_py.load(res_dir, "Grids_11.uai")
    obase = oracle.Model.from_parsed(pm)
    gm = g.Model.NewModelFromFile(os.path.join(res_dir, "Grids_11.uai"))
    cw, nv = 40, 100
    e = L.orc_ensemble_new(obase.ptr, 42, 3, cw, 500)
    assert L.orc_ensemble_advance(e) == 0
    merged = np.zeros(200); coll = np.zeros(nv, dtype=np.int32)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    assert L.orc_ensemble_merge(e, dp(merged), ip(coll)) == 0
    first, second = _oracle_ensemble_halves(oracle, L, e, 3, nv, cw)
    for measure in range(4):
        oc = np.zeros(nv)
        assert L.orc_ensemble_convergence(e, measure, dp(oc)) == 0
        gc = g.ChainConvergence(pm.card, pm.fixed, coll, first, second, merged, cw, measure)
        assert np.allclose(gc, oc, rtol=1e-12), measure
    with pytest.raises(g.GrError):
        g.ChainConvergence(pm.card, pm.fixed, coll, first[:1], second[:1], merged, cw)  # needs >= 2 chains
    picked_all = []
    for rnd in range(2):
        out_vars = np.zeros(8, dtype=np.int32); n_out = C.c_int(); conv = np.zeros(nv)
        L.orc_ensemble_merge(e, dp(merged), ip(coll))
        assert L.orc_ensemble_adapt(e, 4, ip(out_vars), C.byref(n_out), dp(conv)) == 0
        assert g.AdaptSelect(gm, coll, conv, 4) == out_vars[:n_out.value].tolist()
        picked_all += out_vars[:n_out.value].tolist()
        assert L.orc_ensemble_advance(e) == 0
    assert len(set(picked_all)) == 8
    L.orc_ensemble_free(e)
    # fewer candidates than slots: every eligible variable in id order, no scores needed
    small = g.Model.NewModelFromFile(os.path.join(res_dir, "sample.uai"))
    assert g.AdaptSelect(small, [0, 0, 0], None, 5) == [0, 1, 2]
    assert g.AdaptSelect(small, [0, 1, 0], None, 5) == [0, 2]
    with pytest.raises(g.GrError):
        g.AdaptSelect(gm, np.zeros(nv, dtype=np.int32), None, 4)  # ranking needed but no scores
    one = g.Model.NewModelFromFile(os.path.join(res_dir, "one.uai"))
    assert g.AdaptSelect(one, [0], None, 3) == []  # blanket of size 1 is never eligible (adaptive.go:84)


# ------------------------------------------------------------- reader fuzz (deterministic, no GPU)
def _serialise(rng, pm, typ="MARKOV"):
    """UAI text of a model with irregular but legal whitespace (uai.go:20-51 collapses it)."""
    ws = lambda: rng.choice([" ", "  ", "\t", "\n", " \n "])
    toks = [typ, str(len(pm.card))] + [str(c) for c in pm.card] + [str(len(pm.scopes))]
    for s in pm.scopes:
        toks += [str(len(s))] + [str(v) for v in s]
    for t in pm.tables:
        toks += [str(len(t))] + [repr(float(x)) for x in t]
    return "".join(tok + ws() for tok in toks) + "\n"


def test_reader_fuzz_roundtrip_and_mutations():
    """60 random models written with irregular whitespace parse to exactly what was written (C++ reader == Python
    twin == source); 240 mutated texts (token deleted / replaced / truncated) either parse to a structurally valid
    model or raise GrError -- never crash, never accept an inconsistent table."""
    from types import SimpleNamespace
    rng = np.random.default_rng(31337)
    texts = []
    for _ in range(60):
        n = int(rng.integers(1, 9))
        card = [int(c) for c in rng.integers(1, 5, n)]
        scopes = []
        for _ in range(int(rng.integers(1, 7))):
            k = int(rng.integers(1, min(n, 4) + 1))
            scopes.append([int(v) for v in rng.choice(n, size=k, replace=False)])
        tables = [rng.random(int(np.prod([card[v] for v in s]))) * rng.choice([1.0, 1e-7, 1e3]) for s in scopes]
        pm = SimpleNamespace(card=card, scopes=scopes, tables=tables)
        text = _serialise(rng, pm, rng.choice(["MARKOV", "BAYES"]))
        if len(text) < 15:
            continue
        texts.append(text)
        m = g.Model.NewModelFromBuffer(text)
        tw = uai_py.read_model(text)
        assert m.cards().tolist() == card == tw.card and m.n_funcs == len(scopes)
        for f in range(m.n_funcs):
            assert m.func_scope(f) == scopes[f] == tw.scopes[f]
            assert np.array_equal(m.func_table(f), tables[f]) and np.array_equal(tw.tables[f], tables[f])
    accepted = rejected = 0
    for i in range(240):
        toks = texts[i % len(texts)].split()
        op = i % 3
        pos = int(rng.integers(0, len(toks)))
        if op == 0:
            del toks[pos]
        elif op == 1:
            toks[pos] = rng.choice(["-1", "0", "x", "99999999999", "1e400", "nan", "3.5"])
        else:
            toks = toks[:pos]
        bad = " ".join(toks)
        try:
            m = g.Model.NewModelFromBuffer(bad)
        except g.GrError:
            rejected += 1
            continue
        accepted += 1  # a mutation can still be a legal file (e.g. a table value replaced by another number)
        cards = m.cards().tolist()
        assert all(c >= 1 for c in cards) and m.n_funcs >= 1
        for f in range(m.n_funcs):
            sc = m.func_scope(f)
            assert all(0 <= v < len(cards) for v in sc)
            assert len(m.func_table(f)) == int(np.prod([cards[v] for v in sc]))
    assert rejected > 120 and accepted + rejected == 240


def test_systematic_order_is_a_proper_colouring_on_every_fixture(res_dir):
    """GR_FLAG_SCAN_SYSTEMATIC's sweep order (host code, no GPU): on every fixture model, with and without evidence, the order lists
    every selectable variable once, by (colour, id); no factor holds two selectable variables of one colour (so a colour class can
    be resampled side by side); and it equals the Python twin the GPU tests replay with (tests/emulator.py)."""
    from tests.emulator import systematic_order
    seen = 0
    for f in sorted(os.listdir(res_dir)):
        if not f.endswith(".uai"):
            continue
        for ev in (False, True):
            if ev and not os.path.exists(os.path.join(res_dir, f + ".evid")):
                continue
            try:
                pm = uai_py.load(res_dir, f, ev)
                m = g.Model.NewModelFromFile(os.path.join(res_dir, f), ev)
            except Exception:
                continue
            order, color = m.systematic_order()
            free = [v for v in range(len(pm.card)) if pm.fixed[v] < 0]
            assert sorted(order) == free
            assert order == sorted(free, key=lambda v: (color[v], v))
            assert all(color[v] == -1 for v in range(len(pm.card)) if pm.fixed[v] >= 0)
            for sc in pm.scopes:
                cs = [color[v] for v in sc if pm.fixed[v] < 0]
                assert len(cs) == len(set(cs)), (f, ev, sc)
            want, wcol = systematic_order(pm.card, pm.fixed, pm.scopes)
            assert order == want
            seen += 1
    assert seen >= 19
    # Grids: the checkerboard (two colours)
    order, color = g.Model.NewModelFromFile(os.path.join(res_dir, "Grids_11.uai")).systematic_order()
    assert max(color) == 1 and len(order) == 100
