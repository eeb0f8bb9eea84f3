"""ctypes loader for the CPU parity oracle (oracle/grample_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package grample_amd never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so with gcc (no GPU, no reference sources needed)."""
    subprocess.run(["make", "-s", "-C", _HERE, "liboracle.so"], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("ORACLE_LIB", os.path.join(_HERE, "liboracle.so"))
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
    pi, pd = C.POINTER(C.c_int), C.POINTER(C.c_double)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_last_error", C.c_char_p)
    sig("orc_gen_new", vp, i64)
    sig("orc_gen_new_slice", vp, C.POINTER(u64), i32)
    sig("orc_gen_free", None, vp)
    sig("orc_gen_draws", u64, vp)
    sig("orc_int63", i64, vp)
    sig("orc_int63n", i64, vp, i64)
    sig("orc_int31", C.c_int32, vp)
    sig("orc_int31n", C.c_int32, vp, C.c_int32)
    sig("orc_float64", dbl, vp)
    sig("orc_norm_marginal", None, pd, i32)
    sig("orc_model_create", vp, i32, pi, pi, i32, pi, pi, pi, pd)
    sig("orc_model_clone", vp, vp)
    sig("orc_model_free", None, vp)
    sig("orc_model_check", i32, vp)
    sig("orc_model_nvars", i32, vp)
    sig("orc_model_nfuncs", i32, vp)
    sig("orc_var_card", i32, vp, i32)
    sig("orc_var_fixed", i32, vp, i32)
    sig("orc_var_set_fixed", None, vp, i32, i32)
    sig("orc_var_collapsed", i32, vp, i32)
    sig("orc_var_set_collapsed", None, vp, i32, i32)
    sig("orc_var_get_marginal", None, vp, i32, pd)
    sig("orc_var_set_marginal", None, vp, i32, pd)
    sig("orc_var_selections", dbl, vp, i32)
    sig("orc_func_nvars", i32, vp, i32)
    sig("orc_func_tabsize", i32, vp, i32)
    sig("orc_func_is_log", i32, vp, i32)
    sig("orc_func_get_vars", None, vp, i32, pi)
    sig("orc_func_get_table", None, vp, i32, pd)
    sig("orc_func_eval", i32, vp, i32, pi, i32, pd)
    sig("orc_func_use_log_space", i32, vp, i32)
    sig("orc_func_add_value", i32, vp, i32, pi, i32, dbl)
    sig("orc_model_new_function", i32, vp, pi, i32)
    sig("orc_viter_new", vp, vp, pi, i32, i32)
    sig("orc_viter_val", None, vp, pi)
    sig("orc_viter_next", i32, vp)
    sig("orc_viter_free", None, vp)
    sig("orc_uni_sample", i32, vp, i32, pi)
    sig("orc_weighted_sample", i32, vp, i32, pd, i32, pi)
    sig("orc_var_sample", i32, vp, vp, i32, pi)
    sig("orc_simple_new", vp, vp, vp)
    sig("orc_simple_free", None, vp)
    sig("orc_simple_sample", i32, vp, pi, i32)
    sig("orc_simple_sample_var", i32, vp, i32, pi, pd)
    sig("orc_simple_get_last", None, vp, pi)
    sig("orc_simple_set_last", None, vp, pi)
    sig("orc_simple_conditional", i32, vp, i32, pi, pd)
    sig("orc_collapsed_new", vp, vp, vp)
    sig("orc_collapsed_free", None, vp)
    sig("orc_collapsed_base", vp, vp)
    sig("orc_collapsed_blanket_size", i32, vp, i32)
    sig("orc_collapsed_function_count", i32, vp, i32)
    sig("orc_collapsed_collapse", i32, vp, i32)
    sig("orc_collapsed_sample", i32, vp, pi, i32)
    sig("orc_chain_new_simple", vp, vp, vp, i32, i64)
    sig("orc_chain_new_collapsed", vp, vp, vp, i32, i64)
    sig("orc_chain_new_nosampler", vp, vp)
    sig("orc_chain_free", None, vp)
    sig("orc_chain_new_history", vp, vp, i32, i32)
    sig("orc_chain_history_add", i32, vp, i32, i32)
    sig("orc_chain_total_samples", i64, vp)
    sig("orc_chain_get_last", None, vp, pi)
    sig("orc_chain_advance", i32, vp)
    sig("orc_chain_run", i32, vp, i64)
    sig("orc_merge_chains", i32, C.POINTER(vp), i32, pd, pi)
    sig("orc_ring_new", vp, i32)
    sig("orc_ring_free", None, vp)
    sig("orc_ring_add", None, vp, i32)
    sig("orc_ring_half", i32, vp, i32, pi)
    sig("orc_ring_buf_size", i32, vp)
    sig("orc_ring_count", i32, vp)
    sig("orc_chain_convergence", i32, C.POINTER(vp), i32, i32, pd, pi, pd)
    sig("orc_ensemble_new", vp, vp, i64, i32, i32, i64)
    sig("orc_ensemble_free", None, vp)
    sig("orc_ensemble_n", i32, vp)
    sig("orc_ensemble_chain", vp, vp, i32)
    sig("orc_ensemble_advance", i32, vp)
    sig("orc_ensemble_run", i32, vp, i64)
    sig("orc_ensemble_merge", i32, vp, pd, pi)
    sig("orc_ensemble_convergence", i32, vp, i32, pd)
    sig("orc_ensemble_adapt", i32, vp, i32, pi, pi, pd)
    sig("orc_max_abs_diff", dbl, pd, pd, i32)
    sig("orc_mean_abs_diff", dbl, pd, pd, i32)
    sig("orc_hellinger", dbl, pd, pd, i32)
    sig("orc_js_divergence", dbl, pd, pd, i32)
    sig("orc_error_suite", i32, i32, pi, pi, pi, pi, pd, pd, pd)
    sig("orc_run_chains", i32, vp, i64, i32, i64, i64, i32, pi, pd, pi, pi, C.POINTER(i64))
    sig("orc_bench_simple", i64, vp, i64, i32, i32, i64, i64, pd)
    _LIB = L
    return L


def err():
    return lib().orc_last_error().decode()


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iarr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int32))


def darr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64))


class Model:
    """Owns an orc_model*.  Built from the flat arrays a UAI parse yields."""

    def __init__(self, card, scopes, tables, fixed=None, _ptr=None):
        L = lib()
        if _ptr is not None:
            self.ptr = _ptr
        else:
            card = iarr(card)
            fx = iarr(fixed if fixed is not None else [-1] * len(card))
            soff = iarr(np.concatenate([[0], np.cumsum([len(s) for s in scopes])]))
            svar = iarr(np.concatenate([np.asarray(s, dtype=np.int32) for s in scopes]) if scopes else [])
            toff = iarr(np.concatenate([[0], np.cumsum([len(t) for t in tables])]))
            tab = darr(np.concatenate([np.asarray(t, dtype=np.float64) for t in tables]) if tables else [])
            self.ptr = L.orc_model_create(len(card), _ip(card), _ip(fx), len(scopes), _ip(soff), _ip(svar), _ip(toff), _dp(tab))
            if not self.ptr:
                raise ValueError(err())
        self.L = L

    @classmethod
    def from_parsed(cls, pm):
        """pm: object with .card, .scopes, .tables, .fixed (tests/uai_py.py)."""
        return cls(pm.card, pm.scopes, pm.tables, pm.fixed)

    def clone(self):
        return Model(None, None, None, _ptr=self.L.orc_model_clone(self.ptr))

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self.L.orc_model_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    @property
    def nvars(self):
        return self.L.orc_model_nvars(self.ptr)

    @property
    def nfuncs(self):
        return self.L.orc_model_nfuncs(self.ptr)

    def card(self, v):
        return self.L.orc_var_card(self.ptr, v)

    def cards(self):
        return [self.card(v) for v in range(self.nvars)]

    def marginal(self, v):
        out = np.zeros(self.card(v))
        self.L.orc_var_get_marginal(self.ptr, v, _dp(out))
        return out

    def set_marginal(self, v, m):
        m = darr(m)
        self.L.orc_var_set_marginal(self.ptr, v, _dp(m))

    def collapsed(self, v):
        return bool(self.L.orc_var_collapsed(self.ptr, v))

    def func_vars(self, f):
        out = np.zeros(self.L.orc_func_nvars(self.ptr, f), dtype=np.int32)
        self.L.orc_func_get_vars(self.ptr, f, _ip(out))
        return out.tolist()

    def func_table(self, f):
        out = np.zeros(self.L.orc_func_tabsize(self.ptr, f))
        self.L.orc_func_get_table(self.ptr, f, _dp(out))
        return out

    def func_eval(self, f, values):
        v = iarr(values)
        out = C.c_double()
        rc = self.L.orc_func_eval(self.ptr, f, _ip(v), len(v), C.byref(out))
        return rc, out.value


def run_chains(model, seed, n_chains, burn_in, updates, collapse_mode=0, collapse_vars=None):
    """cmd/root.go-style multi-chain run on the oracle; returns (merged marginal
    list per var incl. the per-chain 1/Card prior, collapsed flags, collapsed var per chain, total updates)."""
    L = lib()
    cards = model.cards()
    out = np.zeros(sum(cards))
    coll = np.zeros(len(cards), dtype=np.int32)
    cvar = np.full(n_chains, -1, dtype=np.int32)
    tot = C.c_int64()
    cv = iarr(collapse_vars) if collapse_vars is not None else None
    rc = L.orc_run_chains(model.ptr, seed, n_chains, burn_in, updates, collapse_mode,
                          _ip(cv) if cv is not None else None, _dp(out), _ip(coll), _ip(cvar), C.byref(tot))
    if rc:
        raise RuntimeError(err())
    off = np.concatenate([[0], np.cumsum(cards)])
    return [out[off[i]:off[i + 1]].copy() for i in range(len(cards))], coll, cvar, tot.value


def error_suite(cards, m1, m2, fixed1=None, fixed2=None):
    """m1/m2: list of per-variable marginal arrays. Returns dict like model.ErrorSuite."""
    L = lib()
    c = iarr(cards)
    a = darr(np.concatenate(m1))
    b = darr(np.concatenate(m2))
    f1 = iarr(fixed1) if fixed1 is not None else None
    f2 = iarr(fixed2) if fixed2 is not None else None
    out = np.zeros(8)
    rc = L.orc_error_suite(len(c), _ip(c), _ip(c), _ip(f1) if f1 is not None else None,
                           _ip(f2) if f2 is not None else None, _dp(a), _dp(b), _dp(out))
    if rc:
        raise ValueError(err())
    keys = ["MeanMeanAbsError", "MeanMaxAbsError", "MeanHellinger", "MeanJSDiverge",
            "MaxMeanAbsError", "MaxMaxAbsError", "MaxHellinger", "MaxJSDiverge"]
    return dict(zip(keys, out.tolist()))
