"""Thin Python view of the C ABI, used by tests and bench.py.

Names follow the reference's Go API (sampler.NewGibbsSimple / NewGibbsCollapsed
/ Collapse / Chain.AdvanceChain / MergeChains, model.NewModelFromFile /
NewSolutionFromFile / ErrorSuite) so the parity tests read like the
reference's own.  All work happens behind include/grample_hip.h.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import ModelDesc, Position, SamplerInfo, SamplerOpts

GR_FLAG_DIRECT = 1 << 0
GR_FLAG_SEQUENTIAL = 1 << 1
GR_FLAG_CHAIN_PER_LANE = 1 << 2
GR_FLAG_SCAN_PER_CHAIN = 1 << 3
GR_FLAG_STATE_GLOBAL = 1 << 4
GR_FLAG_SCAN_SYSTEMATIC = 1 << 5

ERROR_KEYS = ["MeanMeanAbsError", "MeanMaxAbsError", "MeanHellinger", "MeanJSDiverge",
              "MaxMeanAbsError", "MaxMaxAbsError", "MaxHellinger", "MaxJSDiverge"]


class GrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def _err(code=-1):
    return GrError(code, _lib.lib().gr_last_error().decode())


def _chk(rc):
    if rc != 0:
        raise _err(rc)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def device_count():
    return _lib.lib().gr_device_count()


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    _lib.lib().gr_philox4x32_10(_p(c, C.c_uint32), _p(k, C.c_uint32), _p(out, C.c_uint32))
    return out


class Model:
    """model.Model (model/model.go:24-29) held by the library."""

    def __init__(self, ptr):
        if not ptr:
            raise _err()
        self.ptr = ptr
        self.L = _lib.lib()

    @classmethod
    def NewModelFromFile(cls, path, useEvidence=False):
        return cls(_lib.lib().gr_model_from_file(path.encode(), int(useEvidence)))

    @classmethod
    def NewModelFromBuffer(cls, text):
        b = text.encode() if isinstance(text, str) else text
        return cls(_lib.lib().gr_model_from_uai(b, len(b)))

    @classmethod
    def from_arrays(cls, card, scopes, tables, fixed=None):
        card = _i32(card)
        fx = _i32(fixed) if fixed is not None else None
        soff = _i32(np.concatenate([[0], np.cumsum([len(s) for s in scopes])]))
        svar = _i32(np.concatenate([np.asarray(s, dtype=np.int32) for s in scopes]))
        toff = _i32(np.concatenate([[0], np.cumsum([len(t) for t in tables])]))
        tab = _f64(np.concatenate([np.asarray(t, dtype=np.float64) for t in tables]))
        d = ModelDesc(len(card), _p(card, C.c_int32), _p(fx, C.c_int32) if fx is not None else None, len(scopes),
                      _p(soff, C.c_int32), _p(svar, C.c_int32), _p(toff, C.c_int32), _p(tab, C.c_double))
        return cls(_lib.lib().gr_model_create(C.byref(d)))

    def ApplyEvidence(self, text):
        b = text.encode() if isinstance(text, str) else text
        _chk(self.L.gr_model_apply_evidence(self.ptr, b, len(b)))

    def Clone(self):
        return Model(self.L.gr_model_clone(self.ptr))

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.gr_model_destroy(self.ptr)
            self.ptr = None

    @property
    def n_vars(self):
        return self.L.gr_model_n_vars(self.ptr)

    @property
    def n_funcs(self):
        return self.L.gr_model_n_funcs(self.ptr)

    @property
    def total_card(self):
        return self.L.gr_model_total_card(self.ptr)

    def cards(self):
        out = np.zeros(self.n_vars, dtype=np.int32)
        _chk(self.L.gr_model_cards(self.ptr, _p(out, C.c_int32)))
        return out

    def fixed(self):
        out = np.zeros(self.n_vars, dtype=np.int32)
        _chk(self.L.gr_model_fixed(self.ptr, _p(out, C.c_int32)))
        return out

    def func_scope(self, f):
        n = self.L.gr_model_func_arity(self.ptr, f)
        if n < 0:
            raise _err(n)
        out = np.zeros(n, dtype=np.int32)
        _chk(self.L.gr_model_func_scope(self.ptr, f, _p(out, C.c_int32)))
        return out.tolist()

    def func_table(self, f):
        n = self.L.gr_model_func_tabsize(self.ptr, f)
        if n < 0:
            raise _err(n)
        out = np.zeros(n)
        _chk(self.L.gr_model_func_table(self.ptr, f, _p(out, C.c_double)))
        return out

    def systematic_order(self):
        """(order, colour per variable) of GR_FLAG_SCAN_SYSTEMATIC for this model"""
        n = self.n_vars
        order, color = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
        k = C.c_int32()
        _chk(self.L.gr_model_systematic_order(self.ptr, _p(order, C.c_int32), C.byref(k), _p(color, C.c_int32)))
        return order[: k.value].tolist(), color.tolist()

    def Eval(self, f, values):
        v = _i32(values)
        out = C.c_double()
        rc = self.L.gr_model_func_eval(self.ptr, f, _p(v, C.c_int32), len(v), C.byref(out))
        return rc, out.value


def split(flat, cards):
    off = np.concatenate([[0], np.cumsum(cards)])
    return [np.array(flat[off[i]:off[i + 1]]) for i in range(len(cards))]


class Solution:
    """model.Solution (model/solution.go)."""

    def __init__(self, ptr):
        if not ptr:
            raise _err()
        self.ptr = ptr
        self.L = _lib.lib()

    @classmethod
    def NewSolutionFromFile(cls, path):
        return cls(_lib.lib().gr_solution_from_file(path.encode()))

    @classmethod
    def from_text(cls, text):
        b = text.encode() if isinstance(text, str) else text
        return cls(_lib.lib().gr_solution_from_text(b, len(b)))

    def __del__(self):
        if getattr(self, "ptr", None):
            self.L.gr_solution_destroy(self.ptr)
            self.ptr = None

    def cards(self):
        out = np.zeros(self.L.gr_solution_n_vars(self.ptr), dtype=np.int32)
        _chk(self.L.gr_solution_cards(self.ptr, _p(out, C.c_int32)))
        return out

    def marginals(self):
        out = np.zeros(self.L.gr_solution_total_card(self.ptr))
        _chk(self.L.gr_solution_marginals(self.ptr, _p(out, C.c_double)))
        return split(out, self.cards())


def NewErrorSuite(cards, m1, m2, fixed1=None, fixed2=None):
    """model.NewErrorSuite (model/error.go:28-78); m1/m2 lists of per-variable arrays."""
    c = _i32(cards)
    a, b = _f64(np.concatenate(m1)), _f64(np.concatenate(m2))
    f1 = _i32(fixed1) if fixed1 is not None else None
    f2 = _i32(fixed2) if fixed2 is not None else None
    out = np.zeros(8)
    _chk(_lib.lib().gr_error_suite(len(c), _p(c, C.c_int32), _p(f1, C.c_int32) if f1 is not None else None,
                                   _p(f2, C.c_int32) if f2 is not None else None, _p(a, C.c_double),
                                   _p(b, C.c_double), _p(out, C.c_double)))
    return dict(zip(ERROR_KEYS, out.tolist()))


class HIPChains:
    """n_chains sampler.Chain objects + their FullSampler on one MI355X."""

    def __init__(self, model, n_chains, n_variants=1, device=0, seed=42, chain_offset=0, variant_offset=0, flags=0,
                 block_threads=0, fast_form=0):
        self.L = _lib.lib()
        o = SamplerOpts(C.sizeof(SamplerOpts), n_chains, n_variants, device, seed, chain_offset, variant_offset, flags,
                        block_threads, fast_form)
        self.ptr = self.L.gr_sampler_create(model.ptr, C.byref(o))
        if not self.ptr:
            raise _err()
        self.n_chains, self.n_variants = n_chains, n_variants
        self.cards = model.cards()
        self.n_vars, self.H = len(self.cards), int(self.cards.sum())

    # constructors named after the reference's
    @classmethod
    def NewGibbsSimple(cls, model, n_chains, **kw):
        return cls(model, n_chains, n_variants=1, **kw)

    @classmethod
    def NewGibbsCollapsed(cls, model, n_chains, n_variants, **kw):
        return cls(model, n_chains, n_variants=n_variants, **kw)

    def close(self):
        if getattr(self, "ptr", None):
            self.L.gr_sampler_destroy(self.ptr)
            self.ptr = None

    __del__ = close

    def Collapse(self, variant, varIdx=-1):
        ov = C.c_int32(-1)
        marg = np.zeros(int(self.cards.max()))
        _chk(self.L.gr_collapse(self.ptr, variant, varIdx, C.byref(ov), _p(marg, C.c_double)))
        return ov.value, marg[: self.cards[ov.value]].copy()

    def BlanketSize(self, variant, v):
        return self.L.gr_blanket_size(self.ptr, variant, v)

    def FunctionCount(self, variant, v):
        return self.L.gr_function_count(self.ptr, variant, v)

    def set_active_groups(self, n):
        _chk(self.L.gr_set_active_groups(self.ptr, n))

    def active_groups(self):
        return self.L.gr_active_groups(self.ptr)

    def AdvanceWindow(self, updates_per_chain):
        _chk(self.L.gr_advance_window(self.ptr, updates_per_chain))

    def AdvanceChain(self, cw):
        """Chain.AdvanceChain with the reference's stopping rule; returns the updates run per chain."""
        n = C.c_uint64()
        _chk(self.L.gr_advance_chain(self.ptr, cw, C.byref(n)))
        return n.value

    def window_halves(self):
        a = np.zeros((self.n_variants, self.H)); b = np.zeros((self.n_variants, self.H))
        cw = C.c_double()
        _chk(self.L.gr_window_halves(self.ptr, _p(a, C.c_double), _p(b, C.c_double), C.byref(cw)))
        return a, b, cw.value

    def Adapt(self, newChainCount):
        """ConvergenceSampler.Adapt: returns (collapsed variables of the groups added, convergence scores or NaNs)."""
        out = np.zeros(max(newChainCount, 1), dtype=np.int32)
        n = C.c_uint32()
        conv = np.zeros(self.n_vars)
        _chk(self.L.gr_adapt(self.ptr, newChainCount, _p(out, C.c_int32), C.byref(n), _p(conv, C.c_double)))
        return out[: n.value].tolist(), conv

    def ActivateGroup(self, var=-1):
        """The activation step of Adapt: next group <- fresh clone with `var` collapsed + burn-in of 2; returns the variable."""
        out = C.c_int32(-1)
        _chk(self.L.gr_activate_group(self.ptr, var, C.byref(out)))
        return out.value

    def BurnIn(self, updates_per_chain):
        _chk(self.L.gr_burnin(self.ptr, updates_per_chain))

    def Advance(self, updates_per_chain):
        _chk(self.L.gr_advance(self.ptr, updates_per_chain))

    def TotalSampleCount(self):
        out = C.c_uint64()
        _chk(self.L.gr_total_samples(self.ptr, C.byref(out)))
        return out.value

    def counts(self):
        out = np.zeros((self.n_variants, self.H), dtype=np.uint64)
        _chk(self.L.gr_counts(self.ptr, _p(out, C.c_uint64), None))
        return out

    def counts_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.L.gr_counts_device(self.ptr, C.byref(p), C.byref(n)))
        return p.value, n.value

    def counts_upload(self, counts):
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        _chk(self.L.gr_counts_upload(self.ptr, _p(c, C.c_uint64)))

    def checkpoint(self):
        """Everything a resumed run needs besides the sampler's construction: chain states, raw counts, position."""
        pos = Position()
        pos.struct_size = C.sizeof(Position)
        cg = np.zeros(self.n_variants, dtype=np.uint64)
        ep = np.zeros(self.n_variants, dtype=np.uint32)
        _chk(self.L.gr_get_position(self.ptr, C.byref(pos), _p(cg, C.c_uint64), _p(ep, C.c_uint32)))
        st = np.zeros((self.n_vars, self.n_chains), dtype=np.uint8)
        _chk(self.L.gr_state_download(self.ptr, _p(st, C.c_uint8)))
        ck = {"states": st, "counts": self.counts(), "counted_per_group": cg, "epoch_per_group": ep,
              "position": {k: getattr(pos, k) for k, _ in Position._fields_ if k != "struct_size"}}
        try:  # the ChainHistory halves of the last AdvanceChain / AdvanceWindow, if there was one
            a, b, cw = self.window_halves()
            ck["window"] = (a, b, cw)
        except GrError:
            pass
        return ck

    def restore(self, ck):
        st = np.ascontiguousarray(ck["states"], dtype=np.uint8)
        _chk(self.L.gr_state_upload(self.ptr, _p(st, C.c_uint8)))
        self.counts_upload(ck["counts"])
        pos = Position(C.sizeof(Position), *[ck["position"][k] for k, _ in Position._fields_ if k != "struct_size"])
        cg = np.ascontiguousarray(ck["counted_per_group"], dtype=np.uint64)
        ep = np.ascontiguousarray(ck["epoch_per_group"], dtype=np.uint32)
        _chk(self.L.gr_set_position(self.ptr, C.byref(pos), _p(cg, C.c_uint64), _p(ep, C.c_uint32)))
        if "window" in ck:
            a, b, cw = ck["window"]
            a, b = _f64(a), _f64(b)
            _chk(self.L.gr_set_window_halves(self.ptr, _p(a, C.c_double), _p(b, C.c_double), float(cw)))

    def merge_local(self):
        s, cm = np.zeros(self.H), np.zeros(self.H)
        cv = np.zeros(self.n_vars, dtype=np.int32)
        _chk(self.L.gr_merge_local(self.ptr, _p(s, C.c_double), _p(cv, C.c_int32), _p(cm, C.c_double)))
        return s, cv, cm

    def LastSample(self, chain):
        out = np.zeros(self.n_vars, dtype=np.int32)
        _chk(self.L.gr_state(self.ptr, chain, _p(out, C.c_int32)))
        return out

    def set_state(self, chain, state):
        s = _i32(state)
        _chk(self.L.gr_set_state(self.ptr, chain, _p(s, C.c_int32)))

    def conditional(self, variant, v, state):
        s = _i32(state)
        out = np.zeros(int(self.cards[v]))
        _chk(self.L.gr_conditional(self.ptr, variant, v, _p(s, C.c_int32), _p(out, C.c_double)))
        return out

    def info(self):
        i = SamplerInfo()
        i.struct_size = C.sizeof(SamplerInfo)
        _chk(self.L.gr_sampler_get_info(self.ptr, C.byref(i)))
        return {k: getattr(i, k) for k, _ in SamplerInfo._fields_ if k != "struct_size"}


def ChainConvergence(cards, fixed, merged_collapsed, first, second, merged, cw, measure=0):
    """sampler.ChainConvergence (chain.go:32-92) as a pure function of the half-window histograms.
    first/second: [n_groups][total_card] arrays; merged: flat [total_card]."""
    c = _i32(cards)
    f = _i32(fixed) if fixed is not None else None
    mc = _i32(merged_collapsed) if merged_collapsed is not None else None
    a, b, m = _f64(first), _f64(second), _f64(merged)
    out = np.zeros(len(c))
    _chk(_lib.lib().gr_chain_convergence(a.shape[0], len(c), _p(c, C.c_int32), _p(f, C.c_int32) if f is not None else None,
                                         _p(mc, C.c_int32) if mc is not None else None, _p(a, C.c_double), _p(b, C.c_double),
                                         _p(m, C.c_double), float(cw), measure, _p(out, C.c_double)))
    return out


def AdaptSelect(model, merged_collapsed, converge, new_chain_count):
    """The selection rule of ConvergenceSampler.Adapt (adaptive.go:78-119)."""
    mc = _i32(merged_collapsed)
    cv = _f64(converge) if converge is not None else None
    out = np.zeros(max(new_chain_count, 1), dtype=np.int32)
    n = C.c_uint32()
    _chk(_lib.lib().gr_adapt_select(model.ptr, _p(mc, C.c_int32), _p(cv, C.c_double) if cv is not None else None,
                                    new_chain_count, _p(out, C.c_int32), C.byref(n)))
    return out[: n.value].tolist()


def MergeChains(samplers):
    """sampler.MergeChains (chain.go:96-148) over every chain of the given samplers.
    Returns (list of un-normalised per-variable marginals, collapsed flags)."""
    L = _lib.lib()
    arr = (C.c_void_p * max(len(samplers), 1))(*[s.ptr for s in samplers])
    out = np.zeros(samplers[0].H if samplers else 1)
    coll = np.zeros(samplers[0].n_vars if samplers else 1, dtype=np.int32)
    _chk(L.gr_merge(arr, len(samplers), _p(out, C.c_double), _p(coll, C.c_int32)))
    return split(out, samplers[0].cards), coll


def rccl_calls():
    """ncclAllReduce calls completed inside gr_allreduce_counts by this process (0 = RCCL never ran)."""
    return int(_lib.lib().gr_rccl_calls())


def allreduce_counts(samplers):
    L = _lib.lib()
    arr = (C.c_void_p * len(samplers))(*[s.ptr for s in samplers])
    out = np.zeros((samplers[0].n_variants, samplers[0].H), dtype=np.uint64)
    _chk(L.gr_allreduce_counts(arr, len(samplers), _p(out, C.c_uint64)))
    return out
