"""Independent pure-Python UAI / evidence / MAR reader used ONLY by tests.

Restates model/uai.go:20-332 (preprocess: drop blank lines and lines starting
with 'c'; whitespace token stream) so that the product's C++ reader can be
cross-checked against a second implementation, and so the oracle can be fed
without going through product code.
"""
import os
from types import SimpleNamespace

import numpy as np


def preprocess(text, req_prefix=""):
    """uaiPreprocess (uai.go:20-50) -> (joined text, line count)."""
    out = []
    found = len(req_prefix) < 1
    for ln in text.split("\n"):
        ln = ln.strip()
        if len(ln) < 1 or ln[0] == "c":
            continue
        if not found:
            if ln.startswith(req_prefix):
                found = True
            else:
                continue
        out.append(ln)
    return "\n".join(out), len(out)


def read_model(text):
    """ReadModel (uai.go:53-179)."""
    if len(text) < 15:
        raise ValueError("Invalid data buffer")
    body, n = preprocess(text)
    if n < 1:
        raise ValueError("No lines found in file")
    t = body.split()
    if len(t) < 6:
        raise ValueError("Invalid data: too few fields")
    p = 0
    typ = t[p]; p += 1
    if typ not in ("BAYES", "MARKOV"):
        raise ValueError("Unknown model type")
    nv = int(t[p]); p += 1
    if nv < 1:
        raise ValueError("Invalid variable count")
    card = [int(x) for x in t[p:p + nv]]; p += nv
    if any(c < 1 for c in card):
        raise ValueError("Invalid card")
    nf = int(t[p]); p += 1
    if nf < 1:
        raise ValueError("Invalid Clique count")
    scopes = []
    for _ in range(nf):
        k = int(t[p]); p += 1
        if k < 1:
            raise ValueError("Invalid variable count for Clique")
        s = [int(x) for x in t[p:p + k]]; p += k
        if any(v < 0 or v >= nv for v in s):
            raise ValueError("Invalid var idx")
        scopes.append(s)
    tables = []
    for s in scopes:
        ts = int(t[p]); p += 1
        exp = int(np.prod([card[v] for v in s]))
        if ts != exp:
            raise ValueError("Read table size != previous Clique size")
        tables.append(np.array([float(x) for x in t[p:p + ts]], dtype=np.float64)); p += ts
    return SimpleNamespace(type=typ, card=card, scopes=scopes, tables=tables, fixed=[-1] * nv)


def apply_evidence(text, m):
    """ApplyEvidence (uai.go:183-249)."""
    body, n = preprocess(text)
    if n < 1:
        raise ValueError("Invalid data buffer: there is no data")
    if n > 2:
        raise ValueError("only understand evidence files with 1 or 2 lines")
    t = body.split()
    p = 0
    if n == 2:
        sc = int(t[p]); p += 1
        if sc == 0:
            return
        if sc > 1:
            raise ValueError("only single sample evidence currently supported")
    cnt = int(t[p]); p += 1
    if cnt < 1:
        return
    for _ in range(cnt):
        idx = int(t[p]); p += 1
        if idx < 0 or idx >= len(m.card):
            raise ValueError("Read incorrect variable index")
        if m.fixed[idx] != -1:
            raise ValueError("variable had previous fixedval")
        val = int(t[p]); p += 1
        if val < 0 or val >= m.card[idx]:
            raise ValueError("Read invalid value")
        m.fixed[idx] = val


def read_mar(text):
    """ReadMargSolution (uai.go:252-332): skips to the first line starting with MAR."""
    body, n = preprocess(text, "MAR")
    if n < 1:
        raise ValueError("No lines in file")
    t = body.split()
    if len(t) < 4 or t[0] != "MAR":
        raise ValueError("Unknown solution file type")
    p = 1
    nv = int(t[p]); p += 1
    out = []
    for _ in range(nv):
        c = int(t[p]); p += 1
        m = np.array([float(x) for x in t[p:p + c]], dtype=np.float64); p += c
        s = 0.0  # NormMarginal (variable.go:106-147): sequential sum like the Go loop (a pairwise sum can flip the 1e-8 test)
        for x in m:
            s += float(x)
        if abs(s - 1.0) >= 1e-8:
            m = np.full(c, 1.0 / c) if abs(s) < 1e-8 else m / s
        out.append(m)
    return out


def load(res_dir, name, use_evidence=False):
    path = os.path.join(res_dir, name)
    m = read_model(open(path).read())
    if use_evidence:
        apply_evidence(open(path + ".evid").read(), m)
    return m


def load_mar(res_dir, name, suffix=".MAR"):
    return read_mar(open(os.path.join(res_dir, name + suffix)).read())
