"""numpy re-statement of the DEVICE sampling process (not of the reference):
Philox4x32-10 streams, per-wavefront scan sequence, threshold draws, counting.

Given the conditional thresholds (fetched from the device with gr_conditional,
so no float arithmetic is re-done here) it replays chains bit for bit.  Used to
check the sweep kernel's indexing, RNG counters, state layout and histogram
against an independent implementation at small sizes.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
DOM_DRAW, DOM_SCAN, DOM_INIT, DOM_SCAN_CHAIN, DOM_DRAW_SYS = 1, 2, 3, 5, 6


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays (or scalars)."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint32) for x in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def mul_hi(a, b):
    return ((np.asarray(a, dtype=np.uint64) * np.uint64(b)) >> np.uint64(32)).astype(np.uint32)


def init_state(card, fixed, n_chains, seed, chain_offset=0, epoch=0):
    """k_init_state: evidence value, else floor(u * Card / 2^32)."""
    n = len(card)
    g = np.arange(n_chains, dtype=np.uint64) + np.uint64(chain_offset)
    st = np.zeros((n, n_chains), dtype=np.int64)
    for v0 in range(0, n, 4):
        r = philox4x32_10(np.full(n_chains, v0 >> 2, dtype=np.uint32), np.full(n_chains, epoch, dtype=np.uint32),
                          g.astype(np.uint32), (np.uint32(DOM_INIT) + ((g >> np.uint64(32)).astype(np.uint32) << np.uint32(4))),
                          seed & 0xFFFFFFFF, seed >> 32)
        for j in range(4):
            v = v0 + j
            if v < n:
                st[v] = fixed[v] if fixed[v] >= 0 else mul_hi(r[j], card[v])
    return st


def systematic_order(card, fixed, scopes, collapsed=()):
    """The sweep order of GR_FLAG_SCAN_SYSTEMATIC (compiled_model.cpp): greedy colouring of the selectable variables in ascending
    id (smallest colour no already-coloured neighbour uses; neighbours = variables sharing a factor), then by (colour, id)."""
    n = len(card)
    free = [v for v in range(n) if fixed[v] < 0 and v not in collapsed]
    nb = [set() for _ in range(n)]
    for sc in scopes:
        for a in sc:
            for b in sc:
                if a != b:
                    nb[a].add(b)
    color = {}
    for v in free:
        used = {color[u] for u in nb[v] if u in color}
        c = 0
        while c in used:
            c += 1
        color[v] = c
    return sorted(free, key=lambda v: (color[v], v)), color


class Emulator:
    """Replays the sweep kernel for n_chains chains of one chain group.

    thresholds(v, state_column) -> uint64 array of Card-1 cumulative thresholds.
    """

    def __init__(self, card, fixed, free_vars, thresholds, n_chains, seed, chain_offset=0, per_chain=False, systematic=None):
        self.per_chain = per_chain
        self.systematic = systematic  # the sweep order (list of variables) of GR_FLAG_SCAN_SYSTEMATIC, or None
        self.card, self.fixed = list(card), list(fixed)
        self.free = list(free_vars)
        self.thr = thresholds
        self.n_chains, self.seed, self.off = n_chains, seed, chain_offset
        self.state = init_state(card, fixed, n_chains, seed, chain_offset)
        self.hoff = np.concatenate([[0], np.cumsum(card)]).astype(int)
        self.counts = np.zeros(int(sum(card)), dtype=np.uint64)
        self.step = 0
        self.record = None  # list of (variable per chain, value per chain, live mask) per counted step when set to []
        self.frozen = np.zeros(n_chains, dtype=bool)  # chains that have finished their AdvanceChain: steps pass, nothing changes

    def run(self, n_steps, counting):
        k0, k1 = self.seed & 0xFFFFFFFF, self.seed >> 32
        g = np.arange(self.n_chains, dtype=np.uint64) + np.uint64(self.off)
        wave = (g >> np.uint64(6)).astype(np.uint32)
        for s in range(self.step, self.step + n_steps):
            if self.systematic is not None:  # update s resamples order[s mod n] with word (sweep & 3) of philox_sys_draw(sweep, pos, chain)
                nfree = len(self.systematic)
                sweep, pos = divmod(s, nfree)
                sq = sweep >> 2
                dr = philox4x32_10(np.full(self.n_chains, sq & 0xFFFFFFFF, dtype=np.uint32), np.full(self.n_chains, pos | ((sq >> 32) << 20), dtype=np.uint32),
                                   g.astype(np.uint32), np.uint32(DOM_DRAW_SYS) + ((g >> np.uint64(32)).astype(np.uint32) << np.uint32(4)), k0, k1)[sweep & 3]
                v = self.systematic[pos]
                vals = np.zeros(self.n_chains, dtype=np.int64)
                for c in range(self.n_chains):
                    if self.frozen[c]:
                        continue
                    T = self.thr(v, self.state[:, c])
                    val = int(np.sum(np.uint64(dr[c]) >= T))
                    self.state[v, c] = val
                    vals[c] = val
                    if counting:
                        self.counts[self.hoff[v] + val] += np.uint64(1)
                if counting and self.record is not None:
                    self.record.append((np.full(self.n_chains, v), vals, ~self.frozen))
                continue
            q, j = s >> 2, s & 3
            ql, qh = np.uint32(q & 0xFFFFFFFF), np.uint32(q >> 32)
            if self.per_chain:  # strict mode: the scan stream is keyed by the chain, not by its 64-chain group
                sc = philox4x32_10(np.full(self.n_chains, ql), np.full(self.n_chains, qh), g.astype(np.uint32),
                                   np.uint32(DOM_SCAN_CHAIN) + ((g >> np.uint64(32)).astype(np.uint32) << np.uint32(4)), k0, k1)[j]
            else:
                sc = philox4x32_10(np.full(self.n_chains, ql), np.full(self.n_chains, qh), wave,
                                   np.full(self.n_chains, DOM_SCAN, dtype=np.uint32), k0, k1)[j]
            dr = philox4x32_10(np.full(self.n_chains, ql), np.full(self.n_chains, qh), g.astype(np.uint32),
                               np.uint32(DOM_DRAW) + ((g >> np.uint64(32)).astype(np.uint32) << np.uint32(4)), k0, k1)[j]
            ks = mul_hi(sc, len(self.free))
            vals = np.zeros(self.n_chains, dtype=np.int64)
            for c in range(self.n_chains):
                if self.frozen[c]:
                    continue
                v = self.free[int(ks[c])]
                T = self.thr(v, self.state[:, c])
                val = int(np.sum(np.uint64(dr[c]) >= T))
                self.state[v, c] = val
                vals[c] = val
                if counting:
                    self.counts[self.hoff[v] + val] += np.uint64(1)
            if counting and self.record is not None:
                self.record.append((np.asarray([self.free[int(k)] for k in ks]), vals, ~self.frozen))
        self.step += n_steps
