#include "compiled_model.hpp"

#include <algorithm>
#include <set>

#include "../../include/grample_hip.h"

namespace gr {

std::vector<std::vector<int>> VarFuncs(const Model &m) {
    std::vector<std::vector<int>> vf(m.Vars.size());
    for (size_t f = 0; f < m.Funcs.size(); ++f)
        for (int v : m.Funcs[f].Vars) vf[v].push_back((int)f);
    return vf;
}

std::vector<std::vector<int>> VarNeighbors(const Model &m) {
    std::vector<std::set<int>> s(m.Vars.size());
    for (const auto &f : m.Funcs)
        for (int v : f.Vars)
            for (int u : f.Vars) s[v].insert(u);
    std::vector<std::vector<int>> out(m.Vars.size());
    for (size_t v = 0; v < s.size(); ++v) out[v].assign(s[v].begin(), s[v].end());
    return out;
}

VariantImage CompileVariant(const Model &m, const CompileLimits &lim) {
    const int N = (int)m.Vars.size();
    VariantImage im;
    im.prog_off.assign(N, 0);
    im.cprog_off.assign(N, 0);
    for (const auto &f : m.Funcs) {
        if (!f.IsLog) throw Error(GR_E_INVALID, "Function " + f.Name + " is not in log space");
        std::set<int> uniq(f.Vars.begin(), f.Vars.end());
        if (uniq.size() != f.Vars.size()) throw Error(GR_E_UNSUPPORTED, "Function " + f.Name + " lists a variable twice");
    }
    const auto vf = VarFuncs(m);
    std::vector<uint32_t> hoff(N, 0);
    for (int v = 1; v < N; ++v) hoff[v] = hoff[v - 1] + (uint32_t)m.Vars[v - 1].Card;

    for (int v = 0; v < N; ++v)
        if (m.Vars[v].FixedVal < 0 && !m.Vars[v].Collapsed) im.free_vars.push_back((uint32_t)v);
    if (im.free_vars.empty()) throw Error(GR_E_INVALID, "No Variables to select");

    {   // ---- systematic scan order: greedy colouring in ascending id, smallest colour no neighbour uses
        const auto nbrs = VarNeighbors(m);
        std::vector<int> color(N, -1);
        int n_col = 0;
        for (uint32_t v : im.free_vars) {
            std::vector<char> used((size_t)n_col + 1, 0);
            for (int u : nbrs[v])
                if (u != (int)v && color[u] >= 0) used[(size_t)color[u]] = 1;
            int c = 0;
            while (used[(size_t)c]) ++c;
            color[v] = c;
            n_col = std::max(n_col, c + 1);
        }
        im.color_off.assign((size_t)n_col + 1, 0);
        for (int c = 0; c < n_col; ++c) {
            im.color_off[(size_t)c] = (uint32_t)im.sys_order.size();
            for (uint32_t v : im.free_vars)
                if (color[v] == c) im.sys_order.push_back(v);
        }
        im.color_off[(size_t)n_col] = (uint32_t)im.sys_order.size();
    }

    // ---- direct programs + view tables
    for (int v = 0; v < N; ++v) {
        const Variable &var = m.Vars[v];
        im.prog_off[v] = (uint32_t)im.prog.size();
        const bool selectable = var.FixedVal < 0 && !var.Collapsed;
        if (selectable && vf[v].empty())
            throw Error(GR_E_INVALID, "There are no functions for var " + var.Name + " (ID=" + std::to_string(v) + ")");
        const int C = var.Card;
        const uint32_t rowlen = RowLen(C);
        const uint32_t nfac = selectable ? (uint32_t)vf[v].size() : 0u;
        if (nfac > 0xFFFFu) throw Error(GR_E_UNSUPPORTED, "variable is in more than 65535 factors");
        im.prog.push_back((uint32_t)C | (rowlen << 8) | (nfac << 16));
        im.prog.push_back(hoff[v]);
        if (!selectable) continue;
        for (int fi : vf[v]) {
            const Function &f = m.Funcs[fi];
            const int ar = (int)f.Vars.size();
            int p = -1;
            for (int j = 0; j < ar; ++j)
                if (f.Vars[j] == v) p = j;
            // strides of every scope position in the ORIGINAL table (last var fastest, function.go:180-202)
            std::vector<long long> tstride(ar);
            long long d = 1;
            for (int j = ar - 1; j >= 0; --j) { tstride[j] = d; d *= m.Vars[f.Vars[j]].Card; }
            // others in scope order; view index o = mixed radix over them (last other fastest)
            std::vector<int> oth;
            for (int j = 0; j < ar; ++j)
                if (j != p) oth.push_back(j);
            std::vector<long long> ostride(oth.size());
            long long od = 1;
            for (int k = (int)oth.size() - 1; k >= 0; --k) { ostride[k] = od; od *= m.Vars[f.Vars[oth[k]]].Card; }
            const long long n_o = od;
            while (im.views.size() % 4) im.views.push_back(0.f); // 16-byte aligned table start
            const size_t tab_off = im.views.size();
            if (tab_off + (size_t)n_o * rowlen > 0x7FFFFFFFull) throw Error(GR_E_UNSUPPORTED, "view tables exceed 2^31 floats");
            im.views.resize(tab_off + (size_t)n_o * rowlen, 0.f);
            std::vector<int> digits(oth.size(), 0);
            for (long long o = 0; o < n_o; ++o) {
                long long base = 0;
                for (size_t k = 0; k < oth.size(); ++k) base += digits[k] * tstride[oth[k]];
                for (int val = 0; val < C; ++val)
                    im.views[tab_off + (size_t)o * rowlen + val] = (float)f.Table[(size_t)(base + val * tstride[p])];
                for (int k = (int)oth.size() - 1; k >= 0; --k) { // odometer over the others
                    if (++digits[k] < m.Vars[f.Vars[oth[k]]].Card) break;
                    digits[k] = 0;
                }
            }
            im.prog.push_back((uint32_t)tab_off);
            im.prog.push_back((uint32_t)oth.size());
            im.max_others = std::max<uint32_t>(im.max_others, (uint32_t)oth.size());
            for (size_t k = 0; k < oth.size(); ++k) {
                im.prog.push_back((uint32_t)f.Vars[oth[k]]);
                im.prog.push_back((uint32_t)(ostride[k] * rowlen));
                im.max_view_stride = std::max<uint32_t>(im.max_view_stride, (uint32_t)(ostride[k] * rowlen));
            }
        }
    }
    while (im.views.size() % 4) im.views.push_back(0.f);
    im.zero_row = (uint32_t)im.views.size();
    im.views.resize(im.views.size() + 64, 0.f); // an all-zero row: what the padding slots of the value-parallel programs read

    // ---- pairwise fixed-stride form of the same programs
    im.pairwise = true;
    for (const auto &f : m.Funcs)
        if (f.Vars.size() > 2) im.pairwise = false;
    im.pprog_off.assign(N, 0);
    if (im.pairwise) {
        for (int v = 0; v < N; ++v) {
            while (im.pprog.size() % 4) im.pprog.push_back(0);
            im.pprog_off[v] = (uint32_t)im.pprog.size();
            const uint32_t *pg = im.prog.data() + im.prog_off[v];
            const uint32_t nf = pg[0] >> 16;
            im.pprog.push_back(pg[0]);
            im.pprog.push_back(pg[1]);
            im.pprog.push_back(0);
            im.pprog.push_back(0);
            const uint32_t *q = pg + 2;
            for (uint32_t f = 0; f < nf; ++f) {
                const uint32_t toff = q[0], no = q[1];
                im.pprog.push_back(toff);
                im.pprog.push_back(no ? q[2] : (uint32_t)v);
                im.pprog.push_back(no ? q[3] : 0u);
                im.pprog.push_back(0);
                q += 2 + 2 * no;
            }
        }
        while (im.pprog.size() % 4) im.pprog.push_back(0);
        // value-parallel form: byte offsets, LDS row offsets, slots padded to 8 with the zero row
        im.vprog_off.assign(N, 0);
        for (int v = 0; v < N; ++v) {
            im.vprog_off[v] = (uint32_t)im.vprog.size();
            const uint32_t *pg = im.pprog.data() + im.pprog_off[v];
            const uint32_t nf = pg[0] >> 16, slots = (nf + 7u) & ~7u;
            im.vprog.push_back((pg[0] & 0xffffu) | (slots << 16));
            im.vprog.push_back(pg[1]);
            // words 2..3: 64-bit filter of the variable's blanket, itself included -- bit (u mod 64) is set for every u whose
            // update does not commute with this one.  A clear bit proves independence (the kernel then runs two consecutive
            // steps together); with more than 64 variables a set bit may be a false alarm, which only costs the pairing.
            uint64_t filt = 1ull << (v & 63);
            for (uint32_t f = 0; f < nf; ++f) filt |= 1ull << (pg[4 + 4 * f + 1] & 63u);
            im.vprog.push_back((uint32_t)filt);
            im.vprog.push_back((uint32_t)(filt >> 32));
            for (uint32_t f = 0; f < slots; ++f) {
                const uint32_t *e = pg + 4 + 4 * f;
                const bool real = f < nf;
                im.vprog.push_back((real ? e[0] : im.zero_row) * 4u);
                im.vprog.push_back((real ? e[1] : (uint32_t)v) * 64u);
                im.vprog.push_back(real ? e[2] * 4u : 0u);
                im.vprog.push_back(0);
            }
        }
    }

    // ---- memoised programs: blanket (minus evidence variables, whose value is a constant)
    im.cpt_ok = true;
    uint64_t words = 0;
    std::vector<uint64_t> tab_words(N, 0); // table size of every selectable variable
    for (int v = 0; v < N; ++v) {
        const Variable &var = m.Vars[v];
        im.cprog_off[v] = (uint32_t)im.cprog.size();
        const bool selectable = var.FixedVal < 0 && !var.Collapsed;
        std::vector<int> nb;
        if (selectable) {
            std::set<int> s;
            for (int fi : vf[v])
                for (int u : m.Funcs[fi].Vars)
                    if (u != v && m.Vars[u].FixedVal < 0) s.insert(u);
            nb.assign(s.begin(), s.end());
        }
        uint64_t cfgs = 1;
        std::vector<uint64_t> mult(nb.size());
        for (int k = (int)nb.size() - 1; k >= 0; --k) {
            mult[k] = cfgs;
            cfgs *= (uint64_t)m.Vars[nb[k]].Card;
            if (cfgs > (1ull << 24)) { cfgs = (1ull << 24); im.cpt_ok = false; }
        }
        if (nb.size() > lim.max_cpt_neighbours) im.cpt_ok = false;
        im.max_neighbours = std::max<uint32_t>(im.max_neighbours, (uint32_t)nb.size());
        im.cprog.push_back((uint32_t)var.Card | ((uint32_t)nb.size() << 8));
        im.cprog.push_back(hoff[v]);
        im.cprog.push_back((uint32_t)words);
        for (size_t k = 0; k < nb.size(); ++k) {
            im.cprog.push_back((uint32_t)nb[k]);
            im.cprog.push_back((uint32_t)mult[k]);
        }
        if (selectable) {
            tab_words[v] = cfgs * (uint64_t)(var.Card > 1 ? var.Card - 1 : 1);
            words += tab_words[v];
            im.max_cfg = std::max<uint32_t>(im.max_cfg, (uint32_t)cfgs);
        }
        if (words > lim.max_cpt_words) im.cpt_ok = false;
    }
    im.cpt_words = im.cpt_ok ? (uint32_t)words : 0u;
    // Pool order: small tables first.  A kernel that can keep only part of the pool in LDS (k_sweep_par4's split mode) stages
    // a PREFIX of it -- with this order the prefix holds as many variables' tables as possible and the few large ones (the
    // neighbours of a collapsed variable) stay in global memory.
    if (im.cpt_ok) {
        std::vector<int> order;
        for (int v = 0; v < N; ++v)
            if (tab_words[v]) order.push_back(v);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return tab_words[a] < tab_words[b]; });
        uint64_t at = 0;
        for (int v : order) {
            im.cprog[im.cprog_off[v] + 2] = (uint32_t)at;
            at += tab_words[v];
        }
    }
    return im;
}

} // namespace gr
