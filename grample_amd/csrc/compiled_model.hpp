// Device image of one chain group's model ("variant"): what NewGibbsSimple's
// bookkeeping (gibbs-simple.go:68-82: log tables + varFuncs) becomes on the
// GPU.  Everything the per-update code needs is laid out so that, with the
// updated variable wave-uniform, all metadata comes through scalar loads and
// the only per-lane traffic is chain state and table rows.
#pragma once
#include <cstdint>
#include <vector>

#include "host_model.hpp"

namespace gr {

// Per-variable "program" (uint32 words), direct path:
//   [0] Card | rowlen << 8 | n_factors << 16
//   [1] histogram offset of the variable (sum of Card of lower ids)
//   per factor containing the variable, in Model.Funcs order:
//     [.] offset (floats) of this (factor, variable) VIEW table inside the variant's view pool
//     [.] n_others
//     n_others x { other var id, stride (floats) }
// A view table is the factor's log table re-laid with the updated variable's
// value fastest: view[o * rowlen + val], o = mixed-radix index of the OTHER
// scope variables in scope order.  rowlen = 2 for Card <= 2, else Card rounded
// up to 4, so the Card candidate values are one aligned vector read.
//
// Memoised path, per variable:
//   [0] Card | n_neighbours << 8
//   [1] histogram offset
//   [2] offset (words) of the variable's threshold table inside the variant's table pool
//   n_neighbours x { neighbour var id, mixed-radix multiplier }
// table[(cfg) * (Card-1) + i] = i-th cumulative threshold (u32) of the
// conditional given blanket configuration cfg = sum mult * x_neighbour.
struct VariantImage {
    std::vector<uint32_t> prog;     // direct programs, all variables
    std::vector<uint32_t> prog_off; // [n_vars] word offset of each variable's direct program
    std::vector<float> views;       // view-table pool (f32 log potentials)
    // pairwise form of the direct program (only when every factor has arity <= 2): per variable
    //   [0] Card | rowlen << 8 | n_factors << 16, [1] histogram offset, [2..3] pad, then per factor one
    //   16-byte entry {view offset, other var (the variable itself for a unary factor), stride (0 for unary), 0}
    // fixed stride => every factor's entry is an independent scalar load.
    bool pairwise = false;
    std::vector<uint32_t> pprog, pprog_off;
    // value-parallel form of the pairwise program (k_sweep_pw4), per variable:
    //   [0] Card | rowlen << 8 | n_slots << 16 (n_slots = factors rounded up to 8), [1] histogram offset, [2..3] 64-bit filter
    //   of the variable's blanket (bit u mod 64; itself included), then per
    //   slot one 16-byte entry {view BYTE offset, other var * 64 (its LDS state row), BYTE stride, 0}; the slots past the
    //   last factor point at the all-zero row `zero_row` with stride 0, so the kernel needs no guards.
    std::vector<uint32_t> vprog, vprog_off;
    uint32_t zero_row = 0;          // float offset of an all-zero row (64 floats) at the end of `views`
    uint32_t max_others = 0;        // largest (arity - 1) of any factor of a selectable variable
    uint32_t max_view_stride = 0;   // largest view-table stride (floats) of an "other" scope variable
    uint32_t max_neighbours = 0;    // largest blanket (free variables only, the variable itself excluded)
    std::vector<uint32_t> free_vars;// selectable variables: FixedVal < 0 and not Collapsed (sampler.go:146-156)
    // GR_FLAG_SCAN_SYSTEMATIC: the selectable variables in the order a systematic sweep updates them -- by colour class of a
    // greedy colouring of the factor graph (variables of one class share no factor, so their updates commute), ascending id
    // inside a class; color_off[k] .. color_off[k + 1] = the positions of class k in sys_order
    std::vector<uint32_t> sys_order, color_off;
    bool cpt_ok = false;            // memoised path available
    std::vector<uint32_t> cprog;    // memoised programs
    std::vector<uint32_t> cprog_off;// [n_vars]
    uint32_t cpt_words = 0;         // size of the threshold pool
    uint32_t max_cfg = 0;           // largest blanket configuration count of any variable
};

struct CompileLimits {
    uint32_t max_cpt_words = 1u << 20;  // threshold pool budget (words) of one chain group for the memoised path: pools up to
                                        // ~28 K words are staged in LDS, larger ones (to 4 MB) are read from global memory / L2
    uint32_t max_cpt_neighbours = 16;
};

inline uint32_t RowLen(int card) { return card <= 2 ? 2u : (uint32_t)((card + 3) & ~3); }

// `m` must be in log space (every Function.IsLog).  Throws gr::Error.
VariantImage CompileVariant(const Model &m, const CompileLimits &lim);

// var -> indices into m.Funcs containing it, in Funcs order (gibbs-simple.go:72-81)
std::vector<std::vector<int>> VarFuncs(const Model &m);
// varNeighbors (gibbs-collapsed.go:44-66): sorted ids of all vars sharing a factor with v, v included
std::vector<std::vector<int>> VarNeighbors(const Model &m);

} // namespace gr
