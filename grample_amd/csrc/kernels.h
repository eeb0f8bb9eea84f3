// Host-visible launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"

namespace gr {

struct CollapseArgs {
    uint32_t n_entries;      // size of the new factor's table
    uint32_t card_v;         // Card of the collapsed variable
    uint32_t n_new;          // arity of the new factor
    uint32_t n_fac;          // factors containing the variable
    const uint32_t *new_card;   // [n_new]
    const int32_t *new_fixed;   // [n_new] FixedVal of each new-factor variable
    const uint32_t *fac_off;    // [n_fac+1] offsets into fac_pos / fac_stride
    const int32_t *fac_pos;     // per factor scope position: index into the new factor's vars, or -1 = the collapsed variable
    const uint32_t *fac_stride; // stride of that scope position in the factor's table
    const uint32_t *fac_tab;    // [n_fac] offset of the factor's table in `tables`
    const double *tables;       // f64 log tables of those factors
    double *out_table;          // [n_entries] log space after the 1e-6 rule (function.go:126-142)
    double *out_partial;        // [gridDim.x][card_v] block sums of p per value
};

constexpr int kMaxSupportedCard = 64; // largest Card the register-resident conditional handles

hipError_t launch_sweep(const SweepParams &p, int max_card, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global, uint32_t blocks,
                        uint32_t threads, uint32_t lds_bytes, hipStream_t s);
hipError_t launch_init_state(unsigned char *state, const int32_t *card, const int32_t *fixed, uint32_t n_vars, uint32_t pitch,
                             uint32_t chain_begin, uint32_t chain_end, uint64_t chain_offset, uint64_t seed, uint32_t epoch,
                             hipStream_t s);
hipError_t launch_build_cpt(const VariantDesc &vd, const uint32_t *progs, const uint32_t *prog_off, const float *views,
                            const uint32_t *cprogs, const uint32_t *cprog_off, uint32_t *cpt, const int32_t *card,
                            const int32_t *fixed, uint32_t n_vars, uint32_t max_cfg, int max_card, hipStream_t s);
hipError_t launch_conditional(const VariantDesc &vd, const uint32_t *progs, const uint32_t *prog_off, const float *views,
                              uint32_t v, const int32_t *state, double *probs, int max_card, hipStream_t s);
hipError_t launch_collapse(const CollapseArgs &a, uint32_t blocks, hipStream_t s);
hipError_t launch_scan_vars(const ScanArgs &a, uint32_t n_quads, uint32_t pitch_quads, unsigned short *out, hipStream_t s);
hipError_t launch_stop_time(const ScanArgs &a, uint32_t cw, uint32_t batch, uint32_t max_batches, uint32_t *out, hipStream_t s);
hipError_t launch_ring_halves(const ScanArgs &a, uint32_t n_steps, uint32_t half, const unsigned char *log, uint32_t log_pitch, uint32_t log_rows,
                              uint32_t log_vals, const uint32_t *limit, uint32_t *hist, uint32_t *status, uint32_t n_seg, uint32_t seg_chunks,
                              uint32_t *seg_cnt, hipStream_t s);
// AdvanceChain under GR_FLAG_SCAN_PER_CHAIN: per-chain stopping times and ChainHistory halves (log: one value byte per chain and step)
hipError_t launch_stop_time_pc(const ScanArgs &a, uint32_t n_chains, uint32_t n_free_max, uint32_t cw, uint32_t batch, uint32_t max_batches, uint32_t *out, hipStream_t s);
hipError_t launch_ring_halves_pc(const ScanArgs &a, uint32_t n_chains, uint32_t n_free_max, uint32_t half, const unsigned char *log, uint32_t log_pitch,
                                 const uint32_t *limit, uint32_t *hist, uint32_t *status, hipStream_t s);
// value-parallel form of the direct path for large-Card pairwise models (sweep_pw4.hip): 64-chain blocks of 256 lanes
hipError_t launch_sweep_pw4(const SweepParams &p, uint32_t blocks, uint32_t lds_bytes, hipStream_t s);
// fast binary / memoised path (sweep_bin.hip)
hipError_t launch_scan_plan(const FastParams &p, hipStream_t s);
hipError_t launch_sweep_strict(const FastParams &p, int slots, bool packed, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_bin(const FastParams &p, int slots, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_par4(const FastParams &p, int slots, bool packed, uint32_t blocks, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_bits(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s); // 4 lanes per chain, bit-packed state (fast_form 11)
// wide step-parallel form (8 or 16 lanes per chain; sweep_bin.hip, GR_PAR4_PART 5): its own scan plan + sweep, burn-in / counting only
hipError_t launch_scan_plan_wide(const FastParams &p, int lpc, hipStream_t s);
hipError_t launch_sweep_wide(const FastParams &p, int lpc, int slots, uint32_t groups, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_wide_dense(const FastParams &p, uint32_t groups, uint32_t lds, hipStream_t s); // 8 lanes, <= 64 registers (experiment)
// chromatic form of the systematic scan (sweep_chroma.hip): one wavefront per chain, blocks of 64 * p.chroma_wpb lanes
hipError_t launch_sweep_chroma(const FastParams &p, int n_lane_slots, int slots, uint32_t blocks, uint32_t lds, hipStream_t s);
// step-parallel form of the direct conditional (sweep_dir4.hip): Card <= 4, arity <= 4, 64-chain blocks of 256 lanes
hipError_t launch_sweep_dir4(const FastParams &p, int max_card, bool views_in_lds, bool entries_in_lds, bool chained, uint32_t blocks, uint32_t lds, hipStream_t s);
constexpr uint32_t kParStateStride = 80; // must match kParStride in sweep_bin.hip

} // namespace gr
