// Structures shared by the host launch code and the gfx950 kernels.
#pragma once
#include <stdint.h>

namespace gr {

// One per chain group; all offsets index the sampler-wide pools below.
struct VariantDesc {
    uint32_t prog_base;    // words, into progs
    uint32_t progoff_base; // entries, into prog_off ([n_vars] per variant)
    uint32_t views_base;   // floats, into views (multiple of 4)
    uint32_t views_len;    // floats (multiple of 4)
    uint32_t free_base;    // entries, into free_vars
    uint32_t n_free;
    uint32_t cprog_base;   // words, into cprogs
    uint32_t cprogoff_base;// entries, into cprog_off
    uint32_t cpt_base;     // words, into cpt
    uint32_t cpt_len;      // words
    uint32_t rec_base;     // records (kRecDwords each), into recs: one per entry of this group's free list
    uint32_t pprog_base;   // words, into pprogs (pairwise fixed-stride direct programs); pprog_off shares progoff_base
    uint32_t dent_base, dent_len; // 16-byte units, into the factor entries of the step-parallel direct form (k_sweep_dir4)
    uint32_t vprog_base;   // words, into vprogs (value-parallel pairwise programs, k_sweep_pw4); vprog_off shares progoff_base
    uint32_t csl_base, csl_n; // k_sweep_chroma: first lane slot of this group (64 entries each) in chroma_slots, number of lane slots
    uint32_t zero_row;     // floats from views_base: the group's all-zero view row (what continuation entries of k_sweep_dir4's chained form load)
};

struct SweepParams {
    const VariantDesc *variants;
    const uint32_t *progs;
    const uint32_t *prog_off;
    const float *views;
    const uint32_t *free_vars;
    const uint32_t *cprogs;
    const uint32_t *cprog_off;
    const uint32_t *cpt;
    const uint32_t *pprogs;     // pairwise fixed-stride direct programs (null unless every group is pairwise)
    const uint32_t *pprog_off;
    const uint32_t *vprogs;     // value-parallel pairwise programs (k_sweep_pw4), same condition
    const uint32_t *vprog_off;
    const unsigned short *scan_vars; // k_sweep_pw4: the launch's variables, 4 per quad and 64-chain group (k_scan_vars)
    uint32_t scan_pitch;             // quads per group row of scan_vars
    unsigned char *state;       // [n_vars][pitch] chain-minor: one coalesced 64-B line per wave per variable
    unsigned long long *counts; // [n_variants][H]
    uint32_t n_vars;
    uint32_t H;                 // sum of Card
    uint32_t chain0;            // first local chain of this launch (multiple of the block size)
    uint32_t n_chains;          // one past the last local chain of this launch
    uint32_t pitch;             // row pitch of state (chains rounded up to 256)
    uint32_t chains_per_variant;
    uint32_t chain_offset_lo, chain_offset_hi; // global id of local chain 0 (multiple of 64)
    uint32_t seed_lo, seed_hi;
    uint32_t step0_lo, step0_hi; // first update index of this launch
    uint32_t n_steps;           // updates per chain in this launch
    uint32_t counting;          // 0 = burn-in (chain.go:166-172), 1 = oneSample(true)
    uint32_t sequential;        // 1 = never co-issue commuting updates
    uint32_t lds_hist_off;      // byte offsets inside dynamic LDS
    uint32_t lds_tab_off;
    // optional draw log (ChainHistory rebuild): log[(chain >> 6) * log_pitch + log_off + step - step0] = number of
    // chains of the wavefront whose binary update drew 1; null = off
    unsigned char *log;
    uint32_t log_pitch, log_off;
    const uint32_t *limit;      // with the log: per 64-chain group, how many steps of the window it executes (its own AdvanceChain
                                // stopping time); later steps of the window leave the group untouched.  win_off = steps of the
                                // window before this launch
    uint32_t win_off;
    uint32_t log_vals;          // bytes per logged step: 1 = the number of ones (all-binary models), else max Card (one count per value)
    uint32_t log_rows;          // rows per 64-chain group in the log (4 when the 4-lane form writes it); this kernel uses the first
    const uint32_t *sys_order;  // GR_FLAG_SCAN_SYSTEMATIC: the sweep order (same pool layout as free_vars), else null
    uint32_t prio_shift;        // wave_prio.h: the wavefront priority rotates every 2^prio_shift trips of the kernel's loop (>= 32: never)
};

// Fast binary/memoised path (sweep_bin.hip).  A RECORD is one 64-byte line per
// selectable variable:
//   w0 state-row byte offset (var * T)      w1 histogram byte offset (4 * offset of value 0)
//   w2 LDS byte offset of the variable's threshold table     w3 number of neighbours
//   unpacked (<= 6 neighbours): w[4+2i] neighbour state-row byte offset, w[5+2i] 4 * mixed-radix multiplier
//   packed (7..12 neighbours):  w[4+i]  = row offset | (4 * multiplier) << 16
// padding slots point at the variable's own row with multiplier 0.
constexpr uint32_t kRecDwords = 16;

struct FastParams {
    const VariantDesc *variants;
    const uint32_t *free_vars;
    const uint32_t *cprogs;
    const uint32_t *cprog_off;
    const uint32_t *cpt;
    const uint32_t *recs;
    uint32_t *plan;             // [n_chains/64][plan_pitch] u32, written by k_scan_plan
    unsigned char *state;
    unsigned long long *counts;
    uint32_t plan_pitch;        // words per wavefront row (multiple of 4)
    uint32_t rec_stride;        // bytes per record: 64, or 128 when a variable has 13..16 neighbours (4-lane form only)
    uint32_t n_vars, H, n_chains, pitch, chains_per_variant;
    uint32_t chain_offset_lo, chain_offset_hi;
    uint32_t seed_lo, seed_hi;
    uint32_t step0_lo, step0_hi; // multiple of 4
    uint32_t n_steps;            // multiple of 4
    uint32_t counting;
    uint32_t lds_hist_off, lds_tab_off;
    uint32_t lds_plan_off;       // per-wave plan ring: 2 x 64 quads x 16 B
    // step-parallel form (k_sweep_par4): its own records (80-byte state rows) and LDS layout
    unsigned char *log;          // optional draw log, see SweepParams (4-lane form: one row per wavefront = 16 chains)
    uint32_t log_pitch, log_off;
    const uint32_t *limit;       // see SweepParams::limit (window offset of the launch = log_off)
    uint32_t log_vals;           // bytes per logged step (k_sweep_dir4: 1 = the number of ones, else max Card: one count per value)
    const uint32_t *recs_par;
    uint32_t par_hist_off, par_tab_off, par_rec_off, par_plan_off, par_scr_off;
    uint32_t *err;               // device error word (bit 0: the dynamic LDS array does not start at LDS address 0)
    uint32_t prio_shift;         // wave_prio.h: the wavefront priority rotates every 2^prio_shift trips of the kernel's loop (>= 32: never)
    unsigned long long *dbg_times; // measurements only (GR_DEV_BLOCK_TIMES): per block of k_sweep_par4 {start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID}, or null
    uint32_t par_recmode;        // k_sweep_par4 record source: 0 LDS per step, 1 LDS fetched together, 2 global prefetched, other = auto
    uint32_t plan_slots, plan_nfree_max; // k_scan_plan's LDS table: most neighbours of a variable, most selectable variables of a group
    uint32_t strict_state_off;   // k_sweep_strict with 4 waves per block: state regions of waves 1..3
    uint32_t strict_rec_stride;  // k_sweep_strict: record stride in LDS (rec_stride + 16 when it fits: bank spreading)
    const uint32_t *dent;        // k_sweep_dir4: factor entries (16 bytes each); its step headers travel in recs_par
    const float *views;          // k_sweep_dir4: the view-table pool (staged in LDS at dir_view_off when it fits)
    uint32_t dir_view_off;       // k_sweep_dir4: LDS byte offset of the staged view tables
    uint32_t dir_ent_off;        // k_sweep_dir4: LDS byte offset of the staged factor entries
    uint32_t par_stride;         // k_sweep_par4: bytes between the state rows of consecutive variables (80, or 64 for very large models)
    uint32_t tab_global;         // k_sweep_par4: threshold tables too large for LDS -- read from global memory (record word 2 = byte
                                 // offset inside the group's pool); records from global memory too (record mode 2)
    uint32_t par_spread;         // k_sweep_par4: 1 = launch the 4 wavefronts of a 64-chain group as 4 blocks of 64 lanes (few groups: spread
                                 // them over the CUs), 0 = one block of 256 lanes per group
    // k_sweep_chroma (systematic scan): lane slots (64 entries per slot: index into the group's free list, or 0xffffffff), its records
    // (state offsets inside a chain's row instead of row offsets), LDS layout {chain rows, tables}, chains (= wavefronts) per block
    const uint32_t *chroma_slots;
    const uint32_t *recs_chroma;
    const uint32_t *sys_pos;     // position in the sweep order of every entry of the free list (the draw stream's key)
    const uint32_t *sys_rank;    // the inverse: free-list index of every position of the sweep order.  Non-null = the plan-based
                                 // kernels (k_scan_plan + k_sweep_dir4) run the systematic scan
    uint32_t chroma_row, chroma_tab_off, chroma_wpb, chroma_hist_off;
    uint32_t chroma_part;        // k_sweep_chroma(_l): 1 = the launch may start / end inside a sweep (per chain group); 0 = whole sweeps from a sweep boundary
    uint32_t chroma_rec_off;     // 0 = the lanes' records live in registers (k_sweep_chroma); else their LDS offset (k_sweep_chroma_l)
    uint32_t par_tab_words;      // k_sweep_par4 with tab_global: the first par_tab_words words of every group's pool are staged in LDS all
                                 // the same (split pool; record word 2 carries bit 31 for a table that stayed in global memory)
};

// A scan-plan word (k_scan_plan -> the fast kernels): byte offset of the step's record / header (a multiple of the record
// stride, 32 at least) in bits 5..29, and
//   bit 0      the step reads a neighbour written earlier in its quad (k_sweep_bin recomputes it)
//   bits 1-2   dependency level of the step inside its quad; bits 3-4 the highest level in the quad (the same in its 4 words)
//   bit 31     level != 0;  bit 30  highest level != 0 -- copies placed so that ONE unsigned compare answers the two questions a
//              4-lane kernel asks at every step: word < 2^31 <=> "commit in the first pass", word >= 2^30 <=> "the quad needs
//              more passes" (bit 31 implies bit 30)
constexpr uint32_t kPlanLevelBit = 1u << 31, kPlanDepsBit = 1u << 30, kPlanOffMask = 0x3fffffffu;

// Arguments of the scan-stream kernels behind gr_advance_chain (k_stop_time, k_ring_halves): one thread per
// 64-chain scan group replays its variable-selection stream.
struct ScanArgs {
    const VariantDesc *variants;
    const uint32_t *free_vars;
    uint32_t n_groups;           // active chains / 64
    uint32_t chains_per_variant;
    uint32_t chain_offset_lo, chain_offset_hi;
    uint32_t seed_lo, seed_hi;
    uint32_t n_vars;
    uint32_t step0_lo, step0_hi; // first step of the window
    const uint32_t *hoff;        // [n_vars + 1] histogram offset of every variable (prefix sum of Card)
    uint32_t H;
    const uint32_t *sys_order;   // GR_FLAG_SCAN_SYSTEMATIC: the sweep order (free_base layout); null = random scan
};

} // namespace gr
