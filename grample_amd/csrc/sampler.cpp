// C-ABI implementation (include/grample_hip.h): host orchestration of the
// gfx950 Gibbs kernels.  No CPU sampling path exists in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../../include/grample_hip.h"
#include "compiled_model.hpp"
#include "device_layout.h"
#include "host_model.hpp"
#include "kernels.h"
#include "philox.h"

using namespace gr;

#define GR_VISIBLE __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;
std::atomic<uint64_t> g_rccl_calls{0}; // ncclAllReduce calls completed by gr_allreduce_counts (gr_rccl_calls)

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Error(GR_E_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

template <class F>
int guarded(F &&f) {
    try {
        f();
        return GR_OK;
    } catch (const Error &e) {
        return fail(e.code, e.what());
    } catch (const std::exception &e) {
        return fail(GR_E_INVALID, e.what());
    }
}

// Pinned host staging for every transfer of the control path.  A copy between pageable host memory and the device makes
// the runtime pin the pages on the fly and unpin them afterwards, and the unmap that follows was measured to hold back the
// NEXT kernel of the stream by 10 - 20 ms (an adaptive run of 126 group activations: 2.4 s of such waits around 0.3 s of
// work).  So uploads are staged here and stay valid until the stream has drained (gr_sampler::sync resets the arena);
// downloads land here and are copied out after the wait (gr_sampler::download).
struct PinnedArena {
    struct Block { unsigned char *p; size_t cap; };
    std::vector<Block> blocks;
    size_t cur = 0, used = 0; // block in use, bytes used in it
    PinnedArena() = default;
    PinnedArena(const PinnedArena &) = delete;
    PinnedArena &operator=(const PinnedArena &) = delete;
    ~PinnedArena() { for (auto &b : blocks) (void)hipHostFree(b.p); }
    void *take(size_t bytes) {
        bytes = (std::max<size_t>(bytes, 1) + 255u) & ~size_t(255);
        while (cur < blocks.size() && used + bytes > blocks[cur].cap) { ++cur; used = 0; }
        if (cur == blocks.size()) {
            Block b{nullptr, std::max<size_t>(bytes, size_t(4) << 20)};
            hip_check(hipHostMalloc(reinterpret_cast<void **>(&b.p), b.cap, hipHostMallocDefault), "hipHostMalloc");
            blocks.push_back(b);
            used = 0;
        }
        void *out = blocks[cur].p + used;
        used += bytes;
        return out;
    }
    // only once nothing in flight reads or writes the arena.  Blocks are kept for the next call (allocating pinned memory is
    // slow) up to a watermark: a one-off bulk transfer -- gr_state_download of a million chains is 400 MB -- must not stay
    // page-locked for the sampler's lifetime.
    void reset() {
        cur = 0; used = 0;
        size_t total = 0;
        for (const auto &b : blocks) total += b.cap;
        while (total > kKeep && !blocks.empty()) {
            total -= blocks.back().cap;
            (void)hipHostFree(blocks.back().p);
            blocks.pop_back();
        }
    }
    static constexpr size_t kKeep = size_t(64) << 20;
};

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;   // elements in use
    size_t cap = 0; // elements allocated
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = cap = 0;
    }
    // (contents undefined afterwards.)  An allocation that is large enough is kept: hipFree synchronises the device and
    // hipMalloc is slow, and rebuild() runs once per activated chain group in an adaptive run.  Safe because every C-ABI
    // call drains the stream before it returns and all work on a buffer is ordered on that one stream.
    void alloc(size_t count) {
        if (p && count <= cap) { n = count; return; }
        reset();
        const size_t want = std::max<size_t>(count + (count < (4u << 20) ? count / 4 : 0), 1); // headroom: the pools of an adaptive run grow group by group
        hip_check(hipMalloc(reinterpret_cast<void **>(&p), want * sizeof(T)), "hipMalloc");
        n = count; cap = want;
    }
    void upload(const std::vector<T> &h, hipStream_t s, PinnedArena &stage) {
        alloc(h.size());
        if (h.empty()) return;
        void *src = stage.take(h.size() * sizeof(T));
        memcpy(src, h.data(), h.size() * sizeof(T));
        hip_check(hipMemcpyAsync(p, src, h.size() * sizeof(T), hipMemcpyHostToDevice, s), "hipMemcpy H2D");
    }
};

constexpr uint32_t kLdsBudget = 160 * 1024;   // LDS per CU on gfx950
constexpr uint32_t kSplitBlocks = 2;          // split threshold pool (k_sweep_par4<.., TABG 2>): resident blocks per CU the LDS part must leave room for
constexpr uint32_t kMaxStepsPerLaunch = 1u << 20; // keeps the u32 wave histograms far from overflow
constexpr uint64_t kMaxCptWordsTotal = 4ull << 20; // memoised threshold pools of all chain groups together: 16 MB (L2 / MALL resident)

} // namespace

struct gr_model {
    Model m;
};
struct gr_solution {
    std::vector<Variable> vars;
};

struct gr_sampler {
    gr_sampler_opts o{};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_vars = 0, H = 0, max_card = 0;
    std::vector<int32_t> card, fixed;
    std::vector<Model> vmodel;        // one log-space clone per chain group (root.go:385, gibbs-simple.go:72-76)
    std::vector<VariantImage> image;
    std::vector<uint32_t> epoch;      // start-state re-draw counter per group
    bool dirty = true;                // device pools need rebuilding
    bool started = false;             // any burn-in / advance done
    // configuration
    uint32_t T = 0, blocks = 0, lds = 0, lds_hist_off = 0, lds_tab_off = 0, lds_plan_off = 0, pitch = 0, cpv = 0;
    bool cpt = false, tab_lds = false;
    bool fast = false;                // binary + memoised + whole wavefronts: sweep_bin.hip
    int fast_slots = 0;               // max neighbours of any selectable variable
    uint32_t plan_steps = 0;          // steps per fast launch (plan row length)
    DevBuf<uint32_t> d_recs, d_plan, d_recs_par, d_err;
    DevBuf<uint32_t> d_recs_bits;     // records of the bit-packed 4-lane form (k_sweep_bits, fast_form 11)
    bool bits = false;                // ... which this sampler runs (outside logged windows)
    uint32_t bits_lds = 0;
    bool par_packed = false;          // step-parallel records use one dword per neighbour
    uint32_t rec_stride = 64;         // bytes per record (128 when a variable has 13..16 neighbours)
    bool par = false;                 // step-parallel form (k_sweep_par4): 4 lanes per chain
    bool par_lean = false;            // ... with the lean LDS layout: records always from global memory (large models)
    uint32_t par_stride = 80;         // ... bytes between the LDS state rows of consecutive variables
    bool sysscan = false;             // GR_FLAG_SCAN_SYSTEMATIC
    bool chroma = false;              // ... on k_sweep_chroma (binary memoised, few lane slots); else the generic kernel's systematic mode
    uint32_t chroma_row = 0, chroma_tab_off = 0, chroma_hist_off = 0, chroma_rec_off = 0, chroma_lds = 0, chroma_nslot = 0, sys_n_free = 0, chroma_wpb = 4;
    DevBuf<uint32_t> d_sys_order, d_sys_pos, d_sys_rank, d_chroma_slots, d_recs_chroma;
    DevBuf<unsigned long long> d_dbg; // GR_DEV_BLOCK_TIMES (measurements only)
    // Plan-ahead for k_sweep_par4: the scan plan of the NEXT launch is computed on a second stream while the current sweep runs (a CU that
    // holds 4 blocks of the sweep has a fifth wavefront slot free), into the other of two plan buffers; across calls the next call's
    // first launch is guessed to repeat this call's (a caller's loop of equal gr_advance calls) and checked before use.
    hipStream_t pstream = nullptr;
    hipEvent_t ev_plan = nullptr, ev_pre = nullptr;   // the plan-ahead kernel has finished / main-stream work enqueued before the current sweep
    DevBuf<uint32_t> d_plan_b;                         // the second plan buffer (d_plan is the first)
    int ahead_buf = -1;                                // buffer holding the plan computed ahead (-1: none), for steps [ahead_step0, + ahead_n) of ahead_act chains
    uint64_t ahead_step0 = 0;
    uint32_t ahead_n = 0, ahead_act = 0;
    bool ahead_inflight = false;                       // ev_plan has been recorded and not yet waited for by the main stream
    void plan_quiesce() {                              // before anything frees or rewrites what the plan-ahead kernel reads or writes
        if (pstream) (void)hipStreamSynchronize(pstream);
        ahead_buf = -1; ahead_inflight = false;
    }
    bool no_wide = false;             // fast_form 3: every automatic choice, but never the wide form (measurements)
    bool wide_ok = false;             // ... the 8 / 16 lanes-per-chain form (k_sweep_wide) may take its place when wavefronts are what a launch lacks
    uint32_t lpc_last = 4;            // lanes per chain of the last step-parallel launch (gr_sampler_get_info)
    bool par_tabg = false;            // ... and the threshold tables too (pools that do not fit LDS)
    uint32_t par_tab_words = 0;       // ... except the first par_tab_words words of every group's pool (split pool; 0 = all global)
    bool stg = false;                 // chain state stays in global memory (k_sweep<.., STG>): models whose state does not fit LDS
    bool pw4 = false;                 // value-parallel form of the direct path (k_sweep_pw4): pairwise, 8 < Card <= 16
    uint32_t pw4_hist_off = 0, pw4_lds = 0;
    bool dir4 = false;                // step-parallel form of the direct conditional (k_sweep_dir4): Card <= 4, blankets too large to memoise
    bool dir4_vlds = false, dir4_elds = false; // ... with the view tables / the factor entries staged in LDS
    // wave_prio.h: the wavefront priority of a sweep kernel rotates every 2^shift trips of its loop; GR_DEV_PRIO_SHIFT overrides
    // the per-kernel default (measurements only; 32 = no rotation)
    static uint32_t prio_shift(uint32_t dflt) {
        static const int env = [] { const char *e = getenv("GR_DEV_PRIO_SHIFT"); return e ? atoi(e) : -1; }();
        return env >= 0 ? (uint32_t)env : dflt;
    }
    // Lanes per chain of the step-parallel form for a launch over `groups` 64-chain groups: 4 (k_sweep_par4), or 8 / 16 (k_sweep_wide)
    // while the launch would otherwise leave SIMDs short of wavefronts.
    // measured (scripts/dev_wide.py -> profiles/r03_dev_wide.txt): 16 lanes pay while 16 x groups <= 4 096 wavefronts AND either
    // the block's LDS footprint lets a CU hold four of them or the whole launch fits the chip as one-wavefront blocks (each
    // with its own copy of the tables: the first rounds of an adaptive run); 8 lanes while 8 x groups <= 4 096; beyond that the
    // 4-lane form has the chip full and the wider forms' extra passes only cost
    uint32_t auto_lpc(uint32_t groups) const {
        if (!(par && wide_ok && !no_wide)) return 4;
        const uint32_t fit = kLdsBudget / std::max(par_rec_off, 1u);
        if (o.fast_form == 8 || o.fast_form == 10) return 8;
        if (o.fast_form == 9) return 16;
        if (o.fast_form == 0 && groups * 16u <= 4096u && (fit >= 4u || groups * 16u <= 256u * fit)) return 16;
        if (o.fast_form == 0 && groups * 8u <= 4096u) return 8;
        return 4;
    }
    bool sys_wide_ok() const { return sysscan && fast && par && wide_ok && !no_wide && !strict_fast && (o.fast_form == 0 || o.fast_form == 8 || o.fast_form == 9); } // the systematic scan on k_sweep_wide
    // The chromatic kernel for this launch?  With the lane slots' records in registers (Grids: 1.3e12) always; with the records staged in
    // LDS (k_sweep_chroma_l: Segmentation_11, collapsed groups -- one fat block per CU) only while the launch is short of wavefronts:
    // one wavefront per chain beats every other form up to ~8 192 chains (an adaptive run's first rounds: 6 ms instead of 27), beyond that
    // the 8-lane form on the sweep order is the faster one (Segmentation_11 collapsed, 32 768 chains: 3.9e11 against 1.6e11)
    bool use_chroma(uint32_t act) const { return sysscan && chroma && (chroma_rec_off == 0 || act <= 8192u || !sys_wide_ok()); }
    bool sys_wide() const { return sys_wide_ok() && !use_chroma(active_chains()); }
    bool sys_bits() const { return sysscan && bits && fast && par && !strict_fast && !use_chroma(active_chains()) && !sys_wide(); }
    bool dir4_chain = false;          // ... with chained factor entries (a factor of arity > 4 or with strides past 16 bits: collapse factors)
    // factor entries of variable v's list in the form this sampler uses: one per factor, or one per pair of other scope variables
    uint32_t dir4_entries(const VariantImage &im, uint32_t v) const {
        const uint32_t *pg = im.prog.data() + im.prog_off[v];
        const uint32_t nf = pg[0] >> 16;
        if (!dir4_chain) return nf;
        uint32_t n = 0;
        const uint32_t *q = pg + 2;
        for (uint32_t f = 0; f < nf; ++f) { n += std::max(1u, (q[1] + 1u) / 2u); q += 2 + 2 * q[1]; }
        return n;
    }
    uint32_t dir_hist_off = 0, dir_view_off = 0, dir_ent_off = 0, dir_lds = 0, dir_slots = 0;
    DevBuf<uint32_t> d_dhdr, d_dent;  // ... its step headers (32 bytes per selectable variable) and factor entries (16 bytes per factor)
    DevBuf<unsigned char> d_log;        // draw log of gr_advance_chain windows: kept between calls (hundreds of MB: allocating and
                                        // freeing it every window cost more than the window's kernels), grown when a window needs more
    DevBuf<unsigned short> d_scan_vars; // k_sweep_pw4: the variables of one launch (k_scan_vars), 8 bytes per quad and 64-chain group
    uint32_t pw4_plan_steps = 0;        // steps per k_sweep_pw4 launch (row length of d_scan_vars, multiple of 4)
    bool strict_fast = false;         // GR_FLAG_SCAN_PER_CHAIN on memoised tables (k_sweep_strict, the 4-lane form's LDS layout)
    uint32_t par_hist_off = 0, par_tab_off = 0, par_rec_off = 0, par_plan_off = 0, par_scr_off = 0, par_lds = 0;
    // device pools
    DevBuf<VariantDesc> d_variants;
    DevBuf<uint32_t> d_progs, d_prog_off, d_free, d_cprogs, d_cprog_off, d_cpt, d_pprogs, d_pprog_off, d_vprogs, d_vprog_off;
    bool pairwise = false;            // every group's model has arity <= 2: fixed-stride direct programs
    DevBuf<float> d_views;
    DevBuf<int32_t> d_card, d_fixed;
    DevBuf<unsigned char> d_state;
    DevBuf<unsigned long long> d_counts;
    DevBuf<unsigned long long> d_reduced; // gr_allreduce_counts: the cross-device total (scratch; d_counts is never overwritten)
    // Scratch of the control-path calls, kept between calls like every other device buffer (an allocation and a free per
    // call cost more than the kernels they serve, and a free synchronises the device): gr_collapse ...
    DevBuf<uint32_t> cs_new_card, cs_fac_off, cs_fac_stride, cs_fac_tab;
    DevBuf<int32_t> cs_new_fixed, cs_fac_pos;
    DevBuf<double> cs_tables, cs_out, cs_part;
    // ... gr_advance_chain ...
    DevBuf<uint32_t> ac_out, ac_limit, ac_hist, ac_status, ac_hoff, ac_segcnt;
    bool cur_log_pc = false;          // the current window's log is per chain (strict mode): [group][step][64 value bytes]
    // ... gr_conditional
    DevBuf<int32_t> cd_state;
    DevBuf<double> cd_probs;
    std::vector<VariantDesc> h_variants;
    uint64_t step = 0, counted = 0;
    // adaptive sampler (N2): groups [0, n_active) run; the rest wait to be activated by gr_adapt
    uint32_t n_active = 0;
    std::vector<uint64_t> counted_g;             // counted updates per chain, per group
    std::vector<double> win_first, win_second;   // [n_variants * H] halves of the last convergence window
    double win_cw = 0;
    Model base_linear;                           // the caller's model as given (linear tables): Adapt clones it
    // draw log of the current gr_advance_chain window (null = off)
    unsigned char *cur_log = nullptr;
    uint32_t cur_log_pitch = 0, cur_log_rows = 1, cur_log_vals = 1;
    const uint32_t *cur_limit = nullptr;  // per scan group: steps of the window it executes (gr_advance_chain)
    uint64_t skipped_updates = 0;         // updates NOT made inside such windows by groups that stopped before the slowest one
    uint64_t cur_log_step0 = 0;
    uint32_t active_chains() const { return o.n_variants > 1 ? n_active * (o.n_chains / o.n_variants) : o.n_chains; }
    double kernel_ms_last = 0;
    uint32_t launches_last = 0;

    ~gr_sampler() {
        if (stream) (void)hipStreamSynchronize(stream); // a failed call may have left copies in flight that use the arena
        if (pstream) { (void)hipStreamSynchronize(pstream); (void)hipStreamDestroy(pstream); }
        if (ev_plan) (void)hipEventDestroy(ev_plan);
        if (ev_pre) (void)hipEventDestroy(ev_pre);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
    void use() { hip_check(hipSetDevice(o.device), "hipSetDevice"); }
    void sync() {
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        stage.reset(); // nothing in flight any more
    }
    // device -> caller memory, through the pinned arena (see PinnedArena); returns after the data has arrived
    void download(void *dst, const void *dsrc, size_t bytes) {
        if (!bytes) { sync(); return; }
        void *tmp = stage.take(bytes);
        hip_check(hipMemcpyAsync(tmp, dsrc, bytes, hipMemcpyDeviceToHost, stream), "D2H");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        memcpy(dst, tmp, bytes); // (before the arena is reset: a reset may give blocks above the watermark back)
        stage.reset();
    }
    PinnedArena stage;

    void configure();
    void configure_paths();
    void configure_impl(bool try_strict_fast);
    void rebuild();
    void init_state(uint32_t variant);
    void run(uint64_t updates, bool counting);
    void burn_groups(uint32_t first, uint32_t last, uint32_t updates);
};

// Pick block size, LDS layout and conditional path for the current images.
void gr_sampler::configure() {
    configure_paths();
    // the wide step-parallel form: packed records read from global memory, record offsets within the 26 bits of its plan word
    {
        uint64_t rec_bytes = 0;
        for (const auto &im : image) rec_bytes += (uint64_t)im.free_vars.size() * rec_stride;
        wide_ok = fast && par && !strict_fast && par_packed && rec_bytes < (1ull << 26) && !o.block_threads &&
                  (o.fast_form == 0 || o.fast_form == 2 || o.fast_form >= 6) && par_rec_off <= kLdsBudget;
        // (fast_form 10: experiment, see launch_sweep_wide_dense)
    }
    // systematic scan: the chromatic kernel when every group is binary + memoised with at most 8 lane slots (colour classes cut
    // into chunks of 64 variables) and 12 neighbours and the threshold pool fits LDS next to the chains' rows (groups may sweep
    // different numbers of variables: the kernel takes any range of steps); otherwise see run()
    chroma = false;
    if (sysscan) {
        if (stg) throw Error(GR_E_UNSUPPORTED, "GR_FLAG_SCAN_SYSTEMATIC needs the chain state in LDS (model too large, or GR_FLAG_STATE_GLOBAL)");
        sys_n_free = (uint32_t)image[0].free_vars.size();
        bool ok = fast && cpt && fast_slots <= 12 && n_vars < 65536 && !(o.flags & (GR_FLAG_DIRECT | GR_FLAG_SEQUENTIAL | GR_FLAG_CHAIN_PER_LANE)) &&
                  (o.n_chains % 16u) == 0 && o.fast_form != 1;
        uint32_t nslot = 0, cpt_max = 0;
        for (const auto &im : image) {
            uint32_t ns = 0;
            for (size_t c = 0; c + 1 < im.color_off.size(); ++c) ns += (im.color_off[c + 1] - im.color_off[c] + 63u) / 64u;
            nslot = std::max(nslot, ns);
            cpt_max = std::max<uint32_t>(cpt_max, (im.cpt_words + 3u) & ~3u);
            sys_n_free = std::max<uint32_t>(sys_n_free, (uint32_t)im.free_vars.size()); // (the longest sweep: sizes the block histogram)
            if (ok)
                for (uint32_t v : im.free_vars) {
                    const uint32_t *cp = im.cprog.data() + im.cprog_off[v];
                    const uint32_t nn = (cp[0] >> 8) & 0xffu;
                    for (uint32_t i = 0; i < nn; ++i) ok = ok && cp[4 + 2 * i] * 4u < 65536u;
                }
        }
        chroma_nslot = nslot;
        chroma_row = ((uint32_t)n_vars + 15u) & ~15u;
        chroma_wpb = 16; // chains (= wavefronts) per block: 16 consecutive chains make the state transfers 16-byte segments of the
                         // global rows, and every block stages its own copy of the threshold tables
        if (const char *e = getenv("GR_DEV_CHROMA_WPB")) chroma_wpb = (uint32_t)atoi(e); // (measurements only)
        if (chroma_wpb != 1 && chroma_wpb != 2 && chroma_wpb != 4 && chroma_wpb != 8 && chroma_wpb != 16) chroma_wpb = 16;
        chroma_tab_off = chroma_wpb * chroma_row;
        chroma_hist_off = chroma_tab_off + cpt_max * 4u;
        chroma_rec_off = 0;
        chroma_lds = chroma_hist_off + 8u * sys_n_free;
        if (nslot > 4 || fast_slots > 8) { // more lane slots / neighbours than registers hold: k_sweep_chroma_l stages the lanes' records in LDS
            const uint32_t rb = 16u + 4u * (fast_slots <= 8 ? 8u : 12u);
            chroma_rec_off = (chroma_lds + 15u) & ~15u;
            chroma_lds = chroma_rec_off + nslot * 64u * rb;
        }
        chroma = ok && nslot <= (chroma_rec_off ? 8u : 4u) && chroma_lds <= kLdsBudget;
    }
    // value-parallel direct form: every group pairwise, the direct path chosen, Card 9..16, whole 64-chain scan groups
    pw4 = false;
    bool all_pw = true;
    for (const auto &im : image) all_pw = all_pw && im.pairwise;
    if (all_pw && !fast && !cpt && !stg && max_card > 8 && max_card <= 16 && !(o.flags & (GR_FLAG_SCAN_PER_CHAIN | GR_FLAG_SEQUENTIAL | GR_FLAG_CHAIN_PER_LANE)) &&
        o.fast_form != 1 && (o.n_chains % 64u) == 0 && (o.n_variants == 1 || cpv % 64u == 0) && !o.block_threads) {
        pw4_hist_off = ((uint32_t)n_vars * 64u + 15u) & ~15u;
        pw4_lds = pw4_hist_off + 4u * (uint32_t)H * 4u;
        // measured on ObjectDetection_13: 1.9x the chain-per-lane form at 32 768 chains (2 wavefronts per SIMD instead of
        // one on half of the SIMDs), still 1.5x at 262 144 (fewer table bytes in flight per wavefront) -- so: whenever it fits
        pw4 = pw4_lds <= kLdsBudget;
    }
    // step-parallel direct form: blankets too large to memoise, at most 4 values per variable, state rows (80 bytes apart)
    // that fit the 16-bit fields of a factor entry: up to 819 variables.  Arity <= 4 with byte strides < 64 KiB: one entry
    // per factor; otherwise (collapse factors) the chained entry form
    dir4 = false;
    // (GR_FLAG_DIRECT keeps the generic kernel: it is the cross-check every other form is compared with)
    // ... and models that are memoisable but not binary (the generic memoised kernel is chain-per-lane too: Pedigree_11 without
    // evidence 3.5e10 there, 9.3e10 here), unless they are so small that most quads are one dependency chain (sample.uai)
    uint32_t nfree_min = 0xffffffffu;
    for (const auto &im : image) nfree_min = std::min<uint32_t>(nfree_min, (uint32_t)im.free_vars.size());
    if (!fast && (!cpt || nfree_min >= 16u) && !stg && !pw4 && max_card <= 4 && !(o.flags & (GR_FLAG_SCAN_PER_CHAIN | GR_FLAG_SEQUENTIAL | GR_FLAG_CHAIN_PER_LANE | GR_FLAG_DIRECT)) && !(sysscan && chroma) &&
        o.fast_form != 1 && (o.n_chains % 64u) == 0 && (o.n_variants == 1 || cpv % 64u == 0) && !o.block_threads &&
        (uint64_t)n_vars * kParStateStride < 65536u && H < 16384) {
        bool ok = true;
        uint32_t views_max = 0, ent_max = 0;
        dir_slots = 1;
        // a factor with more than 3 other scope variables, or a view stride past 16 bits (a collapse factor: up to 12
        // variables, up to 2^23 entries), takes the chained entry form: two others with 32-bit strides per entry, the
        // partial offsets of a factor's entries added up along the chain
        dir4_chain = false;
        for (const auto &im : image) dir4_chain = dir4_chain || im.max_others > 3u || (uint64_t)im.max_view_stride * 4u >= 65536u;
        for (const auto &im : image) {
            ok = ok && im.max_neighbours <= 255u && (uint64_t)im.views.size() * 4u < (1ull << 32);
            views_max = std::max<uint32_t>(views_max, (uint32_t)im.views.size());
            dir_slots = std::max(dir_slots, im.max_neighbours);
            uint32_t n_ent = 0; // factor entries, every selectable variable's list padded to a multiple of 4
            for (uint32_t v : im.free_vars) n_ent += (dir4_entries(im, v) + 3u) & ~3u;
            ent_max = std::max(ent_max, n_ent);
        }
        dir_hist_off = ((uint32_t)n_vars * kParStateStride + 15u) & ~15u;
        dir_view_off = dir_hist_off + (((uint32_t)H * 4u + 15u) & ~15u);
        // Factor entries go to LDS while that still leaves a CU two blocks (or all it needs).  View tables only for binary
        // models (8-byte rows): with 16-byte rows the per-lane row reads are what saturates the LDS pipe, and the otherwise
        // idle vector-memory path serves them better from L1/L2 (CSP_13: 1.85e10 with the views in LDS, 2.93e10 without).
        const uint32_t need = std::min<uint32_t>((o.n_chains / 64u + 255u) / 256u, 2u);
        const uint32_t with_views = dir_view_off + views_max * 4u;
        dir4_vlds = !dir4_chain && max_card <= 2 && with_views <= kLdsBudget && kLdsBudget / with_views >= need;
        dir_ent_off = dir4_vlds ? with_views : dir_view_off;
        const uint32_t with_ent = dir_ent_off + ent_max * 16u;
        dir4_elds = with_ent <= kLdsBudget && kLdsBudget / with_ent >= need;
        dir_lds = dir4_elds ? with_ent : dir_ent_off;
        // the scan-plan kernel keeps every selectable variable's neighbour list in LDS (u16 each)
        uint32_t nfree_max = 1;
        for (const auto &im : image) nfree_max = std::max<uint32_t>(nfree_max, (uint32_t)im.free_vars.size());
        dir4 = ok && dir_lds <= kLdsBudget && (uint64_t)nfree_max * (2u + dir_slots) * 2u <= 64u * 1024u;
    }
}

void gr_sampler::configure_paths() {
    strict_fast = false;
    sysscan = (o.flags & GR_FLAG_SCAN_SYSTEMATIC) != 0;
    if (sysscan && (o.flags & GR_FLAG_SCAN_PER_CHAIN)) throw Error(GR_E_INVALID, "GR_FLAG_SCAN_SYSTEMATIC and GR_FLAG_SCAN_PER_CHAIN exclude each other");
    if ((o.flags & GR_FLAG_SCAN_PER_CHAIN) && !(o.flags & GR_FLAG_DIRECT)) {
        // strict mode first tries the memoised tables in the 4-lane form's LDS layout (k_sweep_strict)
        try {
            configure_impl(true);
        } catch (const Error &) {
            par = false;
        }
        if (par) { strict_fast = true; return; }
    }
    configure_impl(false);
}

void gr_sampler::configure_impl(bool try_strict_fast) {
    cpv = o.n_chains / o.n_variants;
    uint32_t views_max = 0, cpt_max = 0;
    uint64_t cpt_sum = 0;
    bool all_cpt = !(o.flags & GR_FLAG_DIRECT) && (try_strict_fast || !(o.flags & GR_FLAG_SCAN_PER_CHAIN));
    for (const auto &im : image) {
        views_max = std::max<uint32_t>(views_max, (uint32_t)im.views.size());
        cpt_max = std::max<uint32_t>(cpt_max, (im.cpt_words + 3u) & ~3u);
        all_cpt = all_cpt && im.cpt_ok;
        cpt_sum += (uint64_t)((im.cpt_words + 3u) & ~3u);
    }
    if (cpt_sum > kMaxCptWordsTotal) all_cpt = false; // (every chain group has its own pool)
    uint32_t plan_off = 0;
    auto layout = [&](uint32_t t, bool use_cpt, bool tl, uint32_t &hist_off, uint32_t &tab_off) -> uint32_t {
        uint32_t b = (uint32_t)n_vars * t;
        b = (b + 15u) & ~15u;
        hist_off = b;
        b += (t / 64u) * (uint32_t)H * 4u;
        b = (b + 15u) & ~15u;
        tab_off = b;
        if (tl) b += (use_cpt ? cpt_max : views_max) * 4u;
        b = (b + 15u) & ~15u;
        plan_off = b;
        if (use_cpt && tl) b += (t / 64u) * 1024u; // fast path: per-wave scan-plan ring (2 x 32 quads x 16 B)
        return b;
    };
    // State in global memory: forced by flag, or the fallback when no LDS layout fits (below).  Generic kernel only: thresholds /
    // views from global memory, wave-private histograms in LDS.
    stg = false;
    auto state_global = [&]() {
        if (o.flags & GR_FLAG_SCAN_PER_CHAIN) throw Error(GR_E_UNSUPPORTED, "strict per-chain scan needs the chain state in LDS (model too large, or GR_FLAG_STATE_GLOBAL)");
        for (uint32_t t : {256u, 64u}) {
            if (o.n_variants > 1 && cpv % t) continue;
            const uint32_t b = (t / 64u) * (uint32_t)H * 4u;
            if (b > kLdsBudget) continue;
            stg = true; fast = false; par = false; par_lean = false; par_tabg = false;
            T = t; lds = b; lds_hist_off = 0; lds_tab_off = b; lds_plan_off = b; cpt = all_cpt; tab_lds = false;
            blocks = (o.n_chains + T - 1) / T;
            return;
        }
        throw Error(GR_E_UNSUPPORTED, "model too large: the per-wavefront histogram (sum of Card * 4 bytes) does not fit LDS");
    };
    if (o.flags & GR_FLAG_STATE_GLOBAL) { state_global(); return; }
    // fast path (sweep_bin.hip): every selectable variable binary, memoised tables, whole wavefronts
    bool fast_ok = all_cpt && !(o.flags & GR_FLAG_SEQUENTIAL) && (o.n_chains % 64u) == 0 && n_vars < 65536 && H < 16384;
    int slots = 0;
    uint32_t max_mult = 1; // largest mixed-radix multiplier of any neighbour (packed records keep 4*mult in 16 bits)
    if (fast_ok)
        for (uint32_t g = 0; g < o.n_variants && fast_ok; ++g)
            for (uint32_t v : image[g].free_vars) {
                const uint32_t *cp = image[g].cprog.data() + image[g].cprog_off[v];
                if ((cp[0] & 0xffu) != 2u) { fast_ok = false; break; }
                const uint32_t nn = (cp[0] >> 8) & 0xffu;
                slots = std::max<int>(slots, (int)nn);
                for (uint32_t i = 0; i < nn; ++i) max_mult = std::max(max_mult, cp[4 + 2 * i]);
            }
    if (slots > 16) fast_ok = false;
    const bool cpl_slots_ok = slots <= 12; // the chain-per-lane form keeps 4 records in SGPRs: 12 neighbours at most
    std::vector<uint32_t> cands;
    if (o.block_threads) {
        const uint32_t bt = o.block_threads;
        if (bt != 64 && bt != 128 && bt != 256 && bt != 512 && bt != 1024)
            throw Error(GR_E_INVALID, "block_threads must be 0, 64, 128, 256, 512 or 1024");
        if (bt > 256 && !fast_ok) throw Error(GR_E_INVALID, "block_threads > 256 needs the binary memoised path");
        cands = {bt};
    } else {
        // 256-lane blocks share one staged table copy between 4 waves; use them once
        // every CU gets a block anyway, otherwise spread single-wave blocks.
        if (o.n_chains >= 65536) cands = {256, 128, 64};
        else cands = {64};
    }
    for (int pass = 0; pass < 2; ++pass) { // pass 0: memoised path if allowed, pass 1: direct
        const bool use_cpt = (pass == 0) && all_cpt;
        if (pass == 0 && !use_cpt) continue;
        for (bool tl : {true, false}) { // tables (thresholds or views) staged in LDS, else read from global memory / L2
            bool prefer_global_tables = false;
            if (tl && use_cpt && o.fast_form == 7 && fast_ok && !try_strict_fast) continue; // form 7: the split-pool kernel is asked for
            for (uint32_t t : cands) {
                if (o.n_variants > 1 && cpv % t) continue;
                uint32_t ho, to;
                const uint32_t b = layout(t, use_cpt, tl, ho, to);
                if (b > kLdsBudget) continue;
                T = t; lds = b; lds_hist_off = ho; lds_tab_off = to; lds_plan_off = plan_off; cpt = use_cpt; tab_lds = tl;
                blocks = (o.n_chains + T - 1) / T;
                fast = fast_ok && cpl_slots_ok && use_cpt && tl && (o.n_chains % T) == 0;
                if (fast && slots > 6 && ((uint64_t)n_vars * T >= 65536u || max_mult * 4u >= 65536u)) fast = false; // packed slots are 16+16 bit
                const bool cpl_fast = fast;
                rec_stride = slots > 12 ? 128u : 64u;
                fast_slots = slots;
                if (T > 256 && !fast) continue;
                // step-parallel form: worth it while the chain-per-lane form cannot fill the SIMDs
                // step-parallel form (4 lanes per chain): wins while the chain-per-lane form leaves SIMDs with
                // a single wavefront (measured on Grids_13: 4.9e11 vs 4.1e11 updates/s at 65 536 chains,
                // 5.0e11 vs 6.2e11 at 131 072)
                par = false; par_lean = false;
                const uint32_t form = try_strict_fast ? 2u : o.fast_form; // 0 auto, 1 chain per lane, 2 four lanes per chain
                par_tabg = false;
                if (fast_ok && use_cpt && !o.block_threads && !(o.flags & GR_FLAG_CHAIN_PER_LANE) && form != 1 &&
                    (tl || (!try_strict_fast && (o.fast_form < 4 || o.fast_form >= 6))) &&
                    (form >= 2 || o.n_chains / 64u <= 1536u || !cpl_fast || (kLdsBudget / lds) * (T / 64u) < 8u)) {
                    // (last clause: a chain-per-lane block whose LDS footprint leaves a CU fewer than 8 wavefronts -- large models,
                    // Promedus_17: 125 KB behind ONE wavefront -- loses to the 4-lane form at any chain count: 6.8e10 vs 1.1e11)
                    uint32_t nfree_max = 0;
                    for (const auto &im : image) nfree_max = std::max<uint32_t>(nfree_max, (uint32_t)im.free_vars.size());
                    // 80-byte state rows spread the 4 rows of an access over the LDS banks; a model with more than 6 neighbours per
                    // variable packs row offsets into 16 bits, so above 819 variables its rows are 64 bytes apart instead
                    par_stride = (slots > 6 && (uint64_t)n_vars * kParStateStride >= 65536u && !try_strict_fast) ? 64u : kParStateStride;
                    uint32_t b2 = ((uint32_t)n_vars * par_stride + 15u) & ~15u;
                    par_hist_off = b2; b2 += ((uint32_t)H * 4u + 15u) & ~15u;
                    par_tab_off = b2; // (!tl: thresholds from global memory, k_sweep_par4<.., TABG>)
                    // ... all of them, or all but a prefix of the pool (the small tables come first, compiled_model.cpp) that is
                    // staged in LDS while a CU still holds kSplitBlocks blocks: a collapsed Segmentation_11 group has 14 - 17 K
                    // words, nearly all of it in tables of at most 256 words; only the collapsed variable's neighbours are larger
                    par_tab_words = 0;
                    if (!tl && !try_strict_fast && o.fast_form == 0) {
                        // (measured there at 65 536 chains, resident blocks per CU -> updates/s: all global 2.47e11, 4 -> 2.48e11,
                        // 3 -> 2.67e11, 2 -> 3.10e11, 1 = everything in LDS -> 1.93e11)
                        const uint32_t per_block = (kLdsBudget / kSplitBlocks) & ~15u;
                        if (per_block > b2) par_tab_words = std::min<uint32_t>((per_block - b2) / 4u & ~3u, (cpt_max + 3u) & ~3u);
                        // worth it only when most lookups land in LDS (variables are selected uniformly: count variables, not
                        // words): the split costs every step a branch, and the LDS share costs occupancy (dv-rel_3: 4 of its
                        // 1 000 tables would fit -- all global)
                        for (const auto &im : image) {
                            uint32_t in_lds = 0;
                            for (uint32_t v : im.free_vars) {
                                const uint32_t *cp = im.cprog.data() + im.cprog_off[v];
                                if (cp[2] + (1u << ((cp[0] >> 8) & 0xffu)) <= par_tab_words) ++in_lds;
                            }
                            if (in_lds * 2u < im.free_vars.size()) { par_tab_words = 0; break; }
                        }
                    }
                    // form 7 (cross-checks): the split kernel whatever the occupancy -- half of the largest pool in LDS
                    if (!tl && !try_strict_fast && o.fast_form == 7) par_tab_words = std::max<uint32_t>(4u, (cpt_max / 2u) & ~3u);
                    b2 += tl ? cpt_max * 4u : par_tab_words * 4u;
                    par_rec_off = b2; b2 += nfree_max * rec_stride;
                    par_plan_off = b2; b2 += 4u * 1024u;
                    par_scr_off = b2; b2 += 4u * 64u * 16u;
                    par_lds = b2;
                    const bool fits16 = (uint64_t)n_vars * par_stride < 65536u && max_mult * 4u < 65536u;
                    par_packed = fits16; // (unpacked records -- fewer VALU unpack instructions, 48 instead of 32 bytes fetched per
                                         // step -- measured 27 % SLOWER on Grids_13 at 65 536 chains: the record fetch is a co-limiter)
                    // Large models (Promedus_17: 916 variables, dv-rel_2: 1 200): state + tables fit, the LDS copy of the records
                    // does not.  With the records read from global memory (record mode 2) the kernel needs neither that copy nor
                    // the plan ring: the "lean" layout ends after the tables.  (Without it such models ran the chain-per-lane
                    // form with one 64-lane block per CU: 125 KB of LDS behind a single wavefront.)
                    par_lean = (!tl || b2 > kLdsBudget) && par_rec_off <= kLdsBudget && !try_strict_fast && (o.fast_form < 4 || o.fast_form >= 6);
                    // ... and whenever dropping the LDS records lets a CU hold more of the blocks it is going to get (collapsed
                    // Segmentation_11, 128 groups: 100 KB -> 62 KB per block = 2 resident blocks per CU instead of 1)
                    if (!par_lean && b2 <= kLdsBudget && !try_strict_fast && o.fast_form == 0) {
                        const uint32_t need = (o.n_chains / 64u + 255u) / 256u; // blocks per CU if all were resident
                        if (std::min(kLdsBudget / par_rec_off, need) > std::min(kLdsBudget / b2, need)) par_lean = true;
                    }
                    if (par_lean) par_lds = par_rec_off;
                    par_tabg = !tl;
                    par = par_lds <= kLdsBudget && (slots <= 6 || fits16) && (o.n_variants == 1 || cpv % 64u == 0) && (o.n_chains % 64u) == 0;
                    if (par) fast = true;
                    else par_tabg = false;
                    // Thresholds in LDS are faster than thresholds from L2 at equal occupancy (Segmentation_11: 3.6e11 vs 2.9e11),
                    // but not when they keep a CU at ONE block while it has two or more to run (the same model collapsed in
                    // 128 groups: 17 K words per group, 88 KB per block even without records -- 2.0e11 vs 2.2e11): then leave
                    // them in global memory.
                    if (par && tl && !try_strict_fast && o.fast_form == 0) {
                        const uint32_t need = (o.n_chains / 64u + 255u) / 256u;
                        if (std::min(kLdsBudget / par_lds, need) < 2u && std::min(kLdsBudget / std::max(par_tab_off, 1u), need) >= 2u) {
                            prefer_global_tables = true;
                            break;
                        }
                    }
                }
                // direct path, LDS-resident state, but the footprint keeps most wavefronts waiting for a CU (Alchemy_11 at
                // 131 072 chains: 108 KB per 128-lane block = 512 resident wavefronts for 2 048): leave the state in global
                // memory instead (2.3e10 -> 5.5e10 updates/s there; slower whenever the LDS form can keep the chip full)
                if (!fast && !use_cpt && !try_strict_fast && !(o.flags & GR_FLAG_SCAN_PER_CHAIN) && !o.block_threads &&
                    (uint64_t)(o.n_chains / 64u) > 2ull * (kLdsBudget / lds) * (T / 64u) * 256u)
                    state_global();
                return;
            }
            if (prefer_global_tables) continue; // -> the tl = false candidate
        }
    }
    state_global(); // no LDS layout holds the chain state (n_vars * 64 bytes > 160 KiB): leave it in global memory
}

// Concatenate the per-group images into the device pools and (re)build the
// memoised conditional tables on the device.
void gr_sampler::rebuild() {
    use();
    plan_quiesce(); // (the pools the plan-ahead kernel reads are about to be replaced)
    configure();
    std::vector<uint32_t> progs, prog_off, freev, cprogs, cprog_off, pprogs, pprog_off, vprogs, vprog_off;
    pairwise = true;
    for (const auto &im : image) pairwise = pairwise && im.pairwise;
    std::vector<float> views;
    // Only the groups that run are built: the groups an adaptive run has not activated yet are clones of the base model that no
    // kernel touches (every launch covers the active prefix), and an activation rebuilds anyway -- with 2 of 128 groups
    // active a full build was 64 times the work needed.
    const uint32_t n_build = o.n_variants > 1 ? n_active : o.n_variants;
    h_variants.assign(o.n_variants, VariantDesc{});
    uint32_t cpt_total = 0;
    for (uint32_t g = 0; g < n_build; ++g) {
        const VariantImage &im = image[g];
        VariantDesc &vd = h_variants[g];
        vd.prog_base = (uint32_t)progs.size();
        vd.progoff_base = (uint32_t)prog_off.size();
        vd.views_base = (uint32_t)views.size();
        vd.views_len = (uint32_t)im.views.size();
        vd.zero_row = im.zero_row;
        vd.free_base = (uint32_t)freev.size();
        vd.n_free = (uint32_t)im.free_vars.size();
        vd.cprog_base = (uint32_t)cprogs.size();
        vd.cprogoff_base = (uint32_t)cprog_off.size();
        vd.cpt_base = cpt_total;
        vd.cpt_len = cpt ? im.cpt_words : 0u;
        vd.pprog_base = (uint32_t)pprogs.size();
        pprogs.insert(pprogs.end(), im.pprog.begin(), im.pprog.end());
        pprog_off.insert(pprog_off.end(), im.pprog_off.begin(), im.pprog_off.end());
        vd.vprog_base = (uint32_t)vprogs.size();
        vprogs.insert(vprogs.end(), im.vprog.begin(), im.vprog.end());
        vprog_off.insert(vprog_off.end(), im.vprog_off.begin(), im.vprog_off.end());
        progs.insert(progs.end(), im.prog.begin(), im.prog.end());
        prog_off.insert(prog_off.end(), im.prog_off.begin(), im.prog_off.end());
        views.insert(views.end(), im.views.begin(), im.views.end());
        freev.insert(freev.end(), im.free_vars.begin(), im.free_vars.end());
        cprogs.insert(cprogs.end(), im.cprog.begin(), im.cprog.end());
        cprog_off.insert(cprog_off.end(), im.cprog_off.begin(), im.cprog_off.end());
        if (cpt) cpt_total += (im.cpt_words + 3u) & ~3u;
    }
    d_variants.upload(h_variants, stream, stage);
    d_progs.upload(progs, stream, stage);
    d_prog_off.upload(prog_off, stream, stage);
    d_views.upload(views, stream, stage);
    d_free.upload(freev, stream, stage);
    d_cprogs.upload(cprogs, stream, stage);
    d_cprog_off.upload(cprog_off, stream, stage);
    pprogs.resize(pprogs.size() + 4, 0);
    d_pprogs.upload(pprogs, stream, stage);
    d_pprog_off.upload(pprog_off, stream, stage);
    vprogs.resize(vprogs.size() + 4, 0);
    d_vprogs.upload(vprogs, stream, stage);
    d_vprog_off.upload(vprog_off, stream, stage);
    d_cpt.alloc(cpt_total + 4);
    hip_check(hipMemsetAsync(d_cpt.p, 0, (cpt_total + 4) * sizeof(uint32_t), stream), "hipMemset");
    if (cpt)
        for (uint32_t g = 0; g < n_build; ++g)
            hip_check(launch_build_cpt(h_variants[g], d_progs.p, d_prog_off.p, d_views.p, d_cprogs.p, d_cprog_off.p, d_cpt.p,
                                       d_card.p, d_fixed.p, (uint32_t)n_vars, std::max(1u, image[g].max_cfg), max_card, stream),
                      "k_build_cpt");
    if (fast) {
        std::vector<uint32_t> recs;
        uint32_t rec_count = 0;
        for (uint32_t g = 0; g < n_build; ++g) {
            const VariantImage &im = image[g];
            h_variants[g].rec_base = rec_count;
            rec_count += (uint32_t)im.free_vars.size();
            if (fast_slots > 12) continue; // chain-per-lane records hold 12 neighbours; such models run the 4-lane form only
            for (uint32_t v : im.free_vars) {
                const uint32_t *cp = im.cprog.data() + im.cprog_off[v];
                const uint32_t nn = (cp[0] >> 8) & 0xffu;
                uint32_t r[kRecDwords] = {0};
                r[0] = v * T;
                r[1] = cp[1] * 4u;
                r[2] = lds_tab_off + cp[2] * 4u;
                r[3] = nn;
                const bool packed = fast_slots > 6;
                for (uint32_t i = 0; i < (packed ? 12u : 6u); ++i) {
                    const uint32_t off = (i < nn ? cp[3 + 2 * i] : v) * T, m4 = (i < nn ? cp[4 + 2 * i] : 0u) * 4u;
                    if (packed) r[4 + i] = off | (m4 << 16);
                    else { r[4 + 2 * i] = off; r[5 + 2 * i] = m4; }
                }
                recs.insert(recs.end(), r, r + kRecDwords);
            }
        }
        d_variants.upload(h_variants, stream, stage); // now with rec_base
        d_recs.upload(recs, stream, stage);
        if (par) { // same records for the step-parallel layout: 80-byte state rows, block-level histogram
            std::vector<uint32_t> rp;
            for (uint32_t g = 0; g < n_build; ++g) {
                const VariantImage &im = image[g];
                uint32_t rank = 0;
                for (uint32_t v : im.free_vars) {
                    const uint32_t *cp = im.cprog.data() + im.cprog_off[v];
                    const uint32_t nn = (cp[0] >> 8) & 0xffu;
                    uint32_t r[32] = {0};
                    r[0] = v * par_stride;
                    r[1] = par_hist_off + 8u * rank++; // block histogram by rank in the free list (flush_rank_hist, sweep_bin.hip)
                    r[2] = (par_tabg ? 0u : par_tab_off) + cp[2] * 4u;
                    if (par_tabg && par_tab_words) // split pool: the table is in the staged prefix, or stays in global memory (bit 31)
                        r[2] = cp[2] + (1u << nn) <= par_tab_words ? par_tab_off + cp[2] * 4u : (1u << 31) | (cp[2] * 4u);
                    r[3] = nn;
                    const bool packed = par_packed;
                    for (uint32_t i = 0; i < (packed ? (rec_stride == 128u ? 16u : 12u) : 6u); ++i) {
                        const uint32_t off = (i < nn ? cp[3 + 2 * i] : v) * par_stride, m4 = (i < nn ? cp[4 + 2 * i] : 0u) * 4u;
                        if (packed) r[4 + i] = off | (m4 << 16);
                        else { r[4 + 2 * i] = off; r[5 + 2 * i] = m4; }
                    }
                    rp.insert(rp.end(), r, r + rec_stride / 4u);
                }
            }
            d_recs_par.upload(rp, stream, stage);
        }
        // fast_form 11: the 4-lane form with bit-packed state (k_sweep_bits) -- binary memoised models with <= 12 neighbours, tables in LDS
        // Chosen by the sampler itself (fast_form 0) where the STATE is what fills LDS: a model whose byte layout leaves a CU one block
        // and whose bit layout leaves it two (dv-rel_2: 1 200 variables, 144 KB -> 58 KB, 1.4e11 -> 4.8e11 updates/s); everywhere
        // else the form costs 13 more VALU instructions per 64 updates than the byte layout and is slower (profiles/r03_bits_form.txt).
        const bool bits_can = par && (!par_tabg || par_tab_words == 0) && fast_slots <= 12 && (uint64_t)n_vars * 8u < 65536u && !(o.flags & GR_FLAG_SCAN_PER_CHAIN);
        uint32_t bits_need = 0;
        if (bits_can) {
            uint32_t nfree_max = 1, cpt_max = 4;
            for (uint32_t g = 0; g < n_build; ++g) {
                nfree_max = std::max<uint32_t>(nfree_max, (uint32_t)image[g].free_vars.size());
                cpt_max = std::max<uint32_t>(cpt_max, (image[g].cpt_words + 3u) & ~3u);
            }
            const uint32_t hist_off = ((uint32_t)n_vars * 8u + 15u) & ~15u, tab_off = (hist_off + nfree_max * 8u + 15u) & ~15u;
            bits_need = par_tabg ? tab_off : (tab_off + cpt_max * 4u + 15u) & ~15u; // (a pool too large for LDS stays in global memory)
        }
        // (state-dominated: the chains' bytes alone take more than half of a CU's LDS; models whose TABLES fill LDS keep the byte layout and its wide forms)
        const bool bits_auto = o.fast_form == 0 && bits_can && !strict_fast && (uint64_t)n_vars * kParStateStride > kLdsBudget / 2u && par_lds > kLdsBudget / 2u && bits_need <= kLdsBudget / 2u;
        bits = (o.fast_form == 11 || bits_auto) && bits_can;
        if (o.fast_form == 11 && !bits) throw Error(GR_E_UNSUPPORTED, "fast_form 11 (bit-packed state) needs a binary memoised model with at most 12 neighbours per variable and its tables in LDS");
        if (bits) {
            std::vector<uint32_t> rb;
            uint32_t nfree_max = 1, cpt_max = 4;
            for (uint32_t g = 0; g < n_build; ++g) {
                const VariantImage &im = image[g];
                nfree_max = std::max<uint32_t>(nfree_max, (uint32_t)im.free_vars.size());
                cpt_max = std::max<uint32_t>(cpt_max, (im.cpt_words + 3u) & ~3u);
                uint32_t rank = 0;
                for (uint32_t v : im.free_vars) {
                    const uint32_t *cp = im.cprog.data() + im.cprog_off[v];
                    const uint32_t nn = (cp[0] >> 8) & 0xffu;
                    uint32_t r[16] = {0};
                    // absolute LDS addresses of the kernel's layout: masks | histogram pairs (by rank) | tables
                    const uint32_t hist_g = ((uint32_t)n_vars * 8u + 15u) & ~15u, tab_g = (hist_g + (uint32_t)im.free_vars.size() * 8u + 15u) & ~15u;
                    r[0] = v * 8u; r[1] = hist_g + 8u * rank; r[2] = (par_tabg ? 0u : tab_g) + cp[2] * 4u; r[3] = nn;
                    for (uint32_t i = 0; i < 12u; ++i) r[4 + i] = ((i < nn ? cp[3 + 2 * i] : v) * 8u) | ((i < nn ? cp[4 + 2 * i] * 4u : 0u) << 16);
                    rb.insert(rb.end(), r, r + 16);
                    ++rank;
                }
            }
            rb.resize(rb.size() + 16, 0);
            d_recs_bits.upload(rb, stream, stage);
            const uint32_t hist_off = ((uint32_t)n_vars * 8u + 15u) & ~15u, tab_off = (hist_off + nfree_max * 8u + 15u) & ~15u;
            bits_lds = par_tabg ? std::max(tab_off, 64u) : (tab_off + cpt_max * 4u + 15u) & ~15u;
            if (bits_lds > kLdsBudget) throw Error(GR_E_UNSUPPORTED, "fast_form 11 (bit-packed state): the tables do not fit LDS");
        }

        const uint64_t waves = o.n_chains / 64u;
        uint64_t ps = (256ull << 20) / (waves * 4ull); // <= 256 MiB of plan per launch
        ps = std::min<uint64_t>(ps, kMaxStepsPerLaunch) & ~63ull; // (a multiple of every form's iteration: 4, 16, 32 or 64 steps)
        plan_steps = (uint32_t)std::max<uint64_t>(ps, 64);
        d_plan.alloc((size_t)waves * plan_steps + 64);
    }
    if (sysscan) {
        std::vector<uint32_t> order, pos, slots, recs, rank_of_pos;
        for (uint32_t g = 0; g < n_build; ++g) {
            const VariantImage &im = image[g];
            order.insert(order.end(), im.sys_order.begin(), im.sys_order.end()); // (same pool layout as free_vars: free_base)
            std::vector<uint32_t> where((size_t)n_vars, 0);
            for (size_t i = 0; i < im.sys_order.size(); ++i) where[im.sys_order[i]] = (uint32_t)i;
            std::vector<uint32_t> rank((size_t)n_vars, 0);
            for (size_t i = 0; i < im.free_vars.size(); ++i) { pos.push_back(where[im.free_vars[i]]); rank[im.free_vars[i]] = (uint32_t)i; }
            for (uint32_t v : im.sys_order) rank_of_pos.push_back(rank[v]); // free-list index of every position (k_scan_plan's systematic mode)
            if (!chroma) continue;
            h_variants[g].csl_base = (uint32_t)(slots.size() / 64u);
            for (size_t c = 0; c + 1 < im.color_off.size(); ++c)
                for (uint32_t b = im.color_off[c]; b < im.color_off[c + 1]; b += 64u)
                    for (uint32_t l = 0; l < 64u; ++l) slots.push_back(b + l < im.color_off[c + 1] ? rank[im.sys_order[b + l]] : 0xffffffffu);
            h_variants[g].csl_n = (uint32_t)(slots.size() / 64u) - h_variants[g].csl_base;
            for (uint32_t v : im.free_vars) { // records in free-list order (rec_base), 16 words each
                const uint32_t *cp = im.cprog.data() + im.cprog_off[v];
                const uint32_t nn = (cp[0] >> 8) & 0xffu;
                uint32_t r[16] = {0};
                r[0] = v; r[1] = cp[1]; r[2] = cp[2] * 4u; r[3] = nn;
                for (uint32_t i = 0; i < 12u; ++i) r[4 + i] = (i < nn ? cp[3 + 2 * i] : v) | ((i < nn ? cp[4 + 2 * i] * 4u : 0u) << 16);
                recs.insert(recs.end(), r, r + 16);
            }
        }
        order.resize(order.size() + 4, 0); pos.resize(pos.size() + 4, 0); slots.resize(slots.size() + 64, 0xffffffffu); recs.resize(recs.size() + 16, 0);
        d_sys_order.upload(order, stream, stage);
        d_sys_pos.upload(pos, stream, stage);
        rank_of_pos.resize(rank_of_pos.size() + 4, 0);
        d_sys_rank.upload(rank_of_pos, stream, stage);
        d_chroma_slots.upload(slots, stream, stage);
        d_recs_chroma.upload(recs, stream, stage);
        d_variants.upload(h_variants, stream, stage); // now with the lane slots
    }
    if (dir4) {
        // step headers + factor entries of the direct 4-lane form, from the generic programs (compiled_model.hpp)
        std::vector<uint32_t> hdr, ent;
        uint32_t rec_count = 0;
        for (uint32_t g = 0; g < n_build; ++g) {
            const VariantImage &im = image[g];
            h_variants[g].rec_base = rec_count;
            rec_count += (uint32_t)im.free_vars.size();
            h_variants[g].dent_base = (uint32_t)(ent.size() / 4u);
            const uint32_t vbase = dir4_vlds ? dir_view_off : 0u; // views staged in LDS: entries carry absolute LDS addresses
            for (uint32_t v : im.free_vars) {
                const uint32_t *pg = im.prog.data() + im.prog_off[v];
                const uint32_t C = pg[0] & 0xffu, nf = pg[0] >> 16, n_ent = (dir4_entries(im, v) + 3u) & ~3u;
                // entries staged in LDS: absolute LDS address of the variable's list; else byte offset into the entry pool
                const uint32_t eoff = dir4_elds ? dir_ent_off + ((uint32_t)(ent.size() / 4u) - h_variants[g].dent_base) * 16u : (uint32_t)ent.size() * 4u;
                const uint32_t h8[8] = {v * kParStateStride, dir_hist_off + pg[1] * 4u, eoff, n_ent | (C << 16), 0, 0, 0, 0};
                hdr.insert(hdr.end(), h8, h8 + 8);
                const uint32_t *q = pg + 2;
                const uint32_t own = v * kParStateStride;
                if (dir4_chain) {
                    // {view byte offset | 1 = continuation, row1 | row2 << 16, stride1, stride2 (bytes)}: a factor's others two per
                    // entry; a continuation entry hands its partial offset on to the next one and loads the all-zero row itself
                    uint32_t made = 0;
                    for (uint32_t f = 0; f < nf; ++f) {
                        const uint32_t no = q[1], ne = std::max(1u, (no + 1u) / 2u);
                        for (uint32_t e = 0; e < ne; ++e) {
                            uint32_t row[2] = {own, own}, st[2] = {0, 0};
                            for (uint32_t k = 0; k < 2 && 2 * e + k < no; ++k) { row[k] = q[2 + 2 * (2 * e + k)] * kParStateStride; st[k] = q[3 + 2 * (2 * e + k)] * 4u; }
                            const bool last = e + 1 == ne;
                            ent.push_back(last ? vbase + q[0] * 4u : 1u);
                            ent.push_back(row[0] | (row[1] << 16));
                            ent.push_back(st[0]);
                            ent.push_back(st[1]);
                        }
                        made += ne;
                        q += 2 + 2 * no;
                    }
                    for (; made < n_ent; ++made) { ent.push_back(vbase + im.zero_row * 4u); ent.push_back(own | (own << 16)); ent.push_back(0); ent.push_back(0); }
                    continue;
                }
                for (uint32_t f = 0; f < n_ent; ++f) {
                    uint32_t toff = im.zero_row, row[3] = {own, own, own}, st[3] = {0, 0, 0}; // padding: the all-zero row, stride 0
                    if (f < nf) {
                        const uint32_t no = q[1];
                        toff = q[0];
                        for (uint32_t k = 0; k < no && k < 3; ++k) { row[k] = q[2 + 2 * k] * kParStateStride; st[k] = q[3 + 2 * k] * 4u; }
                        q += 2 + 2 * no;
                    }
                    ent.push_back(vbase + toff * 4u);
                    ent.push_back(row[0] | (row[1] << 16));
                    ent.push_back(row[2] | (st[0] << 16));
                    ent.push_back(st[1] | (st[2] << 16));
                }
            }
            h_variants[g].dent_len = (uint32_t)(ent.size() / 4u) - h_variants[g].dent_base;
        }
        ent.resize(ent.size() + 4, 0);
        d_variants.upload(h_variants, stream, stage); // now with rec_base
        d_dhdr.upload(hdr, stream, stage);
        d_dent.upload(ent, stream, stage);
        const uint64_t waves = o.n_chains / 64u;
        uint64_t ps = (256ull << 20) / (waves * 4ull); // <= 256 MiB of plan per launch
        ps = std::min<uint64_t>(ps, kMaxStepsPerLaunch) & ~15ull;
        plan_steps = (uint32_t)std::max<uint64_t>(ps, 16);
        d_plan.alloc((size_t)waves * plan_steps + 16);
    }
    sync();
    dirty = false;
}

void gr_sampler::init_state(uint32_t variant) {
    use();
    const uint32_t b = variant * cpv, e = (variant + 1 == o.n_variants) ? o.n_chains : b + cpv;
    hip_check(launch_init_state(d_state.p, d_card.p, d_fixed.p, (uint32_t)n_vars, pitch, b, e, o.chain_offset, o.seed,
                                epoch[variant], stream),
              "k_init_state");
}

void gr_sampler::run(uint64_t updates, bool counting) {
    use();
    if (dirty) rebuild();
    started = true;
    kernel_ms_last = 0;
    launches_last = 0;
    if (updates == 0) return;
    SweepParams p{};
    p.prio_shift = prio_shift(5);
    p.variants = d_variants.p;
    p.progs = d_progs.p; p.prog_off = d_prog_off.p; p.views = d_views.p; p.free_vars = d_free.p;
    p.cprogs = d_cprogs.p; p.cprog_off = d_cprog_off.p; p.cpt = d_cpt.p;
    p.pprogs = d_pprogs.p; p.pprog_off = d_pprog_off.p;
    p.vprogs = d_vprogs.p; p.vprog_off = d_vprog_off.p;
    p.state = d_state.p; p.counts = d_counts.p;
    const uint32_t act = active_chains();
    p.n_vars = (uint32_t)n_vars; p.H = (uint32_t)H; p.chain0 = 0; p.n_chains = act; p.pitch = pitch;
    p.chains_per_variant = cpv;
    p.chain_offset_lo = (uint32_t)o.chain_offset; p.chain_offset_hi = (uint32_t)(o.chain_offset >> 32);
    p.seed_lo = (uint32_t)o.seed; p.seed_hi = (uint32_t)(o.seed >> 32);
    p.counting = counting ? 1u : 0u;
    p.sequential = (o.flags & GR_FLAG_SEQUENTIAL) ? 1u : 0u;
    p.lds_hist_off = lds_hist_off; p.lds_tab_off = lds_tab_off;
    hip_check(hipEventRecord(ev0, stream), "hipEventRecord");
    if (ahead_inflight) { // whatever runs next may use the plan buffers: the plan computed ahead must have landed
        hip_check(hipStreamWaitEvent(stream, ev_plan, 0), "hipStreamWaitEvent");
        ahead_inflight = false;
    }
    auto generic = [&](uint64_t n_total) {
        uint64_t left = n_total;
        const bool use_pw4 = pw4;
        if (use_pw4 && !d_scan_vars.p) {
            // <= 256 MiB of scan variables per launch (2 bytes per step and 64-chain group), + 2 quads of slack for unaligned ends
            const uint64_t groups = o.n_chains / 64u;
            pw4_plan_steps = (uint32_t)std::max<uint64_t>(std::min<uint64_t>((256ull << 20) / (groups * 2ull), kMaxStepsPerLaunch) & ~3ull, 64);
            d_scan_vars.alloc((size_t)groups * (pw4_plan_steps + 8u));
        }
        while (left) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(left, use_pw4 ? pw4_plan_steps : kMaxStepsPerLaunch);
            p.step0_lo = (uint32_t)step; p.step0_hi = (uint32_t)(step >> 32);
            p.n_steps = n;
            p.log = cur_log; p.log_pitch = cur_log_pitch; p.log_rows = cur_log_rows; p.log_vals = cur_log_vals;
            p.log_off = cur_log_pc ? (uint32_t)(step - cur_log_step0) : (uint32_t)(step - cur_log_step0) * cur_log_vals;
            p.limit = cur_limit; p.win_off = (uint32_t)(step - cur_log_step0);
            // strict mode runs the direct arithmetic (bit-identical to the memoised tables); with strict_fast the LDS table
            // area holds thresholds, so the views are read from global memory
            // ... and a logged window (gr_advance_chain) of the generic kernel runs its direct back-end, the one built with the log code
            const bool plain = strict_fast || cur_log != nullptr;
            if (use_pw4) {
                ScanArgs a{};
                a.variants = d_variants.p; a.free_vars = d_free.p; a.n_groups = act / 64u; a.chains_per_variant = cpv;
                a.chain_offset_lo = p.chain_offset_lo; a.chain_offset_hi = p.chain_offset_hi;
                a.seed_lo = p.seed_lo; a.seed_hi = p.seed_hi; a.n_vars = (uint32_t)n_vars;
                a.step0_lo = p.step0_lo; a.step0_hi = p.step0_hi;
                a.sys_order = p.sys_order; // (systematic scan: set by the caller)
                const uint32_t n_quads = (uint32_t)(((step + n - 1) >> 2) - (step >> 2) + 1);
                p.scan_vars = d_scan_vars.p; p.scan_pitch = pw4_plan_steps / 4u + 2u;
                hip_check(launch_scan_vars(a, n_quads, p.scan_pitch, d_scan_vars.p, stream), "k_scan_vars launch");
                p.lds_hist_off = pw4_hist_off; p.prio_shift = prio_shift(2); // (a trip of k_sweep_pw4's loop is 4 heavy steps)
                hip_check(launch_sweep_pw4(p, act / 64u, pw4_lds, stream), "k_sweep_pw4 launch");
                p.lds_hist_off = lds_hist_off; p.prio_shift = prio_shift(5);
            } else
            hip_check(launch_sweep(p, max_card, cpt && !plain, tab_lds && !plain, pairwise, (o.flags & GR_FLAG_SCAN_PER_CHAIN) != 0, stg,
                                   (act + T - 1) / T, T, lds, stream), "k_sweep launch");
            step += n;
            left -= n;
            launches_last++;
        }
    };
    // the generic kernel on the fast path's geometry: same LDS layout (state stride = block size), <= 256 lanes
    auto generic_small = [&](uint64_t n_total) {
        if (T <= 256) { generic(n_total); return; }
        // blocks of 256 lanes: the layout offsets were sized for T lanes, which is larger -> still fits
        struct Restore { // the geometry is restored even when a launch throws
            uint32_t &t, &b, kt, kb;
            ~Restore() { t = kt; b = kb; }
        } restore{T, blocks, T, blocks};
        T = 256; blocks = (act + 255) / 256;
        generic(n_total);
    };
    auto sys_generic = [&](uint64_t n_total) {
        struct Restore { uint32_t &t, &b, kt, kb; ~Restore() { t = kt; b = kb; } } restore{T, blocks, T, blocks};
        if (T > 256) { T = 256; blocks = (act + 255) / 256; }
        uint64_t left = n_total;
        while (left) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(left, kMaxStepsPerLaunch);
            p.step0_lo = (uint32_t)step; p.step0_hi = (uint32_t)(step >> 32);
            p.n_steps = n;
            p.sys_order = d_sys_order.p;
            hip_check(launch_sweep(p, max_card, false, false, false, false, false, (act + T - 1) / T, T, lds, stream), "k_sweep (systematic) launch");
            step += n; left -= n; launches_last++;
        }
    };
    // k_scan_plan + k_sweep_dir4; sys: the systematic scan (the plan follows the sweep order, draws keyed by the position)
    auto run_dir4 = [&](const uint64_t n_updates, const bool sys) {
        auto ends = [&](uint64_t n) { if (sys) sys_generic(n); else generic_small(n); };
        const uint64_t updates = n_updates;
        // quad-unaligned head and tail go through the generic kernel (bit-identical streams), the rest in launches of
        // k_scan_plan + k_sweep_dir4 (16 steps per inner iteration)
        FastParams f{};
        f.prio_shift = prio_shift(5);
        f.sys_rank = sys ? d_sys_rank.p : nullptr;
        f.variants = d_variants.p; f.free_vars = d_free.p; f.cprogs = d_cprogs.p; f.cprog_off = d_cprog_off.p;
        f.plan = d_plan.p; f.state = d_state.p; f.counts = d_counts.p; f.plan_pitch = plan_steps;
        f.rec_stride = 32; f.recs_par = d_dhdr.p; f.dent = d_dent.p; f.views = d_views.p; f.dir_view_off = dir_view_off; f.dir_ent_off = dir_ent_off;
        if (!d_err.p) {
            d_err.alloc(1);
            hip_check(hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), stream), "hipMemset");
        }
        f.err = d_err.p;
        f.plan_slots = dir_slots;
        f.plan_nfree_max = 1;
        for (const auto &im : image) f.plan_nfree_max = std::max<uint32_t>(f.plan_nfree_max, (uint32_t)im.free_vars.size());
        f.n_vars = (uint32_t)n_vars; f.H = (uint32_t)H; f.n_chains = act; f.pitch = pitch; f.chains_per_variant = cpv;
        f.chain_offset_lo = p.chain_offset_lo; f.chain_offset_hi = p.chain_offset_hi;
        f.seed_lo = p.seed_lo; f.seed_hi = p.seed_hi;
        f.counting = p.counting; f.par_hist_off = dir_hist_off; f.par_stride = kParStateStride;
        uint64_t left = updates;
        const uint64_t head = std::min<uint64_t>(left, (16 - (step % 16)) % 16);
        if (head) { ends(head); left -= head; }
        while (left >= 16) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(left - left % 16, plan_steps);
            f.step0_lo = (uint32_t)step; f.step0_hi = (uint32_t)(step >> 32);
            f.n_steps = n;
            f.log = cur_log; f.log_pitch = cur_log_pitch; f.log_off = (uint32_t)(step - cur_log_step0); f.limit = cur_limit; f.log_vals = cur_log_vals;
            ahead_buf = -1; // (this plan goes into the first buffer)
            hip_check(launch_scan_plan(f, stream), "k_scan_plan launch");
            hip_check(launch_sweep_dir4(f, max_card, dir4_vlds, dir4_elds, dir4_chain, act / 64u, dir_lds, stream), "k_sweep_dir4 launch");
            step += n; left -= n; launches_last++;
        }
        if (left) ends(left);
    };
    // the memoised fast kernels (k_sweep_bin / k_sweep_par4 / k_sweep_wide / k_sweep_strict); sys: the systematic scan on the wide form
    auto run_fast = [&](const uint64_t n_updates, const bool sys) {
        auto ends = [&](uint64_t n) { if (sys) sys_generic(n); else generic_small(n); };
        const uint64_t updates = n_updates;
        // quad-unaligned head and tail go through the generic kernel (bit-identical streams)
        FastParams f{};
        f.prio_shift = prio_shift(5);
        f.sys_rank = sys ? d_sys_rank.p : nullptr;
        f.variants = d_variants.p; f.free_vars = d_free.p; f.cprogs = d_cprogs.p; f.cprog_off = d_cprog_off.p;
        f.cpt = d_cpt.p; f.recs = d_recs.p; f.plan = d_plan.p; f.state = d_state.p; f.counts = d_counts.p;
        f.plan_pitch = plan_steps;
        f.rec_stride = rec_stride;
        f.tab_global = par_tabg ? 1u : 0u;
        f.par_tab_words = par_tabg ? par_tab_words : 0u;
        f.par_stride = par_stride;
        // at most one wavefront per CU in the whole launch: give every wavefront a CU of its own (Grids_11 at 4 096 chains
        // 6.9e10 -> 7.2e10, Segmentation_11 6.3e10 -> 6.7e10; slower as soon as two wavefronts would have to share a CU's LDS)
        f.par_spread = act / 64u * 4u <= 256u ? 1u : 0u; // (set again below for the wide form)
        f.par_recmode = par_lean ? 2u : o.fast_form >= 4 && o.fast_form <= 6 ? o.fast_form - 4u : 99u;
        if (!d_err.p) {
            d_err.alloc(1);
            hip_check(hipMemsetAsync(d_err.p, 0, sizeof(uint32_t), stream), "hipMemset");
        }
        f.err = d_err.p;
        f.plan_slots = (uint32_t)std::max(fast_slots, 1);
        f.plan_nfree_max = 1;
        for (const auto &im : image) f.plan_nfree_max = std::max<uint32_t>(f.plan_nfree_max, (uint32_t)im.free_vars.size());
        f.n_vars = (uint32_t)n_vars; f.H = (uint32_t)H; f.n_chains = act; f.pitch = pitch; f.chains_per_variant = cpv;
        f.chain_offset_lo = p.chain_offset_lo; f.chain_offset_hi = p.chain_offset_hi;
        f.seed_lo = p.seed_lo; f.seed_hi = p.seed_hi;
        f.counting = p.counting; f.lds_hist_off = lds_hist_off; f.lds_tab_off = lds_tab_off; f.lds_plan_off = lds_plan_off;
        f.recs_par = d_recs_par.p; f.par_hist_off = par_hist_off; f.par_tab_off = par_tab_off; f.par_rec_off = par_rec_off;
        f.par_plan_off = par_plan_off; f.par_scr_off = par_scr_off;
        // Lanes per chain of the step-parallel form for THIS launch: 4 (k_sweep_par4), or 8 / 16 (k_sweep_wide) while the launch would
        // otherwise leave SIMDs short of wavefronts.  Wider rounds cost more re-gather passes per update (Grids: x1.13 with 8 lanes,
        // x1.3 with 16), so: 16 lanes up to 2 wavefronts per SIMD (1 024 SIMDs), 8 lanes up to 4 (auto_lpc); a logged window (gr_advance_chain)
        // runs the form gr_advance_chain laid its log out for.  All forms give the same results bit for bit, so launches of a run may differ.
        uint32_t lpc = 4;
        if (par && wide_ok && !no_wide) {
            if (cur_log || cur_limit) lpc = cur_log_rows > 4u ? cur_log_rows : 4u; // a logged window: gr_advance_chain chose (one log row per wavefront)
            else lpc = auto_lpc(act / 64u);
            if (sys && lpc == 4) lpc = 8; // (the systematic scan exists in the wide forms only: k_sweep_par4 is left as it is)
        }
        const bool use_bits = bits && par && lpc == 4 && !cur_log && !cur_limit; // (logged windows keep the byte layout's kernels)
        if (sys && lpc == 4 && !use_bits) throw Error(GR_E_STATE, "systematic scan on the 4-lane form (internal error)");
        if (use_bits) { f.rec_stride = 64; f.recs_par = d_recs_bits.p; }
        if (par) lpc_last = lpc;
        uint32_t first_n = 0;             // plan-ahead: this call's first launch
        FastParams plan_ahead{};          // ... and the plan kernel to launch behind the current sweep
        bool plan_ahead_pending = false;
        f.prio_shift = prio_shift(strict_fast || !par ? 8u : lpc > 4u ? 4u : 5u); // trips of the kernel's loop per priority rotation (wave_prio.h)
        const uint64_t gran = strict_fast ? 4 : par ? 4u * lpc : 4; // steps per inner iteration of the fast kernels
        uint64_t left = updates;
        const uint64_t head = std::min<uint64_t>(left, (gran - (step % gran)) % gran);
        if (head) { ends(head); left -= head; }
        while (left >= gran) {
            const uint32_t n = (uint32_t)std::min<uint64_t>(left - left % gran, plan_steps);
            f.step0_lo = (uint32_t)step; f.step0_hi = (uint32_t)(step >> 32);
            f.n_steps = n;
            f.log = cur_log; f.log_pitch = cur_log_pitch; f.log_off = (uint32_t)(step - cur_log_step0); f.limit = cur_limit;
            if (strict_fast) {
                // 4 waves per block share tables and records once there are more 64-chain groups than the chip holds
                // as single-wave blocks (LDS limits a CU to ~4 blocks)
                const uint32_t region = (((uint32_t)n_vars * kParStateStride) + 15u) & ~15u;
                const bool quad_blocks = act % 256u == 0 && act / 64u > 1024u && (o.n_variants == 1 || cpv % 256u == 0) &&
                                         par_lds + 3u * region <= kLdsBudget;
                f.strict_state_off = par_lds;
                {
                    uint32_t nfree_max = 0;
                    for (const auto &im : image) nfree_max = std::max<uint32_t>(nfree_max, (uint32_t)im.free_vars.size());
                    // the plan ring and Philox scratch areas behind the records are unused by this kernel
                    f.strict_rec_stride = (uint64_t)nfree_max * (rec_stride + 16u) <= par_lds - par_rec_off ? rec_stride + 16u : rec_stride;
                }
                hip_check(launch_sweep_strict(f, fast_slots, par_packed, quad_blocks ? act / 256u : act / 64u, quad_blocks ? 256u : 64u,
                                              quad_blocks ? par_lds + 3u * region : par_lds, stream), "k_sweep_strict launch");
                step += n; left -= n; launches_last++;
                continue;
            }
            if (par && lpc > 4) {
                // one wavefront per block while the whole launch has fewer wavefronts than the chip has SIMDs (every wavefront its own
                // CU's LDS pipe, as for the 4-lane form); else 64 x lpc lanes per block
                // ... and the CUs' LDS holds all of those blocks at once (each stages its own copy of the tables)
                f.par_spread = (act / 64u * lpc <= 1024u && (uint64_t)(act / 64u) * lpc <= 256ull * (kLdsBudget / par_rec_off)) ? 1u : 0u;
                ahead_buf = -1; // (this plan goes into the first buffer)
                hip_check(launch_scan_plan_wide(f, (int)lpc, stream), "k_scan_plan_wide launch");
                if (o.fast_form == 10 && fast_slots <= 4 && !par_tabg) { f.par_spread = 0; hip_check(launch_sweep_wide_dense(f, act / 64u, par_rec_off, stream), "k_sweep_wide (dense) launch"); }
                else
                hip_check(launch_sweep_wide(f, (int)lpc, fast_slots, act / 64u, par_rec_off, stream), "k_sweep_wide launch");
                step += n; left -= n; launches_last++;
                continue;
            }
            // k_sweep_par4 outside logged windows: plan-ahead (see the members' comment)
            static const bool ahead_on = [] { const char *e = getenv("GR_DEV_PLAN_AHEAD"); return !e || atoi(e) != 0; }(); // (measurements only: 0 = off)
            const bool ahead = ahead_on && par && !cur_log && !cur_limit && !sys;
            if (!ahead) {
                ahead_buf = -1;
                hip_check(launch_scan_plan(f, stream), "k_scan_plan launch");
            } else {
                if (!pstream) {
                    hip_check(hipStreamCreateWithFlags(&pstream, hipStreamNonBlocking), "hipStreamCreate");
                    hip_check(hipEventCreateWithFlags(&ev_plan, hipEventDisableTiming), "hipEventCreate");
                    hip_check(hipEventCreateWithFlags(&ev_pre, hipEventDisableTiming), "hipEventCreate");
                }
                d_plan_b.alloc((size_t)(o.n_chains / 64u) * plan_steps + 64);
                uint32_t *const bufs[2] = {d_plan.p, d_plan_b.p};
                int rb = 0; // the buffer this launch reads
                if (ahead_buf >= 0 && ahead_step0 == step && ahead_n >= n && ahead_act == act) {
                    rb = ahead_buf; // computed ahead (a plan of more steps starts with the plan of fewer)
                    if (ahead_inflight) { hip_check(hipStreamWaitEvent(stream, ev_plan, 0), "hipStreamWaitEvent"); ahead_inflight = false; }
                } else {
                    f.plan = bufs[0];
                    hip_check(launch_scan_plan(f, stream), "k_scan_plan launch");
                }
                f.plan = bufs[rb];
                // the next launch: the rest of this call, else a repeat of this call's first launch
                if (first_n == 0) first_n = n;
                const uint64_t rest = left - n;
                const uint32_t nn = rest >= gran ? (uint32_t)std::min<uint64_t>(rest - rest % gran, plan_steps) : first_n;
                hip_check(hipEventRecord(ev_pre, stream), "hipEventRecord"); // (every earlier reader of the other buffer is before this point)
                hip_check(hipStreamWaitEvent(pstream, ev_pre, 0), "hipStreamWaitEvent");
                FastParams fn = f;
                const uint64_t s1 = step + n;
                fn.plan = bufs[1 - rb]; fn.step0_lo = (uint32_t)s1; fn.step0_hi = (uint32_t)(s1 >> 32); fn.n_steps = nn;
                ahead_buf = 1 - rb; ahead_step0 = s1; ahead_n = nn; ahead_act = act;
                // (launched below, right after the sweep, so that the sweep's blocks are resident first)
                plan_ahead = fn; plan_ahead_pending = true;
            }
            if (par) {
                // (measurements only) GR_DEV_BLOCK_TIMES=<file>: start / end time (100 MHz), HW_ID and XCC_ID of every block of the last k_sweep_par4 launch
                const char *bt = getenv("GR_DEV_BLOCK_TIMES");
                if (bt) { d_dbg.alloc((size_t)(act / 64u) * 4u); f.dbg_times = d_dbg.p; }
                if (use_bits) hip_check(launch_sweep_bits(f, fast_slots, act / 64u, bits_lds, stream), "k_sweep_bits launch");
                else
                hip_check(launch_sweep_par4(f, fast_slots, par_packed, act / 64u, par_lds, stream), "k_sweep_par4 launch");
                if (plan_ahead_pending) {
                    plan_ahead_pending = false;
                    hip_check(launch_scan_plan(plan_ahead, pstream), "k_scan_plan (ahead) launch");
                    hip_check(hipEventRecord(ev_plan, pstream), "hipEventRecord");
                    ahead_inflight = true;
                }
                if (bt) {
                    std::vector<unsigned long long> t((size_t)(act / 64u) * 4u);
                    download(t.data(), d_dbg.p, t.size() * 8u);
                    if (FILE *fp = fopen(bt, "w")) {
                        for (size_t b = 0; b < t.size() / 4u; ++b) fprintf(fp, "%zu %llu %llu %llu %llu\n", b, t[4 * b], t[4 * b + 1], t[4 * b + 2], t[4 * b + 3]);
                        fclose(fp);
                    }
                }
            }
            else hip_check(launch_sweep_bin(f, fast_slots, act / T, T, lds, stream), "k_sweep_bin launch");
            step += n;
            left -= n;
            launches_last++;
        }
        if (left) ends(left);
    };
    if (cur_log_pc && !strict_fast) {
        generic_small(updates); // strict mode's logged windows on the direct path: the generic kernel's per-chain scan with the per-chain log
    } else if (sysscan) {
        // systematic scan: the chromatic kernel for the models it takes, the plan-based kernels on the sweep order for the other families,
        // the generic kernel's systematic mode for what is left (and their launch heads / tails) -- same order, same draws, same thresholds
        uint64_t left = updates;
        if (use_chroma(act)) {
            FastParams f{};
            f.prio_shift = prio_shift(3); // (a trip of the chromatic kernels' loop is 4 sweeps)
            f.variants = d_variants.p; f.free_vars = d_free.p; f.cprogs = d_cprogs.p; f.cprog_off = d_cprog_off.p; f.cpt = d_cpt.p;
            f.state = d_state.p; f.counts = d_counts.p;
            f.n_vars = (uint32_t)n_vars; f.H = (uint32_t)H; f.n_chains = act; f.pitch = pitch; f.chains_per_variant = cpv;
            f.chain_offset_lo = p.chain_offset_lo; f.chain_offset_hi = p.chain_offset_hi;
            f.seed_lo = p.seed_lo; f.seed_hi = p.seed_hi; f.counting = p.counting;
            f.chroma_slots = d_chroma_slots.p; f.recs_chroma = d_recs_chroma.p; f.sys_pos = d_sys_pos.p;
            f.chroma_row = chroma_row; f.chroma_tab_off = chroma_tab_off; f.chroma_wpb = chroma_wpb; f.chroma_hist_off = chroma_hist_off; f.chroma_rec_off = chroma_rec_off;
            // all live groups sweep the same number of variables: whole sweeps from a sweep boundary on the kernel's plain form (heads and
            // tails on the generic kernel); else its PART form, which takes any range of steps (the first and last sweep partial, per group)
            bool same_n = true;
            for (uint32_t g = 0; g < (o.n_variants > 1 ? n_active : 1u); ++g) same_n = same_n && image[g].free_vars.size() == image[0].free_vars.size();
            const uint64_t n0 = image[0].free_vars.size();
            f.chroma_part = same_n ? 0u : 1u;
            if (same_n) {
                const uint64_t head = std::min<uint64_t>(left, (n0 - step % n0) % n0);
                if (head) { sys_generic(head); left -= head; }
            }
            while (same_n ? left >= n0 : left > 0) {
                const uint32_t n = same_n ? (uint32_t)(std::min<uint64_t>(left / n0, std::max<uint64_t>(1, kMaxStepsPerLaunch / n0)) * n0)
                                          : (uint32_t)std::min<uint64_t>(left, kMaxStepsPerLaunch);
                f.step0_lo = (uint32_t)step; f.step0_hi = (uint32_t)(step >> 32);
                f.n_steps = n;
                hip_check(launch_sweep_chroma(f, (int)chroma_nslot, fast_slots, act / chroma_wpb, chroma_lds, stream), "k_sweep_chroma launch");
                step += n; left -= n; launches_last++;
            }
        }
        else if (pw4) { p.sys_order = d_sys_order.p; generic(left); left = 0; } // k_scan_vars + k_sweep_pw4<SYS>
        else if (dir4 && !fast) { run_dir4(left, true); left = 0; }
        else if (sys_wide()) { run_fast(left, true); left = 0; } // memoised binary models the chromatic kernel does not take: k_sweep_wide
        else if (sys_bits()) { run_fast(left, true); left = 0; } // ... or, where the wide form's LDS layout does not fit (state-dominated models), k_sweep_bits
        if (left) sys_generic(left);
    } else if (dir4 && !fast) {
        run_dir4(updates, false);
    } else if (!fast) {
        generic(updates);
    } else {
        run_fast(updates, false);
    }
    hip_check(hipEventRecord(ev1, stream), "hipEventRecord");
    uint32_t dev_err = 0;
    if (d_err.p) download(&dev_err, d_err.p, sizeof dev_err);
    else sync();
    if (dev_err) throw Error(GR_E_STATE, "device kernel refused to run: its LDS layout assumption does not hold (internal error)");
    float ms = 0;
    hip_check(hipEventElapsedTime(&ms, ev0, ev1), "hipEventElapsedTime");
    kernel_ms_last = ms;
    if (counting) {
        counted += updates;
        for (uint32_t g = 0; g < (o.n_variants > 1 ? n_active : 1u); ++g) counted_g[g] += updates;
    }
}

// A few uncounted updates on the groups [first, last) only (NewChain's burn-in for the chains added by Adapt), through
// the generic kernel restricted to those groups' chains, at the step positions just before the current one.
void gr_sampler::burn_groups(uint32_t first, uint32_t last, uint32_t updates) {
    if (last <= first) return;
    use();
    if (dirty) rebuild();
    SweepParams p{};
    p.prio_shift = prio_shift(5);
    p.variants = d_variants.p;
    p.progs = d_progs.p; p.prog_off = d_prog_off.p; p.views = d_views.p; p.free_vars = d_free.p;
    p.cprogs = d_cprogs.p; p.cprog_off = d_cprog_off.p; p.cpt = d_cpt.p;
    p.pprogs = d_pprogs.p; p.pprog_off = d_pprog_off.p;
    p.vprogs = d_vprogs.p; p.vprog_off = d_vprog_off.p;
    p.state = d_state.p; p.counts = d_counts.p;
    p.n_vars = (uint32_t)n_vars; p.H = (uint32_t)H; p.pitch = pitch; p.chains_per_variant = cpv;
    p.chain0 = first * cpv; p.n_chains = last * cpv;
    p.chain_offset_lo = (uint32_t)o.chain_offset; p.chain_offset_hi = (uint32_t)(o.chain_offset >> 32);
    p.seed_lo = (uint32_t)o.seed; p.seed_hi = (uint32_t)(o.seed >> 32);
    const uint64_t s0 = step >= updates ? step - updates : 0;
    p.step0_lo = (uint32_t)s0; p.step0_hi = (uint32_t)(s0 >> 32);
    p.n_steps = updates; p.counting = 0; p.sequential = 0;
    p.lds_hist_off = lds_hist_off; p.lds_tab_off = lds_tab_off;
    const uint32_t t = std::min(T, 256u);
    if (sysscan) {
        p.sys_order = d_sys_order.p;
        hip_check(launch_sweep(p, max_card, false, false, false, false, false, ((last - first) * cpv + t - 1) / t, t, lds, stream), "k_sweep launch (group burn-in)");
        sync();
        return;
    }
    hip_check(launch_sweep(p, max_card, cpt && !strict_fast, tab_lds && !strict_fast, pairwise, (o.flags & GR_FLAG_SCAN_PER_CHAIN) != 0, stg, ((last - first) * cpv + t - 1) / t, t, lds, stream),
              "k_sweep launch (group burn-in)");
    sync();
}

// ============================================================== C ABI: general
extern "C" {

GR_VISIBLE const char *gr_version(void) { return "grample-hip 0.1 (gfx950)"; }
GR_VISIBLE const char *gr_last_error(void) { return g_err.c_str(); }

GR_VISIBLE int gr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

GR_VISIBLE void gr_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const u32x4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

// ================================================================ model
GR_VISIBLE gr_model *gr_model_create(const gr_model_desc *d) {
    gr_model *out = nullptr;
    guarded([&] {
        if (!d || !d->card || !d->scope_off || !d->scope_var || !d->tab_off || !d->table) throw Error(GR_E_INVALID, "null model description");
        if (d->n_vars < 1) throw Error(GR_E_INVALID, "Invalid variable count: " + std::to_string(d->n_vars));
        if (d->n_funcs < 1) throw Error(GR_E_INVALID, "Invalid Clique count count: " + std::to_string(d->n_funcs));
        auto gm = std::make_unique<gr_model>();
        Model &m = gm->m;
        m.Type = "MARKOV";
        for (int i = 0; i < d->n_vars; ++i) {
            m.Vars.push_back(Variable::New(i, d->card[i]));
            if (d->fixed) m.Vars.back().FixedVal = d->fixed[i];
        }
        if (d->scope_off[0] < 0 || d->tab_off[0] < 0) throw Error(GR_E_INVALID, "negative offset in the model description");
        for (int f = 0; f < d->n_funcs; ++f) {
            if (d->scope_off[f + 1] <= d->scope_off[f] || d->tab_off[f + 1] < d->tab_off[f])
                throw Error(GR_E_INVALID, "scope_off / tab_off must be increasing (Clique " + std::to_string(f) + ")");
            std::vector<int> sc(d->scope_var + d->scope_off[f], d->scope_var + d->scope_off[f + 1]);
            for (int v : sc)
                if (v < 0 || v >= d->n_vars) throw Error(GR_E_INVALID, "Invalid var idx " + std::to_string(v) + " for Clique " + std::to_string(f));
            Function fn = m.NewFunction(f, sc);
            if ((int)fn.Table.size() != d->tab_off[f + 1] - d->tab_off[f]) throw Error(GR_E_INVALID, "Read table size != previous Clique size on function " + fn.Name);
            std::copy(d->table + d->tab_off[f], d->table + d->tab_off[f + 1], fn.Table.begin());
            m.Funcs.push_back(std::move(fn));
        }
        m.Check();
        out = gm.release();
    });
    return out;
}

GR_VISIBLE gr_model *gr_model_from_uai(const char *text, size_t len) {
    gr_model *out = nullptr;
    guarded([&] {
        auto gm = std::make_unique<gr_model>();
        gm->m = NewModelFromBuffer(std::string(text ? text : "", text ? len : 0));
        out = gm.release();
    });
    return out;
}

GR_VISIBLE gr_model *gr_model_from_file(const char *path, int use_evidence) {
    gr_model *out = nullptr;
    guarded([&] {
        auto gm = std::make_unique<gr_model>();
        gm->m = NewModelFromFile(path ? path : "", use_evidence != 0);
        out = gm.release();
    });
    return out;
}

GR_VISIBLE int gr_model_apply_evidence(gr_model *m, const char *text, size_t len) {
    return guarded([&] {
        if (!m) throw Error(GR_E_INVALID, "null model");
        UAIReader::ApplyEvidence(std::string(text ? text : "", text ? len : 0), m->m);
    });
}

GR_VISIBLE gr_model *gr_model_clone(const gr_model *m) {
    if (!m) { fail(GR_E_INVALID, "null model"); return nullptr; }
    gr_model *out = nullptr;
    guarded([&] { out = new gr_model{m->m}; });
    return out;
}
GR_VISIBLE void gr_model_destroy(gr_model *m) { delete m; }
GR_VISIBLE int gr_model_n_vars(const gr_model *m) { return m ? (int)m->m.Vars.size() : fail(GR_E_INVALID, "null model"); }
GR_VISIBLE int gr_model_n_funcs(const gr_model *m) { return m ? (int)m->m.Funcs.size() : fail(GR_E_INVALID, "null model"); }
GR_VISIBLE int gr_model_total_card(const gr_model *m) { return m ? m->m.TotalCard() : fail(GR_E_INVALID, "null model"); }
GR_VISIBLE int gr_model_cards(const gr_model *m, int32_t *out) {
    if (!m || !out) return fail(GR_E_INVALID, "null argument");
    for (size_t i = 0; i < m->m.Vars.size(); ++i) out[i] = m->m.Vars[i].Card;
    return GR_OK;
}
GR_VISIBLE int gr_model_fixed(const gr_model *m, int32_t *out) {
    if (!m || !out) return fail(GR_E_INVALID, "null argument");
    for (size_t i = 0; i < m->m.Vars.size(); ++i) out[i] = m->m.Vars[i].FixedVal;
    return GR_OK;
}
static const Function *get_func(const gr_model *m, int f) {
    if (!m || f < 0 || f >= (int)m->m.Funcs.size()) { fail(GR_E_INVALID, "bad function index"); return nullptr; }
    return &m->m.Funcs[f];
}
GR_VISIBLE int gr_model_func_arity(const gr_model *m, int f) { auto *fn = get_func(m, f); return fn ? (int)fn->Vars.size() : GR_E_INVALID; }
GR_VISIBLE int gr_model_func_tabsize(const gr_model *m, int f) { auto *fn = get_func(m, f); return fn ? (int)fn->Table.size() : GR_E_INVALID; }
GR_VISIBLE int gr_model_func_scope(const gr_model *m, int f, int32_t *out) {
    auto *fn = get_func(m, f);
    if (!fn || !out) return GR_E_INVALID;
    std::copy(fn->Vars.begin(), fn->Vars.end(), out);
    return GR_OK;
}
GR_VISIBLE int gr_model_func_table(const gr_model *m, int f, double *out) {
    auto *fn = get_func(m, f);
    if (!fn || !out) return GR_E_INVALID;
    std::copy(fn->Table.begin(), fn->Table.end(), out);
    return GR_OK;
}
GR_VISIBLE int gr_model_func_eval(const gr_model *m, int f, const int32_t *values, int n_values, double *out) {
    if (out) *out = NAN;
    auto *fn = get_func(m, f);
    if (!fn || !out || (n_values > 0 && !values)) return GR_E_INVALID;
    return guarded([&] { *out = m->m.Eval(*fn, values, n_values); });
}

GR_VISIBLE int gr_model_systematic_order(const gr_model *m, int32_t *out_order, int32_t *out_n, int32_t *out_color) {
    return guarded([&] {
        if (!m || !out_order || !out_n) throw Error(GR_E_INVALID, "null argument");
        Model lm = m->m;
        for (auto &f : lm.Funcs) Model::UseLogSpace(f);
        const VariantImage im = CompileVariant(lm, CompileLimits{});
        *out_n = (int32_t)im.sys_order.size();
        std::copy(im.sys_order.begin(), im.sys_order.end(), out_order);
        if (out_color) {
            std::fill(out_color, out_color + lm.Vars.size(), -1);
            for (size_t c = 0; c + 1 < im.color_off.size(); ++c)
                for (uint32_t i = im.color_off[c]; i < im.color_off[c + 1]; ++i) out_color[im.sys_order[i]] = (int32_t)c;
        }
    });
}

// ============================================================== solution
GR_VISIBLE gr_solution *gr_solution_from_text(const char *text, size_t len) {
    gr_solution *out = nullptr;
    guarded([&] {
        auto s = std::make_unique<gr_solution>();
        s->vars = UAIReader::ReadMargSolution(std::string(text ? text : "", text ? len : 0));
        out = s.release();
    });
    return out;
}
GR_VISIBLE gr_solution *gr_solution_from_file(const char *path) {
    gr_solution *out = nullptr;
    guarded([&] {
        auto s = std::make_unique<gr_solution>();
        s->vars = NewSolutionFromFile(path ? path : "");
        out = s.release();
    });
    return out;
}
GR_VISIBLE void gr_solution_destroy(gr_solution *s) { delete s; }
GR_VISIBLE int gr_solution_n_vars(const gr_solution *s) { return s ? (int)s->vars.size() : fail(GR_E_INVALID, "null solution"); }
GR_VISIBLE int gr_solution_total_card(const gr_solution *s) {
    if (!s) return fail(GR_E_INVALID, "null solution");
    int t = 0;
    for (auto &v : s->vars) t += v.Card;
    return t;
}
GR_VISIBLE int gr_solution_cards(const gr_solution *s, int32_t *out) {
    if (!s || !out) return fail(GR_E_INVALID, "null argument");
    for (size_t i = 0; i < s->vars.size(); ++i) out[i] = s->vars[i].Card;
    return GR_OK;
}
GR_VISIBLE int gr_solution_marginals(const gr_solution *s, double *out) {
    if (!s || !out) return fail(GR_E_INVALID, "null argument");
    for (auto &v : s->vars) out = std::copy(v.Marginal.begin(), v.Marginal.end(), out);
    return GR_OK;
}

GR_VISIBLE int gr_error_suite(int32_t n_vars, const int32_t *card, const int32_t *fixed1, const int32_t *fixed2, const double *m1,
                              const double *m2, double *out) {
    return guarded([&] {
        if (n_vars < 1 || !card || !m1 || !m2 || !out) throw Error(GR_E_INVALID, "null argument");
        const ErrorSuite es = NewErrorSuite(n_vars, card, fixed1, fixed2, m1, m2);
        out[0] = es.MeanMeanAbsError; out[1] = es.MeanMaxAbsError; out[2] = es.MeanHellinger; out[3] = es.MeanJSDiverge;
        out[4] = es.MaxMeanAbsError; out[5] = es.MaxMaxAbsError; out[6] = es.MaxHellinger; out[7] = es.MaxJSDiverge;
    });
}

// =============================================================== sampler
GR_VISIBLE gr_sampler *gr_sampler_create(const gr_model *m, const gr_sampler_opts *opts) {
    gr_sampler *out = nullptr;
    guarded([&] {
        if (!m) throw Error(GR_E_INVALID, "No model supplied");
        if (!opts || opts->struct_size != sizeof(gr_sampler_opts)) throw Error(GR_E_INVALID, "bad gr_sampler_opts (struct_size)");
        if (opts->n_chains < 1) throw Error(GR_E_INVALID, "n_chains must be >= 1");
        if (opts->n_chains > (1u << 30)) throw Error(GR_E_INVALID, "n_chains must be <= 2^30 per device");
        if (opts->n_variants < 1 || opts->n_chains % opts->n_variants) throw Error(GR_E_INVALID, "n_chains must be a multiple of n_variants");
        if (opts->n_variants > 1 && (opts->n_chains / opts->n_variants) % 64) throw Error(GR_E_INVALID, "chains per variant must be a multiple of 64");
        if (opts->chain_offset % 64) throw Error(GR_E_INVALID, "chain_offset must be a multiple of 64");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
            throw Error(GR_E_NO_DEVICE, "no HIP device: the Gibbs path runs on the GPU only (there is no CPU fallback)");
        if (opts->device < 0 || opts->device >= ndev) throw Error(GR_E_NO_DEVICE, "device ordinal out of range");
        auto s = std::make_unique<gr_sampler>();
        s->o = *opts;
        if (s->o.fast_form == 3) { s->o.fast_form = 0; s->no_wide = true; }
        s->use();
        hip_check(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking), "hipStreamCreate");
        hip_check(hipEventCreate(&s->ev0), "hipEventCreate");
        hip_check(hipEventCreate(&s->ev1), "hipEventCreate");
        Model base = m->m; // Model.Clone (root.go:385)
        base.Check();
        s->base_linear = base;
        s->n_active = opts->n_variants;
        s->counted_g.assign(opts->n_variants, 0);
        s->n_vars = (int)base.Vars.size();
        s->H = base.TotalCard();
        for (auto &v : base.Vars) {
            s->card.push_back(v.Card);
            s->fixed.push_back(v.FixedVal);
            s->max_card = std::max(s->max_card, v.Card);
        }
        if (s->max_card > kMaxSupportedCard) throw Error(GR_E_UNSUPPORTED, "Card > 64 is not supported by this build");
        for (auto &f : base.Funcs) Model::UseLogSpace(f); // gibbs-simple.go:72-76
        const VariantImage im = CompileVariant(base, CompileLimits{});
        s->vmodel.assign(opts->n_variants, base);
        s->image.assign(opts->n_variants, im);
        s->epoch.assign(opts->n_variants, 0);
        s->pitch = (opts->n_chains + 255u) & ~255u;
        s->d_card.upload(s->card, s->stream, s->stage);
        s->d_fixed.upload(s->fixed, s->stream, s->stage);
        s->d_state.alloc((size_t)s->n_vars * s->pitch);
        hip_check(hipMemsetAsync(s->d_state.p, 0, (size_t)s->n_vars * s->pitch, s->stream), "hipMemset");
        s->d_counts.alloc((size_t)opts->n_variants * s->H);
        hip_check(hipMemsetAsync(s->d_counts.p, 0, (size_t)opts->n_variants * s->H * sizeof(unsigned long long), s->stream), "hipMemset");
        s->configure(); // validates the LDS layout early
        for (uint32_t g = 0; g < opts->n_variants; ++g) s->init_state(g);
        s->sync();
        out = s.release();
    });
    return out;
}

GR_VISIBLE void gr_sampler_destroy(gr_sampler *s) {
    if (!s) return;
    (void)hipSetDevice(s->o.device);
    delete s;
}

GR_VISIBLE int gr_blanket_size(gr_sampler *s, uint32_t variant, int32_t v) {
    if (!s || variant >= s->o.n_variants || v < 0 || v >= s->n_vars) return fail(GR_E_INVALID, "bad argument");
    int n = 0;
    const int rc = guarded([&] { n = (int)VarNeighbors(s->vmodel[variant])[v].size(); });
    return rc == GR_OK ? n : rc;
}
GR_VISIBLE int gr_function_count(gr_sampler *s, uint32_t variant, int32_t v) {
    if (!s || variant >= s->o.n_variants || v < 0 || v >= s->n_vars) return fail(GR_E_INVALID, "bad argument");
    int n = 0;
    const int rc = guarded([&] { n = (int)VarFuncs(s->vmodel[variant])[v].size(); });
    return rc == GR_OK ? n : rc;
}

constexpr int kNeighborVarMax = 12; // gibbs-collapsed.go:93

GR_VISIBLE int gr_collapse(gr_sampler *s, uint32_t variant, int32_t var_idx, int32_t *out_var, double *out_marginal) {
    return guarded([&] {
        if (!s) throw Error(GR_E_INVALID, "null sampler");
        if (variant >= s->o.n_variants) throw Error(GR_E_INVALID, "variant out of range");
        if (s->started && variant < s->n_active) throw Error(GR_E_STATE, "gr_collapse must precede burn-in / advance (or target a group that is not active yet)");
        s->use();
        // All surgery happens on a copy: Check(), CompileVariant() and configure() can all refuse the result (last free
        // variable collapsed, LDS layout no longer fits, ...), and a refused Collapse must leave the group exactly as it was.
        Model pgm = s->vmodel[variant];
        const int N = s->n_vars;
        const auto nbrs = VarNeighbors(pgm);
        if (var_idx < 0) { // gibbs-collapsed.go:102-120: up to N tries among free, un-collapsed variables
            std::vector<int> cand;
            for (int i = 0; i < N; ++i)
                if (!pgm.Vars[i].Collapsed && pgm.Vars[i].FixedVal < 0) cand.push_back(i);
            if (cand.empty()) throw Error(GR_E_COLLAPSE, "Failure selecting random variable to collapse: No Variables to select");
            const uint32_t gvar = s->o.variant_offset + variant;
            for (int t = 0; t < N; ++t) {
                int pick = cand[0];
                if (cand.size() > 1) {
                    const u32x4 r = philox4x32_10((uint32_t)t, s->epoch[variant], gvar, DOM_COLLAPSE, (uint32_t)s->o.seed, (uint32_t)(s->o.seed >> 32));
                    pick = cand[mul_hi_u32(r.x, (uint32_t)cand.size())];
                }
                if ((int)nbrs[pick].size() <= kNeighborVarMax) { var_idx = pick; break; }
            }
        }
        if (var_idx < 0) throw Error(GR_E_COLLAPSE, "Failed to randomly select a variable to collapse");
        if (var_idx >= N) throw Error(GR_E_COLLAPSE, "Invalid variable index: max is " + std::to_string(N - 1));
        Variable &cv = pgm.Vars[var_idx];
        if (cv.FixedVal >= 0) throw Error(GR_E_COLLAPSE, "Can not collapse Fixed Val variable " + std::to_string(cv.ID) + ":" + cv.Name);
        if (cv.Collapsed) throw Error(GR_E_COLLAPSE, "Already collapsed variable " + std::to_string(cv.ID) + ":" + cv.Name);
        // blanket in ascending id (the reference iterates a Go map: any order is legal there)
        std::vector<int> newvars;
        bool self = false;
        for (int u : nbrs[var_idx]) {
            if (u == var_idx) self = true; else newvars.push_back(u);
        }
        if (!self) throw Error(GR_E_COLLAPSE, "Collapsing variable not in its own blanket");
        if (newvars.empty()) throw Error(GR_E_COLLAPSE, "New function would have 0 variables");
        if (newvars.size() > 32) throw Error(GR_E_UNSUPPORTED, "blanket larger than 32 variables");
        Function post;
        try {
            post = pgm.NewFunction((int)pgm.Funcs.size(), newvars); // fails above maxTabSize (function.go:77-79)
        } catch (const Error &e) {
            throw Error(GR_E_COLLAPSE, e.what());
        }
        post.Name = "COLLAPSE-" + cv.Name;
        const auto vf = VarFuncs(pgm);
        // ---- K2 on the device
        std::vector<uint32_t> new_card, fac_off{0}, fac_stride, fac_tab;
        std::vector<int32_t> new_fixed, fac_pos;
        std::vector<double> tables;
        for (int u : newvars) { new_card.push_back((uint32_t)pgm.Vars[u].Card); new_fixed.push_back(pgm.Vars[u].FixedVal); }
        for (int fi : vf[var_idx]) {
            const Function &f = pgm.Funcs[fi];
            if (!f.IsLog) throw Error(GR_E_COLLAPSE, "Function " + f.Name + " is not set up for Log Space");
            fac_tab.push_back((uint32_t)tables.size());
            tables.insert(tables.end(), f.Table.begin(), f.Table.end());
            std::vector<uint32_t> stride(f.Vars.size());
            uint32_t d = 1;
            for (int j = (int)f.Vars.size() - 1; j >= 0; --j) { stride[j] = d; d *= (uint32_t)pgm.Vars[f.Vars[j]].Card; }
            for (size_t j = 0; j < f.Vars.size(); ++j) {
                int pos = -1;
                if (f.Vars[j] != var_idx) pos = (int)(std::find(newvars.begin(), newvars.end(), f.Vars[j]) - newvars.begin());
                fac_pos.push_back(pos);
                fac_stride.push_back(stride[j]);
            }
            fac_off.push_back((uint32_t)fac_pos.size());
        }
        const uint32_t n_entries = (uint32_t)post.Table.size();
        const uint32_t nblk = (n_entries + 255) / 256;
        DevBuf<uint32_t> &d_new_card = s->cs_new_card, &d_fac_off = s->cs_fac_off, &d_fac_stride = s->cs_fac_stride, &d_fac_tab = s->cs_fac_tab;
        DevBuf<int32_t> &d_new_fixed = s->cs_new_fixed, &d_fac_pos = s->cs_fac_pos;
        DevBuf<double> &d_tables = s->cs_tables, &d_out = s->cs_out, &d_part = s->cs_part;
        d_new_card.upload(new_card, s->stream, s->stage); d_new_fixed.upload(new_fixed, s->stream, s->stage);
        d_fac_off.upload(fac_off, s->stream, s->stage); d_fac_pos.upload(fac_pos, s->stream, s->stage);
        d_fac_stride.upload(fac_stride, s->stream, s->stage); d_fac_tab.upload(fac_tab, s->stream, s->stage);
        d_tables.upload(tables, s->stream, s->stage);
        d_out.alloc(n_entries);
        d_part.alloc((size_t)nblk * cv.Card);
        CollapseArgs a{};
        a.n_entries = n_entries; a.card_v = (uint32_t)cv.Card; a.n_new = (uint32_t)newvars.size(); a.n_fac = (uint32_t)vf[var_idx].size();
        a.new_card = d_new_card.p; a.new_fixed = d_new_fixed.p; a.fac_off = d_fac_off.p; a.fac_pos = d_fac_pos.p;
        a.fac_stride = d_fac_stride.p; a.fac_tab = d_fac_tab.p; a.tables = d_tables.p; a.out_table = d_out.p; a.out_partial = d_part.p;
        hip_check(launch_collapse(a, nblk, s->stream), "k_collapse");
        std::vector<double> part((size_t)nblk * cv.Card);
        s->download(post.Table.data(), d_out.p, n_entries * sizeof(double));
        s->download(part.data(), d_part.p, part.size() * sizeof(double));
        post.IsLog = true; // the kernel already applied UseLogSpace (gibbs-collapsed.go:269)
        Variable marg = cv;
        for (int k = 0; k < cv.Card; ++k) {
            double acc = 1e-12; // gibbs-collapsed.go:138-140
            for (uint32_t b = 0; b < nblk; ++b) acc += part[(size_t)b * cv.Card + k];
            marg.Marginal[k] = acc;
        }
        marg.NormMarginal();
        // ---- model surgery (gibbs-collapsed.go:275-313)
        std::set<int> del(vf[var_idx].begin(), vf[var_idx].end());
        std::vector<Function> kept;
        for (size_t i = 0; i < pgm.Funcs.size(); ++i)
            if (!del.count((int)i)) kept.push_back(std::move(pgm.Funcs[i]));
        kept.push_back(std::move(post));
        pgm.Funcs = std::move(kept);
        cv.Collapsed = true;
        cv.Marginal = marg.Marginal;
        pgm.Check();
        if (!VarFuncs(pgm)[var_idx].empty()) throw Error(GR_E_COLLAPSE, "Var is collapsed but has a blanket");
        VariantImage im = CompileVariant(pgm, CompileLimits{}); // refuses a model with nothing left to select (sampler.go:158-160)
        std::swap(s->image[variant], im);
        try {
            s->configure(); // the sampler-wide kernel / LDS choice with the new image in place
        } catch (...) {
            std::swap(s->image[variant], im); // refused: put the old image back, and the configuration that goes with it
            s->configure();
            throw;
        }
        // ---- commit
        const std::vector<double> marginal = cv.Marginal;
        s->vmodel[variant] = std::move(pgm);
        s->dirty = true;
        s->epoch[variant]++; // base.FunctionsChanged re-draws the start state (gibbs-simple.go:132-142)
        s->init_state(variant);
        s->sync();
        if (out_var) *out_var = var_idx;
        if (out_marginal) std::copy(marginal.begin(), marginal.end(), out_marginal);
    });
}

GR_VISIBLE int gr_burnin(gr_sampler *s, uint64_t updates_per_chain) {
    return guarded([&] { if (!s) throw Error(GR_E_INVALID, "null sampler"); s->run(updates_per_chain, false); });
}
GR_VISIBLE int gr_advance(gr_sampler *s, uint64_t updates_per_chain) {
    return guarded([&] { if (!s) throw Error(GR_E_INVALID, "null sampler"); s->run(updates_per_chain, true); });
}
GR_VISIBLE int gr_total_samples(gr_sampler *s, uint64_t *out) {
    if (!s || !out) return fail(GR_E_INVALID, "null argument");
    if (s->o.n_variants > 1) {
        uint64_t t = 0;
        for (uint32_t g = 0; g < s->n_active; ++g) t += s->counted_g[g] * (uint64_t)(s->o.n_chains / s->o.n_variants);
        *out = t - s->skipped_updates;
    } else {
        *out = s->counted * (uint64_t)s->o.n_chains - s->skipped_updates;
    }
    return GR_OK;
}

GR_VISIBLE int gr_counts(gr_sampler *s, uint64_t *counts, uint64_t *n_updates) {
    return guarded([&] {
        if (!s) throw Error(GR_E_INVALID, "null sampler");
        s->use();
        if (counts) {
            s->download(counts, s->d_counts.p, (size_t)s->o.n_variants * s->H * sizeof(uint64_t));
        }
        if (n_updates && gr_total_samples(s, n_updates)) throw Error(GR_E_INVALID, g_err);
    });
}
GR_VISIBLE int gr_counts_device(gr_sampler *s, void **dptr, size_t *n_elems) {
    if (!s || !dptr) return fail(GR_E_INVALID, "null argument");
    *dptr = s->d_counts.p;
    if (n_elems) *n_elems = (size_t)s->o.n_variants * s->H;
    return GR_OK;
}
GR_VISIBLE int gr_counts_upload(gr_sampler *s, const uint64_t *counts) {
    return guarded([&] {
        if (!s || !counts) throw Error(GR_E_INVALID, "null argument");
        s->use();
        const size_t bytes = (size_t)s->o.n_variants * s->H * sizeof(uint64_t);
        void *tmp = s->stage.take(bytes);
        memcpy(tmp, counts, bytes);
        hip_check(hipMemcpyAsync(s->d_counts.p, tmp, bytes, hipMemcpyHostToDevice, s->stream), "H2D");
        s->sync();
    });
}

GR_VISIBLE int gr_get_position(gr_sampler *s, gr_position *pos, uint64_t *counted_per_group, uint32_t *epoch_per_group) {
    return guarded([&] {
        if (!s || !pos || pos->struct_size != sizeof(gr_position)) throw Error(GR_E_INVALID, "bad gr_position (struct_size)");
        pos->n_active = s->n_active;
        pos->step = s->step;
        pos->counted = s->counted;
        pos->skipped_updates = s->skipped_updates;
        if (counted_per_group) std::copy(s->counted_g.begin(), s->counted_g.end(), counted_per_group);
        if (epoch_per_group) std::copy(s->epoch.begin(), s->epoch.end(), epoch_per_group);
    });
}
GR_VISIBLE int gr_set_position(gr_sampler *s, const gr_position *pos, const uint64_t *counted_per_group, const uint32_t *epoch_per_group) {
    return guarded([&] {
        if (!s || !pos || pos->struct_size != sizeof(gr_position)) throw Error(GR_E_INVALID, "bad gr_position (struct_size)");
        if (pos->n_active != s->n_active)
            throw Error(GR_E_STATE, "gr_set_position: the sampler runs " + std::to_string(s->n_active) + " chain groups, the checkpoint " +
                                        std::to_string(pos->n_active) + " (replay the collapses / activations first)");
        s->step = pos->step;
        s->counted = pos->counted;
        s->skipped_updates = pos->skipped_updates;
        if (!counted_per_group && s->o.n_variants > 1)
            throw Error(GR_E_INVALID, "gr_set_position: counted_per_group is required for a sampler with several chain groups");
        if (counted_per_group) s->counted_g.assign(counted_per_group, counted_per_group + s->o.n_variants);
        else s->counted_g.assign(1, pos->counted);
        if (epoch_per_group) s->epoch.assign(epoch_per_group, epoch_per_group + s->o.n_variants);
        s->started = s->started || pos->step > 0;
    });
}
GR_VISIBLE int gr_state_download(gr_sampler *s, uint8_t *states) {
    return guarded([&] {
        if (!s || !states) throw Error(GR_E_INVALID, "null argument");
        s->use();
        const size_t bytes = (size_t)s->o.n_chains * s->n_vars;
        void *tmp = s->stage.take(bytes);
        hip_check(hipMemcpy2DAsync(tmp, s->o.n_chains, s->d_state.p, s->pitch, s->o.n_chains, s->n_vars, hipMemcpyDeviceToHost, s->stream), "D2H state");
        hip_check(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
        memcpy(states, tmp, bytes); // (before the arena is reset: this one may be above its watermark)
        s->stage.reset();
    });
}
GR_VISIBLE int gr_state_upload(gr_sampler *s, const uint8_t *states) {
    return guarded([&] {
        if (!s || !states) throw Error(GR_E_INVALID, "null argument");
        for (int v = 0; v < s->n_vars; ++v) {
            const uint8_t *row = states + (size_t)v * s->o.n_chains;
            const int C = s->card[v], f = s->fixed[v];
            for (uint32_t c = 0; c < s->o.n_chains; ++c)
                if (row[c] >= C || (f >= 0 && row[c] != f)) throw Error(GR_E_INVALID, "state value out of range (or contradicting evidence) for variable " + std::to_string(v));
        }
        s->use();
        const size_t bytes = (size_t)s->o.n_chains * s->n_vars;
        void *tmp = s->stage.take(bytes);
        memcpy(tmp, states, bytes);
        hip_check(hipMemcpy2DAsync(s->d_state.p, s->pitch, tmp, s->o.n_chains, s->o.n_chains, s->n_vars, hipMemcpyHostToDevice, s->stream), "H2D state");
        s->sync();
    });
}

GR_VISIBLE int gr_merge_local(gr_sampler *s, double *out_sum, int32_t *out_coll_variant, double *out_coll_marginal) {
    return guarded([&] {
        if (!s || !out_sum || !out_coll_variant || !out_coll_marginal) throw Error(GR_E_INVALID, "null argument");
        std::vector<uint64_t> cnt((size_t)s->o.n_variants * s->H);
        if (gr_counts(s, cnt.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
        std::fill(out_sum, out_sum + s->H, 0.0);
        std::fill(out_coll_marginal, out_coll_marginal + s->H, 0.0);
        const double cpv = (double)(s->o.n_chains / s->o.n_variants);
        int off = 0;
        for (int v = 0; v < s->n_vars; ++v) {
            out_coll_variant[v] = -1;
            const int C = s->card[v];
            for (uint32_t g = 0; g < (s->o.n_variants > 1 ? s->n_active : 1u); ++g) {
                const Variable &var = s->vmodel[g].Vars[v];
                if (var.Collapsed) {
                    if (out_coll_variant[v] < 0) {
                        out_coll_variant[v] = (int32_t)(s->o.variant_offset + g);
                        std::copy(var.Marginal.begin(), var.Marginal.end(), out_coll_marginal + off);
                    }
                    continue; // a collapsed variable is never sampled: it contributes no counts
                }
                // every chain starts from the uniform marginal 1/Card (variable.go:45) and adds its counts (chain.go:230-236)
                for (int k = 0; k < C; ++k) out_sum[off + k] += cpv / (double)C + (double)cnt[(size_t)g * s->H + off + k];
            }
            off += C;
        }
    });
}

GR_VISIBLE int gr_merge(gr_sampler *const *samplers, int n, double *out_marginal, int32_t *out_collapsed) {
    return guarded([&] {
        if (!samplers || n < 1 || !out_marginal) throw Error(GR_E_INVALID, "Can not merge 0 chains");
        if (!samplers[0]) throw Error(GR_E_INVALID, "null sampler");
        const int H = samplers[0]->H, N = samplers[0]->n_vars;
        std::vector<double> sum(H), cm(H), tot(H, 0.0), best_m(H, 0.0);
        std::vector<int32_t> cvv(N), best(N, -1);
        for (int i = 0; i < n; ++i) {
            gr_sampler *s = samplers[i];
            if (!s || s->H != H || s->n_vars != N) throw Error(GR_E_INVALID, "Cannot merge samplers of different models");
            if (gr_merge_local(s, sum.data(), cvv.data(), cm.data())) throw Error(GR_E_NO_DEVICE, g_err);
            for (int k = 0; k < H; ++k) tot[k] += sum[k];
            int off = 0;
            for (int v = 0; v < N; ++v) {
                if (cvv[v] >= 0 && (best[v] < 0 || cvv[v] < best[v])) {
                    best[v] = cvv[v];
                    std::copy(cm.begin() + off, cm.begin() + off + s->card[v], best_m.begin() + off);
                }
                off += s->card[v];
            }
        }
        int off = 0;
        for (int v = 0; v < N; ++v) {
            const int C = samplers[0]->card[v];
            // "collapsed wins" (chain.go:113-129): first collapsed chain's exact marginal, no summation
            for (int k = 0; k < C; ++k) out_marginal[off + k] = best[v] >= 0 ? best_m[off + k] : tot[off + k];
            if (out_collapsed) out_collapsed[v] = best[v] >= 0 ? 1 : 0;
            off += C;
        }
    });
}

GR_VISIBLE int gr_allreduce_counts(gr_sampler *const *samplers, int n, uint64_t *out) {
    return guarded([&] {
        if (!samplers || n < 1 || !samplers[0]) throw Error(GR_E_INVALID, "no samplers");
        const size_t cnt = (size_t)samplers[0]->o.n_variants * samplers[0]->H;
        std::vector<int> devs(n);
        for (int i = 0; i < n; ++i) {
            if (!samplers[i] || (size_t)samplers[i]->o.n_variants * samplers[i]->H != cnt) throw Error(GR_E_INVALID, "samplers differ in shape");
            devs[i] = samplers[i]->o.device;
        }
        // One sampler needs no exchange: its counts are the total.  GR_ALLREDUCE_FORCE_RCCL=1 (read once) takes the RCCL branch
        // all the same -- ncclCommInitAll over one device is legal -- so that the collective of the path can be executed and
        // checked on a one-GPU box (tests/test_gpu_dist.py).
        static const bool force_rccl = [] { const char *e = getenv("GR_ALLREDUCE_FORCE_RCCL"); return e && e[0] == '1'; }();
        if (n > 1 || force_rccl) {
            // RCCL is bound lazily: a host process that brings its own RCCL (e.g. PyTorch) never sees a second copy
            static void *h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) throw Error(GR_E_NO_DEVICE, std::string("cannot load librccl.so: ") + dlerror());
            auto pInitAll = reinterpret_cast<ncclResult_t (*)(ncclComm_t *, int, const int *)>(dlsym(h, "ncclCommInitAll"));
            auto pAllReduce = reinterpret_cast<ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>(dlsym(h, "ncclAllReduce"));
            auto pGroupStart = reinterpret_cast<ncclResult_t (*)()>(dlsym(h, "ncclGroupStart"));
            auto pGroupEnd = reinterpret_cast<ncclResult_t (*)()>(dlsym(h, "ncclGroupEnd"));
            auto pDestroy = reinterpret_cast<ncclResult_t (*)(ncclComm_t)>(dlsym(h, "ncclCommDestroy"));
            if (!pInitAll || !pAllReduce || !pGroupStart || !pGroupEnd || !pDestroy) throw Error(GR_E_NO_DEVICE, "librccl.so lacks the expected symbols");
            for (int i = 0; i < n; ++i)
                for (int k = 0; k < i; ++k)
                    if (devs[i] == devs[k]) throw Error(GR_E_INVALID, "gr_allreduce_counts: one sampler per device (two samplers share device " + std::to_string(devs[i]) + ")");
            // out of place, into a scratch buffer per device: every sampler's own counts stay what its chains sampled, so
            // gr_merge / gr_counts / further gr_advance calls after the reduction see unchanged data
            for (int i = 0; i < n; ++i) { samplers[i]->use(); samplers[i]->d_reduced.alloc(cnt); }
            struct Comms { // destroyed on every way out, also when a sync below throws
                std::vector<ncclComm_t> c;
                ncclResult_t (*destroy)(ncclComm_t);
                ~Comms() { for (ncclComm_t x : c) if (x) destroy(x); }
            } comms{std::vector<ncclComm_t>(n, nullptr), pDestroy};
            if (pInitAll(comms.c.data(), n, devs.data()) != ncclSuccess) throw Error(GR_E_NO_DEVICE, "ncclCommInitAll failed");
            pGroupStart();
            ncclResult_t r = ncclSuccess;
            for (int i = 0; i < n; ++i) {
                const ncclResult_t ri = pAllReduce(samplers[i]->d_counts.p, samplers[i]->d_reduced.p, cnt, ncclUint64, ncclSum, comms.c[i], samplers[i]->stream);
                if (ri != ncclSuccess) r = ri;
            }
            const ncclResult_t re = pGroupEnd();
            if (re != ncclSuccess) r = re;
            for (int i = 0; i < n; ++i) { samplers[i]->use(); samplers[i]->sync(); }
            if (r != ncclSuccess) throw Error(GR_E_NO_DEVICE, "ncclAllReduce failed");
            g_rccl_calls.fetch_add(1, std::memory_order_relaxed);
            if (out) {
                gr_sampler *s0 = samplers[0];
                s0->use();
                s0->download(out, s0->d_reduced.p, cnt * sizeof(uint64_t));
            }
            return;
        }
        if (out && gr_counts(samplers[0], out, nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
    });
}

GR_VISIBLE uint64_t gr_rccl_calls(void) { return g_rccl_calls.load(std::memory_order_relaxed); }

// ============================================================ adaptive (N2)
GR_VISIBLE int gr_set_active_groups(gr_sampler *s, uint32_t n_active) {
    return guarded([&] {
        if (!s) throw Error(GR_E_INVALID, "null sampler");
        if (s->started) throw Error(GR_E_STATE, "gr_set_active_groups must precede burn-in / advance");
        if (n_active < 1 || n_active > s->o.n_variants) throw Error(GR_E_INVALID, "n_active out of range");
        s->n_active = n_active;
        s->dirty = true; // the device pools hold the active groups only
    });
}
GR_VISIBLE int gr_active_groups(gr_sampler *s) { return s ? (int)s->n_active : fail(GR_E_INVALID, "null sampler"); }

// the two halves of a window of `updates_per_chain` counted updates, in time: counts at the start, in the middle, at the end
static void window_in_time(gr_sampler *s, uint64_t updates_per_chain) {
    {
        const size_t n = (size_t)s->o.n_variants * s->H;
        std::vector<uint64_t> c0(n), c1(n), c2(n);
        if (gr_counts(s, c0.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
        const uint64_t h1 = updates_per_chain / 2;
        s->run(h1, true);
        double ms = s->kernel_ms_last;
        uint32_t nl = s->launches_last;
        if (gr_counts(s, c1.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
        s->run(updates_per_chain - h1, true);
        s->kernel_ms_last += ms;
        s->launches_last += nl;
        if (gr_counts(s, c2.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
        s->win_first.assign(n, 0.0);
        s->win_second.assign(n, 0.0);
        for (size_t i = 0; i < n; ++i) {
            s->win_first[i] = (double)(c1[i] - c0[i]);
            s->win_second[i] = (double)(c2[i] - c1[i]);
        }
        // ConvergenceWindow ~ samples of one variable in one chain of the window
        s->win_cw = std::max(2.0, (double)updates_per_chain / (double)std::max<size_t>(1, s->image[0].free_vars.size()));
    }
}

GR_VISIBLE int gr_advance_window(gr_sampler *s, uint64_t updates_per_chain) {
    return guarded([&] {
        if (!s) throw Error(GR_E_INVALID, "null sampler");
        if (updates_per_chain < 2) throw Error(GR_E_INVALID, "a convergence window needs at least 2 updates");
        window_in_time(s, updates_per_chain);
    });
}


GR_VISIBLE int gr_advance_chain(gr_sampler *s, uint32_t cw, uint64_t *out_updates) {
    return guarded([&] {
        if (!s) throw Error(GR_E_INVALID, "null sampler");
        if (cw < 2) throw Error(GR_E_INVALID, "ConvergenceWindow must be >= 2");
        s->use();
        if (s->dirty) s->rebuild();
        if (s->o.flags & GR_FLAG_SCAN_SYSTEMATIC) {
            // AdvanceChain under the systematic scan needs no stopping-time replay and no draw log: a sweep selects every variable once.
            // cw + 1 sweeps of the longest order among the live groups -- chain.go:184-215's criterion (cw + 1 new samples of every
            // variable), met exactly by the groups that sweep n_max variables, passed by a selection or so per n_max / (n_max - n_g)
            // sweeps by a group with fewer (collapsed) ones -- and ChainHistory's halves (circular.go:16-28: the last 2 * (cw / 2)
            // selections of a variable, oldest half / newest half) are the last 2 * (cw / 2) sweeps split in time: any stretch of
            // h * n_max updates holds exactly h selections of every variable of an n_max-variable group, wherever it starts.
            uint64_t n_max = 1;
            for (uint32_t g = 0; g < (s->o.n_variants > 1 ? s->n_active : 1u); ++g) n_max = std::max<uint64_t>(n_max, s->image[g].free_vars.size());
            const uint64_t h = cw / 2u, L = (uint64_t)(cw + 1u) * n_max, W = 2ull * h * n_max;
            double ms = 0; uint32_t nl = 0;
            if (L > W) { s->run(L - W, true); ms = s->kernel_ms_last; nl = s->launches_last; }
            window_in_time(s, W);
            s->kernel_ms_last += ms; s->launches_last += nl;
            s->win_cw = (double)cw;
            if (out_updates) *out_updates = L;
            return;
        }
        const uint32_t act = s->active_chains();
        if (s->o.flags & GR_FLAG_SCAN_PER_CHAIN) {
            // Strict mode: every chain has its own scan stream, so AdvanceChain is the reference's, chain by chain (chain.go:184-215):
            // per-chain stopping times (k_stop_time_pc), a log of one value byte per chain and step written by the generic kernel's
            // per-chain scan, per-chain rings (k_ring_halves_pc) added up per 64-chain group.
            if (cw + 1u > 65535u) throw Error(GR_E_UNSUPPORTED, "ConvergenceWindow too large for the strict mode's 16-bit selection counters");
            if (s->stg) throw Error(GR_E_UNSUPPORTED, "strict per-chain scan needs the chain state in LDS");
            const uint32_t n_groups = (act + 63u) / 64u;
            uint32_t nfree_max = 1;
            for (const auto &im : s->image) nfree_max = std::max<uint32_t>(nfree_max, (uint32_t)im.free_vars.size());
            ScanArgs a{};
            a.variants = s->d_variants.p; a.free_vars = s->d_free.p; a.n_groups = n_groups; a.chains_per_variant = s->cpv;
            a.chain_offset_lo = (uint32_t)s->o.chain_offset; a.chain_offset_hi = (uint32_t)(s->o.chain_offset >> 32);
            a.seed_lo = (uint32_t)s->o.seed; a.seed_hi = (uint32_t)(s->o.seed >> 32);
            a.n_vars = (uint32_t)s->n_vars;
            a.step0_lo = (uint32_t)s->step; a.step0_hi = (uint32_t)(s->step >> 32);
            const uint32_t batch = 2u * (uint32_t)s->n_vars;
            const uint32_t max_batches = (uint32_t)std::min<uint64_t>((1ull << 31) / batch, 1u << 20);
            s->ac_out.alloc(act);
            hip_check(launch_stop_time_pc(a, act, nfree_max, cw, batch, max_batches, s->ac_out.p, s->stream), "k_stop_time_pc");
            std::vector<uint32_t> nb(act);
            s->download(nb.data(), s->ac_out.p, (size_t)act * sizeof(uint32_t));
            uint32_t worst = 0;
            for (uint32_t b : nb) {
                if (b == 0xffffffffu) throw Error(GR_E_UNSUPPORTED, "ConvergenceWindow too large: a chain would need more than 2^31 updates");
                worst = std::max(worst, b);
            }
            const uint64_t L = (uint64_t)worst * batch;
            const uint64_t pitch = L * 64u;
            if ((uint64_t)n_groups * pitch > (16ull << 30) || pitch > 0xfffffff0ull)
                throw Error(GR_E_UNSUPPORTED, "the strict mode's per-chain draw log of this window would exceed 16 GiB: use gr_advance_window");
            s->d_log.alloc((size_t)n_groups * pitch);
            std::vector<uint32_t> lim(act);
            for (uint32_t c = 0; c < act; ++c) lim[c] = nb[c] * batch;
            s->ac_limit.upload(lim, s->stream, s->stage);
            s->cur_log = s->d_log.p; s->cur_log_pitch = (uint32_t)pitch; s->cur_log_rows = 1; s->cur_log_vals = 64; s->cur_log_step0 = s->step;
            s->cur_limit = s->ac_limit.p; s->cur_log_pc = true;
            struct Off { gr_sampler *s; ~Off() { s->cur_log = nullptr; s->cur_limit = nullptr; s->cur_log_rows = 1; s->cur_log_vals = 1; s->cur_log_pc = false; } } off{s};
            s->run(L, true);
            for (uint32_t c = 0; c < act; ++c) s->skipped_updates += L - (uint64_t)nb[c] * batch;
            const size_t nH = (size_t)s->o.n_variants * s->H;
            s->win_first.assign(nH, 0.0);
            s->win_second.assign(nH, 0.0);
            s->win_cw = (double)cw;
            std::vector<uint32_t> hoff(s->n_vars + 1, 0);
            for (int v = 0; v < s->n_vars; ++v) hoff[v + 1] = hoff[v] + (uint32_t)s->card[v];
            s->ac_hoff.upload(hoff, s->stream, s->stage);
            a.hoff = s->ac_hoff.p; a.H = (uint32_t)s->H;
            s->ac_hist.alloc((size_t)n_groups * 2u * s->H);
            s->ac_status.alloc(n_groups);
            hip_check(launch_ring_halves_pc(a, act, nfree_max, cw / 2u, s->d_log.p, (uint32_t)pitch, s->ac_limit.p, s->ac_hist.p, s->ac_status.p, s->stream), "k_ring_halves_pc");
            std::vector<uint32_t> hist((size_t)n_groups * 2u * s->H), status(n_groups);
            s->download(hist.data(), s->ac_hist.p, hist.size() * sizeof(uint32_t));
            s->download(status.data(), s->ac_status.p, n_groups * sizeof(uint32_t));
            for (uint32_t st : status)
                if (st) throw Error(GR_E_STATE, "a ChainHistory ring is not full after AdvanceChain (internal error)");
            for (uint32_t g = 0; g < n_groups; ++g) {
                const uint32_t variant = (g * 64u) / s->cpv;
                for (int i = 0; i < s->H; ++i) {
                    s->win_first[(size_t)variant * s->H + i] += hist[((size_t)g * 2u + 0u) * s->H + i];
                    s->win_second[(size_t)variant * s->H + i] += hist[((size_t)g * 2u + 1u) * s->H + i];
                }
            }
            if (out_updates) *out_updates = L;
            return;
        }
        if (act % 64u) throw Error(GR_E_UNSUPPORTED, "gr_advance_chain needs a multiple of 64 chains");
        const uint32_t n_groups = act / 64u;
        ScanArgs a{};
        a.variants = s->d_variants.p; a.free_vars = s->d_free.p; a.n_groups = n_groups; a.chains_per_variant = s->cpv;
        a.chain_offset_lo = (uint32_t)s->o.chain_offset; a.chain_offset_hi = (uint32_t)(s->o.chain_offset >> 32);
        a.seed_lo = (uint32_t)s->o.seed; a.seed_hi = (uint32_t)(s->o.seed >> 32);
        a.n_vars = (uint32_t)s->n_vars;
        a.step0_lo = (uint32_t)s->step; a.step0_hi = (uint32_t)(s->step >> 32);
        // 1. how long: batches of 2N updates until every variable of every group has cw + 1 new samples (chain.go:184-215)
        const uint32_t batch = 2u * (uint32_t)s->n_vars;
        const uint32_t max_batches = (uint32_t)std::min<uint64_t>((1ull << 31) / batch, 1u << 20);
        DevBuf<uint32_t> &d_out = s->ac_out;
        d_out.alloc(n_groups);
        hip_check(launch_stop_time(a, cw, batch, max_batches, d_out.p, s->stream), "k_stop_time");
        std::vector<uint32_t> nb(n_groups);
        s->download(nb.data(), d_out.p, n_groups * sizeof(uint32_t));
        uint32_t worst = 0;
        for (uint32_t b : nb) {
            if (b == 0xffffffffu) throw Error(GR_E_UNSUPPORTED, "ConvergenceWindow too large: a chain group would need more than 2^31 updates");
            worst = std::max(worst, b);
        }
        const uint64_t L = (uint64_t)worst * batch; // the window: as long as the slowest group needs; every group executes its own nb * batch steps of it
        // 2. run, logging every step's draws per group: the number of ones when all selectable variables are binary,
        //    else one count per value (generic kernel only)
        bool binary = true;
        for (uint32_t g = 0; g < (s->o.n_variants > 1 ? s->n_active : 1u) && binary; ++g)
            for (uint32_t v : s->image[g].free_vars) binary = binary && s->card[v] == 2;
        uint32_t rows = (s->par || s->dir4 || s->pw4) ? 4u : 1u; // the 4-lane forms log one row per wavefront (16 chains)
        const uint32_t log_vals = binary ? 1u : (uint32_t)s->max_card;
        const uint64_t pitch = (L * log_vals + 15u) & ~15ull;
        // ... and the 8 / 16-lane form one per wavefront of 8 / 4 chains, while the window is short of wavefronts (an adaptive run's
        // first rounds) and the log stays small
        if (s->fast && s->par && binary && !s->strict_fast) {
            const uint32_t lpc = s->auto_lpc(n_groups);
            if (lpc > 4u && (uint64_t)n_groups * lpc * pitch <= (2ull << 30)) rows = lpc;
        }
        const bool logging = (binary || !s->fast) && pitch <= 0xfffffff0ull && (uint64_t)n_groups * rows * pitch <= (16ull << 30);
        const size_t nH = (size_t)s->o.n_variants * s->H;
        DevBuf<unsigned char> &d_log = s->d_log;
        std::vector<uint64_t> c0(nH), c1(nH), c2(nH);
        DevBuf<uint32_t> &d_limit = s->ac_limit;
        if (logging) {
            if (d_log.cap > (size_t(256) << 20) && d_log.cap > 4u * ((size_t)n_groups * rows * pitch)) d_log.reset(); // a much shorter window: give the big log back
            d_log.alloc((size_t)n_groups * rows * pitch); // (keeps a large enough allocation of an earlier window)
            hip_check(hipMemsetAsync(d_log.p, 0, (size_t)n_groups * rows * pitch, s->stream), "hipMemset");
            s->cur_log = d_log.p; s->cur_log_pitch = (uint32_t)pitch; s->cur_log_rows = rows; s->cur_log_vals = log_vals; s->cur_log_step0 = s->step;
            // every group stops at its own stopping time, like every reference chain does (chain.go:207-214)
            std::vector<uint32_t> lim(n_groups);
            for (uint32_t g = 0; g < n_groups; ++g) lim[g] = nb[g] * batch;
            d_limit.upload(lim, s->stream, s->stage);
            s->cur_limit = d_limit.p;
        }
        struct Off { gr_sampler *s; ~Off() { s->cur_log = nullptr; s->cur_limit = nullptr; s->cur_log_rows = 1; s->cur_log_vals = 1; } } off{s};
        if (logging) {
            s->run(L, true);
            for (uint32_t g = 0; g < n_groups; ++g) s->skipped_updates += 64ull * (L - (uint64_t)nb[g] * batch);
        } else { // no exact rings: fall back to the two halves of the window in time (gr_advance_window)
            if (gr_counts(s, c0.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
            s->run(L / 2, true);
            const double ms = s->kernel_ms_last; const uint32_t nl = s->launches_last;
            if (gr_counts(s, c1.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
            s->run(L - L / 2, true);
            s->kernel_ms_last += ms; s->launches_last += nl;
            if (gr_counts(s, c2.data(), nullptr)) throw Error(GR_E_NO_DEVICE, g_err);
        }
        s->win_first.assign(nH, 0.0);
        s->win_second.assign(nH, 0.0);
        s->win_cw = (double)cw;
        if (logging) {
            // 3. rebuild every ring's halves: the last 2 * (cw / 2) selections of each variable (circular.go:16-28)
            const uint32_t half = cw / 2u;
            DevBuf<uint32_t> &d_hist = s->ac_hist, &d_status = s->ac_status, &d_hoff = s->ac_hoff;
            std::vector<uint32_t> hoff(s->n_vars + 1, 0);
            for (int v = 0; v < s->n_vars; ++v) hoff[v + 1] = hoff[v] + (uint32_t)s->card[v];
            d_hoff.upload(hoff, s->stream, s->stage);
            a.hoff = d_hoff.p; a.H = (uint32_t)s->H;
            d_hist.alloc((size_t)n_groups * 2u * s->H);
            d_status.alloc(n_groups);
            // one block per group and SEGMENT of the window: enough of them to fill the chip (a lone wavefront per group paid
            // every latency in full: 8 ms for the 512 groups x 500 000 steps of the reference's adaptive experiment)
            const uint32_t n_chunks = (uint32_t)((L + 3u + 255u) / 256u + 1u); // 256-step chunks, counted from the end of the window
            uint32_t n_seg = std::min<uint32_t>({16u, std::max(1u, 4096u / n_groups), std::max(1u, n_chunks / 4u)});
            const uint32_t seg_chunks = (n_chunks + n_seg - 1u) / n_seg;
            n_seg = (n_chunks + seg_chunks - 1u) / seg_chunks;
            if (n_seg > 1u) s->ac_segcnt.alloc((size_t)n_groups * n_seg * (size_t)s->n_vars);
            hip_check(launch_ring_halves(a, (uint32_t)L, half, d_log.p, (uint32_t)pitch, rows, log_vals, d_limit.p, d_hist.p, d_status.p, n_seg, seg_chunks,
                                         s->ac_segcnt.p, s->stream), "k_ring_halves");
            std::vector<uint32_t> hist((size_t)n_groups * 2u * s->H), status(n_groups);
            s->download(hist.data(), d_hist.p, hist.size() * sizeof(uint32_t));
            s->download(status.data(), d_status.p, n_groups * sizeof(uint32_t));
            for (uint32_t st : status)
                if (st) throw Error(GR_E_STATE, "a ChainHistory ring is not full after AdvanceChain (internal error)");
            for (uint32_t g = 0; g < n_groups; ++g) {
                const uint32_t variant = (g * 64u) / s->cpv;
                for (int i = 0; i < s->H; ++i) {
                    s->win_first[(size_t)variant * s->H + i] += hist[((size_t)g * 2u + 0u) * s->H + i];
                    s->win_second[(size_t)variant * s->H + i] += hist[((size_t)g * 2u + 1u) * s->H + i];
                }
            }
        } else {
            for (size_t i = 0; i < nH; ++i) {
                s->win_first[i] = (double)(c1[i] - c0[i]);
                s->win_second[i] = (double)(c2[i] - c1[i]);
            }
        }
        if (out_updates) *out_updates = L;
    });
}

GR_VISIBLE int gr_window_halves(gr_sampler *s, double *first, double *second, double *cw) {
    return guarded([&] {
        if (!s || !first || !second) throw Error(GR_E_INVALID, "null argument");
        if (s->win_first.empty()) throw Error(GR_E_STATE, "no convergence window yet (gr_advance_window / gr_advance_chain)");
        std::copy(s->win_first.begin(), s->win_first.end(), first);
        std::copy(s->win_second.begin(), s->win_second.end(), second);
        if (cw) *cw = s->win_cw;
    });
}

GR_VISIBLE int gr_set_window_halves(gr_sampler *s, const double *first, const double *second, double cw) {
    return guarded([&] {
        if (!s || !first || !second) throw Error(GR_E_INVALID, "null argument");
        if (!(cw >= 2.0)) throw Error(GR_E_INVALID, "ConvergenceWindow must be >= 2");
        const size_t n = (size_t)s->o.n_variants * s->H;
        s->win_first.assign(first, first + n);
        s->win_second.assign(second, second + n);
        s->win_cw = cw;
    });
}

GR_VISIBLE int gr_chain_convergence(int32_t n_groups, int32_t n_vars, const int32_t *card, const int32_t *fixed,
                                    const int32_t *merged_collapsed, const double *first, const double *second,
                                    const double *merged, double cw, int32_t measure, double *out) {
    return guarded([&] {
        if (n_groups < 2) throw Error(GR_E_INVALID, "Convergence requires at least 2 chains");
        if (n_vars < 1 || !card || !first || !second || !merged || !out) throw Error(GR_E_INVALID, "null argument");
        int H = 0;
        for (int v = 0; v < n_vars; ++v) H += card[v];
        auto dist = [&](const double *a, const double *b, int c) -> double {
            switch (measure) {
            case 1: return JSDivergence(a, b, c);
            case 2: return MaxAbsDiff(a, b, c);
            case 3: return MeanAbsDiff(a, b, c);
            default: return HellingerDiff(a, b, c);
            }
        };
        const double m = (double)n_groups;
        const double bNorm = cw / (m - 1.0), wFactor = (cw - 1.0) / cw, bFactor = (m + 1.0) / (m * cw); // chain.go:52-59
        std::vector<double> v1, v2;
        int off = 0;
        for (int i = 0; i < n_vars; ++i) {
            const int c = card[i];
            if ((merged_collapsed && merged_collapsed[i]) || (fixed && fixed[i] >= 0)) { out[i] = 1.0; off += c; continue; }
            double W = 1e-8, B = 1e-8;
            for (int g = 0; g < n_groups; ++g) { // ChainDist (chain.go:253-290)
                v1.assign(first + (size_t)g * H + off, first + (size_t)g * H + off + c);
                v2.assign(second + (size_t)g * H + off, second + (size_t)g * H + off + c);
                for (int k = 0; k < c; ++k) { v1[k] += 1e-8; v2[k] += 1e-8; }
                W += dist(v1.data(), v2.data(), c);
                for (int k = 0; k < c; ++k) v1[k] += v2[k];
                B += dist(merged + off, v1.data(), c);
            }
            W /= m;
            B *= bNorm;
            const double vhat = wFactor * W + bFactor * B;
            out[i] = std::sqrt((4.0 * vhat) / (2.0 * W));
            off += c;
        }
    });
}

GR_VISIBLE int gr_adapt_select(const gr_model *base, const int32_t *merged_collapsed, const double *converge, uint32_t new_chain_count,
                               int32_t *out_vars, uint32_t *out_n) {
    return guarded([&] {
        if (!base || !out_vars || !out_n) throw Error(GR_E_INVALID, "null argument");
        *out_n = 0;
        const Model &m = base->m;
        const auto nbrs = VarNeighbors(m); // BlanketSize (gibbs-collapsed.go:81-83)
        std::vector<int> vars;
        for (size_t v = 0; v < m.Vars.size(); ++v) {
            const int sz = (int)nbrs[v].size();
            if (m.Vars[v].FixedVal < 0 && !(merged_collapsed && merged_collapsed[v]) && sz > 1 && sz <= kNeighborVarMax) vars.push_back((int)v);
        }
        if (vars.empty() || new_chain_count == 0) return;
        if (vars.size() <= new_chain_count) {
            for (int v : vars) out_vars[(*out_n)++] = v;
            return;
        }
        if (!converge) throw Error(GR_E_INVALID, "more candidates than slots: convergence scores are required");
        std::stable_sort(vars.begin(), vars.end(), [&](int a, int b) { return converge[a] > converge[b]; });
        for (uint32_t cc = 0, pos = (uint32_t)vars.size() - 1; cc < new_chain_count; ++cc, --pos) out_vars[(*out_n)++] = vars[pos];
    });
}

// The first half of an activation: the next inactive group becomes a fresh clone of the base model with var_idx collapsed
// (adaptive.go:133-145) and counts as active; its burn-in (and the rebuild of the device pools) is the caller's.
static int activate_collapsed(gr_sampler *s, int32_t var_idx, int32_t *out_var) {
    return guarded([&] {
        if (!s) throw Error(GR_E_INVALID, "null sampler");
        if (s->n_active >= s->o.n_variants) throw Error(GR_E_STATE, "every chain group is already active");
        const uint32_t g = s->n_active;
        int32_t cv = -1;
        if (gr_collapse(s, g, var_idx, &cv, nullptr)) throw Error(GR_E_COLLAPSE, g_err);
        s->n_active = g + 1;
        s->counted_g[g] = 0;
        if (out_var) *out_var = cv;
    });
}

GR_VISIBLE int gr_adapt(gr_sampler *s, uint32_t new_chain_count, int32_t *out_vars, uint32_t *out_n, double *out_converge) {
    return guarded([&] {
        if (!s || !out_vars || !out_n) throw Error(GR_E_INVALID, "null argument");
        *out_n = 0;
        if (out_converge) std::fill(out_converge, out_converge + s->n_vars, std::nan(""));
        if (s->o.n_variants < 2) throw Error(GR_E_INVALID, "adaptation needs a sampler with several chain groups");
        if (s->n_active < 2) throw Error(GR_E_INVALID, "At least 2 chains required for adaptation");
        if (s->n_active >= s->o.n_variants) return; // MaxChains reached (adaptive.go:62-64)
        std::vector<double> sum(s->H), cm(s->H), merged(s->H);
        std::vector<int32_t> cvv(s->n_vars), coll(s->n_vars);
        if (gr_merge_local(s, sum.data(), cvv.data(), cm.data())) throw Error(GR_E_NO_DEVICE, g_err);
        int off = 0;
        for (int v = 0; v < s->n_vars; ++v) {
            coll[v] = cvv[v] >= 0;
            for (int k = 0; k < s->card[v]; ++k) merged[off + k] = coll[v] ? cm[off + k] : sum[off + k];
            off += s->card[v];
        }
        gr_model base{s->base_linear};
        const uint32_t slots = std::min<uint32_t>(new_chain_count, s->o.n_variants - s->n_active);
        std::vector<double> conv;
        uint32_t n_sel = 0;
        std::vector<int32_t> sel(std::max<uint32_t>(slots, 1));
        int rc = gr_adapt_select(&base, coll.data(), nullptr, slots, sel.data(), &n_sel);
        if (rc) { // more candidates than slots: rank by ChainConvergence over the active groups
            if (s->win_first.empty()) throw Error(GR_E_STATE, "gr_adapt needs a preceding gr_advance_window");
            conv.resize(s->n_vars);
            if (gr_chain_convergence((int32_t)s->n_active, s->n_vars, s->card.data(), s->fixed.data(), coll.data(), s->win_first.data(),
                                     s->win_second.data(), merged.data(), s->win_cw, 0, conv.data()))
                throw Error(GR_E_INVALID, g_err);
            if (out_converge) std::copy(conv.begin(), conv.end(), out_converge);
            if (gr_adapt_select(&base, coll.data(), conv.data(), slots, sel.data(), &n_sel)) throw Error(GR_E_INVALID, g_err);
        }
        // adaptive.go:131-153, batched: every new group is collapsed first, then ONE rebuild of the device pools and ONE
        // burn-in launch over all of them (the groups are independent -- streams are keyed by chain and step --, so the result
        // is what activating them one by one gives; rebuild() is the expensive part of an activation)
        struct Finish { // ... also for the groups already added when a later collapse is refused
            gr_sampler *s; uint32_t first;
            void run() {
                const uint32_t f = first;
                first = s->n_active;
                if (s->n_active > f) { if (s->dirty) s->rebuild(); s->burn_groups(f, s->n_active, 2); }
            }
            ~Finish() { try { run(); } catch (...) {} }
        } finish{s, s->n_active};
        for (uint32_t i = 0; i < n_sel; ++i) {
            int32_t cv = -1;
            if (activate_collapsed(s, sel[i], &cv)) throw Error(GR_E_COLLAPSE, g_err);
            out_vars[(*out_n)++] = cv;
        }
        finish.run();
    });
}

GR_VISIBLE int gr_activate_group(gr_sampler *s, int32_t var_idx, int32_t *out_var) {
    return guarded([&] {
        if (activate_collapsed(s, var_idx, out_var)) throw Error(GR_E_COLLAPSE, g_err);
        // NewChain(..., burnIn = 2): two uncounted updates on the new group only, at step positions it has not used
        if (s->dirty) s->rebuild();
        s->burn_groups(s->n_active - 1, s->n_active, 2);
    });
}

GR_VISIBLE int gr_state(gr_sampler *s, uint32_t chain, int32_t *out) {
    return guarded([&] {
        if (!s || !out) throw Error(GR_E_INVALID, "null argument");
        if (chain >= s->o.n_chains) throw Error(GR_E_INVALID, "chain out of range");
        s->use();
        std::vector<unsigned char> col(s->n_vars);
        void *tmp = s->stage.take(s->n_vars);
        hip_check(hipMemcpy2DAsync(tmp, 1, s->d_state.p + chain, s->pitch, 1, s->n_vars, hipMemcpyDeviceToHost, s->stream), "D2H state");
        hip_check(hipStreamSynchronize(s->stream), "hipStreamSynchronize");
        memcpy(col.data(), tmp, s->n_vars);
        s->stage.reset();
        for (int v = 0; v < s->n_vars; ++v) out[v] = col[v];
    });
}
GR_VISIBLE int gr_set_state(gr_sampler *s, uint32_t chain, const int32_t *in) {
    return guarded([&] {
        if (!s || !in) throw Error(GR_E_INVALID, "null argument");
        if (chain >= s->o.n_chains) throw Error(GR_E_INVALID, "chain out of range");
        s->use();
        std::vector<unsigned char> col(s->n_vars);
        for (int v = 0; v < s->n_vars; ++v) {
            if (in[v] < 0 || in[v] >= s->card[v]) throw Error(GR_E_INVALID, "state value out of range");
            col[v] = (unsigned char)in[v];
        }
        void *tmp = s->stage.take(s->n_vars);
        memcpy(tmp, col.data(), s->n_vars);
        hip_check(hipMemcpy2DAsync(s->d_state.p + chain, s->pitch, tmp, 1, 1, s->n_vars, hipMemcpyHostToDevice, s->stream), "H2D state");
        s->sync();
    });
}

GR_VISIBLE int gr_conditional(gr_sampler *s, uint32_t variant, int32_t v, const int32_t *state, double *probs_out) {
    return guarded([&] {
        if (!s || !state || !probs_out) throw Error(GR_E_INVALID, "null argument");
        if (variant >= s->o.n_variants || v < 0 || v >= s->n_vars) throw Error(GR_E_INVALID, "bad variant / variable");
        if (s->o.n_variants > 1 && variant >= s->n_active) throw Error(GR_E_STATE, "chain group is not active (its device tables are built when it is activated)");
        const Variable &var = s->vmodel[variant].Vars[v];
        if (var.FixedVal >= 0) throw Error(GR_E_INVALID, "Selected sample variable " + var.Name + " which has FixedVal=" + std::to_string(var.FixedVal));
        if (var.Collapsed) throw Error(GR_E_INVALID, "variable is collapsed");
        for (int u = 0; u < s->n_vars; ++u)
            if (state[u] < 0 || state[u] >= s->card[u]) throw Error(GR_E_INVALID, "state value out of range");
        s->use();
        if (s->dirty) s->rebuild();
        DevBuf<int32_t> &d_st = s->cd_state;
        DevBuf<double> &d_pr = s->cd_probs;
        d_st.upload(std::vector<int32_t>(state, state + s->n_vars), s->stream, s->stage);
        d_pr.alloc(var.Card);
        hip_check(launch_conditional(s->h_variants[variant], s->d_progs.p, s->d_prog_off.p, s->d_views.p, (uint32_t)v, d_st.p, d_pr.p, s->max_card, s->stream), "k_conditional");
        s->download(probs_out, d_pr.p, var.Card * sizeof(double));
    });
}

GR_VISIBLE int gr_sampler_get_info(gr_sampler *s, gr_sampler_info *out) {
    return guarded([&] {
        if (!s || !out || out->struct_size != sizeof(gr_sampler_info)) throw Error(GR_E_INVALID, "bad gr_sampler_info");
        if (s->dirty) { s->use(); s->configure(); }
        out->block_threads = s->T; out->n_blocks = s->blocks;
        out->lds_bytes = s->use_chroma(s->active_chains()) ? s->chroma_lds : s->bits ? s->bits_lds : s->fast && s->par ? s->par_lds : s->pw4 ? s->pw4_lds : s->dir4 ? s->dir_lds : s->lds; // of the kernel that actually runs
        if (s->fast && s->par) { out->block_threads = s->strict_fast ? s->T : 64u * s->lpc_last; out->n_blocks = s->active_chains() / 64u; } // (64 x lanes per chain of the last launch)
        out->path = s->sysscan ? (s->use_chroma(s->active_chains()) ? 9u : s->pw4 ? 6u : (s->dir4 && !s->fast) ? 8u : s->sys_wide() ? 3u : s->sys_bits() ? 11u : 0u) : s->fast ? (s->strict_fast ? 5u : s->bits ? 11u : s->par ? 3u : 2u) : (s->pw4 ? 6u : s->dir4 ? 8u : s->cpt ? 1u : 0u) + (s->stg ? 16u : 0u); out->max_card = (uint32_t)s->max_card; out->tables_in_lds = s->tab_lds ? 1u : (s->par && s->par_tabg && s->par_tab_words) ? 2u : 0u;
        out->n_launches_last = s->launches_last; out->kernel_ms_last = s->kernel_ms_last; out->step = s->step;
    });
}

} // extern "C"
