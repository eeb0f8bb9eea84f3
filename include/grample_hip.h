/*
 * grample_hip.h -- C ABI of the MI355X-native Gibbs hot path for grample.
 *
 * This is the drop-in boundary: a Go host (cgo), a C++ host or Python (ctypes)
 * binds exactly these symbols.  Plain pointers and sizes only; no C++ or torch
 * types.  Every entry point cites the reference interface it replaces
 * (file:line relative to CraigKelly/grample).
 *
 * Conventions
 *   - return 0 = OK, negative = error (GR_E_*); gr_last_error() gives the text
 *     (thread-local).  Nothing aborts or throws across the ABI.
 *   - the library never keeps a caller pointer after the call returns (cgo
 *     pointer rule); outputs are caller-allocated.
 *   - handles are not thread-safe; distinct handles may be used from distinct
 *     OS threads (one gr_sampler per device).
 *   - calls are synchronous: the sampler's HIP stream is synchronised before
 *     a call returns.
 *   - there is NO CPU fallback: every sampling entry point fails with
 *     GR_E_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef GRAMPLE_HIP_H
#define GRAMPLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR_OK 0
#define GR_E_INVALID (-1)    /* bad argument / malformed input (Go: errors.Errorf paths) */
#define GR_E_NO_DEVICE (-2)  /* no HIP device / HIP runtime failure */
#define GR_E_UNSUPPORTED (-3)/* valid for the reference, outside this build's limits (e.g. Card > 64) */
#define GR_E_STATE (-4)      /* call order violation (e.g. collapse after advance) */
#define GR_E_COLLAPSE (-5)   /* Collapse refused, same cases as gibbs-collapsed.go:122-196 */

typedef struct gr_model gr_model;       /* model.Model (model/model.go:24-29) */
typedef struct gr_solution gr_solution; /* model.Solution (model/solution.go:16-18) */
typedef struct gr_sampler gr_sampler;   /* a set of sampler.Chain + their FullSampler on ONE device */

/* ------------------------------------------------------------------ general */
const char *gr_version(void);
const char *gr_last_error(void);
/* number of usable HIP devices (0 when there is none; never an error) */
int gr_device_count(void);

/* -------------------------------------------------------------------- model
 * Flat image of what UAIReader.ReadModel builds (model/uai.go:53-179):
 * table[] holds LINEAR-space values exactly as parsed, last scope variable
 * fastest (model/function.go:13-36).  fixed may be NULL (= no evidence). */
typedef struct gr_model_desc {
    int32_t n_vars;
    const int32_t *card;      /* [n_vars]  Variable.Card */
    const int32_t *fixed;     /* [n_vars]  Variable.FixedVal, -1 = free; may be NULL */
    int32_t n_funcs;
    const int32_t *scope_off; /* [n_funcs+1] */
    const int32_t *scope_var; /* [scope_off[n_funcs]] Function.Vars, most significant first */
    const int32_t *tab_off;   /* [n_funcs+1] */
    const double *table;      /* [tab_off[n_funcs]] Function.Table */
} gr_model_desc;

gr_model *gr_model_create(const gr_model_desc *desc);                /* + Model.Check (model.go:115-157) */
gr_model *gr_model_from_uai(const char *text, size_t len);           /* NewModelFromBuffer (model.go:79-91) */
gr_model *gr_model_from_file(const char *path, int use_evidence);    /* NewModelFromFile (model.go:52-76) */
int gr_model_apply_evidence(gr_model *m, const char *text, size_t len); /* UAIReader.ApplyEvidence (uai.go:183-249) */
gr_model *gr_model_clone(const gr_model *m);                         /* Model.Clone (model.go:32-49) */
void gr_model_destroy(gr_model *m);
int gr_model_n_vars(const gr_model *m);
int gr_model_n_funcs(const gr_model *m);
int gr_model_total_card(const gr_model *m);                          /* sum of Card = length of a marginal vector */
int gr_model_cards(const gr_model *m, int32_t *out /*[n_vars]*/);
int gr_model_fixed(const gr_model *m, int32_t *out /*[n_vars]*/);
int gr_model_func_arity(const gr_model *m, int f);
int gr_model_func_tabsize(const gr_model *m, int f);
int gr_model_func_scope(const gr_model *m, int f, int32_t *out /*[arity]*/);
int gr_model_func_table(const gr_model *m, int f, double *out /*[tabsize]*/);
/* Function.Eval (model/function.go:146-157): *out = NaN and GR_E_INVALID on bad input */
int gr_model_func_eval(const gr_model *m, int f, const int32_t *values, int n_values, double *out);
/* The sweep order of GR_FLAG_SCAN_SYSTEMATIC for this model (pure host function): the selectable variables (FixedVal < 0) by
 * (colour, id), colours from a greedy colouring of the factor graph in ascending id.  out_order [n_vars] receives *out_n
 * variables, out_color [n_vars] (may be NULL) the colour of every variable (-1 for evidence). */
int gr_model_systematic_order(const gr_model *m, int32_t *out_order, int32_t *out_n, int32_t *out_color);

/* ----------------------------------------------------------------- solution */
gr_solution *gr_solution_from_text(const char *text, size_t len);    /* ReadMargSolution (uai.go:252-332) */
gr_solution *gr_solution_from_file(const char *path);                /* NewSolutionFromFile (solution.go:21-33) */
void gr_solution_destroy(gr_solution *s);
int gr_solution_n_vars(const gr_solution *s);
int gr_solution_total_card(const gr_solution *s);
int gr_solution_cards(const gr_solution *s, int32_t *out);
int gr_solution_marginals(const gr_solution *s, double *out /*[total_card]*/);

/* model.NewErrorSuite (model/error.go:28-78).  m1/m2 = concatenated marginals
 * (any positive scale).  fixed1/fixed2 may be NULL.  out[8] = MeanMeanAbsError,
 * MeanMaxAbsError, MeanHellinger, MeanJSDiverge, MaxMeanAbsError,
 * MaxMaxAbsError, MaxHellinger, MaxJSDiverge. */
int gr_error_suite(int32_t n_vars, const int32_t *card, const int32_t *fixed1, const int32_t *fixed2,
                   const double *m1, const double *m2, double *out);

/* ------------------------------------------------------------------ sampler
 * One gr_sampler = n_chains independent chains (sampler.Chain, chain.go:13-20)
 * on one device, in n_variants equal groups.  All chains of a group share one
 * model clone (root.go:385): group g is what the reference builds with
 * NewGibbsSimple / NewGibbsCollapsed (+ Collapse) for each of its chains. */
#define GR_FLAG_DIRECT (1u << 0)      /* never memoise conditionals into tables (debug / cross-check) */
#define GR_FLAG_SEQUENTIAL (1u << 1)  /* never co-issue commuting updates (debug / cross-check) */
#define GR_FLAG_CHAIN_PER_LANE (1u << 2) /* never use the step-parallel (4 lanes per chain) kernel */
#define GR_FLAG_SCAN_PER_CHAIN (1u << 3) /* strict mode: every chain draws its own variable at every step, exactly like
                                            UniformSampler.VarSample per chain (sampler.go:135-174); the default shares the
                                            draw inside a 64-chain group (same per-chain law).  About 1.6x slower on binary
                                            memoised models (k_sweep_strict), 20x on the direct path. */

#define GR_FLAG_STATE_GLOBAL (1u << 4)  /* keep the chain state in global memory (one coalesced row per wavefront and variable) instead of
                                            staging it in LDS: chosen automatically for models whose state does not fit LDS
                                            (n_vars * 64 B > 160 KiB per wavefront); as a flag it forces that kernel (cross-checks) */

#define GR_FLAG_SCAN_SYSTEMATIC (1u << 5) /* systematic (chromatic) scan instead of the reference's random scan (sampler.go:135-174): the selectable
                                            variables are coloured so that variables of one colour share no factor; update t of a chain resamples
                                            variable order[t mod n] -- colour class by colour class, ascending id inside a class -- so updates
                                            per chain still count single-variable updates and n of them are one sweep.  The stationary law is the
                                            model's, like the random scan's; the chain's law is NOT the reference's (off by default).
                                            Binary memoisable models (up to 8 lane slots of 64 variables per colour chunk, 12 neighbours) run k_sweep_chroma / k_sweep_chroma_l: one wavefront per chain, one lane per
                                            variable of a colour class, all of a class updated side by side -- full occupancy from 4 096 chains up.
                                            The other model families keep their random-scan kernels (paths 6, 8 and the 8 / 16-lane form of path 3) with
                                            the plan following the sweep order; the generic kernel (path 0) serves what is left.
                                            Not combinable with GR_FLAG_SCAN_PER_CHAIN.  gr_advance_chain(cw) runs cw + 1 sweeps of the longest
                                            order among the live groups (every variable gets its cw + 1 new samples) and takes the ring halves
                                            from the last 2 * (cw / 2) sweeps split in time (exactly cw / 2 selections of every variable per half
                                            for the groups that sweep that many variables); gr_advance_window and gr_adapt work as ever. */

typedef struct gr_sampler_opts {
    uint32_t struct_size;    /* = sizeof(gr_sampler_opts) */
    uint32_t n_chains;       /* chains on this device */
    uint32_t n_variants;     /* chain groups (>=1); n_chains % n_variants == 0 */
    int32_t device;          /* HIP device ordinal */
    uint64_t seed;           /* rand.NewGenerator(seed) (rand.go:47-49); Philox4x32-10 key */
    uint64_t chain_offset;   /* global index of this device's chain 0 (multi-GPU: results do not depend on the split) */
    uint32_t variant_offset; /* global index of this device's variant 0 */
    uint32_t flags;          /* GR_FLAG_* */
    uint32_t block_threads;  /* 0 = auto; else 64/128/256 */
    uint32_t fast_form;      /* kernel form: 0 = auto; 1 = chain per lane; 2 = 4 lanes per chain (binary memoised models: one lane per step
                                of a quad; large-Card pairwise models on the direct path: one lane per block of 4 values);
                                4, 5, 6 = 4 lanes per chain with the step records read from LDS one by one / from LDS together /
                                from global memory one iteration ahead (2 picks by occupancy); 7 = as 6 with the threshold pool split
                                between LDS (its first half) and global memory whatever the occupancy (cross-checks of that kernel);
                                3 = as 0, but never more than 4 lanes per chain (measurements); 11 = 4 lanes per chain with bit-packed state (one 16-bit mask per variable and wavefront; binary memoised models, <= 12 neighbours, tables in LDS; GR_E_UNSUPPORTED otherwise); 8, 9 = 8 / 16 lanes per chain (binary memoised models; chosen automatically while a launch has too few
                                wavefronts for the chip: small chain counts, the first rounds of an adaptive run, models whose LDS
                                footprint leaves a CU two 64-chain blocks) */
} gr_sampler_opts;

/* NewGibbsSimple for every chain (gibbs-simple.go:25-115): clones the model,
 * UseLogSpace on every factor, var->factor adjacency, uniform random start
 * state honouring evidence.  The model handle may be destroyed afterwards. */
gr_sampler *gr_sampler_create(const gr_model *m, const gr_sampler_opts *opts);
void gr_sampler_destroy(gr_sampler *s);

/* (*GibbsCollapsed).Collapse (gibbs-collapsed.go:98-314) applied to the model
 * of chain group `variant`: var_idx < 0 picks a random tractable variable.
 * Writes the variable's index and its exact marginal [Card].  Like the
 * reference it re-draws that group's start state.  Only before sampling. */
int gr_collapse(gr_sampler *s, uint32_t variant, int32_t var_idx, int32_t *out_var, double *out_marginal);
/* BlanketSize / FunctionCount (gibbs-collapsed.go:81-88) on a group's current model */
int gr_blanket_size(gr_sampler *s, uint32_t variant, int32_t var_idx);
int gr_function_count(gr_sampler *s, uint32_t variant, int32_t var_idx);

/* NewChain's burn-in loop (chain.go:166-172): updates_per_chain single-variable
 * updates on every chain, nothing counted. */
int gr_burnin(gr_sampler *s, uint64_t updates_per_chain);
/* Chain.AdvanceChain + wg.Wait (chain.go:180-218, root.go:476-479) with an
 * explicit length: updates_per_chain x oneSample(true) on every chain. */
int gr_advance(gr_sampler *s, uint64_t updates_per_chain);
/* Chain.TotalSampleCount summed over this sampler's chains (root.go:486-490) */
int gr_total_samples(gr_sampler *s, uint64_t *out);

/* Raw sample counts: counts[g * total_card + off(v) + val] = number of counted
 * updates that set v := val in group g.  (Variable.Marginal minus its 1/Card
 * prior, chain.go:230-236.)  counts may be NULL. */
int gr_counts(gr_sampler *s, uint64_t *counts /*[n_variants*total_card]*/, uint64_t *n_updates);
/* device-resident copy of the same array, for an external collective (RCCL
 * through the host framework): pointer stays valid until destroy. */
int gr_counts_device(gr_sampler *s, void **dptr, size_t *n_elems);
int gr_counts_upload(gr_sampler *s, const uint64_t *counts); /* overwrite the raw counts (one part of a checkpoint restore, below) */

/* ------------------------------------------------------ checkpoint / resume
 * The reference has none (its trace is write-only); with a counter-based RNG a
 * sampler's whole condition is: the chain states, the raw counts, this position
 * and -- for an adaptive sampler between gr_advance_chain and gr_adapt -- the
 * ChainHistory halves of the last window (gr_window_halves / gr_set_window_halves).  To resume: build the sampler exactly as before (same model, opts,
 * and the same gr_collapse / gr_activate_group calls with explicit variables),
 * then gr_state_upload + gr_counts_upload + gr_set_position; every later
 * gr_burnin / gr_advance continues the Philox streams where the checkpointed
 * run left them, so the resumed run's counts equal the uninterrupted run's bit
 * for bit (tests/test_gpu_parity.py::test_checkpoint_resume_is_bit_exact). */
typedef struct gr_position {
    uint32_t struct_size;     /* = sizeof(gr_position) */
    uint32_t n_active;        /* chain groups running (gr_active_groups) */
    uint64_t step;            /* updates per chain executed so far, burn-in included = position in every Philox stream */
    uint64_t counted;         /* counted updates per chain (samplers with one group) */
    uint64_t skipped_updates; /* updates not made inside gr_advance_chain windows by groups that stopped early */
} gr_position;
/* counted_per_group [n_variants] (counted updates per chain of each group) and
 * epoch_per_group [n_variants] (start-state re-draw counter, gibbs-simple.go:132-142) may be NULL in gr_get_position;
 * gr_set_position requires counted_per_group for a sampler with more than one chain group */
int gr_get_position(gr_sampler *s, gr_position *pos, uint64_t *counted_per_group, uint32_t *epoch_per_group);
/* pos->n_active must equal the sampler's: groups are re-created by replaying the collapses, not by this call */
int gr_set_position(gr_sampler *s, const gr_position *pos, const uint64_t *counted_per_group, const uint32_t *epoch_per_group);
/* Chain.LastSample of every chain: states[v * n_chains + chain], one byte per variable and chain */
int gr_state_download(gr_sampler *s, uint8_t *states /*[n_vars*n_chains]*/);
int gr_state_upload(gr_sampler *s, const uint8_t *states /*[n_vars*n_chains]*/);

/* MergeChains (chain.go:96-148) over all chains of `n` samplers (= devices),
 * in global variant order: a variable collapsed in ANY group takes the first
 * such group's exact marginal; otherwise marginal = sum over chains of
 * (1/Card + counts).  out_marginal [total_card] is NOT normalised (like the
 * reference); out_collapsed [n_vars] may be NULL. */
int gr_merge(gr_sampler *const *samplers, int n, double *out_marginal, int32_t *out_collapsed);
/* Same, but only the device-local half: per-variable partial sums and, for
 * collapsed variables, the lowest global variant that collapsed them and its
 * marginal.  Lets a multi-PROCESS host finish the merge with one all-reduce. */
int gr_merge_local(gr_sampler *s, double *out_sum /*[total_card]*/, int32_t *out_coll_variant /*[n_vars], -1 = none*/,
                   double *out_coll_marginal /*[total_card]*/);
/* Sum of the raw count arrays of n samplers living in ONE process, one per
 * device, with a single ncclAllReduce(sum, uint64) over RCCL, OUT OF PLACE: the
 * total lands in a scratch buffer per device and in out; every sampler's own
 * counts stay what its chains sampled, so gr_merge, gr_counts and further
 * gr_advance calls are unaffected.  out may be NULL. */
int gr_allreduce_counts(gr_sampler *const *samplers, int n, uint64_t *out /*[n_variants*total_card]*/);
/* With n == 1 there is nothing to exchange and the call returns the sampler's own counts without touching RCCL;
 * the environment variable GR_ALLREDUCE_FORCE_RCCL=1 (read once per process) makes it run the collective all the
 * same (a one-device communicator), which is how the RCCL branch is exercised on a one-GPU box.
 * gr_rccl_calls: ncclAllReduce calls this process has completed inside gr_allreduce_counts (0 = RCCL never ran). */
uint64_t gr_rccl_calls(void);

/* Chain.LastSample of one chain (chain.go:19) */
int gr_state(gr_sampler *s, uint32_t chain, int32_t *out /*[n_vars]*/);
int gr_set_state(gr_sampler *s, uint32_t chain, const int32_t *in /*[n_vars]*/);

/* The conditional law GibbsSimple.SampleVar draws from (gibbs-simple.go:175-258)
 * for variable var_idx of group `variant` given a full state, evaluated ON THE
 * DEVICE by the same code the sampling kernels use: probs_out[Card]. */
int gr_conditional(gr_sampler *s, uint32_t variant, int32_t var_idx, const int32_t *state, double *probs_out);

/* Introspection for benchmarks: milliseconds spent in the sweep kernel(s) of the
 * last gr_burnin/gr_advance call (HIP events on the sampler's stream), number
 * of launches, and the configuration chosen. */
typedef struct gr_sampler_info {
    uint32_t struct_size;
    uint32_t block_threads;
    uint32_t n_blocks;
    uint32_t lds_bytes;
    uint32_t path;          /* 0 = direct conditional, 1 = memoised conditional tables, 2 = memoised + planned scan (binary fast path), 3 = same, 4 lanes per chain, 5 = strict per-chain scan on memoised tables,
                               6 = direct conditional, value-parallel (4 lanes per chain, pairwise models with 8 < Card <= 16);
                               8 = direct conditional, step-parallel (4 lanes per chain: Card <= 4, blankets too large to memoise; collapse factors as chained entries);
                               9 = systematic (chromatic) scan on memoised tables, one wavefront per chain (GR_FLAG_SCAN_SYSTEMATIC);
                               11 = as 3 with the chain state bit-packed in LDS (chosen for models whose state bytes would take more than half of a CU's LDS; fast_form 11 forces it);
                               + 16 when the chain state stays in global memory (paths 0 and 1 only) */
    uint32_t max_card;
    uint32_t tables_in_lds; /* threshold / view tables: 1 = staged in LDS, 0 = read from global memory, 2 = split (4-lane memoised form:
                               the small tables in LDS, the large ones from global memory) */
    uint32_t n_launches_last;
    double kernel_ms_last;
    uint64_t step;          /* updates per chain executed so far (burn-in included) */
} gr_sampler_info;
int gr_sampler_get_info(gr_sampler *s, gr_sampler_info *out);

/* ------------------------------------------------------- adaptive sampler (N2)
 * ConvergenceSampler.Adapt (sampler/adaptive.go:57-157) + ChainConvergence (sampler/chain.go:32-92).
 * A chain GROUP plays the role of one reference chain: same model clone, same collapsed variable.
 * The sampler is created with n_variants = the most groups it may ever hold (reference MaxChains = 128,
 * adaptive.go:49); only the first gr_active_groups() run, and gr_adapt activates more. */

/* number of groups that sample / are merged; set before the first burn-in (default: all) */
int gr_set_active_groups(gr_sampler *s, uint32_t n_active);
int gr_active_groups(gr_sampler *s);
/* AdvanceChain over a convergence window: like gr_advance, and additionally keeps, per group, the
 * histograms of the first and of the second half of this window -- the two halves ChainDist
 * (chain.go:253-290) reads from ChainHistory.  (The reference's ring holds each variable's last
 * cw selections; here the halves are the two halves of the window in time.) */
int gr_advance_window(gr_sampler *s, uint64_t updates_per_chain);
/* Chain.AdvanceChain (chain.go:180-218) with its own stopping rule: batches of 2 N updates until every
 * selectable variable has been selected cw + 1 more times -- evaluated per 64-chain scan group from the
 * scan stream; *out_updates = the slowest group's length, by which every chain's stream position advances, while
 * every group samples only up to its own stopping time (gr_total_samples counts what was sampled).  The kernels
 * log every step's draws per group and the two halves of every ChainHistory ring (the variable's last
 * 2*(cw/2) selections, buffer/circular.go) are rebuilt exactly for gr_adapt / ChainConvergence; a window
 * whose log would exceed 16 GiB runs every group for the slowest group's length and keeps the halves of the
 * window in time instead.
 * With GR_FLAG_SCAN_PER_CHAIN (round 3) the rule is the reference's chain by chain: every chain has its own scan stream, its own
 * stopping time and its own rings (one logged value byte per chain and step; k_stop_time_pc / k_ring_halves_pc); the halves
 * returned are still summed per chain group.  Models whose chain state stays in global memory are served too.  Not available
 * with GR_FLAG_SCAN_SYSTEMATIC (the rule belongs to the random scan). */
int gr_advance_chain(gr_sampler *s, uint32_t cw, uint64_t *out_updates);
/* the half-window histograms kept by the last gr_advance_window / gr_advance_chain:
 * first/second [n_variants * total_card] (inactive groups: zeros); *cw = the ConvergenceWindow they stand for */
int gr_window_halves(gr_sampler *s, double *first, double *second, double *cw);
/* restore them (checkpoint / resume of an adaptive sampler: the next gr_adapt ranks with these) */
int gr_set_window_halves(gr_sampler *s, const double *first, const double *second, double cw);
/* ChainConvergence (chain.go:32-92) as a pure function: first/second [n_groups*total_card] half-window
 * histograms (any non-negative numbers; 1e-8 is added like ChainDist does), merged [total_card],
 * merged_collapsed [n_vars] (may be NULL), cw = ConvergenceWindow, measure 0 Hellinger / 1 JS /
 * 2 MaxAbs / 3 MeanAbs.  out [n_vars]; collapsed or fixed variables score 1.0. */
int gr_chain_convergence(int32_t n_groups, int32_t n_vars, const int32_t *card, const int32_t *fixed,
                         const int32_t *merged_collapsed, const double *first, const double *second,
                         const double *merged, double cw, int32_t measure, double *out);
/* The selection rule of Adapt (adaptive.go:78-119): eligible = free, not collapsed in the merge,
 * blanket size 2..12 on the base model; if there are no more candidates than slots all are taken in
 * id order, else they are sorted by descending score and taken from the TAIL (lowest scores first).
 * converge may be NULL when no ranking is needed.  out_vars [new_chain_count]. */
int gr_adapt_select(const gr_model *base, const int32_t *merged_collapsed, const double *converge,
                    uint32_t new_chain_count, int32_t *out_vars, uint32_t *out_n);
/* Adapt: merge the active groups, rank with the halves kept by the last gr_advance_window, and for
 * every selected variable activate one more group: Collapse(var) on a fresh clone of the base model,
 * NewChain with a burn-in of 2 updates (adaptive.go:131-153).  Nothing happens once all n_variants
 * groups are active.  out_converge [n_vars] may be NULL. */
int gr_adapt(gr_sampler *s, uint32_t new_chain_count, int32_t *out_vars, uint32_t *out_n, double *out_converge);
/* The activation step of Adapt alone (adaptive.go:131-153): the next inactive group gets a fresh clone of the
 * base model with var_idx collapsed (-1 = random), a start state and NewChain's burn-in of 2 updates.  With
 * gr_merge_local, gr_window_halves, gr_chain_convergence and gr_adapt_select this lets a host run Adapt over
 * several devices or processes (grample_amd/dist.py: adapt_plan / adapt_distributed). */
int gr_activate_group(gr_sampler *s, int32_t var_idx, int32_t *out_var);

/* Philox4x32-10 block function on the host (same code the device runs);
 * exported so the known-answer vectors can be checked without a GPU. */
void gr_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* GRAMPLE_HIP_H */
