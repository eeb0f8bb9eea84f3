// gfx950 (MI355X / CDNA4) kernels of the Gibbs hot path.
//
// Execution shape: one chain per lane, 64 chains per wavefront.  Every lane of
// a wavefront updates the SAME variable at each step (the scan sequence is a
// per-wavefront Philox stream), so
//   * all model metadata (which factors, which neighbours, strides, table
//     offsets) is wave-uniform -> scalar loads + SALU, no VALU work;
//   * chain state lives in LDS as state[var][lane] (one byte per variable and
//     chain): every state access is a conflict-free 64-byte row;
//   * table rows are per-lane LDS gathers with at most prod(Card of the other
//     scope vars) distinct addresses per wave (broadcast on equal addresses);
//   * the histogram update for binary variables is one ballot + popcount.
// There is no MFMA here: this is gather / reduce work, not a contraction.
//
// Reference behaviour implemented (file:line in CraigKelly/grample):
//   select variable      sampler/sampler.go:135-174  (UniformSampler.VarSample)
//   gather + log-sum     sampler/gibbs-simple.go:175-220, model/function.go:146-202
//   stabilise/exp/floor  sampler/gibbs-simple.go:227-258
//   categorical draw     sampler/sampler.go:90-130   (WeightedSample)
//   write back + count   sampler/gibbs-simple.go:267, sampler/chain.go:221-246
//   start state          sampler/gibbs-simple.go:84-112
//   collapse             sampler/gibbs-collapsed.go:206-272
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"
#include "philox.h"
#include "wave_prio.h"
#include "cond.h"

// Compiled three times (Makefile): GR_KSWEEP_PART 0 = everything, with the instantiations of k_sweep for Card <= 4 only;
// parts 1 and 2 = the instantiations for Card <= 16 and Card <= 64 (72 of them in one translation unit took two minutes).
#ifndef GR_KSWEEP_PART
#define GR_KSWEEP_PART 0
#endif

namespace gr {

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// Programs, program offsets and the free list are read-only for a launch.  k_sweep reads them through the constant address
// space: a uniform address is then always a scalar load.  Without it the instantiations that store to global memory inside
// the step loop (STG: chain state; LOG: the draw log) lose every scalar load -- those stores could alias the programs as far
// as the compiler knows -- and walk the programs with vector loads instead (84 vs 13 in the STG variant).
#define GR_CONST_AS __attribute__((address_space(4)))
typedef const GR_CONST_AS uint32_t *cu32p;
typedef uint32_t u32x4c __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cu32p as_const(const uint32_t *p) { return (cu32p)(uintptr_t)p; }

// Summed log potentials of variable `pg` (its direct program) for every
// candidate value, reading neighbour values through `get(var)`.
// `tab` = this variant's view-table pool (LDS or global).
template <int MAXC, class Getter, class PG>
__device__ __forceinline__ void log_weights(PG pg, const float *tab, Getter get, float (&w)[MAXC],
                                            const uint32_t p0, const uint32_t p1, const uint32_t p2, bool &conflict) {
    const uint32_t h = pg[0];
    const uint32_t rowlen = (h >> 8) & 0xffu;
    const uint32_t nf = h >> 16;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) w[i] = 0.f;
    PG q = pg + 2;
    for (uint32_t f = 0; f < nf; ++f) { // factors in Model.Funcs order (gibbs-simple.go:186)
        uint32_t o = q[0];
        const uint32_t no = q[1];
        q += 2;
        for (uint32_t j = 0; j < no; ++j) {
            const uint32_t u = q[0], st = q[1];
            q += 2;
            conflict |= (u == p0) | (u == p1) | (u == p2);
            o += st * get(u);
        }
        const float *row = tab + o;
        if constexpr (MAXC <= 2) {
            const float2 r = *reinterpret_cast<const float2 *>(row);
            w[0] += r.x;
            w[1] += r.y;
        } else {
#pragma unroll
            for (int c = 0; c < MAXC; c += 4)
                if ((uint32_t)c < rowlen) {
                    const float4 r = *reinterpret_cast<const float4 *>(row + c);
                    w[c] += r.x; w[c + 1] += r.y; w[c + 2] += r.z; w[c + 3] += r.w;
                }
        }
    }
}

// Pairwise models (arity <= 2), fixed-stride program: every factor's entry is an independent
// scalar load, and the table rows of up to GF factors are all requested before the first one is
// consumed -- one table round trip per group instead of one per factor (the rows come from L2 when
// the view tables do not fit LDS: ObjectDetection_13).  Accumulation stays in Model.Funcs order, so
// the sums are bit-identical to log_weights().
template <int MAXC, int GF, bool FULLROW, class Getter>
__device__ __forceinline__ void log_weights_pw_impl(cu32p pg, const float *tab, Getter get, float (&w)[MAXC],
                                                    const uint32_t p0, const uint32_t p1, const uint32_t p2, bool &conflict) {
    const uint32_t h = pg[0];
    const uint32_t rowlen = (h >> 8) & 0xffu;
    const uint32_t nf = h >> 16;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) w[i] = 0.f;
    const GR_CONST_AS u32x4c *ent = reinterpret_cast<const GR_CONST_AS u32x4c *>(pg + 4);
    constexpr int NV = MAXC <= 2 ? 1 : MAXC / 4;
    for (uint32_t f0 = 0; f0 < nf; f0 += GF) {
        // Slots past the last factor re-read the last entry (clamped index) instead of being branched around: the GF
        // entry loads, then the GF neighbour reads, then the GF x NV row loads are each issued back to back with one
        // wait per stage.  (Guarding every slot made the compiler wait for each scalar entry load on its own -- a scalar
        // cache round trip per factor with a lone wavefront per SIMD.)  Only the accumulation is guarded.
        uint4 e[GF];
#pragma unroll
        for (int i = 0; i < GF; ++i) {
            const u32x4c t = ent[min(f0 + (uint32_t)i, nf - 1u)];
            e[i] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        const float *row[GF];
#pragma unroll
        for (int i = 0; i < GF; ++i) {
            conflict |= (e[i].y == p0) | (e[i].y == p1) | (e[i].y == p2);
            row[i] = tab + (e[i].x + e[i].z * get(e[i].y));
        }
        float4 r[GF][NV];
#pragma unroll
        for (int i = 0; i < GF; ++i) {
            if constexpr (MAXC <= 2) {
                const float2 t = *reinterpret_cast<const float2 *>(row[i]);
                r[i][0] = make_float4(t.x, t.y, 0.f, 0.f);
            } else {
#pragma unroll
                for (int c = 0; c < NV; ++c)
                    if (FULLROW || (uint32_t)(4 * c) < rowlen) r[i][c] = *reinterpret_cast<const float4 *>(row[i] + 4 * c);
            }
        }
#pragma unroll
        for (int i = 0; i < GF; ++i)
            if (f0 + i < nf) {
                if constexpr (MAXC <= 2) {
                    w[0] += r[i][0].x;
                    w[1] += r[i][0].y;
                } else {
#pragma unroll
                    for (int c = 0; c < NV; ++c)
                        if (FULLROW || (uint32_t)(4 * c) < rowlen) {
                            w[4 * c] += r[i][c].x; w[4 * c + 1] += r[i][c].y; w[4 * c + 2] += r[i][c].z; w[4 * c + 3] += r[i][c].w;
                        }
                }
            }
    }
}
// FULLROW = the variable's rows use all MAXC values: one uniform test up front instead of a scalar guard per row
// chunk and factor (those guards were most of the kernel's 313 scalar instructions per ObjectDetection_13 step).
template <int MAXC, int GF, class Getter>
__device__ __forceinline__ void log_weights_pw(cu32p pg, const float *tab, Getter get, float (&w)[MAXC],
                                               const uint32_t p0, const uint32_t p1, const uint32_t p2, bool &conflict) {
    if (MAXC > 2 && ((pg[0] >> 8) & 0xffu) == (uint32_t)MAXC) log_weights_pw_impl<MAXC, GF, true>(pg, tab, get, w, p0, p1, p2, conflict);
    else log_weights_pw_impl<MAXC, GF, false>(pg, tab, get, w, p0, p1, p2, conflict);
}

// ------------------------------------------------------------------- sweep
// K1: n_steps single-variable updates on every chain of the block.
//   CPT = false: conditionals computed from the view tables at every update;
//   CPT = true : conditionals memoised per (variable, blanket configuration)
//                by k_build_cpt -- one table word per binary update.
//   TAB_LDS    : tables staged in LDS (else read from global through L1/L2).
// PC (strict mode, GR_FLAG_SCAN_PER_CHAIN): every chain draws its own variable at every step, exactly like
// UniformSampler.VarSample per chain (sampler.go:135-174); nothing is wave-uniform any more (divergent
// program walks, per-lane histogram atomics).  Slow; exists so that the reference's exact process law is
// available and the wave-uniform scan can be checked against it.
// LOG: write the draw log of gr_advance_chain (only instantiated for the direct back-end, tables from global memory:
// every back-end draws the same values, and keeping the log code out of the other instantiations keeps them compact --
// with it inlined everywhere ObjectDetection_13 ran 17 % slower)
// STG: the chain state stays in global memory as it is laid out there -- state[var][chain], one coalesced 64-byte row per
// wavefront and variable -- instead of being staged in LDS: for models whose state does not fit LDS (n_vars * 64 B per
// wavefront > 160 KiB) and as GR_FLAG_STATE_GLOBAL for cross-checks.  Same arithmetic, same order: bit-identical results.
// SCAN 0: one variable per 64-chain group and step (shared scan); 1 = PC, every chain its own variable; 2 = SYS,
// GR_FLAG_SCAN_SYSTEMATIC: update t resamples variable sys_order[t mod n] with the draw philox_sys_draw(t / n, t mod n, chain) --
// the cross-check of k_sweep_chroma and its head / tail launches (instantiated for the direct back-end only).
template <int MAXC, bool CPT, bool TAB_LDS, bool PW, int SCAN, bool LOG = false, bool STG = false>
__global__ void __launch_bounds__(256) k_sweep(const SweepParams p) {
    constexpr bool PC = SCAN == 1, SYS = SCAN == 2;
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t T = blockDim.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t chain = p.chain0 + blockIdx.x * T + tid; // local chain index
    const bool active = chain < p.n_chains;
    const uint32_t variant = (p.chain0 + blockIdx.x * T) / p.chains_per_variant; // blocks never straddle groups
    const VariantDesc vd = p.variants[variant];

    unsigned char *st = lds + tid;                                    // st[var * T]
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + p.lds_hist_off) + wave * p.H; // wave-private
    const float *tab_f;
    const uint32_t *tab_u;
    if constexpr (TAB_LDS) {
        // stage this group's tables: 16 B per lane per iteration, coalesced
        const uint32_t n4 = (CPT ? (vd.cpt_len + 3u) : vd.views_len) >> 2;
        const uint4 *src = CPT ? reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base)
                               : reinterpret_cast<const uint4 *>(p.views + vd.views_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.lds_tab_off);
        for (uint32_t i = tid; i < n4; i += T) dst[i] = src[i];
        tab_f = reinterpret_cast<const float *>(lds + p.lds_tab_off);
        tab_u = reinterpret_cast<const uint32_t *>(lds + p.lds_tab_off);
    } else {
        tab_f = p.views + vd.views_base;
        tab_u = p.cpt + vd.cpt_base;
    }
    if constexpr (!STG)
        for (uint32_t v = 0; v < p.n_vars; ++v) st[v * T] = active ? p.state[(size_t)v * p.pitch + chain] : (unsigned char)0;
    unsigned char *gst = p.state + chain; // STG: this chain's column of the global state (inactive lanes stay inside the row pitch)
    for (uint32_t i = lane; i < p.H; i += 64) hist[i] = 0;
    __syncthreads();

    const unsigned long long amask = __ballot(active);
    const uint32_t nact = (uint32_t)__popcll(amask);
    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint32_t wave_gid = uni((uint32_t)(gchain >> 6));
    const uint64_t step0 = ((uint64_t)p.step0_hi << 32) | p.step0_lo;
    const uint64_t step1 = step0 + p.n_steps;
    const uint32_t *progs = CPT ? p.cprogs + vd.cprog_base : (PW ? p.pprogs + vd.pprog_base : p.progs + vd.prog_base);
    const uint32_t *poff = CPT ? p.cprog_off + vd.cprogoff_base : (PW ? p.pprog_off + vd.progoff_base : p.prog_off + vd.progoff_base);
    const uint32_t *freev = p.free_vars + vd.free_base;
    const bool seq = p.sequential != 0;
    const bool counting = p.counting != 0;

    auto get = [&](uint32_t u) -> uint32_t {
        if constexpr (STG) return (uint32_t)gst[(size_t)u * p.pitch];
        else return (uint32_t)st[u * T];
    };

    // new value of variable v (program pg) given the current state and the draw u
    auto sample = [&](cu32p pg, const uint32_t u, const uint32_t p0, const uint32_t p1, const uint32_t p2,
                      bool &conflict) -> uint32_t {
        const int C = (int)(pg[0] & 0xffu);
        if constexpr (CPT) {
            const uint32_t nn = (pg[0] >> 8) & 0xffu;
            uint32_t cfg = 0;
            for (uint32_t j = 0; j < nn; ++j) {
                const uint32_t nb = pg[3 + 2 * j], mult = pg[4 + 2 * j];
                conflict |= (nb == p0) | (nb == p1) | (nb == p2);
                cfg += mult * get(nb);
            }
            const uint32_t *row = tab_u + pg[2] + cfg * (uint32_t)(C - 1);
            uint32_t val = 0;
#pragma unroll
            for (int i = 0; i < MAXC - 1; ++i)
                if (i < C - 1) val += (u >= row[i]) ? 1u : 0u;
            return val;
        } else {
            float w[MAXC];
            if constexpr (PW) log_weights_pw<MAXC, (MAXC <= 16 ? 8 : (MAXC <= 32 ? 4 : 2))>(pg, tab_f, get, w, p0, p1, p2, conflict);
            else log_weights<MAXC>(pg, tab_f, get, w, p0, p1, p2, conflict);
            uint32_t cum[MAXC > 1 ? MAXC - 1 : 1], tot;
            cumulative<MAXC>(w, C, cum, tot);
            return pick<MAXC>(cum, tot, C, u);
        }
    };

    // g.last[v] = val (gibbs-simple.go:267) and Marginal[val] += 1 (chain.go:230-236)
    // (PC: the log holds one VALUE byte per chain and step, 64 bytes per group and step -- a wavefront's store is one line)
    unsigned char *logrow = LOG && p.log ? (PC ? p.log + (size_t)(chain >> 6) * p.log_pitch + (size_t)p.log_off * 64u + lane
                                               : p.log + (size_t)(chain >> 6) * p.log_rows * p.log_pitch + p.log_off) : nullptr;
    // LOG: the group's (PC: the chain's) own stopping time inside the window (steps of this launch it still executes)
    uint32_t lim = 0xffffffffu;
    if constexpr (LOG)
        if (p.limit) {
            const uint32_t t = PC ? (active ? p.limit[chain] : 0u) : p.limit[chain >> 6];
            lim = t > p.win_off ? t - p.win_off : 0u;
        }
    auto commit = [&](const uint32_t v, cu32p pg, const uint32_t val, const uint32_t sidx) {
        if (LOG && sidx >= lim) return; // the group has finished its AdvanceChain: state, counts and log stay as they are
        if constexpr (STG) { if (active) gst[(size_t)v * p.pitch] = (unsigned char)val; }
        else st[v * T] = (unsigned char)val;
        if (counting) {
            const uint32_t C = pg[0] & 0xffu, ho = pg[1];
            if (!PC && C == 2) {
                const uint32_t ones = (uint32_t)__popcll(__ballot(active && val != 0));
                if (lane == 0) {
                    hist[ho] += nact - ones;
                    hist[ho + 1] += ones;
                    if (LOG && logrow) {
                        if (p.log_vals == 1) logrow[sidx] = (unsigned char)ones;
                        else { logrow[sidx * p.log_vals] = (unsigned char)(nact - ones); logrow[sidx * p.log_vals + 1] = (unsigned char)ones; }
                    }
                }
            } else {
                if (active) atomicAdd(&hist[ho + val], 1u);
                if (LOG && PC && logrow && active) logrow[(size_t)sidx * 64u] = (unsigned char)val;
                if (LOG && !PC && logrow) // one count per value (log_vals = max Card)
                    for (uint32_t c = 0; c < C; ++c) {
                        const uint32_t n = (uint32_t)__popcll(__ballot(active && val == c));
                        if (lane == 0) logrow[sidx * p.log_vals + c] = (unsigned char)n;
                    }
            }
        }
    };

    // SYS: position in the sweep order and sweep number of the first step of the current quad (uniform), kept by increments
    uint32_t sys_pos = 0;
    uint64_t sys_sweep = 0;
    if constexpr (SYS) {
        const uint64_t s_first = (step0 >> 2) << 2;
        sys_sweep = s_first / vd.n_free;
        sys_pos = (uint32_t)(s_first - sys_sweep * vd.n_free);
    }
    const PrioRotor prio; // (wave_prio.h: the blocks of a CU progress as equals)
    for (uint64_t q = step0 >> 2; (q << 2) < step1; ++q) {
        const uint32_t q_lo = (uint32_t)q, q_hi = (uint32_t)(q >> 32);
        prio.tick(q_lo, p.prio_shift);
        uint32_t su[4] = {0, 0, 0, 0}, du[4];
        uint32_t v[4];
        cu32p pg[4];
        bool valid[4];
        if constexpr (SYS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t s = (q << 2) + j;
                valid[j] = (s >= step0) && (s < step1);
                v[j] = as_const(p.sys_order + vd.free_base)[sys_pos];
                pg[j] = as_const(progs) + as_const(poff)[v[j]];
                const u32x4 dr = philox_sys_draw(sys_sweep, sys_pos, gchain, p.seed_lo, p.seed_hi);
                const uint32_t w = (uint32_t)sys_sweep & 3u;
                du[j] = w == 0 ? dr.x : w == 1 ? dr.y : w == 2 ? dr.z : dr.w;
                if (++sys_pos == vd.n_free) { sys_pos = 0; ++sys_sweep; }
            }
        } else {
        // which variable each of the next 4 steps updates: wave-uniform stream -> SALU
        const u32x4 sc = PC ? philox4x32_10(q_lo, q_hi, (uint32_t)gchain, DOM_SCAN_CHAIN + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi)
                            : philox4x32_10<true>(q_lo, q_hi, wave_gid, DOM_SCAN, p.seed_lo, p.seed_hi);
        // the 4 categorical draws of this chain
        const u32x4 dr = philox4x32_10(q_lo, q_hi, (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
        su[0] = sc.x; su[1] = sc.y; su[2] = sc.z; su[3] = sc.w;
        du[0] = dr.x; du[1] = dr.y; du[2] = dr.z; du[3] = dr.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t s = (q << 2) + j;
            valid[j] = (s >= step0) && (s < step1);
            // VarSample: uniform over the selectable variables
            if constexpr (PC) v[j] = as_const(freev)[mul_hi_u32(su[j], vd.n_free)];
            else v[j] = as_const(freev)[uni(mul_hi_u32(uni(su[j]), vd.n_free))];
            pg[j] = as_const(progs) + as_const(poff)[v[j]];
        }
        }
        // Phase A: conditionals of all 4 steps from the current state.  Step j's
        // result is exact unless an earlier step of the quad updates one of its
        // neighbours (then `conf[j]`); updates of non-adjacent variables commute.
        uint32_t nv[4] = {0, 0, 0, 0};
        bool conf[4] = {true, true, true, true};
        if (!seq) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (valid[j]) {
                    bool c = false;
                    nv[j] = sample(pg[j], du[j], j > 0 ? v[0] : 0xffffffffu, j > 1 ? v[1] : 0xffffffffu,
                                   j > 2 ? v[2] : 0xffffffffu, c);
                    conf[j] = c;
                }
        }
        // Phase B: commit in order; recompute the (rare) steps that did not commute.
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (valid[j]) {
                if (conf[j]) {
                    bool c = false;
                    nv[j] = sample(pg[j], du[j], 0xffffffffu, 0xffffffffu, 0xffffffffu, c);
                }
                commit(v[j], pg[j], nv[j], (uint32_t)(((q << 2) + j) - step0));
            }
    }

    // flush: state back to HBM (coalesced rows), wave-private histograms -> global counts
    if constexpr (!STG)
        for (uint32_t vv = 0; vv < p.n_vars; ++vv)
            if (active) p.state[(size_t)vv * p.pitch + chain] = st[vv * T];
    if (counting) {
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = lane; i < p.H; i += 64) {
            const uint32_t c = hist[i];
            if (c) atomicAdd(&gc[i], (unsigned long long)c);
        }
    }
}

#if GR_KSWEEP_PART == 0
// K3: start state (gibbs-simple.go:84-112 / 132-142): evidence value, else
// uniform over [0, Card).  `epoch` separates the re-draw after a collapse.
__global__ void k_init_state(unsigned char *state, const int32_t *card, const int32_t *fixed, uint32_t n_vars, uint32_t pitch,
                             uint32_t chain_begin, uint32_t chain_end, uint64_t chain_offset, uint32_t seed_lo, uint32_t seed_hi,
                             uint32_t epoch) {
    const uint32_t chain = chain_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= chain_end) return;
    const uint64_t g = chain_offset + chain;
    for (uint32_t v0 = 0; v0 < n_vars; v0 += 4) {
        const u32x4 r = philox4x32_10(v0 >> 2, epoch, (uint32_t)g, DOM_INIT + ((uint32_t)(g >> 32) << 4), seed_lo, seed_hi);
        const uint32_t rr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t v = v0 + j;
            if (v < n_vars) {
                const int f = fixed[v];
                state[(size_t)v * pitch + chain] = (unsigned char)(f >= 0 ? (uint32_t)f : mul_hi_u32(rr[j], (uint32_t)card[v]));
            }
        }
    }
}

// K5: memoise the conditional of every (variable, blanket configuration):
// thread = one configuration of blockIdx.y's blanket; writes Card-1 thresholds.
template <int MAXC>
__global__ void k_build_cpt(const VariantDesc vd, const uint32_t *progs_all, const uint32_t *prog_off_all, const float *views_all,
                            const uint32_t *cprogs_all, const uint32_t *cprog_off_all, uint32_t *cpt_all, const int32_t *card,
                            const int32_t *fixed) {
    const uint32_t v = blockIdx.y;
    const uint32_t *cp = cprogs_all + vd.cprog_base + cprog_off_all[vd.cprogoff_base + v];
    const uint32_t *pg = progs_all + vd.prog_base + prog_off_all[vd.progoff_base + v];
    if ((pg[0] >> 16) == 0) return; // not selectable
    const int C = (int)(cp[0] & 0xffu);
    const uint32_t nn = (cp[0] >> 8) & 0xffu;
    uint32_t ncfg = 1;
    for (uint32_t j = 0; j < nn; ++j) ncfg *= (uint32_t)card[cp[3 + 2 * j]];
    const uint32_t cfg = blockIdx.x * blockDim.x + threadIdx.x;
    if (cfg >= ncfg) return;
    auto get = [&](uint32_t u) -> uint32_t {
        if (fixed[u] >= 0) return (uint32_t)fixed[u];
        for (uint32_t j = 0; j < nn; ++j)
            if (cp[3 + 2 * j] == u) return (cfg / cp[4 + 2 * j]) % (uint32_t)card[u];
        return 0u;
    };
    float w[MAXC];
    bool c = false;
    log_weights<MAXC>(pg, views_all + vd.views_base, get, w, 0xffffffffu, 0xffffffffu, 0xffffffffu, c);
    uint32_t cum[MAXC > 1 ? MAXC - 1 : 1], tot;
    cumulative<MAXC>(w, C, cum, tot);
    uint32_t *row = cpt_all + vd.cpt_base + cp[2] + cfg * (uint32_t)(C - 1);
#pragma unroll
    for (int i = 0; i < MAXC - 1; ++i)
        if (i < C - 1) row[i] = threshold_u32(cum[i], tot); // u >= row[i] is the direct path's mulhi(u, tot) >= cum[i]
}

// Probe: the conditional the sampling kernels use, for one variable and one
// explicit state (gr_conditional).  One thread.
template <int MAXC>
__global__ void k_conditional(const VariantDesc vd, const uint32_t *progs_all, const uint32_t *prog_off_all, const float *views_all,
                              uint32_t v, const int32_t *state, double *probs) {
    const uint32_t *pg = progs_all + vd.prog_base + prog_off_all[vd.progoff_base + v];
    const int C = (int)(pg[0] & 0xffu);
    auto get = [&](uint32_t u) -> uint32_t { return (uint32_t)state[u]; };
    float w[MAXC];
    bool c = false;
    log_weights<MAXC>(pg, views_all + vd.views_base, get, w, 0xffffffffu, 0xffffffffu, 0xffffffffu, c);
    uint32_t cum[MAXC > 1 ? MAXC - 1 : 1], tot;
    cumulative<MAXC>(w, C, cum, tot);
    double prev = 0.0;
    for (int i = 0; i < C; ++i) {
        const double cur = (i < C - 1) ? (double)threshold_u32(cum[i < MAXC - 1 ? i : 0], tot) : 4294967296.0;
        probs[i] = (cur - prev) / 4294967296.0;
        prev = cur;
    }
}

// K2: Collapse (gibbs-collapsed.go:206-272).  Thread = one entry e of the new
// factor (a configuration of the blanket minus the variable).  For every value
// of the collapsed variable: p = exp(sum of its factors' log potentials);
// new[e] = sum_val p; partial marginals are reduced per block (fixed order).
// Entries that contradict evidence are never enumerated by the reference's
// VariableIter(honorFixed) and stay 0.

__global__ void __launch_bounds__(256) k_collapse(const CollapseArgs a) {
    __shared__ double red[256];
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t digits[32];
    bool enumerated = e < a.n_entries;
    if (enumerated) {
        uint32_t r = e;
        for (int k = (int)a.n_new - 1; k >= 0; --k) { // last variable fastest
            digits[k] = r % a.new_card[k];
            r /= a.new_card[k];
            if (a.new_fixed[k] >= 0 && (uint32_t)a.new_fixed[k] != digits[k]) enumerated = false;
        }
    }
    double entry = 0.0;
    for (uint32_t val = 0; val < a.card_v; ++val) {
        double p = 0.0;
        if (enumerated) {
            double s = 0.0;
            for (uint32_t f = 0; f < a.n_fac; ++f) {
                uint32_t idx = 0;
                for (uint32_t j = a.fac_off[f]; j < a.fac_off[f + 1]; ++j)
                    idx += a.fac_stride[j] * (a.fac_pos[j] < 0 ? val : digits[a.fac_pos[j]]);
                s += a.tables[a.fac_tab[f] + idx];
            }
            p = exp(s);
            entry += p;
        }
        red[threadIdx.x] = p;
        __syncthreads();
        for (uint32_t w = 128; w > 0; w >>= 1) {
            if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.out_partial[(size_t)blockIdx.x * a.card_v + val] = red[0];
        __syncthreads();
    }
    if (e < a.n_entries) {
        double t = entry;
        if (t < 1e-6) t += 1e-6;
        a.out_table[e] = log(t);
    }
}

// ------------------------------------------------------------ scan-stream kernels
// The variable a 64-chain group updates at step s (same arithmetic as k_sweep / k_scan_plan).
// One wavefront replays the scan stream of one 64-chain group: lane l draws the Philox block of quad q0 + l.
struct ScanReplay {
    const uint32_t *freev;
    uint32_t n_free, gid, seed_lo, seed_hi;
    __device__ ScanReplay(const ScanArgs &a, uint32_t group) {
        const VariantDesc vd = a.variants[(group * 64u) / a.chains_per_variant];
        freev = a.free_vars + vd.free_base;
        n_free = vd.n_free;
        const uint64_t gchain0 = (((uint64_t)a.chain_offset_hi << 32) | a.chain_offset_lo) + (uint64_t)group * 64u;
        gid = (uint32_t)(gchain0 >> 6);
        seed_lo = a.seed_lo; seed_hi = a.seed_hi;
    }
    // the four variables of quad q (steps 4q .. 4q+3)
    __device__ void quad(uint64_t q, uint32_t v[4]) const {
        const u32x4 w = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), gid, DOM_SCAN, seed_lo, seed_hi);
        v[0] = freev[mul_hi_u32(w.x, n_free)]; v[1] = freev[mul_hi_u32(w.y, n_free)];
        v[2] = freev[mul_hi_u32(w.z, n_free)]; v[3] = freev[mul_hi_u32(w.w, n_free)];
    }
};

// The variables a launch of the value-parallel kernel (k_sweep_pw4) will update, evaluated once at full occupancy instead of
// by every one of its wavefronts on the scalar unit (~100 scalar instructions per quad and wavefront, and the Philox key
// schedule's registers): out[(group * pitch_quads + quad) * 4 + j] = variable of step 4 * (q0 + quad) + j.
__global__ void __launch_bounds__(256) k_scan_vars(const ScanArgs a, uint32_t n_quads, uint32_t pitch_quads, unsigned short *out) {
    const uint32_t qi = blockIdx.x * 256u + threadIdx.x, g = blockIdx.y;
    if (qi >= n_quads) return;
    const ScanReplay sr(a, g);
    const uint64_t q = ((((uint64_t)a.step0_hi << 32) | a.step0_lo) >> 2) + qi;
    uint32_t v[4];
    if (a.sys_order) { // systematic scan: step t updates sys_order[t mod n]
        const uint32_t *ord = a.sys_order + (sr.freev - a.free_vars);
        uint32_t pos = (uint32_t)((q << 2) % sr.n_free);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = ord[pos];
            if (++pos == sr.n_free) pos = 0;
        }
    } else
    sr.quad(q, v);
    reinterpret_cast<uint2 *>(out)[(size_t)g * pitch_quads + qi] = make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
}

// AdvanceChain's stopping rule (chain.go:184-215): batches of `batch` (= 2 N) updates until every selectable
// variable of the group has been selected cw + 1 more times.  out[group] = number of batches.
// One wavefront per group; selection counters in LDS (n_vars words).
__global__ __launch_bounds__(256) void k_stop_time(const ScanArgs a, uint32_t cw, uint32_t batch, uint32_t max_batches, uint32_t *out) {
    extern __shared__ uint32_t sm_scan[];
    uint32_t *cnt = sm_scan;                    // [n_vars] selections so far
    unsigned short *fv = reinterpret_cast<unsigned short *>(cnt + a.n_vars); // the group's free list (the replay indexes it with a Philox word)
    const uint32_t g = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    const ScanReplay sr(a, g);
    for (uint32_t v = tid; v < a.n_vars; v += T) cnt[v] = 0;
    for (uint32_t i = tid; i < sr.n_free; i += T) fv[i] = (unsigned short)sr.freev[i];
    __syncthreads();
    // count the selections of steps [s, e): 4 wavefronts, one quad of the scan stream per lane and trip
    auto count_range = [&](const uint64_t s, const uint64_t e) {
        for (uint64_t q = (s >> 2) + tid; q <= ((e - 1u) >> 2); q += T) {
            const u32x4 w = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), sr.gid, DOM_SCAN, sr.seed_lo, sr.seed_hi);
            const uint32_t v[4] = {fv[mul_hi_u32(w.x, sr.n_free)], fv[mul_hi_u32(w.y, sr.n_free)], fv[mul_hi_u32(w.z, sr.n_free)],
                                   fv[mul_hi_u32(w.w, sr.n_free)]};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint64_t st = q * 4u + j;
                if (st >= s && st < e) atomicAdd(&cnt[v[j]], 1u);
            }
        }
    };
    // (cw + 1) * n_free selections are needed in all: no batch before that many steps can be the last one, so the batches up
    // to there are counted in one go (no barrier, no check per batch)
    const uint64_t need_steps = (uint64_t)(cw + 1u) * sr.n_free;
    uint64_t s = ((uint64_t)a.step0_hi << 32) | a.step0_lo;
    uint32_t b = 0;
    bool done = sr.n_free == 0;
    if (!done && batch) {
        const uint64_t bulk = std::min<uint64_t>(need_steps / batch, max_batches);
        if (bulk > 1u) {
            b = (uint32_t)(bulk - 1u); // (the batch that reaches need_steps is checked like every later one)
            count_range(s, s + (uint64_t)b * batch);
            s += (uint64_t)b * batch;
            __syncthreads();
        }
    }
    while (b < max_batches && !done) {
        const uint64_t e = s + batch;
        count_range(s, e);
        s = e;
        b++;
        __syncthreads();
        if ((uint64_t)b * batch >= need_steps) {
            bool need = false;
            for (uint32_t i = tid; i < sr.n_free; i += T) need = need || cnt[fv[i]] < cw + 1u;
            done = __syncthreads_or(need ? 1 : 0) == 0;
        }
    }
    if (tid == 0) out[g] = done ? b : 0xffffffffu;
}

// ChainHistory's two halves (buffer/circular.go:54-84, chain.go:270-277), rebuilt from the draw log: walking the
// window backwards, the newest `half` selections of a variable are its second half, the `half` before them its
// first half.  hist[g][h][H] = how many chains drew each value in half h; status[g] = variables whose ring is not full.
// The log holds per step either the number of ones (log_vals == 1, all-binary models, 64 chains per group) or one
// count per value (log_vals = max Card).
// One wavefront per group, 256 steps at a time: (0) every lane replays one quad of the scan stream and stages
// (variable, counts) per step in LDS; (1) every lane lists, newest first, the steps whose variable it owns
// (v mod 64 == lane); (2) the lanes walk their lists side by side -- a variable is only ever touched by its owner,
// so the per-variable order of the serial walk is kept without atomics.
// A long window is cut into `n_seg` SEGMENTS of `seg_chunks` chunks (counted from the end of the group's window), one block
// each: the one wavefront per group this kernel used to be leaves most of the chip idle and pays every latency in full.  A
// segment needs to know how often each variable was selected in the NEWER segments -- k_ring_count, a first pass over the scan
// stream alone -- and adds its two histograms to the group's with atomics.  hist must be zero on entry.
__global__ __launch_bounds__(64) void k_ring_count(const ScanArgs a, uint32_t n_steps, const uint32_t *limit, uint32_t n_seg, uint32_t seg_chunks,
                                                    uint32_t *seg_cnt /*[n_groups][n_seg][n_vars]*/) {
    extern __shared__ uint32_t sm_scan[];
    const uint32_t n = a.n_vars, g = blockIdx.x, seg = blockIdx.y, lane = threadIdx.x;
    uint32_t *cnt = sm_scan;
    const ScanReplay sr(a, g);
    for (uint32_t v = lane; v < n; v += 64u) cnt[v] = 0;
    __syncthreads();
    const int64_t step0 = (int64_t)(((uint64_t)a.step0_hi << 32) | a.step0_lo);
    const int64_t E = (int64_t)(limit ? limit[g] : n_steps);
    if (E > 0) {
        const int64_t qhi0 = (step0 + E - 1) >> 2;
        for (uint32_t c = seg * seg_chunks; c < (seg + 1u) * seg_chunks; ++c) {
            const int64_t q = qhi0 - 64 * (int64_t)c - 63 + (int64_t)lane;
            if ((qhi0 - 64 * (int64_t)c) * 4 + 3 - step0 < 0) break; // the chunk lies before the window
            uint32_t v[4] = {0, 0, 0, 0};
            if (q >= 0) sr.quad((uint64_t)q, v);
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const int64_t k = q * 4 + j - step0;
                if (k >= 0 && k < E) atomicAdd(&cnt[v[j]], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t v = lane; v < n; v += 64u) seg_cnt[((size_t)g * n_seg + seg) * n + v] = cnt[v];
}

__global__ __launch_bounds__(64) void k_ring_halves(const ScanArgs a, uint32_t n_steps, uint32_t half, const unsigned char *log, uint32_t log_pitch,
                                                     uint32_t log_rows, uint32_t log_vals, const uint32_t *limit /*[n_groups] or null*/, uint32_t *hist /*[n_groups][2][H]*/,
                                                     uint32_t *status, uint32_t n_seg, uint32_t seg_chunks, const uint32_t *seg_cnt) {
    extern __shared__ uint32_t sm_scan[];
    const uint32_t n = a.n_vars, H = a.H, g = blockIdx.x, seg = blockIdx.y, lane = threadIdx.x;
    uint32_t *sn = sm_scan, *o1 = sn + n, *o2 = o1 + H, *hf = o2 + H, *kc = hf + n + 1u, *rem = kc + n;
    unsigned short *cv = reinterpret_cast<unsigned short *>(rem + 1), *fv = cv + 256; // fv[n]: the group's free list
    unsigned char *cl = reinterpret_cast<unsigned char *>(fv + ((n + 1u) & ~1u)); // [256][log_vals]
    const ScanReplay sr(a, g);
    // (the free list in LDS too: the scan replay indexes it with a Philox word, from global memory one more dependent round
    // trip per chunk)
    for (uint32_t i = lane; i < sr.n_free; i += 64u) fv[i] = (unsigned short)sr.freev[i];
    // (histogram offsets in LDS: the walk below looks one up per visited step, right between two LDS accesses -- from global
    // memory that dependent round trip was most of this kernel's time)
    for (uint32_t v = lane; v <= n; v += 64u) hf[v] = a.hoff[v];
    for (uint32_t v = lane; v < n; v += 64u) { // selections in the newer segments
        uint32_t c = 0;
        for (uint32_t t = 0; t < seg; ++t) c += seg_cnt[((size_t)g * n_seg + t) * n + v];
        sn[v] = c;
        kc[v] = 0;
    }
    for (uint32_t i = lane; i < H; i += 64u) { o1[i] = 0; o2[i] = 0; }
    if (lane == 0) *rem = 0;
    __syncthreads();
    { // rings not yet full when this segment starts
        uint32_t open = 0;
        for (uint32_t i = lane; i < sr.n_free; i += 64u) open += sn[fv[i]] < 2u * half ? 1u : 0u;
        if (open) atomicAdd(rem, open);
    }
    if (seg == 0 && n_seg > 1u) { // the window's verdict needs every segment's counts: rings that never fill
        uint32_t open = 0;
        for (uint32_t i = lane; i < sr.n_free; i += 64u) {
            uint32_t c = 0;
            for (uint32_t t = 0; t < n_seg; ++t) c += seg_cnt[((size_t)g * n_seg + t) * n + fv[i]];
            open += c < 2u * half ? 1u : 0u;
        }
        if (open && half) atomicAdd(&status[g], open); // (status is zero on entry)
    }
    __syncthreads();
    const int64_t step0 = (int64_t)(((uint64_t)a.step0_hi << 32) | a.step0_lo);
    const unsigned char *lg = log + (size_t)g * log_rows * log_pitch;
    const int64_t E = half ? (int64_t)(limit ? limit[g] : n_steps) : 0;
    // this segment: chunks [seg * seg_chunks, (seg + 1) * seg_chunks) counted from the end of the window; e = steps [0, e) still
    // to be visited when it starts, e_end = where the next (older) segment takes over
    int64_t e = E, e_end = 0;
    if (E > 0) {
        const int64_t qhi0 = (step0 + E - 1) >> 2;
        if (seg > 0) e = (qhi0 - 64 * (int64_t)(seg * seg_chunks) + 1) * 4 - step0;
        e_end = (qhi0 - 64 * (int64_t)((seg + 1u) * seg_chunks) + 1) * 4 - step0;
        if (e_end < 0) e_end = 0;
        if (e < 0) e = 0;
    }
    // All-binary models (one logged byte per step and row): the 4 bytes of my quad are one (unaligned) dword per row, and the
    // rows' dwords simply add up (a byte is at most 16 chains, 4 rows at most 64: no carry between bytes).  The NEXT chunk's
    // dwords are requested before the current chunk is walked -- their addresses do not depend on the scan stream --, so the
    // one wavefront this group has on its SIMD no longer sits out a global-memory round trip per chunk.
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    const bool packed_log = log_vals == 1u;
    auto quad_ones = [&](const int64_t qlo) -> uint32_t { // sum over the rows of the 4 logged bytes of quad qlo + lane (steps outside the log: 0)
        const int64_t k0 = (qlo + (int64_t)lane) * 4 - step0;
        uint32_t sum = 0;
        if (k0 >= 0 && k0 + 4 <= (int64_t)log_pitch) {
            for (uint32_t r = 0; r < log_rows; ++r) sum += *reinterpret_cast<const u32_unaligned *>(lg + (size_t)r * log_pitch + (size_t)k0);
        } else {
            for (uint32_t j = 0; j < 4; ++j) {
                const int64_t k = k0 + j;
                if (k >= 0 && k < (int64_t)log_pitch)
                    for (uint32_t r = 0; r < log_rows; ++r) sum += (uint32_t)lg[(size_t)r * log_pitch + (size_t)k] << (8u * j);
            }
        }
        return sum;
    };
    // the variables of quad q of the group's scan stream (ScanReplay::quad with the free list read from LDS)
    auto quad_vars = [&](const int64_t q, uint32_t (&v)[4]) {
        v[0] = v[1] = v[2] = v[3] = 0;
        if (q < 0) return;
        const u32x4 w = philox4x32_10((uint32_t)q, (uint32_t)((uint64_t)q >> 32), sr.gid, DOM_SCAN, sr.seed_lo, sr.seed_hi);
        v[0] = fv[mul_hi_u32(w.x, sr.n_free)]; v[1] = fv[mul_hi_u32(w.y, sr.n_free)];
        v[2] = fv[mul_hi_u32(w.z, sr.n_free)]; v[3] = fv[mul_hi_u32(w.w, sr.n_free)];
    };
    uint32_t ones_next = 0, v_next[4] = {0, 0, 0, 0};
    if (e > e_end) {
        const int64_t qlo0 = ((step0 + e - 1) >> 2) - 63;
        if (packed_log) ones_next = quad_ones(qlo0);
        quad_vars(qlo0 + lane, v_next);
    }
    while (e > e_end && *rem) {
        const int64_t qhi = (step0 + e - 1) >> 2, qlo = qhi - 63;
        { // (0) stage the chunk; request / replay the one after it (walking backwards) before this one is walked
            const int64_t q = qlo + lane;
            const uint32_t ones = ones_next;
            const uint32_t v[4] = {v_next[0], v_next[1], v_next[2], v_next[3]};
            if (qlo * 4 - step0 > e_end) {
                if (packed_log) ones_next = quad_ones(qlo - 64);
                quad_vars(qlo - 64 + lane, v_next);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const int64_t k = q * 4 + j - step0;
                const bool valid = k >= 0 && k < e;
                cv[lane * 4u + j] = valid ? (unsigned short)v[j] : (unsigned short)0xffffu;
                if (packed_log) {
                    cl[lane * 4u + j] = valid ? (unsigned char)(ones >> (8u * j)) : (unsigned char)0;
                    continue;
                }
                if ((log_vals & 3u) == 0u) { // one count per value, 4 values per dword: the rows' dwords add up bytewise (no carries, see above)
                    for (uint32_t c4 = 0; c4 < log_vals; c4 += 4u) {
                        uint32_t n4 = 0;
                        if (valid) for (uint32_t r = 0; r < log_rows; ++r) n4 += *reinterpret_cast<const uint32_t *>(lg + (size_t)r * log_pitch + (size_t)k * log_vals + c4);
                        *reinterpret_cast<uint32_t *>(cl + (lane * 4u + j) * log_vals + c4) = n4;
                    }
                    continue;
                }
                for (uint32_t c = 0; c < log_vals; ++c) {
                    uint32_t n1 = 0;
                    if (valid) for (uint32_t r = 0; r < log_rows; ++r) n1 += lg[(size_t)r * log_pitch + (size_t)k * log_vals + c];
                    cl[(lane * 4u + j) * log_vals + c] = (unsigned char)n1;
                }
            }
        }
        __syncthreads();
        // (1) A variable's selections inside this chunk need their order only when the chunk straddles one of its two cuts
        // (selection `half` or 2 * `half`, counted from the end of the window) -- at most two chunks per variable out of the
        // hundreds a window has.  Everywhere else all of them fall into the same half (or are too old to count), so the 256
        // steps are classified side by side: kc[v] = selections of v in this chunk (LDS atomics), sn[v] = selections in the
        // newer chunks; fast steps add their counts with LDS atomics, only the steps of a straddling variable go through the
        // ordered walk below.  (Walking every step in order was this kernel: ~12 dependent LDS round-trip chains per chunk.)
        uint32_t mv[4];
        bool slow[4], fast[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            mv[j] = cv[lane * 4u + j];
            if (mv[j] != 0xffffu) atomicAdd(&kc[mv[j]], 1u);
        }
        __syncthreads();
        uint32_t full = 0;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            slow[j] = fast[j] = false;
            const uint32_t v = mv[j];
            if (v == 0xffffu) continue;
            const uint32_t c0 = sn[v], k = kc[v];
            if (c0 >= 2u * half) continue;                        // its ring is full already
            const bool second = c0 + k <= half, first = c0 >= half && c0 + k <= 2u * half;
            if (!second && !first) { slow[j] = true; continue; } // the chunk straddles a cut of v
            fast[j] = true;
            uint32_t *o = (second ? o2 : o1) + hf[v];
            const uint32_t idx = lane * 4u + j;
            if (log_vals == 1) { atomicAdd(&o[1], (uint32_t)cl[idx]); atomicAdd(&o[0], 64u - cl[idx]); }
            else for (uint32_t x = 0; x < hf[v + 1] - hf[v]; ++x) atomicAdd(&o[x], (uint32_t)cl[idx * log_vals + x]);
        }
        __syncthreads(); // every sn / kc of the chunk has been read
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t v = mv[j];
            if (v == 0xffffu) continue;
            kc[v] = 0; // (every step of v writes the same 0)
            if (fast[j]) full += (atomicAdd(&sn[v], 1u) + 1u == 2u * half) ? 1u : 0u; // exactly one step sees the ring fill up
        }
        // (2) the straddling variables' steps, in order: step (l, j) is owned by lane o = v & 63, so my mask over the 64 lanes of
        // plane j is the AND over the 6 bits of o of "lanes whose bit b equals my bit b" -- 6 ballots per plane; then the lanes
        // walk their steps newest first (step index = 4 * lane + plane), side by side: a variable is only ever touched by its owner
        unsigned long long M[4] = {0, 0, 0, 0};
        if (__any(slow[0] || slow[1] || slow[2] || slow[3])) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t v = mv[j];
                unsigned long long m = __ballot(slow[j]);
#pragma unroll
                for (uint32_t b = 0; b < 6; ++b) {
                    const unsigned long long bb = __ballot((v >> b) & 1u);
                    m &= ((lane >> b) & 1u) ? bb : ~bb;
                }
                M[j] = m;
            }
        }
        for (;;) {
            int best = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (M[j]) best = max(best, (63 - __clzll((long long)M[j])) * 4 + j);
            if (!__any(best >= 0)) break;
            if (best >= 0) {
                const unsigned long long bit = ~(1ull << (best >> 2));
                const int j = best & 3;
                M[0] = j == 0 ? M[0] & bit : M[0]; M[1] = j == 1 ? M[1] & bit : M[1];
                M[2] = j == 2 ? M[2] & bit : M[2]; M[3] = j == 3 ? M[3] & bit : M[3];
                const uint32_t idx = (uint32_t)best, v = cv[idx], c = sn[v];
                if (c < 2u * half) {
                    uint32_t *o = (c < half ? o2 : o1) + hf[v];
                    if (log_vals == 1) { o[1] += cl[idx]; o[0] += 64u - cl[idx]; }
                    else for (uint32_t x = 0; x < hf[v + 1] - hf[v]; ++x) o[x] += cl[idx * log_vals + x];
                    sn[v] = c + 1u;
                    full += (c + 1u == 2u * half) ? 1u : 0u;
                }
            }
        }
        if (full) atomicSub(rem, full);
        __syncthreads();
        e = qlo * 4 - step0; // everything from quad qlo on is done
    }
    __syncthreads();
    for (uint32_t i = lane; i < H; i += 64u) { // (hist is zero on entry; one block per segment adds its share)
        if (o1[i]) atomicAdd(&hist[((size_t)g * 2u + 0u) * H + i], o1[i]);
        if (o2[i]) atomicAdd(&hist[((size_t)g * 2u + 1u) * H + i], o2[i]);
    }
    if (n_seg == 1u && lane == 0 && half) status[g] = *rem; // (several segments: the verdict comes from the counts, above)
}

// ---- AdvanceChain under GR_FLAG_SCAN_PER_CHAIN: every chain has its own scan stream, hence its own stopping time and its own
// rings (chain.go:184-215 and buffer/circular.go per chain, literally).  One thread per chain; the per-variable counters of a
// block's chains sit in LDS as u16 (cw + 1 < 65 536 is checked by the host).
__global__ void k_stop_time_pc(const ScanArgs a, uint32_t n_chains, uint32_t cw, uint32_t batch, uint32_t max_batches, uint32_t *out) {
    extern __shared__ unsigned short sm_pc[];
    const uint32_t chain = blockIdx.x * blockDim.x + threadIdx.x;
    if (chain >= n_chains) return;
    const uint32_t variant = chain / a.chains_per_variant;
    const VariantDesc vd = a.variants[variant];
    unsigned short *cnt = sm_pc + (size_t)threadIdx.x * vd.n_free; // by rank in the free list
    for (uint32_t i = 0; i < vd.n_free; ++i) cnt[i] = 0;
    const uint64_t gchain = (((uint64_t)a.chain_offset_hi << 32) | a.chain_offset_lo) + chain;
    uint64_t s = ((uint64_t)a.step0_hi << 32) | a.step0_lo;
    uint32_t remaining = vd.n_free, b = 0;
    while (remaining && b < max_batches) {
        const uint64_t e = s + batch;
        for (uint64_t q = s >> 2; (q << 2) < e; ++q) {
            const u32x4 w = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain, DOM_SCAN_CHAIN + ((uint32_t)(gchain >> 32) << 4), a.seed_lo, a.seed_hi);
            const uint32_t k[4] = {mul_hi_u32(w.x, vd.n_free), mul_hi_u32(w.y, vd.n_free), mul_hi_u32(w.z, vd.n_free), mul_hi_u32(w.w, vd.n_free)};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint64_t st = (q << 2) + j;
                if (st < s || st >= e) continue;
                const uint32_t c = cnt[k[j]];
                if (c <= cw) {
                    cnt[k[j]] = (unsigned short)(c + 1u);
                    if (c == cw) --remaining; // its (cw + 1)-th selection
                }
            }
        }
        s = e;
        ++b;
    }
    out[chain] = remaining ? 0xffffffffu : b;
}

// log: one value byte per chain and step, [group][step][64 chains]; limit[chain] = steps of the window the chain executed.
// Walking the chain's own window backwards: the newest `half` selections of a variable are its ring's second half, the `half`
// before them the first; hist[g][h][H] adds up the 64 chains of group g.  status[g] counts rings that are not full.
__global__ void __launch_bounds__(64) k_ring_halves_pc(const ScanArgs a, uint32_t n_chains, uint32_t half, const unsigned char *log, uint32_t log_pitch,
                                                      const uint32_t *limit, uint32_t *hist, uint32_t *status) {
    extern __shared__ unsigned short sm_pc[];
    const uint32_t g = blockIdx.x, lane = threadIdx.x, chain = g * 64u + lane;
    const uint32_t variant = (g * 64u) / a.chains_per_variant;
    const VariantDesc vd = a.variants[variant];
    uint32_t *h = reinterpret_cast<uint32_t *>(sm_pc);                  // [2][H] of the group
    unsigned short *cnt = sm_pc + 4u * a.H + (size_t)lane * vd.n_free; // selections seen so far (walking backwards), by rank
    for (uint32_t i = lane; i < 2u * a.H; i += 64u) h[i] = 0;
    for (uint32_t i = 0; i < vd.n_free; ++i) cnt[i] = 0;
    __syncthreads();
    if (chain < n_chains && half) {
        const uint64_t gchain = (((uint64_t)a.chain_offset_hi << 32) | a.chain_offset_lo) + chain;
        const uint64_t s0 = ((uint64_t)a.step0_hi << 32) | a.step0_lo;
        const uint32_t n = limit[chain];
        const unsigned char *lg = log + (size_t)g * log_pitch + lane;
        uint32_t remaining = vd.n_free;
        for (uint64_t q = (s0 + n - 1u) >> 2; remaining && n && q + 1u > (s0 >> 2); --q) {
            const u32x4 w = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain, DOM_SCAN_CHAIN + ((uint32_t)(gchain >> 32) << 4), a.seed_lo, a.seed_hi);
            const uint32_t k[4] = {mul_hi_u32(w.x, vd.n_free), mul_hi_u32(w.y, vd.n_free), mul_hi_u32(w.z, vd.n_free), mul_hi_u32(w.w, vd.n_free)};
            for (int j = 3; j >= 0; --j) {
                const uint64_t st = (q << 2) + (uint32_t)j;
                if (st < s0 || st >= s0 + n) continue;
                const uint32_t c = cnt[k[j]];
                if (c >= 2u * half) continue;
                cnt[k[j]] = (unsigned short)(c + 1u);
                if (c + 1u == 2u * half) --remaining;
                const uint32_t v = a.free_vars[vd.free_base + k[j]];
                const uint32_t val = lg[(size_t)(st - s0) * 64u];
                atomicAdd(&h[(c < half ? a.H : 0u) + a.hoff[v] + val], 1u); // the newest `half` -> second half
            }
            if (q == 0) break;
        }
        if (remaining) atomicAdd(&status[g], remaining);
    }
    __syncthreads();
    for (uint32_t i = lane; i < 2u * a.H; i += 64u)
        if (h[i]) atomicAdd(&hist[(size_t)g * 2u * a.H + i], h[i]);
}

#endif // GR_KSWEEP_PART == 0

// ------------------------------------------------------------ host launchers
#define GR_DISPATCH_MAXC(maxc, ...)                       \
    do {                                                    \
        if ((maxc) <= 2) { constexpr int MC = 2; __VA_ARGS__; }    \
        else if ((maxc) <= 4) { constexpr int MC = 4; __VA_ARGS__; } \
        else if ((maxc) <= 8) { constexpr int MC = 8; __VA_ARGS__; } \
        else if ((maxc) <= 16) { constexpr int MC = 16; __VA_ARGS__; } \
        else if ((maxc) <= 32) { constexpr int MC = 32; __VA_ARGS__; } \
        else { constexpr int MC = 64; __VA_ARGS__; }               \
    } while (0)

template <int MC, bool CPT, bool TL, bool PW, int PC = 0, bool LOG = false, bool STG = false>
static hipError_t launch_sweep_t(const SweepParams &p, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep<MC, CPT, TL, PW, PC, LOG, STG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sweep<MC, CPT, TL, PW, PC, LOG, STG>), dim3(blocks), dim3(threads), lds, s, p);
    return hipGetLastError();
}

template <int MC>
static hipError_t launch_sweep_mc(const SweepParams &p, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global,
                                  uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t s) {
    if (state_global && p.log) // ... with the draw log: direct back-end only (like the LDS-state log instantiation)
        return (per_chain || cpt || tab_lds) ? hipErrorInvalidValue : launch_sweep_t<MC, false, false, false, 0, true, true>(p, blocks, threads, lds_bytes, s);
    if (state_global) // state in global memory: tables from global memory too; no strict mode
        return (per_chain || tab_lds) ? hipErrorInvalidValue
               : cpt ? launch_sweep_t<MC, true, false, false, 0, false, true>(p, blocks, threads, lds_bytes, s)
               : pairwise ? launch_sweep_t<MC, false, false, true, 0, false, true>(p, blocks, threads, lds_bytes, s)
                          : launch_sweep_t<MC, false, false, false, 0, false, true>(p, blocks, threads, lds_bytes, s);
    if (p.sys_order) // systematic scan: direct back-end, tables from global memory, chain state in LDS
        return (p.log || per_chain || cpt || tab_lds) ? hipErrorInvalidValue : launch_sweep_t<MC, false, false, false, 2>(p, blocks, threads, lds_bytes, s);
    if (p.log && per_chain) return (cpt || tab_lds) ? hipErrorInvalidValue : launch_sweep_t<MC, false, false, false, 1, true>(p, blocks, threads, lds_bytes, s);
    if (p.log) return (per_chain || cpt || tab_lds) ? hipErrorInvalidValue : launch_sweep_t<MC, false, false, false, 0, true>(p, blocks, threads, lds_bytes, s);
    if (per_chain) return tab_lds ? launch_sweep_t<MC, false, true, false, 1>(p, blocks, threads, lds_bytes, s)
                                  : launch_sweep_t<MC, false, false, false, 1>(p, blocks, threads, lds_bytes, s);
    if (cpt) return tab_lds ? launch_sweep_t<MC, true, true, false>(p, blocks, threads, lds_bytes, s)
                            : launch_sweep_t<MC, true, false, false>(p, blocks, threads, lds_bytes, s);
    if (pairwise) return tab_lds ? launch_sweep_t<MC, false, true, true>(p, blocks, threads, lds_bytes, s)
                                 : launch_sweep_t<MC, false, false, true>(p, blocks, threads, lds_bytes, s);
    return tab_lds ? launch_sweep_t<MC, false, true, false>(p, blocks, threads, lds_bytes, s)
                   : launch_sweep_t<MC, false, false, false>(p, blocks, threads, lds_bytes, s);
}

#if GR_KSWEEP_PART == 1
hipError_t launch_sweep_card16(const SweepParams &p, int max_card, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global,
                               uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t s) {
    return max_card <= 8 ? launch_sweep_mc<8>(p, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s)
                         : launch_sweep_mc<16>(p, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s);
}
#elif GR_KSWEEP_PART == 2
hipError_t launch_sweep_card64(const SweepParams &p, int max_card, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global,
                               uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t s) {
    return max_card <= 32 ? launch_sweep_mc<32>(p, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s)
                          : launch_sweep_mc<64>(p, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s);
}
#else
hipError_t launch_sweep_card16(const SweepParams &p, int max_card, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global,
                               uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t s);
hipError_t launch_sweep_card64(const SweepParams &p, int max_card, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global,
                               uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t s);

hipError_t launch_sweep(const SweepParams &p, int max_card, bool cpt, bool tab_lds, bool pairwise, bool per_chain, bool state_global,
                        uint32_t blocks, uint32_t threads, uint32_t lds_bytes, hipStream_t s) {
    if (max_card <= 2) return launch_sweep_mc<2>(p, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s);
    if (max_card <= 4) return launch_sweep_mc<4>(p, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s);
    if (max_card <= 16) return launch_sweep_card16(p, max_card, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s);
    return launch_sweep_card64(p, max_card, cpt, tab_lds, pairwise, per_chain, state_global, blocks, threads, lds_bytes, s);
}

hipError_t launch_init_state(unsigned char *state, const int32_t *card, const int32_t *fixed, uint32_t n_vars, uint32_t pitch,
                             uint32_t chain_begin, uint32_t chain_end, uint64_t chain_offset, uint64_t seed, uint32_t epoch,
                             hipStream_t s) {
    const uint32_t n = chain_end - chain_begin;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_init_state, dim3((n + 255) / 256), dim3(256), 0, s, state, card, fixed, n_vars, pitch, chain_begin,
                       chain_end, chain_offset, (uint32_t)seed, (uint32_t)(seed >> 32), epoch);
    return hipGetLastError();
}

hipError_t launch_build_cpt(const VariantDesc &vd, const uint32_t *progs, const uint32_t *prog_off, const float *views,
                            const uint32_t *cprogs, const uint32_t *cprog_off, uint32_t *cpt, const int32_t *card,
                            const int32_t *fixed, uint32_t n_vars, uint32_t max_cfg, int max_card, hipStream_t s) {
    const dim3 grid((max_cfg + 127) / 128, n_vars);
    GR_DISPATCH_MAXC(max_card, {
        hipLaunchKernelGGL((k_build_cpt<MC>), grid, dim3(128), 0, s, vd, progs, prog_off, views, cprogs, cprog_off, cpt, card, fixed);
    });
    return hipGetLastError();
}

hipError_t launch_conditional(const VariantDesc &vd, const uint32_t *progs, const uint32_t *prog_off, const float *views,
                              uint32_t v, const int32_t *state, double *probs, int max_card, hipStream_t s) {
    GR_DISPATCH_MAXC(max_card, {
        hipLaunchKernelGGL((k_conditional<MC>), dim3(1), dim3(1), 0, s, vd, progs, prog_off, views, v, state, probs);
    });
    return hipGetLastError();
}

hipError_t launch_scan_vars(const ScanArgs &a, uint32_t n_quads, uint32_t pitch_quads, unsigned short *out, hipStream_t s) {
    if (n_quads == 0 || a.n_groups == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_vars, dim3((n_quads + 255u) / 256u, a.n_groups), dim3(256), 0, s, a, n_quads, pitch_quads, out);
    return hipGetLastError();
}

hipError_t launch_stop_time(const ScanArgs &a, uint32_t cw, uint32_t batch, uint32_t max_batches, uint32_t *out, hipStream_t s) {
    const size_t lds = (size_t)a.n_vars * sizeof(uint32_t) + (((size_t)a.n_vars + 1u) & ~(size_t)1) * sizeof(unsigned short);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stop_time), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stop_time, dim3(a.n_groups), dim3(256), lds, s, a, cw, batch, max_batches, out);
    return hipGetLastError();
}

hipError_t launch_ring_halves(const ScanArgs &a, uint32_t n_steps, uint32_t half, const unsigned char *log, uint32_t log_pitch, uint32_t log_rows,
                              uint32_t log_vals, const uint32_t *limit, uint32_t *hist, uint32_t *status, uint32_t n_seg, uint32_t seg_chunks,
                              uint32_t *seg_cnt, hipStream_t s) {
    // hist and status are accumulated into: zero them; with several segments the per-segment selection counts come first
    hipError_t e0 = hipMemsetAsync(hist, 0, (size_t)a.n_groups * 2u * a.H * sizeof(uint32_t), s);
    if (e0 != hipSuccess) return e0;
    e0 = hipMemsetAsync(status, 0, (size_t)a.n_groups * sizeof(uint32_t), s);
    if (e0 != hipSuccess) return e0;
    if (n_seg > 1u) {
        const size_t lc = (size_t)a.n_vars * sizeof(uint32_t);
        e0 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_count), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lc);
        if (e0 != hipSuccess) return e0;
        hipLaunchKernelGGL(k_ring_count, dim3(a.n_groups, n_seg), dim3(64), lc, s, a, n_steps, limit, n_seg, seg_chunks, seg_cnt);
        e0 = hipGetLastError();
        if (e0 != hipSuccess) return e0;
    }
    const size_t lds = (3u * (size_t)a.n_vars + 2u * a.H + 2u) * sizeof(uint32_t) + (256u + (((size_t)a.n_vars + 1u) & ~(size_t)1)) * sizeof(unsigned short) + 256u * log_vals;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_halves), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ring_halves, dim3(a.n_groups, n_seg), dim3(64), lds, s, a, n_steps, half, log, log_pitch, log_rows, log_vals, limit, hist, status,
                       n_seg, seg_chunks, seg_cnt);
    return hipGetLastError();
}

hipError_t launch_stop_time_pc(const ScanArgs &a, uint32_t n_chains, uint32_t n_free_max, uint32_t cw, uint32_t batch, uint32_t max_batches, uint32_t *out, hipStream_t s) {
    // threads per block: as many chains' u16 counters as fit 64 KiB of LDS
    uint32_t t = 64;
    while (t > 1 && (size_t)t * n_free_max * 2u > 64u * 1024u) t >>= 1;
    const size_t lds = (size_t)t * n_free_max * 2u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_stop_time_pc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_stop_time_pc, dim3((n_chains + t - 1u) / t), dim3(t), lds, s, a, n_chains, cw, batch, max_batches, out);
    return hipGetLastError();
}

hipError_t launch_ring_halves_pc(const ScanArgs &a, uint32_t n_chains, uint32_t n_free_max, uint32_t half, const unsigned char *log, uint32_t log_pitch,
                                 const uint32_t *limit, uint32_t *hist, uint32_t *status, hipStream_t s) {
    hipError_t e0 = hipMemsetAsync(hist, 0, (size_t)a.n_groups * 2u * a.H * sizeof(uint32_t), s);
    if (e0 != hipSuccess) return e0;
    e0 = hipMemsetAsync(status, 0, (size_t)a.n_groups * sizeof(uint32_t), s);
    if (e0 != hipSuccess) return e0;
    const size_t lds = (size_t)2u * a.H * 4u + (size_t)64u * n_free_max * 2u;
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ring_halves_pc), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ring_halves_pc, dim3(a.n_groups), dim3(64), lds, s, a, n_chains, half, log, log_pitch, limit, hist, status);
    return hipGetLastError();
}

hipError_t launch_collapse(const CollapseArgs &a, uint32_t blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_collapse, dim3(blocks), dim3(256), 0, s, a);
    return hipGetLastError();
}
#endif // GR_KSWEEP_PART

} // namespace gr
