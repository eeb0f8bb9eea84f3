// K1, step-parallel form of the DIRECT conditional (k_sweep_dir4): 4 lanes per chain, one per step of a quad, for models
// whose blankets are too large to memoise and whose variables have at most 4 values (CSP: Card 2..4 with up to 31 factors
// on a variable, Alchemy: binary with blankets of up to 59 variables).
//
// The generic direct kernel walks the updated variable's program on the scalar unit -- fine with many wavefronts, but at the
// BASELINE sizes (32 768 chains = 512 chain-per-lane wavefronts) half of the chip's SIMDs have no wavefront and the others a
// lone one that issues an instruction every ~10 cycles: ~900 instructions per 64-chain step on CSP_13, most of them scalar
// address work and branches.  Here the k_sweep_par4 shape is used instead (sweep_bin.hip): a 256-lane block owns the 64
// chains of one scan group, lane = step slot * 16 + chain, the scan plan (k_scan_plan) names each step's variable and its
// dependency level inside the quad, steps that do not commute are re-evaluated level by level -- but a step's conditional
// is EVALUATED, not looked up: the lane walks the variable's factor list (16-byte entries {view offset, other variables'
// state rows, strides}, read with row-uniform vector loads), gathers the neighbour states from LDS, sums the view rows in
// Model.Funcs order and runs cond.h itself.  Every lane does exactly the arithmetic one lane of the generic kernel does, in
// the same order: bit-identical results.  32 768 chains are 2 048 wavefronts and 4 steps of every chain are in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "cond.h"
#include "device_layout.h"
#include "kernels.h"
#include "philox.h"
#include "wave_prio.h"

namespace gr {

namespace {

__device__ __forceinline__ uint32_t uni_d(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
typedef __attribute__((address_space(3))) uint32_t lds_u32d_t;
__device__ __forceinline__ uint32_t lds_addr_d(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ lds_u32d_t *lds_at_d(uint32_t addr) { return (lds_u32d_t *)(uintptr_t)addr; }

} // namespace

// Step header (32 bytes, one per selectable variable, addressed by the plan word like a k_sweep_par4 record):
//   w0 state-row byte offset (var * 80)     w1 absolute LDS byte address of the variable's histogram slots
//   w2 byte offset of its factor entries (absolute LDS address when they are staged)
//   w3 number of entries (factors padded to a multiple of 4 with entries that read an all-zero row) | Card << 16
// Factor entry (16 bytes): {view-table byte offset (LDS address when the views are staged, else offset into the group's
// view pool), row1 | row2 << 16, row3 | stride1 << 16, stride2 | stride3 << 16} -- the state rows of up to three other
// scope variables and their byte strides in the view table; unused slots name the variable's own row with stride 0.
// VLDS / ELDS: the group's view tables / factor entries are staged in LDS (else read from global memory through L1/L2).
// MODE 0: burn-in, 1: count, 2: count + draw log and per-group stopping time (gr_advance_chain; SweepParams::log / limit)
// CHAIN: chained entries {view byte offset | 1 = continuation, row1 | row2 << 16, stride1, stride2} -- two other scope
// variables with 32-bit byte strides per entry.  A continuation entry adds its partial offset to the next entry of the list
// and loads the group's all-zero row itself, so a factor of any arity still contributes ONE view row, at its place in
// Model.Funcs order (a collapse factor: up to 12 variables and 2^23 entries, gibbs-collapsed.go:93 / function.go:59).
template <int MAXC, bool VLDS, bool ELDS, int MODE, bool CHAIN = false>
__global__ void __launch_bounds__(256) k_sweep_dir4(const FastParams p) {
    static_assert(!(CHAIN && VLDS), "chained entries address the view pool in global memory");
    constexpr bool COUNT = MODE != 0, LOG = MODE == 2;
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t j = lane >> 4;                    // step slot inside the quad
    const uint32_t cb = wave * 16u + (lane & 15u);   // chain inside the block
    const uint32_t chain = blockIdx.x * 64u + cb;
    const uint32_t variant = (blockIdx.x * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    {
        if constexpr (VLDS) { // this group's view tables
            const uint4 *src = reinterpret_cast<const uint4 *>(p.views + vd.views_base);
            uint4 *dst = reinterpret_cast<uint4 *>(lds + p.dir_view_off);
            for (uint32_t i = tid; i < (vd.views_len >> 2); i += 256) dst[i] = src[i];
        }
        if constexpr (ELDS) { // ... and its factor entries
            const uint4 *src = reinterpret_cast<const uint4 *>(p.dent) + vd.dent_base;
            uint4 *dst = reinterpret_cast<uint4 *>(lds + p.dir_ent_off);
            for (uint32_t i = tid; i < vd.dent_len; i += 256) dst[i] = src[i];
        }
        for (uint32_t v = j; v < p.n_vars; v += 4) lds[v * p.par_stride + cb] = p.state[(size_t)v * p.pitch + chain];
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + p.par_hist_off);
        for (uint32_t i = tid; i < p.H; i += 256) h[i] = 0;
    }
    __syncthreads();
    // headers and entries hold offsets into the dynamic LDS array, used as absolute LDS addresses: it must start at 0 (no
    // static LDS in this kernel); fail loudly otherwise (error flag -> GR_E_STATE), like k_sweep_par4
    if (lds_addr_d(lds) != 0u) {
        if (threadIdx.x == 0) atomicOr(p.err, 1u);
        return;
    }

    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint64_t q0 = (((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2;
    const uint32_t nq = p.n_steps >> 2; // multiple of 4
    const unsigned char *lds_c = lds + cb;
    const unsigned char *hdr_g = reinterpret_cast<const unsigned char *>(p.recs_par);
    const unsigned char *ent_g = reinterpret_cast<const unsigned char *>(p.dent);
    const unsigned char *views_g = reinterpret_cast<const unsigned char *>(p.views + vd.views_base);
    const uint32_t *gplan_w = p.plan + (size_t)blockIdx.x * p.plan_pitch + j;
    const uint32_t zrow = vd.zero_row * 4u;
    const bool tail = (lane & 15u) == 15u;
    // draw log: one row per wavefront (the 16 chains of a step slot), log_vals bytes per step -- the number of ones, or one
    // count per value; a group executes only the first `lim` steps of the window (its own AdvanceChain stopping time)
    unsigned char *logrow = LOG && p.log ? p.log + (size_t)(blockIdx.x * 4u + wave) * p.log_pitch + (size_t)p.log_off * p.log_vals : nullptr;
    uint32_t lim = 0xffffffffu;
    if constexpr (LOG)
        if (p.limit) {
            const uint32_t t = p.limit[blockIdx.x];
            lim = t > p.log_off ? t - p.log_off : 0u;
        }

    // headers of the NEXT iteration in flight while the current one runs; plan words two iterations ahead
    uint32_t pwn[4] = {0, 0, 0, 0}, pwa[4] = {0, 0, 0, 0};
    uint4 hn[4];
    auto prefetch_plan = [&](const uint32_t gnext) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pwa[k] = gplan_w[(size_t)min(gnext + (uint32_t)k, nq - 1u) * 4u]; // never past the plan
    };
    auto prefetch = [&]() {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pwn[k] = pwa[k];
            hn[k] = *reinterpret_cast<const uint4 *>(hdr_g + (pwn[k] & (kPlanOffMask & ~31u)));
        }
    };
    prefetch_plan(0); prefetch(); prefetch_plan(4);

    // the conditional of one step, evaluated from the factor tables (gibbs-simple.go:175-258): gather, log sums in
    // Model.Funcs order, cond.h.  The loop runs to the largest factor count among the wavefront's four step slots.
    // `mine`: only the step slots whose turn it is take part (the others walk no factors: no LDS traffic for them)
    auto eval = [&](const uint4 &h, const uint32_t u, const bool mine) -> uint32_t {
        const uint32_t nf = mine ? h.w & 0xffffu : 0u;
        const int C = (int)(h.w >> 16);
        const uint32_t n01 = max((uint32_t)__builtin_amdgcn_readlane((int)nf, 0), (uint32_t)__builtin_amdgcn_readlane((int)nf, 16));
        const uint32_t n23 = max((uint32_t)__builtin_amdgcn_readlane((int)nf, 32), (uint32_t)__builtin_amdgcn_readlane((int)nf, 48));
        const uint32_t nmax = max(n01, n23);
        float w[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) w[i] = 0.f;
        uint32_t carry = 0; // CHAIN: partial view offset handed over from the last entry of the previous trip
        // four factors per trip: their entries, then their 12 neighbour bytes, then their 4 view rows are each requested
        // together (one memory round trip per stage and trip instead of three per factor); summed in factor order
        for (uint32_t f = 0; f < nmax; f += 4) {
            if (f < nf) {
                uint4 e[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if constexpr (ELDS) e[i] = *reinterpret_cast<const uint4 *>(lds + h.z + 16u * (f + (uint32_t)i));
                    else e[i] = *reinterpret_cast<const uint4 *>(ent_g + h.z + 16u * (f + (uint32_t)i));
                }
                uint32_t off[4];
                if constexpr (CHAIN) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t x1 = lds_c[e[i].y & 0xffffu], x2 = lds_c[e[i].y >> 16];
                        off[i] = (e[i].x & ~1u) + x1 * e[i].z + x2 * e[i].w + carry;
                        carry = (e[i].x & 1u) ? off[i] : 0u;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (e[i].x & 1u) off[i] = zrow;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t x1 = lds_c[e[i].y & 0xffffu], x2 = lds_c[e[i].y >> 16], x3 = lds_c[e[i].z & 0xffffu];
                        off[i] = e[i].x + __umul24(x1, e[i].z >> 16) + __umul24(x2, e[i].w & 0xffffu) + __umul24(x3, e[i].w >> 16);
                    }
                }
                if constexpr (MAXC <= 2) {
                    float2 r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (VLDS) r[i] = *reinterpret_cast<const float2 *>(lds + off[i]);
                        else r[i] = *reinterpret_cast<const float2 *>(views_g + off[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { w[0] += r[i].x; w[1] += r[i].y; }
                } else {
                    float4 r[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (VLDS) r[i] = *reinterpret_cast<const float4 *>(lds + off[i]);
                        else r[i] = *reinterpret_cast<const float4 *>(views_g + off[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { w[0] += r[i].x; w[1] += r[i].y; w[2] += r[i].z; w[3] += r[i].w; }
                }
            }
        }
        uint32_t cum[MAXC > 1 ? MAXC - 1 : 1], tot;
        cumulative<MAXC>(w, C, cum, tot);
        return pick<MAXC>(cum, tot, C, u);
    };

    // GR_FLAG_SCAN_SYSTEMATIC (p.sys_rank set; never with the draw log): the plan follows the sweep order and update t of a chain
    // takes word (sweep & 3) of philox_sys_draw(sweep = t / n, pos = t mod n, chain) -- one Philox call per step.  My steps are
    // 4 apart: position and sweep number are kept by increments.
    const bool sys = !LOG && p.sys_rank != nullptr;
    uint64_t sys_sweep = 0;
    uint32_t sys_pos = 0;
    if (sys) {
        const uint64_t s_first = (q0 << 2) + j;
        sys_sweep = s_first / vd.n_free;
        sys_pos = (uint32_t)(s_first - sys_sweep * vd.n_free);
    }
    const PrioRotor prio; // (wave_prio.h: the blocks of a CU progress as equals)
    for (uint32_t g0 = 0; g0 < nq; g0 += 4) {
        prio.tick(g0, p.prio_shift);
        uint32_t U[4];
        if (sys) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 d = philox_sys_draw(sys_sweep, sys_pos, gchain, p.seed_lo, p.seed_hi);
                const uint32_t w = (uint32_t)sys_sweep & 3u;
                U[k] = w == 0 ? d.x : w == 1 ? d.y : w == 2 ? d.z : d.w;
                sys_pos += 4u;
                while (sys_pos >= vd.n_free) { sys_pos -= vd.n_free; ++sys_sweep; }
            }
        } else {
            // draws: my Philox call is quad g0 + j; a 4x4 transpose (step slots x words) with gfx950's lane swaps leaves lane
            // (j, c) with word j of quads g0 .. g0 + 3 (see k_sweep_par4)
            const uint64_t q = q0 + g0 + j;
            const u32x4 W = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
            const auto a = __builtin_amdgcn_permlane32_swap(W.x, W.z, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(W.y, W.w, false, false);
            const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
            const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
            U[0] = c[0]; U[1] = c[1]; U[2] = d[0]; U[3] = d[1];
        }
        uint32_t pw[4];
        uint4 hq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pw[k] = pwn[k]; hq[k] = hn[k]; }
        prefetch(); prefetch_plan(g0 + 8u);
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const uint4 h = k == 0 ? hq[0] : k == 1 ? hq[1] : k == 2 ? hq[2] : hq[3];
            const uint32_t pwk = k == 0 ? pw[0] : k == 1 ? pw[1] : k == 2 ? pw[2] : pw[3];
            const uint32_t u = k == 0 ? U[0] : k == 1 ? U[1] : k == 2 ? U[2] : U[3];
            const uint32_t lv = (pwk >> 1) & 3u;
            const uint32_t maxlv = uni_d(pwk >> 3) & 3u;
            unsigned char *wr = lds + h.x + cb;
            // level by level: a step is evaluated exactly once, in the pass of its own dependency level, after the steps of the
            // lower levels (the ones it does not commute with) have committed
            const uint32_t sidx = (g0 + (uint32_t)k) * 4u + j;     // step index inside this launch
            const bool runs = !LOG || sidx < lim;                   // (limits are even: a quad is cut between its steps 1 and 2 at most)
            uint32_t val = 0;
            for (uint32_t pass = 0; pass <= maxlv; ++pass) {
                // (a pass reads what OTHER lanes committed in the pass before: in-order LDS makes that work, the fence keeps the
                // compiler from reasoning per lane about which of those loads are redundant -- see k_sweep_par4)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const bool mine = lv == pass && runs;
                const uint32_t v2 = eval(h, u, mine);
                if (mine) { *wr = (unsigned char)v2; val = v2; }
            }
            if constexpr (COUNT) {
                if (LOG && uni_d(sidx & ~3u) >= lim) continue;      // the whole quad is past the group's stopping time
                if constexpr (MAXC <= 2) {
                    // ones of the step slot's 16 draws: prefix popcount of the draw mask minus the prefix at the end of the
                    // previous DPP row, valid in lane 15 of every row (k_sweep_par4's count)
                    const unsigned long long drawn = __ballot(val != 0u);
                    const int incl = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(drawn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)drawn, 0u)) + (int)val;
                    const int s1 = incl - __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xe, 0xf, false);
                    if (tail && runs) {
                        lds_u32d_t *hp = lds_at_d(h.y);
                        __hip_atomic_fetch_add(hp, 16u - (uint32_t)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(hp + 1, (uint32_t)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (LOG && logrow) {
                            // one byte per step (the ones) when every selectable variable is binary; a model with max Card 2
                            // that has a selectable Card-1 variable is logged with one count per value like the generic kernel
                            // does (gr_advance_chain: log_vals = max Card), or k_ring_halves would read a pitch this kernel did not write
                            if (p.log_vals == 1) logrow[sidx] = (unsigned char)s1;
                            else { logrow[(size_t)sidx * p.log_vals] = (unsigned char)(16 - s1); logrow[(size_t)sidx * p.log_vals + 1] = (unsigned char)s1; }
                        }
                    }
                } else {
                    if (runs) __hip_atomic_fetch_add(lds_at_d(h.y + 4u * val), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if constexpr (LOG) { // one count per value among the 16 chains of my step slot
                        const unsigned long long rowmask = 0xffffull << (16u * j);
#pragma unroll
                        for (uint32_t c = 0; c < (uint32_t)MAXC; ++c) {
                            const unsigned long long m = __ballot(runs && val == c) & rowmask;
                            if (tail && runs && logrow && c < p.log_vals) logrow[(size_t)sidx * p.log_vals + c] = (unsigned char)__popcll(m);
                        }
                    }
                }
            }
        }
    }

    for (uint32_t v = j; v < p.n_vars; v += 4) p.state[(size_t)v * p.pitch + chain] = lds[v * p.par_stride + cb];
    if constexpr (COUNT) {
        __syncthreads();
        const uint32_t *h = reinterpret_cast<const uint32_t *>(lds + p.par_hist_off);
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = tid; i < p.H; i += 256) {
            const uint32_t c = h[i];
            if (c) atomicAdd(&gc[i], (unsigned long long)c);
        }
    }
}

template <int MAXC, bool VLDS, bool ELDS, bool CHAIN = false>
static hipError_t launch_dir4_t(const FastParams &p, uint32_t blocks, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, p);
        return hipGetLastError();
    };
    return !p.counting ? go(&k_sweep_dir4<MAXC, VLDS, ELDS, 0, CHAIN>) : p.log ? go(&k_sweep_dir4<MAXC, VLDS, ELDS, 2, CHAIN>) : go(&k_sweep_dir4<MAXC, VLDS, ELDS, 1, CHAIN>);
}

hipError_t launch_sweep_dir4(const FastParams &p, int max_card, bool views_in_lds, bool entries_in_lds, bool chained, uint32_t blocks, uint32_t lds, hipStream_t s) {
    if (max_card > 4 || (chained && views_in_lds)) return hipErrorInvalidValue;
    auto pick = [&](auto mc) -> hipError_t {
        constexpr int MC = decltype(mc)::value;
        if (chained) return entries_in_lds ? launch_dir4_t<MC, false, true, true>(p, blocks, lds, s) : launch_dir4_t<MC, false, false, true>(p, blocks, lds, s);
        if (views_in_lds) return entries_in_lds ? launch_dir4_t<MC, true, true>(p, blocks, lds, s) : launch_dir4_t<MC, true, false>(p, blocks, lds, s);
        return entries_in_lds ? launch_dir4_t<MC, false, true>(p, blocks, lds, s) : launch_dir4_t<MC, false, false>(p, blocks, lds, s);
    };
    return max_card <= 2 ? pick(std::integral_constant<int, 2>{}) : pick(std::integral_constant<int, 4>{});
}

} // namespace gr
