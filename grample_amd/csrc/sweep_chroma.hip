// K1, chromatic form (GR_FLAG_SCAN_SYSTEMATIC): systematic scan over the colour classes of the factor graph,
// all variables of a class updated side by side.
//
// The reference draws a random variable per update (sampler.go:135-174); a chain's updates are then sequential by construction
// and the kernels of sweep_bin.hip find their parallelism ACROSS chains (and across a few commuting steps of one chain).  Under
// the systematic scan the order is fixed -- colour 0's variables, colour 1's, ... -- and the variables of one colour share no
// factor, so a whole class can be resampled at once: ONE WAVEFRONT PER CHAIN, one lane per variable of the class.  A lane slot
// = (colour class, 64-variable chunk of it); each lane owns the same variable in every sweep, so everything about it -- its
// row offsets, multipliers, threshold-table address, its place in the histogram -- is loaded ONCE into registers: no scan
// stream, no plan, no records per step, no dependency levels, no histogram atomics (a lane counts its own variable's ones in
// a register).  4 096 chains are 4 096 wavefronts: the chip is full from BASELINE config 1 up.
//   chain state   lds[wave * row + var]  (one row of n_vars bytes per chain; the lanes of a class read distinct bytes)
//   tables        memoised thresholds (k_build_cpt), staged in LDS per block
//   draws         philox_sys_draw(sweep, position in the order, chain): word (sweep & 3) -- one Philox call per lane and 4 sweeps
//   classes       one wavefront executes its LDS operations in order, so class k + 1 sees class k's writes without a barrier
//                 (a wavefront-scope fence keeps the compiler from moving loads across the stores)
// Same thresholds, same draw stream and same order as the generic kernel's systematic mode (k_sweep<.., SCAN 2>): bit-identical
// results (tests/test_gpu_systematic.py); launches that do not start or end on a sweep boundary go through that kernel.
// Limits of this form: every selectable variable binary and memoised, at most NSLOT lane slots (2 or 4) and 8 neighbours.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"
#include "philox.h"
#include "wave_prio.h"

namespace gr {

namespace {
typedef __attribute__((address_space(3))) uint32_t lds_u32c_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8c_t;
__device__ __forceinline__ uint32_t lds_addr_c(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ uint32_t mad_u16_hi_c(const uint32_t w, const uint32_t b, const uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
    return d;
}
} // namespace

// chroma record (64 bytes per entry of the free list): w0 = the variable's byte inside a chain's row, w1 = its histogram offset
// (model layout), w2 = byte offset of its threshold table inside the staged pool, w3 = neighbours, w[4 + i] = neighbour's byte
// inside the row | (4 * multiplier) << 16 (unused slots: the variable's own byte, multiplier 0)
// PART: the launch may start and end inside a sweep (see below); without it the host launches whole sweeps from a sweep boundary
template <int NSLOT, int SLOTS, bool COUNT, bool PART = false>
__global__ void __launch_bounds__(1024) k_sweep_chroma(const FastParams p) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nthr = blockDim.x;
    const uint32_t chain = blockIdx.x * p.chroma_wpb + wave;
    const uint32_t variant = (blockIdx.x * p.chroma_wpb) / p.chains_per_variant; // blocks never straddle groups
    const VariantDesc vd = p.variants[variant];
    {   // stage the threshold tables and this wavefront's chain
        const uint32_t n4 = (vd.cpt_len + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.chroma_tab_off);
        for (uint32_t i = tid; i < n4; i += nthr) dst[i] = src[i];
        // the block's chains are consecutive: a lane group of chroma_wpb lanes reads one variable's bytes of all of them (one
        // segment of the global row) -- per lane and variable it would be one 64-byte line per byte
        const uint32_t wpb = p.chroma_wpb, chain0 = blockIdx.x * wpb;
        for (uint32_t i = tid; i < p.n_vars * wpb; i += nthr) {
            const uint32_t v = i / wpb, c = i % wpb;
            lds[c * p.chroma_row + v] = p.state[(size_t)v * p.pitch + chain0 + c];
        }
    }
    __syncthreads();
    const uint32_t row = lds_addr_c(lds) + wave * p.chroma_row;          // absolute LDS address of my chain's row
    const uint32_t tab = lds_addr_c(lds) + p.chroma_tab_off;
    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint64_t step0 = ((uint64_t)p.step0_hi << 32) | p.step0_lo;
    // Any range of steps [step0, step1): whole sweeps in the main loop, and -- so that chain groups that sweep different numbers of
    // variables (collapsed groups) share a launch, and launches need no generic head or tail -- a partial first and / or last sweep
    // around it, in which a lane takes part when its position lies in the range (the same code, with the range test; at most two
    // sweeps per launch).  [The range test inside the main loop, or one loop with two copies of its body, cost the kernel a third
    // of its rate: 1.27e12 -> 8.8e11 on Grids_13.]
    const uint64_t step1 = step0 + p.n_steps;                            // (n_steps >= 1)
    const uint64_t sw_first = step0 / vd.n_free, sw_last = (step1 - 1u) / vd.n_free;
    const uint32_t p_lo = (uint32_t)(step0 - sw_first * vd.n_free), p_hi = (uint32_t)(step1 - 1u - sw_last * vd.n_free);
    const bool part_first = PART && (p_lo != 0u || (sw_first == sw_last && p_hi != vd.n_free - 1u)); // the first sweep is partial (or the only, partial one)
    const bool part_last = PART && sw_last != sw_first && p_hi != vd.n_free - 1u;
    const uint64_t sweep0 = sw_first + (part_first ? 1u : 0u);          // the whole sweeps: [sweep0, sweep0 + n_sweeps)
    const uint32_t n_sweeps = (uint32_t)(sw_last + 1u - (part_last ? 1u : 0u) - sweep0);

    // my variable of every lane slot: loaded once
    bool on[NSLOT];
    uint32_t own[NSLOT], tbl[NSLOT], nb[NSLOT][SLOTS], mw[NSLOT][SLOTS], pos[NSLOT], rk[NSLOT], ones[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
        const uint32_t k = (uint32_t)s < vd.csl_n ? p.chroma_slots[(size_t)(vd.csl_base + s) * 64u + lane] : 0xffffffffu;
        on[s] = k != 0xffffffffu;
        const uint32_t kk = on[s] ? k : 0u;
        const uint32_t *r = p.recs_chroma + (size_t)(vd.rec_base + kk) * 16u;
        own[s] = row + r[0];
        rk[s] = kk;
        tbl[s] = tab + r[2];
        pos[s] = p.sys_pos[vd.free_base + kk];
        ones[s] = 0;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const uint32_t w = r[4 + i];
            nb[s][i] = row + (w & 0xffffu);
            mw[s][i] = w; // the multiplier sits in the high half: V_MAD_U32_U16 op_sel takes it from there
        }
    }

    // one sweep restricted to the positions [lo, hi]
    auto partial_sweep = [&](const uint64_t sweep, const uint32_t lo, const uint32_t hi) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            if ((uint32_t)s >= vd.csl_n) break; // (uniform)
            const u32x4 d = philox_sys_draw(sweep, pos[s], gchain, p.seed_lo, p.seed_hi);
            const uint32_t wq = (uint32_t)sweep & 3u;
            const uint32_t u = wq == 0 ? d.x : wq == 1 ? d.y : wq == 2 ? d.z : d.w;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            uint32_t a = tbl[s];
#pragma unroll
            for (int i = SLOTS - 1; i >= 0; --i) a = mad_u16_hi_c(mw[s][i], (uint32_t)*(lds_u8c_t *)(uintptr_t)nb[s][i], a);
            const uint32_t T = *(lds_u32c_t *)(uintptr_t)a;
            const bool val = u >= T;
            if (on[s] && pos[s] >= lo && pos[s] <= hi) {
                *(lds_u8c_t *)(uintptr_t)own[s] = val ? (unsigned char)1 : (unsigned char)0;
                if constexpr (COUNT) ones[s] += val ? 1u : 0u;
            }
        }
    };
    if constexpr (PART) if (part_first) partial_sweep(sw_first, p_lo, sw_first == sw_last ? p_hi : vd.n_free - 1u);
    const PrioRotor prio; // (wave_prio.h)
    for (uint64_t sq = sweep0 >> 2; (sq << 2) < sweep0 + n_sweeps; ++sq) {
        prio.tick((uint32_t)sq, p.prio_shift);
        uint32_t U[NSLOT][4];
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const u32x4 d = philox_sys_draw(sq << 2, pos[s], gchain, p.seed_lo, p.seed_hi);
            U[s][0] = d.x; U[s][1] = d.y; U[s][2] = d.z; U[s][3] = d.w;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t sweep = (sq << 2) + (uint64_t)w;
            if (sweep < sweep0 || sweep >= sweep0 + n_sweeps) continue; // (uniform)
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                if ((uint32_t)s >= vd.csl_n) break; // (uniform)
                // the class before this one has committed: in-order LDS in hardware, the fence for the compiler
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                uint32_t x[SLOTS];
#pragma unroll
                for (int i = 0; i < SLOTS; ++i) x[i] = (uint32_t)*(lds_u8c_t *)(uintptr_t)nb[s][i];
                uint32_t a = tbl[s];
#pragma unroll
                for (int i = SLOTS - 1; i >= 0; --i) a = mad_u16_hi_c(mw[s][i], x[i], a);
                const uint32_t T = *(lds_u32c_t *)(uintptr_t)a;
                const bool val = U[s][w] >= T;
                if (on[s]) {
                    *(lds_u8c_t *)(uintptr_t)own[s] = val ? (unsigned char)1 : (unsigned char)0;
                    if constexpr (COUNT) ones[s] += val ? 1u : 0u;
                }
            }
        }
    }
    if constexpr (PART) if (part_last) partial_sweep(sw_last, 0u, p_hi);
    __syncthreads();
    {
        const uint32_t wpb = p.chroma_wpb, chain0 = blockIdx.x * wpb;
        for (uint32_t i = tid; i < p.n_vars * wpb; i += nthr) {
            const uint32_t v = i / wpb, c = i % wpb;
            p.state[(size_t)v * p.pitch + chain0 + c] = lds[c * p.chroma_row + v];
        }
    }
    if constexpr (COUNT) {
        // Every chain of the block counted its own variables' ones in registers; the block adds them up in LDS (histogram indexed
        // by rank in the free list: (zeros, ones) pairs) and sends ONE atomic per counter to the global histogram -- per lane and
        // chain it was 52 M atomics on 200 addresses for 262 144 chains, three times the sweeps' own time.
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + p.chroma_hist_off);
        for (uint32_t i = tid; i < 2u * vd.n_free; i += nthr) h[i] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
            if (on[s] && (uint32_t)s < vd.csl_n) {
                // updates of my variable in this launch: the whole sweeps + the partial ones my position lies in
                uint32_t done = n_sweeps;
                if constexpr (PART) done += ((part_first && pos[s] >= p_lo && (sw_first != sw_last || pos[s] <= p_hi)) ? 1u : 0u) + ((part_last && pos[s] <= p_hi) ? 1u : 0u);
                atomicAdd(&h[2u * rk[s] + 1u], ones[s]);
                atomicAdd(&h[2u * rk[s]], done - ones[s]);
            }
        __syncthreads();
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = tid; i < vd.n_free; i += nthr) {
            const uint32_t v = p.free_vars[vd.free_base + i];
            const uint32_t ho = p.cprogs[vd.cprog_base + p.cprog_off[vd.cprogoff_base + v] + 1u];
            if (h[2u * i]) atomicAdd(&gc[ho], (unsigned long long)h[2u * i]);
            if (h[2u * i + 1u]) atomicAdd(&gc[ho + 1u], (unsigned long long)h[2u * i + 1u]);
        }
    }
}

// The same sweep for models with more lane slots or neighbours than fit registers (Segmentation_11: 7 lane slots, 8 neighbours;
// any model after a collapse): the lane slots' records -- {own byte, rank, table address, position; SLOTS x (neighbour byte |
// 4 * multiplier << 16)} -- are staged ONCE per block in LDS, [slot][lane], and read back per update (three 16-byte reads at
// a 48-byte lane stride: conflict-free); only the draws of the current 4 sweeps and the ones counters stay in registers.
template <int SLOTS, bool COUNT, bool PART = false> // SLOTS = 8 or 12 neighbour words per record; PART: see k_sweep_chroma
__global__ void __launch_bounds__(1024) k_sweep_chroma_l(const FastParams p) {
    constexpr uint32_t NSLOT = 8, RB = 16u + 4u * SLOTS;
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, nthr = blockDim.x;
    const uint32_t chain = blockIdx.x * p.chroma_wpb + wave;
    const uint32_t variant = (blockIdx.x * p.chroma_wpb) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    const uint32_t tab = lds_addr_c(lds) + p.chroma_tab_off;
    {
        const uint32_t n4 = (vd.cpt_len + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.chroma_tab_off);
        for (uint32_t i = tid; i < n4; i += nthr) dst[i] = src[i];
        const uint32_t wpb = p.chroma_wpb, chain0 = blockIdx.x * wpb;
        for (uint32_t i = tid; i < p.n_vars * wpb; i += nthr) {
            const uint32_t v = i / wpb, c = i % wpb;
            lds[c * p.chroma_row + v] = p.state[(size_t)v * p.pitch + chain0 + c];
        }
        // the lane slots' records: [slot][lane], RB bytes each; an unused lane gets a record that reads and "writes" nothing
        for (uint32_t i = tid; i < vd.csl_n * 64u; i += nthr) {
            const uint32_t k = p.chroma_slots[(size_t)vd.csl_base * 64u + i];
            uint32_t *d = reinterpret_cast<uint32_t *>(lds + p.chroma_rec_off + (size_t)i * RB);
            const uint32_t kk = k == 0xffffffffu ? 0u : k;
            const uint32_t *r = p.recs_chroma + (size_t)(vd.rec_base + kk) * 16u;
            d[0] = r[0];
            d[1] = k;                                   // rank in the free list, 0xffffffff = lane not in use
            d[2] = tab + r[2];
            d[3] = p.sys_pos[vd.free_base + kk];
            for (uint32_t j = 0; j < (uint32_t)SLOTS; ++j) d[4 + j] = r[4 + j];
        }
    }
    __syncthreads();
    const uint32_t row = lds_addr_c(lds) + wave * p.chroma_row;
    const uint32_t recs = lds_addr_c(lds) + p.chroma_rec_off + lane * RB;
    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint64_t step0 = ((uint64_t)p.step0_hi << 32) | p.step0_lo;
    const uint64_t step1 = step0 + p.n_steps; // any range of steps: partial first / last sweep around the whole ones (see k_sweep_chroma)
    const uint64_t sw_first = step0 / vd.n_free, sw_last = (step1 - 1u) / vd.n_free;
    const uint32_t p_lo = (uint32_t)(step0 - sw_first * vd.n_free), p_hi = (uint32_t)(step1 - 1u - sw_last * vd.n_free);
    const bool part_first = PART && (p_lo != 0u || (sw_first == sw_last && p_hi != vd.n_free - 1u));
    const bool part_last = PART && sw_last != sw_first && p_hi != vd.n_free - 1u;
    const uint64_t sweep0 = sw_first + (part_first ? 1u : 0u);
    const uint32_t n_sweeps = (uint32_t)(sw_last + 1u - (part_last ? 1u : 0u) - sweep0);
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4u lds_u4_t;
    uint32_t ones[NSLOT], posr[NSLOT];
#pragma unroll
    for (int s = 0; s < (int)NSLOT; ++s) {
        ones[s] = 0;
        posr[s] = (uint32_t)s < vd.csl_n ? ((lds_u4_t *)(uintptr_t)(recs + (uint32_t)s * 64u * RB))->w : 0u;
    }
    auto partial_sweep = [&](const uint64_t sweep, const uint32_t lo, const uint32_t hi) { // one sweep restricted to the positions [lo, hi]
#pragma unroll
        for (int s = 0; s < (int)NSLOT; ++s) {
            if ((uint32_t)s >= vd.csl_n) break;
            const u32x4 d = philox_sys_draw(sweep, posr[s], gchain, p.seed_lo, p.seed_hi);
            const uint32_t wq = (uint32_t)sweep & 3u;
            const uint32_t u = wq == 0 ? d.x : wq == 1 ? d.y : wq == 2 ? d.z : d.w;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t ra = recs + (uint32_t)s * 64u * RB;
            const v4u h = *(lds_u4_t *)(uintptr_t)ra;
            uint32_t a = h.z;
#pragma unroll
            for (int i = SLOTS - 1; i >= 0; --i) {
                const uint32_t wv = *(lds_u32c_t *)(uintptr_t)(ra + 16u + 4u * (uint32_t)i);
                a = mad_u16_hi_c(wv, (uint32_t)*(lds_u8c_t *)(uintptr_t)(row + (wv & 0xffffu)), a);
            }
            const uint32_t T = *(lds_u32c_t *)(uintptr_t)a;
            const bool val = u >= T;
            if (h.y != 0xffffffffu && posr[s] >= lo && posr[s] <= hi) {
                *(lds_u8c_t *)(uintptr_t)(row + h.x) = val ? (unsigned char)1 : (unsigned char)0;
                if constexpr (COUNT) ones[s] += val ? 1u : 0u;
            }
        }
    };
    if constexpr (PART) if (part_first) partial_sweep(sw_first, p_lo, sw_first == sw_last ? p_hi : vd.n_free - 1u);
    const PrioRotor prio; // (wave_prio.h)
    for (uint64_t sq = sweep0 >> 2; (sq << 2) < sweep0 + n_sweeps; ++sq) {
        prio.tick((uint32_t)sq, p.prio_shift);
        uint32_t U[NSLOT][4];
#pragma unroll
        for (int s = 0; s < (int)NSLOT; ++s) {
            if ((uint32_t)s >= vd.csl_n) break;
            const u32x4 d = philox_sys_draw(sq << 2, posr[s], gchain, p.seed_lo, p.seed_hi);
            U[s][0] = d.x; U[s][1] = d.y; U[s][2] = d.z; U[s][3] = d.w;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t sweep = (sq << 2) + (uint64_t)w;
            if (sweep < sweep0 || sweep >= sweep0 + n_sweeps) continue;
#pragma unroll
            for (int s = 0; s < (int)NSLOT; ++s) {
                if ((uint32_t)s >= vd.csl_n) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const uint32_t ra = recs + (uint32_t)s * 64u * RB;
                const v4u h = *(lds_u4_t *)(uintptr_t)ra;
                v4u sw[SLOTS / 4];
#pragma unroll
                for (int i = 0; i < SLOTS / 4; ++i) sw[i] = *(lds_u4_t *)(uintptr_t)(ra + 16u + 16u * (uint32_t)i);
                uint32_t x[SLOTS];
#pragma unroll
                for (int i = 0; i < SLOTS; ++i) {
                    const uint32_t wv = i % 4 == 0 ? sw[i / 4].x : i % 4 == 1 ? sw[i / 4].y : i % 4 == 2 ? sw[i / 4].z : sw[i / 4].w;
                    x[i] = (uint32_t)*(lds_u8c_t *)(uintptr_t)(row + (wv & 0xffffu));
                }
                uint32_t a = h.z;
#pragma unroll
                for (int i = SLOTS - 1; i >= 0; --i) {
                    const uint32_t wv = i % 4 == 0 ? sw[i / 4].x : i % 4 == 1 ? sw[i / 4].y : i % 4 == 2 ? sw[i / 4].z : sw[i / 4].w;
                    a = mad_u16_hi_c(wv, x[i], a);
                }
                const uint32_t T = *(lds_u32c_t *)(uintptr_t)a;
                const bool val = U[s][w] >= T;
                if (h.y != 0xffffffffu) {
                    *(lds_u8c_t *)(uintptr_t)(row + h.x) = val ? (unsigned char)1 : (unsigned char)0;
                    if constexpr (COUNT) ones[s] += val ? 1u : 0u;
                }
            }
        }
    }
    if constexpr (PART) if (part_last) partial_sweep(sw_last, 0u, p_hi);
    __syncthreads();
    {
        const uint32_t wpb = p.chroma_wpb, chain0 = blockIdx.x * wpb;
        for (uint32_t i = tid; i < p.n_vars * wpb; i += nthr) {
            const uint32_t v = i / wpb, c = i % wpb;
            p.state[(size_t)v * p.pitch + chain0 + c] = lds[c * p.chroma_row + v];
        }
    }
    if constexpr (COUNT) {
        uint32_t *hh = reinterpret_cast<uint32_t *>(lds + p.chroma_hist_off);
        for (uint32_t i = tid; i < 2u * vd.n_free; i += nthr) hh[i] = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < (int)NSLOT; ++s) {
            if ((uint32_t)s >= vd.csl_n) break;
            const uint32_t k = ((lds_u4_t *)(uintptr_t)(recs + (uint32_t)s * 64u * RB))->y;
            if (k != 0xffffffffu) {
                uint32_t done = n_sweeps;
                if constexpr (PART) done += ((part_first && posr[s] >= p_lo && (sw_first != sw_last || posr[s] <= p_hi)) ? 1u : 0u) + ((part_last && posr[s] <= p_hi) ? 1u : 0u);
                atomicAdd(&hh[2u * k + 1u], ones[s]);
                atomicAdd(&hh[2u * k], done - ones[s]);
            }
        }
        __syncthreads();
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = tid; i < vd.n_free; i += nthr) {
            const uint32_t v = p.free_vars[vd.free_base + i];
            const uint32_t ho = p.cprogs[vd.cprog_base + p.cprog_off[vd.cprogoff_base + v] + 1u];
            if (hh[2u * i]) atomicAdd(&gc[ho], (unsigned long long)hh[2u * i]);
            if (hh[2u * i + 1u]) atomicAdd(&gc[ho + 1u], (unsigned long long)hh[2u * i + 1u]);
        }
    }
}

template <int NSLOT, int SLOTS>
static hipError_t launch_chroma_t(const FastParams &p, uint32_t blocks, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64u * p.chroma_wpb), lds, s, p);
        return hipGetLastError();
    };
    if (p.chroma_part) return p.counting ? go(&k_sweep_chroma<NSLOT, SLOTS, true, true>) : go(&k_sweep_chroma<NSLOT, SLOTS, false, true>);
    return p.counting ? go(&k_sweep_chroma<NSLOT, SLOTS, true>) : go(&k_sweep_chroma<NSLOT, SLOTS, false>);
}

template <int SLOTS>
static hipError_t launch_chroma_l_t(const FastParams &p, uint32_t blocks, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64u * p.chroma_wpb), lds, s, p);
        return hipGetLastError();
    };
    if (p.chroma_part) return p.counting ? go(&k_sweep_chroma_l<SLOTS, true, true>) : go(&k_sweep_chroma_l<SLOTS, false, true>);
    return p.counting ? go(&k_sweep_chroma_l<SLOTS, true>) : go(&k_sweep_chroma_l<SLOTS, false>);
}

hipError_t launch_sweep_chroma(const FastParams &p, int n_lane_slots, int slots, uint32_t blocks, uint32_t lds, hipStream_t s) {
    if (p.chroma_rec_off) { // records staged in LDS: up to 8 lane slots, 12 neighbours
        if (n_lane_slots > 8 || slots > 12) return hipErrorInvalidValue;
        return slots <= 8 ? launch_chroma_l_t<8>(p, blocks, lds, s) : launch_chroma_l_t<12>(p, blocks, lds, s);
    }
    if (n_lane_slots > 4 || slots > 8) return hipErrorInvalidValue;
    if (n_lane_slots <= 2) return slots <= 4 ? launch_chroma_t<2, 4>(p, blocks, lds, s) : launch_chroma_t<2, 8>(p, blocks, lds, s);
    return slots <= 4 ? launch_chroma_t<4, 4>(p, blocks, lds, s) : launch_chroma_t<4, 8>(p, blocks, lds, s);
}

} // namespace gr
