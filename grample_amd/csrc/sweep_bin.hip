// Fast path of K1 for the headline shape: all selectable variables binary,
// conditionals memoised (k_build_cpt), chain count a multiple of 64.
//
// Why a second kernel: at the BASELINE size (65 536 chains = 1 024 wavefronts)
// there is ONE wavefront per SIMD.  Nothing hides latency and every instruction,
// scalar ones included, costs a 4-cycle issue slot.  The generic k_sweep spends
// ~1 600 cycles per step there on a chain of dependent scalar loads
// (free[k] -> prog_off[v] -> program words) and ~100 SALU instructions of
// scan-stream Philox per 4 steps.  Here:
//   * K0 k_scan_plan (full-occupancy pre-kernel) evaluates the per-wavefront
//     scan stream for the whole launch: plan[wave][step] = byte offset of the
//     selected variable's RECORD | "does not commute with an earlier step of
//     its quad".  The sweep reads 4 steps with one s_load_dwordx4.
//   * a record is one 64-byte line holding everything a step needs (state row
//     offset, histogram offset, threshold-table offset, neighbour row offsets
//     and multipliers), fetched with a single scalar load one quad AHEAD, so
//     its latency hides behind the draw-stream Philox of the next iteration.
//   * the 4 steps of a quad are straight-line code: 4*SLOTS state-byte reads,
//     4 table words, 4 compares in flight together; a step that does not
//     commute (plan bit) is recomputed after the commits before it.
// Same Philox streams, same thresholds, same commit order as k_sweep: results
// are bit-identical (tests: memoised == direct == sequential == emulator).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"
#include "philox.h"
#include "wave_prio.h"

// This file is compiled five times (Makefile): GR_PAR4_PART 0 = everything but the instantiations of k_sweep_par4, parts
// 1..4 = those, by record shape -- 330 instantiations of that kernel in one translation unit took four minutes to build.
#ifndef GR_PAR4_PART
#define GR_PAR4_PART 0
#endif

namespace gr {

__device__ __forceinline__ uint32_t uni_b(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

#if GR_PAR4_PART == 0
// K0: scan plan.  One thread per (64-chain group, quad of steps); blockIdx.y = group.  The block first stages its
// group's adjacency -- per selectable variable {id, neighbour count, neighbour ids} as u16 -- in LDS, so the
// dependency tests below are LDS reads instead of chains of dependent global loads.
__global__ void __launch_bounds__(256) k_scan_plan(const FastParams p) {
    extern __shared__ unsigned short sp_tab[];
    const uint32_t nq = p.n_steps >> 2;
    const uint32_t wave = blockIdx.y, qi = blockIdx.x * 256u + threadIdx.x;
    const uint32_t variant = (wave * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    const uint32_t S = 2u + p.plan_slots;
    for (uint32_t k = threadIdx.x; k < vd.n_free; k += 256u) {
        const uint32_t v = p.free_vars[vd.free_base + k];
        const uint32_t *cp = p.cprogs + vd.cprog_base + p.cprog_off[vd.cprogoff_base + v];
        const uint32_t nn = (cp[0] >> 8) & 0xffu;
        sp_tab[k * S] = (unsigned short)v;
        sp_tab[k * S + 1u] = (unsigned short)nn;
        for (uint32_t i = 0; i < nn; ++i) sp_tab[k * S + 2u + i] = (unsigned short)cp[3 + 2 * i];
    }
    __syncthreads();
    if (qi >= nq) return;
    const uint64_t q = ((((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2) + qi;
    const uint64_t gchain0 = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + (uint64_t)wave * 64u;
    const u32x4 sc = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)(gchain0 >> 6), DOM_SCAN, p.seed_lo, p.seed_hi);
    const uint32_t su[4] = {sc.x, sc.y, sc.z, sc.w};
    uint32_t k[4], v[4], out[4], level[4], maxlv = 0;
    uint32_t sys_p = p.sys_rank ? (uint32_t)((q << 2) % vd.n_free) : 0u; // systematic scan: step t updates position t mod n of the order
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        k[j] = mul_hi_u32(su[j], vd.n_free);
        if (p.sys_rank) {
            k[j] = p.sys_rank[vd.free_base + sys_p];
            if (++sys_p == vd.n_free) sys_p = 0;
        }
        const unsigned short *e = sp_tab + k[j] * S;
        v[j] = e[0];
        // dependency level of step j inside its quad: 1 + the highest level of an earlier step that
        // updates the same variable or one of its neighbours (blankets are symmetric, so this also
        // covers "a later step overwrites what an earlier one still has to read").  Level 0 = commutes
        // with everything before it.
        const uint32_t nn = e[1];
        uint32_t lv = 0, adj_any = 0;
        for (int pj = 0; pj < j; ++pj) {
            uint32_t dep = (v[pj] == v[j]) ? 1u : 0u, adj = 0;
            for (uint32_t i = 0; i < nn; ++i) adj |= (e[2 + i] == v[pj]) ? 1u : 0u;
            adj_any |= adj;
            if (dep | adj) lv = max(lv, level[pj] + 1u);
        }
        level[j] = lv;
        maxlv = max(maxlv, lv);
        // bit 0: reads a neighbour written earlier in the quad (k_sweep_bin recomputes it);
        // bits 1-2: level (k_sweep_par4)
        out[j] = (vd.rec_base + k[j]) * p.rec_stride | adj_any | (lv << 1) | (lv ? kPlanLevelBit : 0u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] |= maxlv << 3 | (maxlv ? kPlanDepsBit : 0u); // bits 3-4: highest level in the quad, in every word
    reinterpret_cast<uint4 *>(p.plan + (size_t)wave * p.plan_pitch)[qi] = make_uint4(out[0], out[1], out[2], out[3]);
}
#endif // GR_PAR4_PART == 0

struct Rec {
    uint32_t w[kRecDwords];
};

// Records are read-only for the whole launch: load them through the constant address space so
// that a uniform address always becomes a scalar (s_load) access served by the scalar cache.
typedef const __attribute__((address_space(4))) uint32_t *const_u32_ptr;
__device__ __forceinline__ void load_rec(Rec &r, const uint32_t *recs, uint32_t plan_word) {
    const_u32_ptr rp = (const_u32_ptr)(uintptr_t)(reinterpret_cast<const unsigned char *>(recs) + (plan_word & kPlanOffMask & ~63u));
#pragma unroll
    for (int i = 0; i < kRecDwords; ++i) r.w[i] = rp[i];
}

// Philox4x32-10 with the rounds exposed, so that groups of rounds can be placed
// where the kernel has latency to hide (same arithmetic as philox4x32_10()).
struct PhiloxSteps {
    uint32_t c0, c1, c2, c3, k0, k1;
    template <int N>
    __device__ __forceinline__ void rounds() {
#pragma unroll
        for (int r = 0; r < N; ++r) {
            const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
            const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
            const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
            const uint32_t n1 = (uint32_t)p1;
            const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
            const uint32_t n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += PHILOX_W0;
            k1 += PHILOX_W1;
        }
    }
};

// a * b + c on 24-bit operands in ONE instruction (left alone the compiler pairs the products of a configuration index
// through V_ADD3: 4 multiplies + 2 adds where 4 multiply-adds do)
__device__ __forceinline__ uint32_t mad_u24(const uint32_t a, const uint32_t b, const uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// (high half of w) * (low half of b) + c: the multiplier of a PACKED slot word (row offset | 4 * multiplier << 16) without
// extracting it first
__device__ __forceinline__ uint32_t mad_u16_hi(const uint32_t w, const uint32_t b, const uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
    return d;
}
// ... with the multiplier in a scalar register (chain-per-lane form: the record is wave-uniform)
__device__ __forceinline__ uint32_t mad_u24_s(const uint32_t a_uniform, const uint32_t b, const uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "s"(a_uniform), "v"(b), "v"(c));
    return d;
}

// gathers the blanket configuration of one step: issues the neighbour-state reads
template <int SLOTS, bool PACKED>
__device__ __forceinline__ void read_states(const Rec &r, const unsigned char *lds_tid, uint32_t (&x)[SLOTS]) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const uint32_t off = PACKED ? (r.w[4 + i] & 0xffffu) : r.w[4 + 2 * i];
        x[i] = (uint32_t)lds_tid[off];
    }
}
template <int SLOTS, bool PACKED>
__device__ __forceinline__ uint32_t table_word(const Rec &r, const uint32_t (&x)[SLOTS], const unsigned char *lds) {
    uint32_t cfg4 = 0;
    if constexpr (PACKED) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) cfg4 += __umul24(r.w[4 + i] >> 16, x[i]);
    } else { // one multiply-add per neighbour (the compiler's own choice is multiplies paired through V_ADD3: 3 per 2 neighbours)
        cfg4 = __umul24(r.w[5], x[0]);
#pragma unroll
        for (int i = 1; i < SLOTS; ++i) cfg4 = mad_u24_s(r.w[5 + 2 * i], x[i], cfg4);
    }
    return *reinterpret_cast<const uint32_t *>(lds + r.w[2] + cfg4);
}
template <int SLOTS, bool PACKED>
__device__ __forceinline__ bool sample_rec(const Rec &r, const unsigned char *lds_tid, const unsigned char *lds, const uint32_t u) {
    uint32_t x[SLOTS];
    read_states<SLOTS, PACKED>(r, lds_tid, x);
    return u >= table_word<SLOTS, PACKED>(r, x, lds);
}

constexpr uint32_t kPlanChunk = 32; // quads per staged plan chunk (lanes 0..31 carry one quad each); the LDS ring holds two chunks per wave

template <int SLOTS, bool PACKED, int MODE> // MODE 0: burn-in, 1: count, 2: count + draw log (gr_advance_chain)
__global__ void __launch_bounds__(1024) k_sweep_bin(const FastParams p) {
    constexpr bool COUNT = MODE != 0, LOG = MODE == 2;
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t T = blockDim.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;
    const uint32_t chain = blockIdx.x * T + tid;
    const uint32_t variant = (blockIdx.x * T) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    unsigned char *st = lds + tid;
    {   // stage this group's threshold tables, the block's chain state, zero the wave histogram
        const uint32_t n4 = (vd.cpt_len + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.lds_tab_off);
        for (uint32_t i = tid; i < n4; i += T) dst[i] = src[i];
        for (uint32_t v = 0; v < p.n_vars; ++v) st[v * T] = p.state[(size_t)v * p.pitch + chain];
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + p.lds_hist_off) + wave * p.H;
        for (uint32_t i = lane; i < p.H; i += 64) h[i] = 0;
    }
    __syncthreads();

    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint32_t wave_local = uni_b(chain >> 6);
    const uint32_t hist_base = uni_b(p.lds_hist_off + wave * p.H * 4u); // LDS byte offset of this wave's histogram
    const uint64_t q0 = (((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2;
    const uint32_t nq = p.n_steps >> 2;
    // lanes 0/1 carry the (zeros, ones) histogram increments: d = sign * ones + base
    const int h_sign = lane == 0 ? -1 : 1;
    const int h_base = lane == 0 ? 64 : 0;
    const uint32_t h_addr = hist_base + lane * 4u;

    // ---- the wave's scan plan, staged through a private LDS ring of 2 x 64 quads.  Lane l of the
    // wave carries quad (chunk * 64 + l); one global_load_dwordx4 + one ds_write_b128 per 64 quads.
    // Vector loads on purpose: vmcnt is separate from the LDS/scalar counter and in order, so a
    // chunk prefetched half a chunk ahead never stalls an LDS wait.
    const uint4 *gplan = reinterpret_cast<const uint4 *>(p.plan + (size_t)wave_local * p.plan_pitch);
    uint4 *ring = reinterpret_cast<uint4 *>(lds + p.lds_plan_off + wave * (2u * kPlanChunk * 16u));
    const uint32_t nchunks = (nq + kPlanChunk - 1) / kPlanChunk;
    auto gload = [&](uint32_t chunk) -> uint4 {
        const uint32_t i = chunk * kPlanChunk + (lane & (kPlanChunk - 1u));
        return gplan[i < nq ? i : nq - 1];
    };
    if (lane < kPlanChunk) ring[lane] = gload(0);
    uint4 stage = gload(nchunks > 1 ? 1 : 0);
    auto ring_read = [&](uint32_t g) -> uint4 { return ring[(g < nq ? g : nq - 1) & (2u * kPlanChunk - 1u)]; };

    Rec R0, R1, R2, R3;
    uint32_t flags;
    {
        const uint4 P0 = ring_read(0);
        const uint32_t a = uni_b(P0.x), b = uni_b(P0.y), c = uni_b(P0.z), d = uni_b(P0.w);
        load_rec(R0, p.recs, a); load_rec(R1, p.recs, b); load_rec(R2, p.recs, c); load_rec(R3, p.recs, d);
        flags = (b & 1u) | ((c & 1u) << 1) | ((d & 1u) << 2);
    }
    uint4 PV = ring_read(1); // plan of the NEXT quad, as uniform VGPRs

    unsigned char *logrow = LOG && p.log ? p.log + (size_t)wave_local * p.log_pitch + p.log_off : nullptr;
    uint32_t lim = 0xffffffffu; // LOG: the group's own stopping time inside the window (SweepParams::limit)
    if constexpr (LOG)
        if (p.limit) {
            const uint32_t t = p.limit[wave_local];
            lim = t > p.log_off ? t - p.log_off : 0u;
        }
    auto commit = [&](const Rec &r, const bool val, const uint32_t sidx) { // val: the compare's own wave mask, counted as it is
        if (LOG && sidx >= lim) return;
        st[r.w[0]] = val ? (unsigned char)1 : (unsigned char)0;
        if constexpr (COUNT) {
            const uint32_t ones = (uint32_t)__popcll(__ballot(val));
            if (lane < 2) {
                uint32_t *hp = reinterpret_cast<uint32_t *>(lds + h_addr + r.w[1]);
                __hip_atomic_fetch_add(hp, (uint32_t)(h_sign * (int)ones + h_base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (LOG && logrow && lane == 1) logrow[sidx] = (unsigned char)ones;
            }
        }
    };

    const PrioRotor prio; // (wave_prio.h)
    for (uint32_t qi = 0; qi < nq; ++qi) {
        prio.tick(qi, p.prio_shift);
        const uint64_t q = q0 + qi;
        if ((qi & (kPlanChunk - 1u)) == kPlanChunk / 2u) {
            // mid-chunk: park the prefetched next chunk in the other half of the ring, fetch the one after it
            const uint32_t nc = qi / kPlanChunk + 1u;
            if (nc < nchunks) {
                if (lane < kPlanChunk) ring[(nc & 1u) * kPlanChunk + lane] = stage;
                if (nc + 1u < nchunks) stage = gload(nc + 1u);
            }
        }
        PhiloxSteps ph{(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi};
        // rounds 1-4 cover what is left of the record loads' scalar-cache latency (issued at the end of the previous quad)
        ph.rounds<4>();
        __builtin_amdgcn_sched_barrier(0);
        uint32_t x0[SLOTS], x1[SLOTS], x2[SLOTS], x3[SLOTS];
        read_states<SLOTS, PACKED>(R0, st, x0);
        read_states<SLOTS, PACKED>(R1, st, x1);
        read_states<SLOTS, PACKED>(R2, st, x2);
        read_states<SLOTS, PACKED>(R3, st, x3);
        __builtin_amdgcn_sched_barrier(0);
        ph.rounds<4>(); // rounds 5-8 cover the state reads
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t t0 = table_word<SLOTS, PACKED>(R0, x0, lds);
        const uint32_t t1 = table_word<SLOTS, PACKED>(R1, x1, lds);
        const uint32_t t2 = table_word<SLOTS, PACKED>(R2, x2, lds);
        const uint32_t t3 = table_word<SLOTS, PACKED>(R3, x3, lds);
        __builtin_amdgcn_sched_barrier(0);
        ph.rounds<2>(); // rounds 9-10 cover the table reads
        __builtin_amdgcn_sched_barrier(0);
        bool v0 = ph.c0 >= t0, v1 = ph.c1 >= t1, v2 = ph.c2 >= t2, v3 = ph.c3 >= t3;
        // commit in order; a step that reads what an earlier step of the quad wrote is recomputed first
        const uint32_t s0 = qi * 4u;
        if (flags == 0) {
            commit(R0, v0, s0); commit(R1, v1, s0 + 1); commit(R2, v2, s0 + 2); commit(R3, v3, s0 + 3);
        } else {
            commit(R0, v0, s0);
            if (flags & 1u) v1 = sample_rec<SLOTS, PACKED>(R1, st, lds, ph.c1);
            commit(R1, v1, s0 + 1);
            if (flags & 2u) v2 = sample_rec<SLOTS, PACKED>(R2, st, lds, ph.c2);
            commit(R2, v2, s0 + 2);
            if (flags & 4u) v3 = sample_rec<SLOTS, PACKED>(R3, st, lds, ph.c3);
            commit(R3, v3, s0 + 3);
        }
        // next quad: plan words -> SGPRs -> record loads (scalar cache), then the plan of the quad after it
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t a = uni_b(PV.x), b = uni_b(PV.y), c = uni_b(PV.z), d = uni_b(PV.w);
        load_rec(R0, p.recs, a); load_rec(R1, p.recs, b); load_rec(R2, p.recs, c); load_rec(R3, p.recs, d);
        flags = (b & 1u) | ((c & 1u) << 1) | ((d & 1u) << 2);
        PV = ring_read(qi + 2u);
    }

    for (uint32_t vv = 0; vv < p.n_vars; ++vv) p.state[(size_t)vv * p.pitch + chain] = st[vv * T];
    if constexpr (COUNT) {
        const uint32_t *h = reinterpret_cast<const uint32_t *>(lds + hist_base);
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = lane; i < p.H; i += 64) {
            const uint32_t c = h[i];
            if (c) atomicAdd(&gc[i], (unsigned long long)c);
        }
    }
}

// ---------------------------------------------------------------------------
// K1, step-parallel form: 4 lanes per chain, one per step of a quad.
//
// With few chains (65 536 = one wavefront per SIMD in the chain-per-lane form) a lone wavefront
// issues one instruction per 4 cycles and nothing hides its latencies.  Here a 256-lane block owns
// 64 chains (one scan group); lane = (chain c, step slot j): the 4 steps of a quad are gathered,
// drawn and committed side by side, so the same 65 536 chains run as 4 096 wavefronts (4 per SIMD).
// Steps of a quad that do not commute carry a dependency LEVEL in the plan (k_scan_plan): pass 0
// commits level 0, and only when the quad has higher levels are the conditionals re-gathered and
// the next level committed.  Every (chain, step) uses the same Philox word and the same threshold
// table as in the other kernels: bit-identical results.
//   state     lds[var * 80 + chain]           (80-byte rows: 4 different rows per access land on different banks; 64-byte rows
//             -- FastParams::par_stride -- for models so large that 80-byte row offsets no longer fit a packed record's 16 bits)
//   records   staged in LDS, read per lane (4 distinct records per access, broadcast inside a step slot)
//   lanes     lane = j * 16 + c: a 16-lane LDS lane group / DPP row is one step slot
//   draws     lane (j, c) computes Philox for quad 4*SQ + j of chain c; two pairs of gfx950 lane swaps transpose
//             the 4x4 words so that lane (j, c) ends up with word j of quads 4*SQ .. 4*SQ+3
//   counts    popcount of the compare mask's 16 bits of a step slot's DPP row; its lane 15 adds (zeros, ones) with one 64-bit LDS atomic
constexpr uint32_t kParStride = 80;

// LDS accesses by absolute LDS address.  The dynamic LDS array's base is a link-time symbol, so `lds + x` costs a
// VALU add (of 0) at every use; the per-lane kernels stage their records with the base folded into the histogram and
// table offsets and then address LDS by integer (the 4-lane form is VALU-issue-bound: 2 of ~28 instructions per step).
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}
__device__ __forceinline__ lds_u32_t *lds_at(uint32_t addr) { return (lds_u32_t *)(uintptr_t)addr; }

template <int SLOTS, bool PACKED>
struct RecL { // a record as per-lane VGPR values
    uint4 h;                                 // state row offset, hist offset, table offset, nn
    uint4 s[PACKED ? (SLOTS + 3) / 4 : (SLOTS + 1) / 2];
};

template <int SLOTS, bool PACKED>
__device__ __forceinline__ uint32_t comp4(const uint4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// The block histogram of the kernels that use the 4-lane records (k_sweep_par4, k_sweep_strict) is indexed by the
// variable's RANK in its group's free list -- words 2 * rank (zeros) and 2 * rank + 1 (ones), always an aligned 8-byte pair,
// whatever odd-Card evidence variables sit in front of it in the model's own layout -- and mapped to the global counters
// (the model's layout: histogram offset = word 1 of the variable's memoised program) when the block ends.
__device__ __forceinline__ void flush_rank_hist(const FastParams &p, const VariantDesc &vd, const uint32_t *h, unsigned long long *gc,
                                                const uint32_t tid, const uint32_t nthr) {
    for (uint32_t i = tid; i < vd.n_free; i += nthr) {
        const uint32_t v = p.free_vars[vd.free_base + i];
        const uint32_t hoff = p.cprogs[vd.cprog_base + p.cprog_off[vd.cprogoff_base + v] + 1u];
        const uint32_t c0 = h[2u * i], c1 = h[2u * i + 1u];
        if (c0) atomicAdd(&gc[hoff], (unsigned long long)c0);
        if (c1) atomicAdd(&gc[hoff + 1u], (unsigned long long)c1);
    }
}

// TABG 1: the threshold word comes from global memory (tabg + the record's byte offset) instead of LDS
// TABG 2: split pool -- record word 2 with bit 31 set is a byte offset into the global pool, otherwise an absolute LDS address
template <int SLOTS, bool PACKED, int TABG = 0>
__device__ __forceinline__ uint32_t sample_l(const RecL<SLOTS, PACKED> &r, const unsigned char *lds_c, const unsigned char *tabg, const uint32_t u) {
    // The configuration index is a multiply-add chain that starts at the table address (record word 2), one instruction per
    // neighbour: V_MAD_U32_U24 on unpacked records, V_MAD_U32_U16 taking the multiplier from the high half of a packed slot
    // word.  The state bytes are read in slot order and consumed last first: LDS returns in order, so the first multiply-add
    // waits for all the reads once instead of each of them waiting for its own.
    uint32_t a = r.h.z; // absolute LDS address (staged records), or offset into the global pool
    auto walk = [&](auto lo, auto hi) {
        constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
        uint32_t x[HI - LO];
#pragma unroll
        for (int i = LO; i < HI; ++i) {
            const uint32_t off = PACKED ? (comp4<SLOTS, PACKED>(r.s[i / 4], i % 4) & 0xffffu) : comp4<SLOTS, PACKED>(r.s[i / 2], 2 * (i % 2));
            x[i - LO] = (uint32_t)lds_c[off];
        }
#pragma unroll
        for (int i = HI - 1; i >= LO; --i) {
            if constexpr (PACKED) a = mad_u16_hi(comp4<SLOTS, PACKED>(r.s[i / 4], i % 4), x[i - LO], a);
            else a = mad_u24(comp4<SLOTS, PACKED>(r.s[i / 2], 2 * (i % 2) + 1), x[i - LO], a);
        }
    };
    // 128-byte records (13..16 slots: a collapsed group, where only the few neighbours of the collapsed variable have more
    // than 8 neighbours): slots 9.. are walked only when one of the wavefront's four steps needs them
    constexpr int HEAD = SLOTS > 12 ? 8 : SLOTS;
    walk(std::integral_constant<int, 0>{}, std::integral_constant<int, HEAD>{});
    if constexpr (SLOTS > HEAD)
        if (__any(r.h.w > (uint32_t)HEAD)) walk(std::integral_constant<int, HEAD>{}, std::integral_constant<int, SLOTS>{});
    uint32_t T;
    if constexpr (TABG == 2) {
        if ((int)a < 0) T = *reinterpret_cast<const uint32_t *>(tabg + (a & 0x7fffffffu));
        else T = *lds_at(a);
    } else {
        T = TABG ? *reinterpret_cast<const uint32_t *>(tabg + a) : *lds_at(a);
    }
    return (u >= T) ? 1u : 0u;
}

// sample_l in two halves, so that independent work can be placed under the state reads' latency
template <int SLOTS, bool PACKED>
__device__ __forceinline__ void gather_l(const RecL<SLOTS, PACKED> &r, const unsigned char *lds_c, uint32_t (&x)[SLOTS]) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const uint32_t off = PACKED ? (comp4<SLOTS, PACKED>(r.s[i / 4], i % 4) & 0xffffu) : comp4<SLOTS, PACKED>(r.s[i / 2], 2 * (i % 2));
        x[i] = (uint32_t)lds_c[off];
    }
}
template <int SLOTS, bool PACKED>
__device__ __forceinline__ uint32_t table_l(const RecL<SLOTS, PACKED> &r, const uint32_t (&x)[SLOTS], const unsigned char *lds) {
    uint32_t cfg4 = 0; // (a product tree, not sample_l's multiply-add chain: this helper serves the strict kernel, one wavefront per
                       // SIMD at the headline size, where the chain's dependent latency shows -- measured 3.7e11 vs 2.9e11)
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const uint32_t m4 = PACKED ? (comp4<SLOTS, PACKED>(r.s[i / 4], i % 4) >> 16) : comp4<SLOTS, PACKED>(r.s[i / 2], 2 * (i % 2) + 1);
        cfg4 += __umul24(m4, x[i]);
    }
    return *lds_at(r.h.z + cfg4); // r.h.z: absolute LDS address (staged records)
}

// HOIST: fetch the four records of an iteration together -- one LDS round trip instead of four.  Worth 7-12 % while a
// SIMD holds fewer than ~3 wavefronts (latency-bound); with 4 per SIMD (65 536 chains) the VALU port is the limit and the
// extra registers cost 1 %, so the launcher picks by occupancy.
// RECMODE 2: the records come from global memory instead (vector loads through L1, prefetched one iteration ahead):
// the LDS pipe is this form's tightest resource and the record reads are 14 of its 48 cycles per step, while the
// vector-memory path is otherwise idle.
// TABG (with RECMODE 2 only): threshold pools too large for LDS (Pedigree without evidence: 36 K words, dv-rel_3: 512 K) are read
// from global memory -- one dependent L2 access per step instead of an LDS one, hidden by the block's other wavefronts; still
// an order of magnitude cheaper than evaluating the conditional from the factor tables.  TABG 2: the first p.par_tab_words
// words of the pool (the small tables: compiled_model.cpp orders the pool by table size) are staged in LDS, the rest is read
// from global memory; each record says where its table lives.
template <int SLOTS, bool PACKED, int MODE, int RECMODE, int TABG = 0> // MODE 0: burn-in, 1: count, 2: count + draw log (gr_advance_chain)
// (launch bounds: the instantiations of the small-blanket models -- Grids: 98 registers -- are held to 96, so that a CU takes FIVE blocks
// when a launch has them: 81 920 chains 5.9e11 -> 9.1e11, 66 560 chains 4.8e11 -> 7.4e11; at 65 536 chains -- 4 blocks per CU by the chain
// count -- nothing changes (8.4e11).  The logged form would spill 21 registers and keeps its 4.)
__global__ void __launch_bounds__(256, (SLOTS <= 4 && TABG == 0 && RECMODE == 2 && MODE != 2) ? 5 : 1) k_sweep_par4(const FastParams p) {
    constexpr bool COUNT = MODE != 0, LOG = MODE == 2;
    extern __shared__ __align__(16) unsigned char lds[];
    // Block shape: 256 lanes = one 64-chain scan group (4 wavefronts of 16 chains), or -- "spread", when the launch has
    // fewer wavefronts than the chip has SIMDs -- 64 lanes = one of the group's 4 wavefronts per block, so that the four land
    // on different CUs (each with its own LDS pipe) instead of on the four SIMDs of one.
    const uint32_t tid = threadIdx.x, lane = tid & 63u, nthr = blockDim.x;
    const bool spread = nthr == 64u;
    const uint32_t group = spread ? blockIdx.x >> 2 : blockIdx.x;    // 64-chain scan group
    const uint32_t wave = spread ? blockIdx.x & 3u : tid >> 6;        // which 16 chains of the group
    const uint32_t lwave = tid >> 6;                                  // wavefront inside the block (its LDS plan ring)
    // lane = step slot * 16 + chain: every 16-lane LDS lane group reads ONE record / plan word
    // (broadcast, no bank conflict) and a DPP row reduction sums a step's draws
    const uint32_t j = lane >> 4;                    // step slot inside the quad
    const uint32_t cb = wave * 16u + (lane & 15u);   // chain inside the block
    const uint32_t chain = group * 64u + cb;
    const uint32_t variant = (group * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime(); // (for p.dbg_times)
    {   // stage tables + records, this block's 64 chains, zero the block histogram
        const uint32_t n4 = ((TABG == 2 ? min(vd.cpt_len, p.par_tab_words) : vd.cpt_len) + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.par_tab_off);
        if constexpr (TABG != 1)
            for (uint32_t i = tid; i < n4; i += nthr) dst[i] = src[i];
        const uint4 *rs = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(p.recs_par) + (size_t)vd.rec_base * p.rec_stride);
        uint4 *rd = reinterpret_cast<uint4 *>(lds + p.par_rec_off);
        const uint32_t per = p.rec_stride / 16u, base = lds_addr_of(lds);
        if constexpr (RECMODE != 2)
            for (uint32_t i = tid; i < vd.n_free * per; i += nthr) {
                uint4 q = rs[i];
                if (i % per == 0) { q.y += base; q.z += base; } // histogram and table offsets -> absolute LDS addresses
                rd[i] = q;
            }
        for (uint32_t v = j; v < p.n_vars; v += 4) lds[v * p.par_stride + cb] = p.state[(size_t)v * p.pitch + chain];
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + p.par_hist_off);
        for (uint32_t i = tid; i < 2u * vd.n_free; i += nthr) h[i] = 0;
    }
    __syncthreads();

    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint64_t q0 = (((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2;
    const uint32_t nq = p.n_steps >> 2; // multiple of 4
    const unsigned char *lds_c = lds + cb;
    const unsigned char *tabp = TABG ? reinterpret_cast<const unsigned char *>(p.cpt + vd.cpt_base) : lds; // where sample_l finds the thresholds
    const uint32_t rec_bias = p.par_rec_off - vd.rec_base * p.rec_stride;

    // the block's scan plan (all 4 waves walk the same plan), staged per wave through an LDS ring
    const uint4 *gplan = reinterpret_cast<const uint4 *>(p.plan + (size_t)group * p.plan_pitch);
    uint4 *ring = reinterpret_cast<uint4 *>(lds + p.par_plan_off + lwave * (2u * kPlanChunk * 16u));
    const uint32_t *ring_w = reinterpret_cast<const uint32_t *>(ring) + j;
    const uint32_t nchunks = (nq + kPlanChunk - 1) / kPlanChunk;
    auto gload = [&](uint32_t chunk) -> uint4 {
        const uint32_t i = chunk * kPlanChunk + (lane & (kPlanChunk - 1u));
        return gplan[i < nq ? i : nq - 1];
    };
    uint4 stage = make_uint4(0, 0, 0, 0);
    if constexpr (RECMODE != 2) {
        if (lane < kPlanChunk) ring[lane] = gload(0);
        stage = gload(nchunks > 1 ? 1 : 0);
    }
    const bool tail = (lane & 15u) == 15u;                 // holds the row sums after the DPP reduction

    // commit the step slots whose level equals `pass`: write the new values, count them.
    // The 16 chains of a step slot are one DPP row: row_shr 1,2,4,8 leave the number of ones in lane 15.
    unsigned char *logrow = LOG && p.log ? p.log + (size_t)(group * 4u + wave) * p.log_pitch + p.log_off : nullptr;
    // A step's value is written at its own level's pass; the counting happens once per step slot after the last pass
    // (histogram adds commute).
    // LOG: a group executes only the steps below its own stopping time `lim`.  The check is made per ITERATION (16 steps): the
    // iterations entirely below it run the unchecked code (CHK = false), the one it cuts the per-step checks, and the loop ends
    // there -- the later steps of the window leave the group untouched.
    uint32_t lim = 0xffffffffu; // LOG: the group's own stopping time inside the window (SweepParams::limit)
    if constexpr (LOG)
        if (p.limit) {
            const uint32_t t = p.limit[group];
            lim = t > p.log_off ? t - p.log_off : 0u;
        }
    auto store = [&](auto chk, const RecL<SLOTS, PACKED> &r, const bool val, const bool mine, const uint32_t sidx) {
        constexpr bool CHK = LOG && decltype(chk)::value;
        if (mine && !(CHK && sidx >= lim)) lds[r.h.x + cb] = val ? (unsigned char)1 : (unsigned char)0;
    };
    // The draws of a step are a wave mask (the compare's own result, in scalar registers; level passes merge masks with scalar
    // logic).  Ones of my step slot = popcount of the mask's 16 bits of my DPP row: (lo & rm_lo) | (hi & rm_hi) with two
    // per-lane constants selects them (V_AND, V_AND_OR), V_BCNT counts -- 4 VALU instructions per step with the 16 - ones
    // below, where prefix counts (V_MBCNT x 2, + own bit, row_bcast:15 difference) took 9.  [Tried: four scalar popcounts packed
    // into a word of which each lane takes its byte -- 2 VALU but 14 scalar instructions, 6 % slower: with 4 wavefronts per
    // SIMD every instruction of any kind costs an issue slot.  A DPP subtract (incl - incl of the previous row's lane 15 in
    // one instruction) does not exist: V_SUB and V_SUBREV with DPP both compute dpp(src0) - src1.]
    const uint32_t rm_lo = j < 2u ? 0xffffu << (16u * (j & 1u)) : 0u, rm_hi = j >= 2u ? 0xffffu << (16u * (j & 1u)) : 0u;
    // returns the ones of the step slot (the same in all 16 lanes of the row)
    auto count = [&](auto chk, const RecL<SLOTS, PACKED> &r, const unsigned long long drawn, const uint32_t sidx) -> uint32_t {
        constexpr bool CHK = LOG && decltype(chk)::value;
        uint32_t s1 = 0;
        if constexpr (COUNT) {
            if (CHK && uni_b(sidx & ~3u) >= lim) return 0u; // the whole quad is past the group's stopping time (limits are even,
                                                            // so a quad is cut between its steps 1 and 2 at most)
            s1 = (uint32_t)__builtin_popcount(((uint32_t)drawn & rm_lo) | ((uint32_t)(drawn >> 32) & rm_hi));
            if (tail && !(CHK && sidx >= lim)) {
                uint32_t zeros; // 16 - s1 (as C++ the compiler emits a negate and an add)
                asm("v_sub_u32_e32 %0, 16, %1" : "=v"(zeros) : "v"(s1));
                // (zeros, ones) of the variable are an aligned 8-byte pair of the rank-indexed histogram: one LDS atomic
                __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>((uintptr_t)r.h.y),
                                       ((unsigned long long)s1 << 32) | zeros, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // (MODE 2 is only launched with a log.)  The iteration the stopping time cuts logs step by step; the others log
                // their 16 steps with one store, below.  (A buffer store for the byte -- row as the resource, quad as scalar
                // offset, slot as a constant vector offset: no 64-bit address add -- measured 8 % SLOWER than the plain store.)
                if constexpr (CHK) logrow[sidx] = (unsigned char)s1;
            }
        }
        return s1;
    };

    // RECMODE 2: records of the NEXT iteration, in flight while the current one runs
    const unsigned char *recs_g = reinterpret_cast<const unsigned char *>(p.recs_par);
    // The global records hold offsets into the dynamic LDS array.  It starts at LDS address 0 because this kernel has no
    // static LDS (the dynamic array follows the static segment), so the offsets are used as absolute addresses without
    // two adds per step; should that ever change, fail loudly (error flag -> GR_E_STATE) instead of sampling from the
    // wrong table.
    if (RECMODE == 2 && lds_addr_of(lds) != 0u) { // every wave leaves at once; the host turns the flag into an error
        if (threadIdx.x == 0) atomicOr(p.err, 1u);
        return;
    }
    // Two register sets (A, B) hold the records and plan words of the current and of the next iteration and swap roles every
    // iteration (the loop below is unrolled by two): with one "next" set copied into one "current" set per iteration the
    // copy was 42 of the loop's 188 VALU instructions -- on the resource this kernel is bound by.
    constexpr int NREC = RECMODE == 2 ? 4 : 1;
    uint32_t pwA[4] = {0, 0, 0, 0}, pwB[4] = {0, 0, 0, 0};
    RecL<SLOTS, PACKED> rA[NREC], rB[NREC];
    // ... and its plan words straight from global memory, two iterations ahead (no LDS ring in this mode)
    // (buffer loads: the group's plan row is the resource, the quad a scalar offset, the lane's word j * 4 a constant vector
    // offset -- no address arithmetic on the VALU)
    const __amdgpu_buffer_rsrc_t plan_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(p.plan + (size_t)group * p.plan_pitch), 0, 0x7fffffff, 0x00020000);
    const uint32_t j4 = j * 4u;
    uint32_t pwa[4] = {0, 0, 0, 0};
    auto prefetch_plan = [&](const uint32_t gnext) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pwa[k] = __builtin_amdgcn_raw_buffer_load_b32(plan_rsrc, j4, min(gnext + (uint32_t)k, nq - 1u) * 16u, 0); // never past the plan
    };
    auto prefetch = [&](RecL<SLOTS, PACKED> (&rx)[NREC], uint32_t (&pwx)[4]) { // the iteration whose plan words are in pwa
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pwx[k] = pwa[k];
            const unsigned char *ga = recs_g + (pwx[k] & (kPlanOffMask & ~63u));
            rx[k % NREC].h = *reinterpret_cast<const uint4 *>(ga);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(rx[0].s) / sizeof(uint4)); ++i) rx[k % NREC].s[i] = *reinterpret_cast<const uint4 *>(ga + 16 + 16 * i);
        }
    };
    if constexpr (RECMODE == 2) { prefetch_plan(0); prefetch(rA, pwA); prefetch_plan(4); }

    // one iteration = 4 quads = 16 steps of every chain: (rc, pwc) hold its records and plan words, (rx, pwx) receive the next one's
    auto iteration = [&](auto chk, const uint32_t g0, RecL<SLOTS, PACKED> (&rc)[NREC], uint32_t (&pwc)[4], RecL<SLOTS, PACKED> (&rx)[NREC], uint32_t (&pwx)[4]) {
        if (RECMODE != 2 && (g0 & (kPlanChunk - 1u)) == kPlanChunk / 2u) {
            const uint32_t nc = g0 / kPlanChunk + 1u;
            if (nc < nchunks) {
                if (lane < kPlanChunk) ring[(nc & 1u) * kPlanChunk + lane] = stage;
                if (nc + 1u < nchunks) stage = gload(nc + 1u);
            }
        }
        // draws: my Philox call is quad g0 + j; transpose so that U[k] = my word of quad g0 + k
        const uint64_t q = q0 + g0 + j;
        const u32x4 W = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain,
                                      DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
        // 4x4 transpose (step slots x words) with gfx950's lane swaps instead of an LDS round trip: V_PERMLANE32_SWAP
        // exchanges the upper 32 lanes of one register with the lower 32 of another (slot bit 1 <-> word bit 1),
        // V_PERMLANE16_SWAP the odd 16-lane rows with the even rows (slot bit 0 <-> word bit 0)
        uint32_t pw[4], U[4];
        {
            const auto a = __builtin_amdgcn_permlane32_swap(W.x, W.z, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(W.y, W.w, false, false);
            const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
            const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
            U[0] = c[0]; U[1] = c[1]; U[2] = d[0]; U[3] = d[1];
        }
        constexpr bool HOIST = RECMODE != 0;
        RecL<SLOTS, PACKED> rq[RECMODE == 1 ? 4 : 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (RECMODE == 2) pw[k] = pwc[k];
            else pw[k] = ring_w[((g0 + k) & (2u * kPlanChunk - 1u)) * 4u];
        }
        // (the offsets in the global records are absolute LDS addresses as they are: see the base check above)
        if constexpr (RECMODE == 2) { prefetch(rx, pwx); prefetch_plan(g0 + 8u); }
        // RECMODE 1: the four records only depend on the plan: fetch them together (one LDS round trip instead of four)
        if constexpr (RECMODE == 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned char *ra = lds + rec_bias + (pw[k] & (kPlanOffMask & ~63u));
                rq[k].h = *reinterpret_cast<const uint4 *>(ra);
#pragma unroll
                for (int i = 0; i < (int)(sizeof(rq[k].s) / sizeof(uint4)); ++i) rq[k].s[i] = *reinterpret_cast<const uint4 *>(ra + 16 + 16 * i);
            }
        }
        uint32_t ones_k[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            RecL<SLOTS, PACKED> r;
            if constexpr (RECMODE == 2) {
                r = rc[k % NREC];
            } else if constexpr (HOIST) {
                r = rq[k];
            } else {
                const unsigned char *ra = lds + rec_bias + (pw[k] & (kPlanOffMask & ~63u));
                r.h = *reinterpret_cast<const uint4 *>(ra);
#pragma unroll
                for (int i = 0; i < (int)(sizeof(r.s) / sizeof(uint4)); ++i) r.s[i] = *reinterpret_cast<const uint4 *>(ra + 16 + 16 * i);
            }
            const uint32_t sidx = (g0 + k) * 4u + j;
            const bool val = sample_l<SLOTS, PACKED, TABG>(r, lds_c, tabp, U[k]) != 0u;
            unsigned long long drawn = __ballot(val); // the compare's own mask
            store(chk, r, val, pw[k] < kPlanLevelBit, sidx); // level 0: commit in the first pass
            if (__any(pw[k] >= kPlanDepsBit)) { // the quad has steps that do not commute (uniform: the same bit in its 4 words): re-gather after the lower levels committed
                const uint32_t lv = (pw[k] >> 1) & 3u;
                const uint32_t maxlv = uni_b(pw[k] >> 3) & 3u;
                auto again = [&](const uint32_t level) {
                    // The lanes of a wavefront talk to each other through the LDS state: this re-gather reads bytes that OTHER
                    // lanes (the lower levels' step slots) have just committed.  The hardware keeps a wavefront's LDS operations
                    // in order; the fence is for the COMPILER, which reasons per lane -- "this lane has not stored since it last
                    // loaded these bytes, so loading them again is redundant unless it took the store branch" -- and did exactly
                    // that when commit and re-gather once sat in one region: the re-gather ran only in the lanes that had just
                    // stored, i.e. in none of those that needed it (caught by the counts hashes of scripts/dev_spread.py).
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const bool v2 = sample_l<SLOTS, PACKED, TABG>(r, lds_c, tabp, U[k]) != 0u;
                    store(chk, r, v2, lv == level, sidx);
                    const unsigned long long mine = __ballot(lv == level);
                    drawn = (drawn & ~mine) | (__ballot(v2) & mine);
                };
                again(1);
                if (maxlv >= 2) {
                    again(2);
                    if (maxlv >= 3) again(3);
                }
            }
            ones_k[k] = count(chk, r, drawn, sidx);
        }
        // LOG: the 16 logged bytes of the iteration -- step (g0 + k) * 4 + j, slot j's count of quad k sitting in row j's
        // register k -- are 16 consecutive bytes.  The same two pairs of lane swaps that transpose the draws turn (register k,
        // row j) into (register j, row k): every row then packs ITS quad's four bytes and its tail lane stores one dword, 4
        // lanes = one 16-byte run per iteration instead of 16 single-byte stores (as many write requests: the logged sweep
        // cost 20 % over the plain one, most of it there).
        if constexpr (LOG && COUNT && !decltype(chk)::value) {
            const auto a = __builtin_amdgcn_permlane32_swap(ones_k[0], ones_k[2], false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(ones_k[1], ones_k[3], false, false);
            const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
            const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
            const uint32_t quad_bytes = c[0] | (c[1] << 8) | (d[0] << 16) | (d[1] << 24);
            typedef uint32_t __attribute__((aligned(1))) u32_unaligned; // (a window may start at any step)
            if (tail) *reinterpret_cast<u32_unaligned *>(logrow + (size_t)(g0 + j) * 4u) = quad_bytes;
        }
    };
    // one iteration, checked against the group's stopping time only where it cuts it; false = the group is done
    auto run = [&](const uint32_t g0, RecL<SLOTS, PACKED> (&rc)[NREC], uint32_t (&pwc)[4], RecL<SLOTS, PACKED> (&rx)[NREC], uint32_t (&pwx)[4]) -> bool {
        if constexpr (LOG) {
            if (g0 * 4u >= lim) return false;
            if (g0 * 4u + 16u > lim) { iteration(std::true_type{}, g0, rc, pwc, rx, pwx); return false; }
        }
        iteration(std::false_type{}, g0, rc, pwc, rx, pwx);
        return true;
    };
    const PrioRotor prio; // (wave_prio.h: the blocks of a CU progress as equals)
    if constexpr (RECMODE == 2) {
        for (uint32_t g0 = 0; g0 < nq; g0 += 8) {
            prio.tick(g0, p.prio_shift);
            if (!run(g0, rA, pwA, rB, pwB)) break;
            if (g0 + 4u < nq && !run(g0 + 4u, rB, pwB, rA, pwA)) break;
        }
    } else {
        for (uint32_t g0 = 0; g0 < nq; g0 += 4) {
            prio.tick(g0, p.prio_shift);
            if (!run(g0, rA, pwA, rB, pwB)) break;
        }
    }

    for (uint32_t v = j; v < p.n_vars; v += 4) p.state[(size_t)v * p.pitch + chain] = lds[v * p.par_stride + cb];
    if constexpr (COUNT) {
        __syncthreads();
        const uint32_t *h = reinterpret_cast<const uint32_t *>(lds + p.par_hist_off);
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        flush_rank_hist(p, vd, h, gc, tid, nthr);
    }
    if (p.dbg_times && tid == 0) { // measurements only: when this block ran, and where
        p.dbg_times[blockIdx.x * 4u] = dbg_t0;
        p.dbg_times[blockIdx.x * 4u + 1u] = __builtin_amdgcn_s_memrealtime();
        p.dbg_times[blockIdx.x * 4u + 2u] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_ID
        p.dbg_times[blockIdx.x * 4u + 3u] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // XCC_ID
    }
}

// K1, strict form (GR_FLAG_SCAN_PER_CHAIN): every chain draws its own variable at every step, exactly like
// UniformSampler.VarSample (sampler/sampler.go:135-174) -- no wave-uniform scan, hence no scan plan and no
// scalar records.  One chain per lane, one 64-chain wavefront per block, the 4-lane form's LDS layout
// (80-byte state rows, records and threshold tables in LDS, read per lane); two Philox blocks per quad
// (DOM_SCAN_CHAIN for the variables, DOM_DRAW for the values); per-lane LDS atomics for the histogram.
// Steps of a chain are sequential, so nothing needs re-gathering.  Bit-identical to the generic
// kernel's strict mode (k_sweep<.., PC = true>), which handles quad-unaligned heads and tails.
// LOG (gr_advance_chain in strict mode): every chain executes only the first limit[chain] steps of the window (its own
// stopping time) and logs the value of each of its updates, one byte per chain and step, [group][step][64 chains].
template <int SLOTS, bool PACKED, bool COUNT, bool LOG = false>
__global__ void __launch_bounds__(256) k_sweep_strict(const FastParams p) {
    extern __shared__ __align__(16) unsigned char lds[];
    // 1 or 4 wavefronts per block share the tables and records; wave 0 keeps its 64 chains in the layout's state
    // region at offset 0, waves 1..3 in extra regions behind the 4-lane form's layout (p.strict_state_off)
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, T = blockDim.x;
    const uint32_t chain = blockIdx.x * T + tid;
    const uint32_t variant = (blockIdx.x * T) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    unsigned char *st = lds + (wave ? p.strict_state_off + (wave - 1u) * (((p.n_vars * kParStride) + 15u) & ~15u) : 0u) + lane;
    {
        const uint32_t n4 = (vd.cpt_len + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.par_tab_off);
        for (uint32_t i = tid; i < n4; i += T) dst[i] = src[i];
        const uint4 *rs = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(p.recs_par) + (size_t)vd.rec_base * p.rec_stride);
        uint4 *rd = reinterpret_cast<uint4 *>(lds + p.par_rec_off);
        const uint32_t per = p.rec_stride / 16u, per_lds = p.strict_rec_stride / 16u; // records are padded in LDS, see below
        const uint32_t base = lds_addr_of(lds);
        for (uint32_t i = tid; i < vd.n_free * per; i += T) {
            uint4 q = rs[i];
            if (i % per == 0) { q.y += base; q.z += base; } // histogram and table offsets -> absolute LDS addresses
            rd[(i / per) * per_lds + (i % per)] = q;
        }
        for (uint32_t v = 0; v < p.n_vars; ++v) st[v * kParStride] = p.state[(size_t)v * p.pitch + chain];
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + p.par_hist_off);
        for (uint32_t i = tid; i < 2u * vd.n_free; i += T) h[i] = 0;
    }
    __syncthreads();
    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint32_t g_lo = (uint32_t)gchain, g_hi4 = (uint32_t)(gchain >> 32) << 4;
    const uint64_t q0 = (((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2;
    const uint32_t nq = p.n_steps >> 2;
    const unsigned char *recs = lds + p.par_rec_off;
    // The two Philox blocks of the NEXT quad (20 rounds) are computed in four slices, one under the state-read
    // latency of each step of the current quad: a lone wavefront per SIMD has nothing else to hide it with.
    u32x4 sc = philox4x32_10((uint32_t)q0, (uint32_t)(q0 >> 32), g_lo, DOM_SCAN_CHAIN + g_hi4, p.seed_lo, p.seed_hi);
    u32x4 dr = philox4x32_10((uint32_t)q0, (uint32_t)(q0 >> 32), g_lo, DOM_DRAW + g_hi4, p.seed_lo, p.seed_hi);
    uint32_t lim = 0xffffffffu;
    unsigned char *logp = nullptr;
    if constexpr (LOG) {
        const uint32_t t = p.limit[chain];
        lim = t > p.log_off ? t - p.log_off : 0u;
        logp = p.log + (size_t)(chain >> 6) * p.log_pitch + (size_t)p.log_off * 64u + lane;
    }
    const PrioRotor prio; // (wave_prio.h)
    for (uint32_t qi = 0; qi < nq; ++qi) {
        prio.tick(qi, p.prio_shift);
        const uint64_t qn = q0 + qi + 1u;
        const uint32_t su[4] = {sc.x, sc.y, sc.z, sc.w}, du[4] = {dr.x, dr.y, dr.z, dr.w};
        RecL<SLOTS, PACKED> r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { // the records only depend on the scan stream: fetch all four first
            // per-lane record reads: with 64-byte strides every record would start in one of two bank groups; the
            // LDS copy is padded to 80 (144) bytes so that the lanes' reads spread over all banks
            const unsigned char *ra = recs + mul_hi_u32(su[j], vd.n_free) * p.strict_rec_stride;
            r[j].h = *reinterpret_cast<const uint4 *>(ra);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(r[j].s) / sizeof(uint4)); ++i) r[j].s[i] = *reinterpret_cast<const uint4 *>(ra + 16 + 16 * i);
        }
        PhiloxSteps ps{(uint32_t)qn, (uint32_t)(qn >> 32), g_lo, DOM_SCAN_CHAIN + g_hi4, p.seed_lo, p.seed_hi};
        PhiloxSteps pd{(uint32_t)qn, (uint32_t)(qn >> 32), g_lo, DOM_DRAW + g_hi4, p.seed_lo, p.seed_hi};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x[SLOTS];
            gather_l<SLOTS, PACKED>(r[j], st, x);
            __builtin_amdgcn_sched_barrier(0);
            if (j == 0) ps.template rounds<5>();
            else if (j == 1) pd.template rounds<5>();
            else if (j == 2) ps.template rounds<5>();
            else pd.template rounds<5>();
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t val = (du[j] >= table_l<SLOTS, PACKED>(r[j], x, lds)) ? 1u : 0u;
            if (!LOG || qi * 4u + (uint32_t)j < lim) {
                st[r[j].h.x] = (unsigned char)val;
                if constexpr (COUNT) {
                    __hip_atomic_fetch_add(lds_at(r[j].h.y + val * 4u), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                if constexpr (LOG) logp[(size_t)(qi * 4u + (uint32_t)j) * 64u] = (unsigned char)val;
            }
        }
        sc = u32x4{ps.c0, ps.c1, ps.c2, ps.c3};
        dr = u32x4{pd.c0, pd.c1, pd.c2, pd.c3};
    }
    for (uint32_t v = 0; v < p.n_vars; ++v) p.state[(size_t)v * p.pitch + chain] = st[v * kParStride];
    if constexpr (COUNT) {
        __syncthreads();
        const uint32_t *h = reinterpret_cast<const uint32_t *>(lds + p.par_hist_off);
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        flush_rank_hist(p, vd, h, gc, tid, T);
    }
}

#if GR_PAR4_PART == 0
template <int SLOTS, bool PACKED>
static hipError_t launch_strict_t(const FastParams &p, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, s, p);
        return hipGetLastError();
    };
    if (p.log && p.limit) return go(&k_sweep_strict<SLOTS, PACKED, true, true>); // logged windows always count
    return p.counting ? go(&k_sweep_strict<SLOTS, PACKED, true>) : go(&k_sweep_strict<SLOTS, PACKED, false>);
}

hipError_t launch_sweep_strict(const FastParams &p, int slots, bool packed, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s) {
    if (!packed && slots > 6) return hipErrorInvalidValue;
    switch (slots) {
    case 0: case 1: return packed ? launch_strict_t<1, true>(p, blocks, threads, lds, s) : launch_strict_t<1, false>(p, blocks, threads, lds, s);
    case 2: return packed ? launch_strict_t<2, true>(p, blocks, threads, lds, s) : launch_strict_t<2, false>(p, blocks, threads, lds, s);
    case 3: return packed ? launch_strict_t<3, true>(p, blocks, threads, lds, s) : launch_strict_t<3, false>(p, blocks, threads, lds, s);
    case 4: return packed ? launch_strict_t<4, true>(p, blocks, threads, lds, s) : launch_strict_t<4, false>(p, blocks, threads, lds, s);
    case 5: return packed ? launch_strict_t<5, true>(p, blocks, threads, lds, s) : launch_strict_t<5, false>(p, blocks, threads, lds, s);
    case 6: return packed ? launch_strict_t<6, true>(p, blocks, threads, lds, s) : launch_strict_t<6, false>(p, blocks, threads, lds, s);
    case 7: case 8: return launch_strict_t<8, true>(p, blocks, threads, lds, s);
    case 9: case 10: case 11: case 12: return launch_strict_t<12, true>(p, blocks, threads, lds, s);
    case 13: case 14: case 15: case 16: return launch_strict_t<16, true>(p, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
    }
}

#endif // GR_PAR4_PART == 0

template <int SLOTS, bool PACKED>
static hipError_t launch_par_t(const FastParams &p, uint32_t blocks, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(p.par_spread ? blocks * 4u : blocks), dim3(p.par_spread ? 64 : 256), lds, s, p);
        return hipGetLastError();
    };
    // fewer than 3 wavefronts per SIMD (256 CUs x 4 SIMDs, 4 wavefronts per block): latency-bound -> hoisted record reads
    // where the records come from (p.par_recmode: 0..2 forced, else by occupancy): with fewer than 3 wavefronts per SIMD
    // (256 CUs x 4 SIMDs, 4 wavefronts per block) the kernel is latency-bound -> LDS records fetched together; with more it
    // sits on the LDS pipe -> records from global memory, prefetched an iteration ahead
    if (p.tab_global && p.par_tab_words) // thresholds partly in LDS, partly from global memory: record mode 2
        return !p.counting ? go(&k_sweep_par4<SLOTS, PACKED, 0, 2, 2>) : p.log ? go(&k_sweep_par4<SLOTS, PACKED, 2, 2, 2>) : go(&k_sweep_par4<SLOTS, PACKED, 1, 2, 2>);
    if (p.tab_global) // thresholds from global memory: record mode 2
        return !p.counting ? go(&k_sweep_par4<SLOTS, PACKED, 0, 2, 1>) : p.log ? go(&k_sweep_par4<SLOTS, PACKED, 2, 2, 1>) : go(&k_sweep_par4<SLOTS, PACKED, 1, 2, 1>);
    const int rm = p.par_recmode <= 2u ? (int)p.par_recmode : (blocks * 4u < 3u * 1024u ? 1 : 2);
    if (rm == 2) return !p.counting ? go(&k_sweep_par4<SLOTS, PACKED, 0, 2>) : p.log ? go(&k_sweep_par4<SLOTS, PACKED, 2, 2>) : go(&k_sweep_par4<SLOTS, PACKED, 1, 2>);
    if (rm == 1) return !p.counting ? go(&k_sweep_par4<SLOTS, PACKED, 0, 1>) : p.log ? go(&k_sweep_par4<SLOTS, PACKED, 2, 1>) : go(&k_sweep_par4<SLOTS, PACKED, 1, 1>);
    return !p.counting ? go(&k_sweep_par4<SLOTS, PACKED, 0, 0>) : p.log ? go(&k_sweep_par4<SLOTS, PACKED, 2, 0>) : go(&k_sweep_par4<SLOTS, PACKED, 1, 0>);
}

// the instantiations, one translation unit per record shape
#if GR_PAR4_PART == 1
hipError_t launch_sweep_par4_p1(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s) { // unpacked, 1..6 neighbours
    switch (slots) {
    case 0: case 1: return launch_par_t<1, false>(p, blocks, lds, s);
    case 2: return launch_par_t<2, false>(p, blocks, lds, s);
    case 3: return launch_par_t<3, false>(p, blocks, lds, s);
    case 4: return launch_par_t<4, false>(p, blocks, lds, s);
    case 5: return launch_par_t<5, false>(p, blocks, lds, s);
    case 6: return launch_par_t<6, false>(p, blocks, lds, s);
    default: return hipErrorInvalidValue;
    }
}
#elif GR_PAR4_PART == 2
hipError_t launch_sweep_par4_p2(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s) { // packed, 1..6 neighbours
    switch (slots) {
    case 0: case 1: return launch_par_t<1, true>(p, blocks, lds, s);
    case 2: return launch_par_t<2, true>(p, blocks, lds, s);
    case 3: return launch_par_t<3, true>(p, blocks, lds, s);
    case 4: return launch_par_t<4, true>(p, blocks, lds, s);
    case 5: return launch_par_t<5, true>(p, blocks, lds, s);
    case 6: return launch_par_t<6, true>(p, blocks, lds, s);
    default: return hipErrorInvalidValue;
    }
}
#elif GR_PAR4_PART == 3
hipError_t launch_sweep_par4_p3(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s) { // packed, 7..12 neighbours
    switch (slots) {
    case 7: return launch_par_t<7, true>(p, blocks, lds, s);
    case 8: return launch_par_t<8, true>(p, blocks, lds, s);
    case 9: return launch_par_t<9, true>(p, blocks, lds, s);
    case 10: return launch_par_t<10, true>(p, blocks, lds, s);
    case 11: return launch_par_t<11, true>(p, blocks, lds, s);
    case 12: return launch_par_t<12, true>(p, blocks, lds, s);
    default: return hipErrorInvalidValue;
    }
}
#elif GR_PAR4_PART == 4
hipError_t launch_sweep_par4_p4(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s) { // packed, 128-byte records (rec_stride = 128)
    switch (slots) {
    case 13: return launch_par_t<13, true>(p, blocks, lds, s);
    case 14: return launch_par_t<14, true>(p, blocks, lds, s);
    case 15: return launch_par_t<15, true>(p, blocks, lds, s);
    case 16: return launch_par_t<16, true>(p, blocks, lds, s);
    default: return hipErrorInvalidValue;
    }
}
#elif GR_PAR4_PART == 5
// ---------------------------------------------------------------------------
// K1, WIDE step-parallel form: 8 or 16 lanes per chain (round 3).
//
// k_sweep_par4 turns 64 chains into 4 wavefronts.  Launches that still leave the chip under-filled with that -- BASELINE config 1
// (Grids_11, 4 096 chains = 256 wavefronts for 1 024 SIMDs), the first rounds of an adaptive run (2 groups), collapsed
// Segmentation_11 (LDS lets a CU hold two 64-chain blocks = 2 wavefronts per SIMD) -- get LPC = 8 or 16 lanes per chain here: a
// block of 64 x LPC lanes still owns the 64 chains of one scan group (same LDS footprint), but LPC consecutive steps of every
// chain are gathered, drawn and committed side by side, so the same chains are 2x / 4x the wavefronts.  The dependency levels
// of the plan (k_scan_plan_wide) now span LPC steps: more re-gather passes per 64 updates (Grids: 1.26 with 4 steps side by
// side, 1.85 with 8, 2.7 with 16), which is why this form is only chosen while wavefronts are what is missing.
//   lanes     lane = slot * W + c, W = 64 / LPC chains per wavefront; slot s handles step j(s) = (s % E) * 4 + s / E of a round,
//             E = LPC / 4 -- chosen so that the draw transpose stays the two lane-swap pairs of k_sweep_par4: lane slot s computes
//             the Philox call of quad Q0 + s (slot bits = (k1 k0 | j_hi)), the swaps exchange (k1 k0) with the word index
//             (j1 j0), and lane slot (j1 j0 | j_hi) ends up with word (j1 j0) of quads Q0 + 4 k' + .. for the rounds k' = 0 .. 3
//   iteration 4 rounds of LPC steps (one Philox call per lane); records and plan words from global memory, one iteration ahead
// Counting, records, threshold tables (LDS, global, split) and the commit order are k_sweep_par4's: bit-identical results.
// Draw log (MODE 2): one row per wavefront; gr_advance_chain picks this form for a window by the same rule as any launch.
constexpr uint32_t kWideOffMask = 0x03ffffc0u; // plan word of the wide form: record offset (< 64 MiB, multiple of 64) in bits 6..25,
                                               // level in bits 0..3, highest level of the round in bits 26..29, bits 30 / 31 as kPlanDepsBit / kPlanLevelBit

template <int LPC>
__global__ void __launch_bounds__(256) k_scan_plan_wide(const FastParams p) {
    extern __shared__ unsigned short sp_tab[];
    const uint32_t ng = p.n_steps / (uint32_t)LPC; // rounds in this launch
    const uint32_t wave = blockIdx.y, gi = blockIdx.x * 256u + threadIdx.x;
    const uint32_t variant = (wave * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    const uint32_t S = 2u + p.plan_slots;
    for (uint32_t k = threadIdx.x; k < vd.n_free; k += 256u) {
        const uint32_t v = p.free_vars[vd.free_base + k];
        const uint32_t *cp = p.cprogs + vd.cprog_base + p.cprog_off[vd.cprogoff_base + v];
        const uint32_t nn = (cp[0] >> 8) & 0xffu;
        sp_tab[k * S] = (unsigned short)v;
        sp_tab[k * S + 1u] = (unsigned short)nn;
        for (uint32_t i = 0; i < nn; ++i) sp_tab[k * S + 2u + i] = (unsigned short)cp[3 + 2 * i];
    }
    __syncthreads();
    if (gi >= ng) return;
    const uint64_t q0 = ((((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2) + (uint64_t)gi * (LPC / 4);
    const uint64_t gchain0 = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + (uint64_t)wave * 64u;
    uint32_t k[LPC], v[LPC], level[LPC], maxlv = 0;
#pragma unroll
    for (int c = 0; c < LPC / 4; ++c) {
        const uint64_t q = q0 + (uint64_t)c;
        const u32x4 sc = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)(gchain0 >> 6), DOM_SCAN, p.seed_lo, p.seed_hi);
        k[4 * c] = mul_hi_u32(sc.x, vd.n_free); k[4 * c + 1] = mul_hi_u32(sc.y, vd.n_free);
        k[4 * c + 2] = mul_hi_u32(sc.z, vd.n_free); k[4 * c + 3] = mul_hi_u32(sc.w, vd.n_free);
    }
    if (p.sys_rank) { // systematic scan: step t updates position t mod n of the sweep order
        uint32_t sys_p = (uint32_t)((q0 << 2) % vd.n_free);
#pragma unroll
        for (int j = 0; j < LPC; ++j) {
            k[j] = p.sys_rank[vd.free_base + sys_p];
            if (++sys_p == vd.n_free) sys_p = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < LPC; ++j) {
        const unsigned short *e = sp_tab + k[j] * S;
        v[j] = e[0];
        // level of step j inside its round: 1 + the highest level of an earlier step that updates the same variable or one of
        // its neighbours (k_scan_plan's rule over LPC steps instead of 4)
        const uint32_t nn = e[1];
        uint32_t lv = 0;
#pragma unroll
        for (int pj = 0; pj < j; ++pj)
            if (v[pj] == v[j]) lv = max(lv, level[pj] + 1u);
        for (uint32_t i = 0; i < nn; ++i) {
            const uint32_t u = e[2 + i];
#pragma unroll
            for (int pj = 0; pj < j; ++pj)
                if (u == v[pj]) lv = max(lv, level[pj] + 1u);
        }
        level[j] = lv;
        maxlv = max(maxlv, lv);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(p.plan + (size_t)wave * p.plan_pitch) + (size_t)gi * (LPC / 4);
#pragma unroll
    for (int c = 0; c < LPC / 4; ++c) {
        uint32_t o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = 4 * c + t;
            o[t] = (vd.rec_base + k[j]) * p.rec_stride | level[j] | (maxlv << 26) | (level[j] ? kPlanLevelBit : 0u) | (maxlv ? kPlanDepsBit : 0u);
        }
        dst[c] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// DENSE (8 lanes per chain, 4-slot records, tables in LDS): records one round ahead and at most 64 registers, so that a SIMD holds 8
// wavefronts -- 65 536 chains as 8 192 resident wavefronts (experiment: fast_form 10)
// MODE 2 (gr_advance_chain's windows while wavefronts are what a launch lacks -- the first rounds of an adaptive run): count + draw
// log, one row per wavefront (LPC rows of W chains per 64-chain group: SweepParams::log_rows = LPC), one byte per step = the ones among
// the wavefront's W chains; a group executes only the steps below its own stopping time (p.limit), checked per step.
template <int LPC, int SLOTS, int MODE, int TABG, bool DENSE = false> // MODE 0: burn-in, 1: count, 2: count + draw log
__global__ void __launch_bounds__(64 * LPC, DENSE ? 8 : 4) k_sweep_wide(const FastParams p) { // (4 wavefronts per SIMD: <= 128 registers)
    constexpr bool COUNT = MODE != 0, LOG = MODE == 2, PACKED = true;
    constexpr uint32_t W = 64u / LPC, E = LPC / 4u;
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, nthr = blockDim.x;
    const bool spread = nthr == 64u;                                   // one wavefront per block (launches with few groups)
    const uint32_t group = spread ? blockIdx.x / LPC : blockIdx.x;    // 64-chain scan group
    const uint32_t wave = spread ? blockIdx.x % LPC : tid >> 6;       // which W chains of the group
    const uint32_t slot = lane / W;                                    // lane slot
    const uint32_t j = (slot % E) * 4u + slot / E;                     // the step of a round this lane slot handles
    const uint32_t cb = wave * W + (lane % W);                         // chain inside the group
    const uint32_t chain = group * 64u + cb;
    const uint32_t variant = (group * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    {   // stage the threshold tables (all, or the LDS part of a split pool), this block's chains, zero the block histogram
        const uint32_t n4 = ((TABG == 2 ? min(vd.cpt_len, p.par_tab_words) : vd.cpt_len) + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + p.par_tab_off);
        if constexpr (TABG != 1)
            for (uint32_t i = tid; i < n4; i += nthr) dst[i] = src[i];
        for (uint32_t v = slot; v < p.n_vars; v += LPC) lds[v * p.par_stride + cb] = p.state[(size_t)v * p.pitch + chain];
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + p.par_hist_off);
        for (uint32_t i = tid; i < 2u * vd.n_free; i += nthr) h[i] = 0;
    }
    __syncthreads();
    // the global records hold offsets into the dynamic LDS array, used as absolute LDS addresses (see k_sweep_par4)
    if (lds_addr_of(lds) != 0u) {
        if (threadIdx.x == 0) atomicOr(p.err, 1u);
        return;
    }
    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint64_t q0 = (((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2;
    const uint32_t n_it = p.n_steps / (4u * LPC);  // iterations: 4 rounds of LPC steps each
    const uint32_t n_rounds = n_it * 4u;
    const unsigned char *lds_c = lds + cb;
    const unsigned char *tabp = TABG ? reinterpret_cast<const unsigned char *>(p.cpt + vd.cpt_base) : lds;
    const unsigned char *recs_g = reinterpret_cast<const unsigned char *>(p.recs_par);
    const bool tail = (lane % W) == W - 1u;
    // ones of my lane slot = popcount of the compare mask's W bits of my slot
    const unsigned long long slot_mask = ((1ull << W) - 1ull) << (slot * W);
    const uint32_t rm_lo = (uint32_t)slot_mask, rm_hi = (uint32_t)(slot_mask >> 32);

    const __amdgpu_buffer_rsrc_t plan_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t *>(p.plan + (size_t)group * p.plan_pitch), 0, 0x7fffffff, 0x00020000);
    const uint32_t j4 = j * 4u;
    uint32_t pwa[4] = {0, 0, 0, 0};
    auto prefetch_plan = [&](const uint32_t round0) { // the plan words of rounds round0 .. round0 + 3 (never past the plan)
#pragma unroll
        for (int k = 0; k < 4; ++k) pwa[k] = __builtin_amdgcn_raw_buffer_load_b32(plan_rsrc, j4, min(round0 + (uint32_t)k, n_rounds - 1u) * (4u * LPC), 0);
    };
    // Records: SLOTS <= 4 -- the four records of the NEXT iteration in flight while the current one runs (two register sets of
    // four, like k_sweep_par4); SLOTS > 4 -- ONE record ahead (two sets of one, the next round's fetched while the current round
    // runs): two sets of four 12..20-word records are 96..160 registers, and above 128 a CU holds one 512-lane block where its
    // LDS would take two (Segmentation_11, Promedus, collapsed groups).
    constexpr bool ROUND_AHEAD = SLOTS > 4 || DENSE;
    constexpr int NSET = ROUND_AHEAD ? 1 : 4;
    uint32_t pwA[4] = {0, 0, 0, 0}, pwB[4] = {0, 0, 0, 0};
    RecL<SLOTS, PACKED> rA[NSET], rB[NSET];
    auto fetch = [&](RecL<SLOTS, PACKED> &r, const uint32_t pw) {
        const unsigned char *ga = recs_g + (pw & kWideOffMask);
        r.h = *reinterpret_cast<const uint4 *>(ga);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(r.s) / sizeof(uint4)); ++i) r.s[i] = *reinterpret_cast<const uint4 *>(ga + 16 + 16 * i);
    };
    auto prefetch = [&](RecL<SLOTS, PACKED> (&rx)[NSET], uint32_t (&pwx)[4]) { // the iteration whose plan words are in pwa
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pwx[k] = pwa[k];
            if constexpr (!ROUND_AHEAD) fetch(rx[k], pwx[k]);
        }
    };
    prefetch_plan(0); prefetch(rA, pwA); prefetch_plan(4);
    if constexpr (ROUND_AHEAD) fetch(rA[0], pwA[0]); // round 0's record; from here on set A serves the even rounds, set B the odd ones

    // GR_FLAG_SCAN_SYSTEMATIC (p.sys_rank set): the plan follows the sweep order and update t of a chain takes word (sweep & 3) of
    // philox_sys_draw(sweep = t / n, pos = t mod n, chain) -- one Philox call per step.  My steps are LPC apart.
    unsigned char *logrow = nullptr; // LOG: my wavefront's row of the draw log, and the group's stopping time inside this launch
    uint32_t lim = 0xffffffffu;
    if constexpr (LOG) {
        if (p.log) logrow = p.log + (size_t)(group * LPC + wave) * p.log_pitch + p.log_off;
        if (p.limit) {
            const uint32_t t = p.limit[group];
            lim = t > p.log_off ? t - p.log_off : 0u;
        }
    }
    const bool sys = !DENSE && !LOG && p.sys_rank != nullptr;
    uint64_t sys_sweep = 0;
    uint32_t sys_pos = 0;
    if (sys) {
        const uint64_t s_first = (q0 << 2) + j;
        sys_sweep = s_first / vd.n_free;
        sys_pos = (uint32_t)(s_first - sys_sweep * vd.n_free);
    }
    auto iteration = [&](const uint32_t it, RecL<SLOTS, PACKED> (&rc)[NSET], uint32_t (&pwc)[4], RecL<SLOTS, PACKED> (&rx)[NSET], uint32_t (&pwx)[4]) {
        uint32_t U[4];
        if (sys) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 d = philox_sys_draw(sys_sweep, sys_pos, gchain, p.seed_lo, p.seed_hi);
                const uint32_t w = (uint32_t)sys_sweep & 3u;
                U[k] = w == 0 ? d.x : w == 1 ? d.y : w == 2 ? d.z : d.w;
                sys_pos += (uint32_t)LPC;
                while (sys_pos >= vd.n_free) { sys_pos -= vd.n_free; ++sys_sweep; }
            }
        } else {
            // draws: my Philox call is quad it * LPC + slot; two lane-swap pairs leave U[k] = my word of round k (see the header)
            const uint64_t q = q0 + (uint64_t)it * LPC + slot;
            const u32x4 Wd = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
            const auto a = __builtin_amdgcn_permlane32_swap(Wd.x, Wd.z, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(Wd.y, Wd.w, false, false);
            const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
            const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
            U[0] = c[0]; U[1] = c[1]; U[2] = d[0]; U[3] = d[1];
        }
        prefetch(rx, pwx); prefetch_plan(it * 4u + 8u);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // ROUND_AHEAD: rc[0] = set A (even rounds), rx[0] = set B (odd rounds): fetch the next round's record into the other set
            if constexpr (ROUND_AHEAD) fetch((k & 1) ? rc[0] : rx[0], k < 3 ? pwc[k + 1] : pwx[0]);
            const RecL<SLOTS, PACKED> r = ROUND_AHEAD ? ((k & 1) ? rx[0] : rc[0]) : rc[ROUND_AHEAD ? 0 : k];
            const uint32_t pw = pwc[k];
            const uint32_t sidx = (it * 4u + (uint32_t)k) * LPC + j; // step index inside this launch
            const bool runs = !LOG || sidx < lim;                     // (LOG: steps past the group's stopping time leave it untouched)
            const bool val = sample_l<SLOTS, PACKED, TABG>(r, lds_c, tabp, U[k]) != 0u;
            unsigned long long drawn = __ballot(val);
            if (pw < kPlanLevelBit && runs) lds[r.h.x + cb] = val ? (unsigned char)1 : (unsigned char)0; // level 0: commit in the first pass
            if (__any(pw >= kPlanDepsBit)) { // the round has steps that do not commute (uniform: the same bits in all its words)
                const uint32_t lv = pw & 15u;
                const uint32_t maxlv = (uni_b(pw) >> 26) & 15u;
                for (uint32_t level = 1; level <= maxlv; ++level) {
                    // (the re-gather reads bytes other lanes have just committed: in-order LDS in hardware, the fence for the
                    // compiler -- see k_sweep_par4)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    const bool v2 = sample_l<SLOTS, PACKED, TABG>(r, lds_c, tabp, U[k]) != 0u;
                    if (lv == level && runs) lds[r.h.x + cb] = v2 ? (unsigned char)1 : (unsigned char)0;
                    const unsigned long long mine = __ballot(lv == level);
                    drawn = (drawn & ~mine) | (__ballot(v2) & mine);
                }
            }
            if constexpr (COUNT) {
                const uint32_t s1 = (uint32_t)__builtin_popcount(((uint32_t)drawn & rm_lo) | ((uint32_t)(drawn >> 32) & rm_hi));
                if (tail && runs) {
                    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>((uintptr_t)r.h.y),
                                           ((unsigned long long)s1 << 32) | (unsigned long long)(W - s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (LOG && logrow) logrow[sidx] = (unsigned char)s1;
                }
            }
        }
    };
    const PrioRotor prio; // (wave_prio.h)
    if constexpr (ROUND_AHEAD) {
        // (the record sets do not swap roles between iterations: 4 rounds per iteration, set A always serves rounds 0 and 2)
        for (uint32_t it = 0; it < n_it; it += 2) {
            if (LOG && it * (4u * LPC) >= lim) break; // the group has finished its AdvanceChain
            prio.tick(it, p.prio_shift);
            iteration(it, rA, pwA, rB, pwB);
            if (it + 1u < n_it) {
                // second iteration of the pair: plan words live in pwB now, the record sets keep their parity
                iteration(it + 1u, rA, pwB, rB, pwA);
            }
        }
    } else {
        for (uint32_t it = 0; it < n_it; it += 2) {
            if (LOG && it * (4u * LPC) >= lim) break; // the group has finished its AdvanceChain
            prio.tick(it, p.prio_shift);
            iteration(it, rA, pwA, rB, pwB);
            if (it + 1u < n_it) iteration(it + 1u, rB, pwB, rA, pwA);
        }
    }

    for (uint32_t v = slot; v < p.n_vars; v += LPC) p.state[(size_t)v * p.pitch + chain] = lds[v * p.par_stride + cb];
    if constexpr (COUNT) {
        __syncthreads();
        const uint32_t *h = reinterpret_cast<const uint32_t *>(lds + p.par_hist_off);
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        flush_rank_hist(p, vd, h, gc, tid, nthr);
    }
}

template <int LPC, int SLOTS>
static hipError_t launch_wide_t(const FastParams &p, uint32_t groups, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(p.par_spread ? groups * LPC : groups), dim3(p.par_spread ? 64 : 64 * LPC), lds, s, p);
        return hipGetLastError();
    };
    const bool log = p.counting && (p.log || p.limit);
    if (p.tab_global && p.par_tab_words) return log ? go(&k_sweep_wide<LPC, SLOTS, 2, 2>) : p.counting ? go(&k_sweep_wide<LPC, SLOTS, 1, 2>) : go(&k_sweep_wide<LPC, SLOTS, 0, 2>);
    if (p.tab_global) return log ? go(&k_sweep_wide<LPC, SLOTS, 2, 1>) : p.counting ? go(&k_sweep_wide<LPC, SLOTS, 1, 1>) : go(&k_sweep_wide<LPC, SLOTS, 0, 1>);
    return log ? go(&k_sweep_wide<LPC, SLOTS, 2, 0>) : p.counting ? go(&k_sweep_wide<LPC, SLOTS, 1, 0>) : go(&k_sweep_wide<LPC, SLOTS, 0, 0>);
}

template <int LPC>
static hipError_t launch_wide_l(const FastParams &p, int slots, uint32_t groups, uint32_t lds, hipStream_t s) {
    // slot classes: padding slots read the variable's own row with multiplier 0 (harmless)
    if (slots <= 4) return launch_wide_t<LPC, 4>(p, groups, lds, s);
    if (slots <= 8) return launch_wide_t<LPC, 8>(p, groups, lds, s);
    if (slots <= 12) return launch_wide_t<LPC, 12>(p, groups, lds, s);
    if (slots <= 16) return launch_wide_t<LPC, 16>(p, groups, lds, s);
    return hipErrorInvalidValue;
}

hipError_t launch_sweep_wide_dense(const FastParams &p, uint32_t groups, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(groups), dim3(512), lds, s, p);
        return hipGetLastError();
    };
    return p.counting ? go(&k_sweep_wide<8, 4, 1, 0, true>) : go(&k_sweep_wide<8, 4, 0, 0, true>);
}

hipError_t launch_sweep_wide(const FastParams &p, int lpc, int slots, uint32_t groups, uint32_t lds, hipStream_t s) {
    return lpc == 8 ? launch_wide_l<8>(p, slots, groups, lds, s) : lpc == 16 ? launch_wide_l<16>(p, slots, groups, lds, s) : hipErrorInvalidValue;
}

hipError_t launch_scan_plan_wide(const FastParams &p, int lpc, hipStream_t s) {
    const uint32_t n_waves = p.n_chains >> 6, ng = p.n_steps / (uint32_t)lpc;
    if (n_waves == 0 || ng == 0) return hipSuccess;
    const size_t lds = (size_t)p.plan_nfree_max * (2u + p.plan_slots) * sizeof(unsigned short);
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((ng + 255u) / 256u, n_waves), dim3(256), lds, s, p);
        return hipGetLastError();
    };
    return lpc == 8 ? go(&k_scan_plan_wide<8>) : lpc == 16 ? go(&k_scan_plan_wide<16>) : hipErrorInvalidValue;
}
#else
hipError_t launch_sweep_par4_p1(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_par4_p2(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_par4_p3(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s);
hipError_t launch_sweep_par4_p4(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s);

hipError_t launch_sweep_par4(const FastParams &p, int slots, bool packed, uint32_t blocks, uint32_t lds, hipStream_t s) {
    if (slots > 16) return hipErrorInvalidValue;
    if (slots > 12) return launch_sweep_par4_p4(p, slots, blocks, lds, s);
    if (slots > 6) return launch_sweep_par4_p3(p, slots, blocks, lds, s);
    // packed: one dword per neighbour -- fewer record bytes per step
    return packed ? launch_sweep_par4_p2(p, slots, blocks, lds, s) : launch_sweep_par4_p1(p, slots, blocks, lds, s);
}

template <int SLOTS, bool PACKED>
static hipError_t launch_bin_t(const FastParams &p, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, s, p);
        return hipGetLastError();
    };
    return !p.counting ? go(&k_sweep_bin<SLOTS, PACKED, 0>) : p.log ? go(&k_sweep_bin<SLOTS, PACKED, 2>) : go(&k_sweep_bin<SLOTS, PACKED, 1>);
}

hipError_t launch_scan_plan(const FastParams &p, hipStream_t s) {
    const uint32_t n_waves = p.n_chains >> 6, nq = p.n_steps >> 2;
    if (n_waves == 0 || nq == 0) return hipSuccess;
    const size_t lds = (size_t)p.plan_nfree_max * (2u + p.plan_slots) * sizeof(unsigned short);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan_plan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_scan_plan, dim3((nq + 255u) / 256u, n_waves), dim3(256), lds, s, p);
    return hipGetLastError();
}

hipError_t launch_sweep_bin(const FastParams &p, int slots, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s) {
    switch (slots) {
    case 0: case 1: return launch_bin_t<1, false>(p, blocks, threads, lds, s);
    case 2: return launch_bin_t<2, false>(p, blocks, threads, lds, s);
    case 3: return launch_bin_t<3, false>(p, blocks, threads, lds, s);
    case 4: return launch_bin_t<4, false>(p, blocks, threads, lds, s);
    case 5: return launch_bin_t<5, false>(p, blocks, threads, lds, s);
    case 6: return launch_bin_t<6, false>(p, blocks, threads, lds, s);
    case 7: return launch_bin_t<7, true>(p, blocks, threads, lds, s);
    case 8: return launch_bin_t<8, true>(p, blocks, threads, lds, s);
    case 9: return launch_bin_t<9, true>(p, blocks, threads, lds, s);
    case 10: return launch_bin_t<10, true>(p, blocks, threads, lds, s);
    case 11: return launch_bin_t<11, true>(p, blocks, threads, lds, s);
    case 12: return launch_bin_t<12, true>(p, blocks, threads, lds, s);
    default: return hipErrorInvalidValue;
    }
}

#endif // GR_PAR4_PART

} // namespace gr
