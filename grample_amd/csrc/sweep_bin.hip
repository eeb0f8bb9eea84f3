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

namespace gr {

__device__ __forceinline__ uint32_t uni_b(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// K0: scan plan.  One thread per (wavefront, quad of steps).
__global__ void __launch_bounds__(256) k_scan_plan(const FastParams p) {
    const uint32_t nq = p.n_steps >> 2;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_waves = p.n_chains >> 6;
    if (idx >= n_waves * nq) return;
    const uint32_t wave = idx / nq, qi = idx - wave * nq;
    const uint64_t q = ((((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2) + qi;
    const uint64_t gchain0 = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + (uint64_t)wave * 64u;
    const uint32_t variant = (wave * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    const u32x4 sc = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)(gchain0 >> 6), DOM_SCAN, p.seed_lo, p.seed_hi);
    const uint32_t su[4] = {sc.x, sc.y, sc.z, sc.w};
    uint32_t k[4], v[4], out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        k[j] = mul_hi_u32(su[j], vd.n_free);
        v[j] = p.free_vars[vd.free_base + k[j]];
        // does step j read a variable that an earlier step of the quad writes?
        const uint32_t *cp = p.cprogs + vd.cprog_base + p.cprog_off[vd.cprogoff_base + v[j]];
        const uint32_t nn = (cp[0] >> 8) & 0xffu;
        uint32_t conf = 0;
        for (uint32_t i = 0; i < nn; ++i) {
            const uint32_t nb = cp[3 + 2 * i];
            for (int pj = 0; pj < j; ++pj) conf |= (nb == v[pj]) ? 1u : 0u;
        }
        out[j] = (vd.rec_base + k[j]) * (kRecDwords * 4u) | conf;
    }
    reinterpret_cast<uint4 *>(p.plan + (size_t)wave * p.plan_pitch)[qi] = make_uint4(out[0], out[1], out[2], out[3]);
}

struct Rec {
    uint32_t w[kRecDwords];
};

__device__ __forceinline__ void load_rec(Rec &r, const uint32_t *recs, uint32_t plan_word) {
    const uint32_t *rp = reinterpret_cast<const uint32_t *>(
        __builtin_assume_aligned(reinterpret_cast<const unsigned char *>(recs) + (plan_word & ~1u), 64));
#pragma unroll
    for (int i = 0; i < kRecDwords; ++i) r.w[i] = rp[i];
}

template <int SLOTS, bool PACKED>
__device__ __forceinline__ uint32_t sample_rec(const Rec &r, const unsigned char *lds_tid, const unsigned char *lds, const uint32_t u) {
    uint32_t cfg4 = 0;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        uint32_t off, m4;
        if constexpr (PACKED) {
            off = r.w[4 + i] & 0xffffu;
            m4 = r.w[4 + i] >> 16;
        } else {
            off = r.w[4 + 2 * i];
            m4 = r.w[5 + 2 * i];
        }
        cfg4 += __umul24(m4, (uint32_t)lds_tid[off]);
    }
    const uint32_t T = *reinterpret_cast<const uint32_t *>(lds + r.w[2] + cfg4);
    return (u >= T) ? 1u : 0u;
}

template <int SLOTS, bool PACKED, bool COUNT>
__global__ void __launch_bounds__(1024) k_sweep_bin(const FastParams p) {
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
    const uint4 *plan = reinterpret_cast<const uint4 *>(p.plan + (size_t)wave_local * p.plan_pitch);
    // lanes 0/1 carry the (zeros, ones) histogram increments: d = sign * ones + base
    const int h_sign = lane == 0 ? -1 : 1;
    const int h_base = lane == 0 ? 64 : 0;
    const uint32_t h_lane = lane * 4u;

    uint4 P = plan[0];
    Rec R0, R1, R2, R3;
    load_rec(R0, p.recs, P.x); load_rec(R1, p.recs, P.y); load_rec(R2, p.recs, P.z); load_rec(R3, p.recs, P.w);
    uint4 Pn = plan[nq > 1 ? 1 : 0];

    auto commit = [&](const Rec &r, const uint32_t val) {
        st[r.w[0]] = (unsigned char)val;
        if constexpr (COUNT) {
            const uint32_t ones = (uint32_t)__popcll(__ballot(val != 0));
            if (lane < 2) {
                uint32_t *hp = reinterpret_cast<uint32_t *>(lds + hist_base + r.w[1] + h_lane);
                __hip_atomic_fetch_add(hp, (uint32_t)(h_sign * (int)ones + h_base), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };

    for (uint32_t qi = 0; qi < nq; ++qi) {
        const uint64_t q = q0 + qi;
        const u32x4 dr = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain,
                                       DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
        // the draw-stream Philox (pure VALU) is issued BEFORE the first use of the prefetched records:
        // it is what hides their scalar-load latency
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t c1 = P.y & 1u, c2 = P.z & 1u, c3 = P.w & 1u;
        // phase A: all four conditionals from the current state
        uint32_t v0 = sample_rec<SLOTS, PACKED>(R0, st, lds, dr.x);
        uint32_t v1 = sample_rec<SLOTS, PACKED>(R1, st, lds, dr.y);
        uint32_t v2 = sample_rec<SLOTS, PACKED>(R2, st, lds, dr.z);
        uint32_t v3 = sample_rec<SLOTS, PACKED>(R3, st, lds, dr.w);
        // phase B: commit in order; recompute a step that reads what an earlier step of the quad wrote
        commit(R0, v0);
        if (c1) v1 = sample_rec<SLOTS, PACKED>(R1, st, lds, dr.y);
        commit(R1, v1);
        if (c2) v2 = sample_rec<SLOTS, PACKED>(R2, st, lds, dr.z);
        commit(R2, v2);
        if (c3) v3 = sample_rec<SLOTS, PACKED>(R3, st, lds, dr.w);
        commit(R3, v3);
        // fetch the next quad's records (and the plan one further ahead): in flight across the back edge,
        // hidden by the next iteration's draw-stream Philox
        __builtin_amdgcn_sched_barrier(0);
        P = Pn;
        load_rec(R0, p.recs, P.x); load_rec(R1, p.recs, P.y); load_rec(R2, p.recs, P.z); load_rec(R3, p.recs, P.w);
        Pn = plan[(qi + 2 < nq) ? qi + 2 : qi];
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

template <int SLOTS, bool PACKED>
static hipError_t launch_bin_t(const FastParams &p, uint32_t blocks, uint32_t threads, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, s, p);
        return hipGetLastError();
    };
    return p.counting ? go(&k_sweep_bin<SLOTS, PACKED, true>) : go(&k_sweep_bin<SLOTS, PACKED, false>);
}

hipError_t launch_scan_plan(const FastParams &p, hipStream_t s) {
    const uint32_t n = (p.n_chains >> 6) * (p.n_steps >> 2);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_plan, dim3((n + 255) / 256), dim3(256), 0, s, p);
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

} // namespace gr
