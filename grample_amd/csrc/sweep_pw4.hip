// K1, value-parallel form for large-Card pairwise models (ObjectDetection: Card 16, 3..13 factors per variable):
// 4 lanes per chain, each lane owning one BLOCK of 4 candidate values of the variable being updated.
//
// Why: the direct conditional costs ~35 VALU instructions per candidate value (f32 log-sum, exp, f64 floor and
// cumulative thresholds, cond.h), so a chain-per-lane wavefront spends ~850 instructions per step at Card 16 -- and at
// the BASELINE size (32 768 chains per GPU = 512 wavefronts) half of the chip's 1 024 SIMDs have no wavefront at all
// while the other half run a lone one that nothing overlaps with (57 % issue-active).  Here the same 32 768 chains are
// 2 048 wavefronts, each ~4x shorter per step: every SIMD is busy and always has a second wavefront to issue from.
//   lanes     lane = 4 * chain + block: a DPP quad is one chain, so every cross-block step (max, the block-sum scan of
//             cond.h, the floor count prefix, the drawn value) is a quad_perm move -- no LDS traffic
//   block     256 lanes = 4 wavefronts = the 64 chains of one scan group (same scan stream as every other kernel)
//   state     lds[var * 64 + chain] (the 4 lanes of a quad read the same byte: broadcast)
//   tables    view-table rows from global memory (L1/L2): the quad's 4 lanes read the 64 contiguous bytes of one row
//   floor     the sequential 1e-6 rule (cond.h) couples the blocks only through the number of floors so far: each block needs
//             the number of floors in the blocks before it.  Guess it from the un-grown threshold (a lower bound), run the
//             block's own 4 values sequentially from there, and check the guess; a wrong guess (a value inside the
//             1e-6-relative window the threshold moves through) repeats the pass, 4 times at most.
// Same Philox streams, same program order of the f32 log sums, same f64 block sums and floor recurrence as cond.h:
// results are bit-identical to the chain-per-lane kernel (tests/test_gpu_parity.py::test_value_parallel_form_*).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cond.h"
#include "device_layout.h"
#include "kernels.h"
#include "philox.h"
#include "wave_prio.h"

namespace gr {

namespace {

// quad_perm:[a,b,c,d]: lane i of every quad reads lane {a,b,c,d}[i] of the same quad
constexpr int QP(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int kShr1 = QP(0, 0, 1, 2); // lane c <- lane c-1 (lane 0 keeps its own)
constexpr int kShr2 = QP(0, 1, 0, 1); // lane c <- lane c-2 (lanes 0, 1 keep their own)
constexpr int kLast = QP(3, 3, 3, 3);
constexpr int kXor1 = QP(1, 0, 3, 2);
constexpr int kXor2 = QP(2, 3, 0, 1);

// (every lane of a quad_perm has a valid source lane, so the `old` operand never shows: mov_dpp = update_dpp without one)
template <int CTRL>
__device__ __forceinline__ int qperm_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float qperm_f(float x) { return __int_as_float(qperm_i<CTRL>(__float_as_int(x))); }

// cond.h's combination of the 4 block sums of a chain, lane c holding q_c: inclusive prefix z_c with
//   z_1 = q1 + q0, z_2 = (q2 + q1) + q0, z_3 = (q3 + q2) + (q1 + q0)   (two Hillis-Steele steps)
// m1 / m2 = 1.0 on the lanes that take part in a step, else 0.0 (fma(x, 1.0, y) is x + y; fma(x, 0.0, y) is y).
__device__ __forceinline__ float quad_scan(const float q, const float m1, const float m2) {
    const float y = fmaf(qperm_f<kShr1>(q), m1, q);
    return fmaf(qperm_f<kShr2>(y), m2, y);
}
// the same for integers (k1 / k2 = all ones on the lanes that take part, else 0): exact, any order
__device__ __forceinline__ int quad_scan_i(const int q, const int k1, const int k2) {
    const int y = q + (k1 & qperm_i<kShr1>(q));
    return y + (k2 & qperm_i<kShr2>(y));
}

} // namespace

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
// The launch's read-only inputs (scan variables, programs) are read through the constant address space: a uniform address is
// then always a scalar load, also in the LOG instantiation, whose byte stores to the draw log could alias anything as far
// as the compiler knows (it had turned every program load into a vector load there: twice the VMEM instructions, +70 % scalar
// and +120 % branch instructions, 1.6x the time).
#define GR_CONST_AS __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const GR_CONST_AS T *as_const(const void *p) { return (const GR_CONST_AS T *)(uintptr_t)p; }

// One step's wave-uniform inputs (scalar registers)
struct StepHdr {
    uint32_t v;          // the variable updated
    const uint32_t *pg;  // its value-parallel program
    uint32_t h, ho;      // Card | rowlen << 8 | n_slots << 16, histogram offset
    uint32_t f_lo, f_hi; // filter of its blanket (compiled_model.hpp)
};

// cond.h across a DPP quad for NP independent steps at once (NP = 2: two consecutive steps of a chain whose variables are
// not adjacent -- both conditionals depend on the state before either commit, so their long dependent f64 chains can be
// interleaved instruction by instruction; with only two wavefronts per SIMD that interleaving is what fills the VALU).
// w[p][i]: f32 log sums of my 4 values of step p; returns the drawn values.
// FULL = every step has Card 16 (all 16 values of the quad live): none of the per-value guards.
template <int NP, bool FULL>
__device__ __forceinline__ void pw4_cond_impl(float (&w)[NP][4], const int (&C)[NP], const uint32_t (&du)[NP], const uint32_t blk, const float m1,
                                              const float m2, const int k1, const int k2, uint32_t (&val)[NP]) {
    bool live[NP][4];
    float d[NP][4], t0[NP], te[NP], e4[NP][4];
    int guess[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            live[p][i] = FULL || (int)(4u * blk) + i < C[p];
            w[p][i] = live[p][i] ? w[p][i] : -INFINITY; // absent values: exp -> +0.0, never floored, mass 0, never drawn
        }
        float m = fmaxf(fmaxf(w[p][0], w[p][1]), fmaxf(w[p][2], w[p][3]));
        m = fmaxf(m, qperm_f<kXor1>(m));
        m = fmaxf(m, qperm_f<kXor2>(m));
#pragma unroll
        for (int i = 0; i < 4; ++i) d[p][i] = __expf(w[p][i] - m);
        const float tot = qperm_f<kLast>(quad_scan(((d[p][0] + d[p][1]) + d[p][2]) + d[p][3], m1, m2));
        t0[p] = kFloorEps * tot;
        te[p] = kFloorEps * t0[p];
        guess[p] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) guess[p] += (live[p][i] && d[p][i] < t0[p]) ? 1 : 0;
    }
    for (;;) {
        bool again = false;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            // floors in the blocks before mine = exclusive quad prefix of the guesses
            float k = (float)(quad_scan_i(guess[p], k1, k2) - guess[p]);
            int n = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                e4[p][i] = d[p][i];
                n += (live[p][i] && e4[p][i] < floor_thr(t0[p], te[p], k)) ? 1 : 0;
                floor_step(e4[p][i], k, t0[p], te[p], live[p][i]);
            }
            again = again || n != guess[p];
            guess[p] = n;
        }
        if (!__any(again)) break;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        // cond.h's masses: total' re-summed in the blocked order, n_i = floor(e_i * (2^32 - 2^20) / total' + 0.5), then integers only
        const float scale = mass_scale(qperm_f<kLast>(quad_scan(((e4[p][0] + e4[p][1]) + e4[p][2]) + e4[p][3], m1, m2)));
        uint32_t n4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) n4[i] = mass_of(e4[p][i], scale);
        const int nb = (int)(((n4[0] + n4[1]) + n4[2]) + n4[3]);
        const int zs = quad_scan_i(nb, k1, k2);                      // masses up to and including my block
        const uint32_t r = __umulhi(du[p], (uint32_t)qperm_i<kLast>(zs)); // pick(): r = floor(u * N / 2^32)
        uint32_t cum = (uint32_t)(zs - nb);                          // masses of the blocks before mine
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            cum += n4[i];
            cnt += ((FULL ? (blk != 3u || i != 3) : (int)(4u * blk) + i < C[p] - 1) && r >= cum) ? 1 : 0;
        }
        cnt += qperm_i<kXor1>(cnt);
        val[p] = (uint32_t)(cnt + qperm_i<kXor2>(cnt));
    }
}

template <int NP>
__device__ __forceinline__ void pw4_cond(float (&w)[NP][4], const int (&C)[NP], const uint32_t (&du)[NP], const uint32_t blk, const float m1,
                                         const float m2, const int k1, const int k2, uint32_t (&val)[NP]) {
    bool full = true;
#pragma unroll
    for (int p = 0; p < NP; ++p) full = full && C[p] == 16;
    if (full) pw4_cond_impl<NP, true>(w, C, du, blk, m1, m2, k1, k2, val); // (uniform: Card is a property of the step's variable)
    else pw4_cond_impl<NP, false>(w, C, du, blk, m1, m2, k1, k2, val);
}

// LOG: write the draw log of gr_advance_chain (one row per wavefront = 16 chains, one count per value and step) and stop at
// the group's own stopping time (SweepParams::log / limit).
// SYS: GR_FLAG_SCAN_SYSTEMATIC -- the launch's variables (k_scan_vars) follow the sweep order, and update t of a chain takes
// word (sweep & 3) of philox_sys_draw(sweep = t / n, pos = t mod n, chain): one Philox call per step instead of one per quad
// (the stream is keyed by the position so that k_sweep_chroma's lanes, which own positions, need one call per 4 sweeps).
// Consecutive positions of a colour class never share a factor, so nearly every step runs paired with the next one.
template <bool LOG, bool SYS = false>
__global__ void __launch_bounds__(256) k_sweep_pw4(const SweepParams p) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t blk = lane & 3u;                   // which block of 4 values this lane owns
    const uint32_t cb = wave * 16u + (lane >> 2);     // chain inside the block
    const uint32_t chain = p.chain0 + blockIdx.x * 64u + cb;
    const uint32_t variant = (p.chain0 + blockIdx.x * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    unsigned char *st = lds + cb;                                        // st[var * 64]
    uint32_t *hist = reinterpret_cast<uint32_t *>(lds + p.lds_hist_off) + wave * p.H; // wave-private
    for (uint32_t v = blk; v < p.n_vars; v += 4u) st[v * 64u] = p.state[(size_t)v * p.pitch + chain];
    for (uint32_t i = lane; i < p.H; i += 64u) hist[i] = 0;
    __syncthreads();

    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const GR_CONST_AS u32x2v *plan = as_const<u32x2v>(reinterpret_cast<const uint2 *>(p.scan_vars) + (size_t)blockIdx.x * p.scan_pitch); // the block's scan stream
    const uint64_t step0 = ((uint64_t)p.step0_hi << 32) | p.step0_lo;
    const uint64_t step1 = step0 + p.n_steps;
    const uint32_t *progs = p.vprogs + vd.vprog_base;
    const uint32_t *poff = p.vprog_off + vd.progoff_base;
    const unsigned char *tab = reinterpret_cast<const unsigned char *>(p.views + vd.views_base);
    const bool counting = p.counting != 0;
    const float m1 = blk >= 1u ? 1.0f : 0.0f, m2 = blk >= 2u ? 1.0f : 0.0f;
    const int k1 = blk >= 1u ? -1 : 0, k2 = blk >= 2u ? -1 : 0; // the same as integer masks

    // the selected variable's (VarSample, sampler.go:135-174: evaluated by k_scan_vars for the whole launch) program header
    auto header = [&](const uint32_t v) -> StepHdr {
        StepHdr h;
        h.v = v;
        h.pg = progs + as_const<uint32_t>(poff)[h.v];
        const u32x4v w = *as_const<u32x4v>(h.pg);
        h.h = w[0]; h.ho = w[1]; h.f_lo = w[2]; h.f_hi = w[3];
        return h;
    };
    // f32 log sums of my 4 values, factors in Model.Funcs order (gibbs-simple.go:186).  The padding slots of the program read
    // an all-zero row (w + 0.0f is w), so nothing here is guarded: per slot one state byte, one multiply-add for the row
    // offset, one 16-byte load, two packed adds.
    // With two wavefronts per SIMD every instruction of a wavefront costs issue time, scalar ones included (measured: ~1
    // instruction per 4 cycles and SIMD whatever the mix), so the loop is built to need few of them: the 8 entries of a slot
    // group arrive as two s_load_dwordx16, and the row loads are buffer loads -- the entry's table offset goes into the
    // instruction's scalar-offset operand instead of a 64-bit scalar add per slot.
    const __amdgpu_buffer_rsrc_t views_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(tab), 0, 0x7fffffff, 0x00020000);
    auto gather = [&](const StepHdr &h, float (&w)[4]) {
        const uint32_t rowlen = (h.h >> 8) & 0xffu, n_slots = h.h >> 16; // slots = factors padded to a multiple of 8
        // byte offset of my block inside a row; a lane whose block lies past the variable's rows reads block 0 instead
        // (valid memory; its values are marked absent in pw4_cond)
        const uint32_t col = 4u * blk < rowlen ? 16u * blk : 0u;
        w[0] = w[1] = w[2] = w[3] = 0.f;
        const GR_CONST_AS u32x16 *ent = as_const<u32x16>(h.pg + 4); // 4 entries of {table offset, state row, stride, 0} each
        for (uint32_t f0 = 0; f0 < n_slots; f0 += 8) {
            const u32x16 ea = ent[f0 >> 2], eb = ent[(f0 >> 2) + 1u];
            uint32_t ex[8], ey[8], ez[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ex[i] = ea[4 * i]; ey[i] = ea[4 * i + 1]; ez[i] = ea[4 * i + 2];
                ex[4 + i] = eb[4 * i]; ey[4 + i] = eb[4 * i + 1]; ez[4 + i] = eb[4 * i + 2];
            }
            uint32_t x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = (uint32_t)st[ey[i]];
            u32x4v r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(views_rsrc, __umul24(x[i], ez[i]) + col, ex[i], 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                w[0] += __uint_as_float(r[i][0]); w[1] += __uint_as_float(r[i][1]);
                w[2] += __uint_as_float(r[i][2]); w[3] += __uint_as_float(r[i][3]);
            }
        }
    };
    // g.last[v] = val (gibbs-simple.go:267) and Marginal[val] += 1 (chain.go:230-236)
    auto commit = [&](const StepHdr &h, const uint32_t val) {
        if (blk == 0u) {
            st[h.v * 64u] = (unsigned char)val;
            if (counting) atomicAdd(&hist[h.ho + val], 1u);
        }
        // (the three other lanes of the chain read this byte in later steps: in-order LDS makes that work in hardware; the
        // fence keeps the compiler, which reasons per lane, from treating a later load of it as redundant in the lanes
        // that did not store -- see k_sweep_par4's re-gather)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // LOG: how many of the wavefront's 16 chains drew each value -- 4 ballots (one per bit of the 4-bit value, over the chains'
    // block-0 lanes); lane c (< 16) ANDs together, bit by bit, the ballot or its complement according to the bits of c, and
    // the popcount of that mask is the number of chains that drew c.  No LDS round trip.
    unsigned char *logrow = nullptr;
    uint32_t lim = 0xffffffffu;
    if constexpr (LOG) {
        const uint32_t group = (p.chain0 >> 6) + blockIdx.x;
        if (p.log) logrow = p.log + ((size_t)group * p.log_rows + wave) * p.log_pitch + p.log_off;
        if (p.limit) {
            const uint32_t t = p.limit[group];
            lim = t > p.win_off ? t - p.win_off : 0u;
        }
    }
    auto log_step = [&](const uint32_t val, const uint32_t sidx) {
        unsigned long long m = 0x1111111111111111ull; // the chains' block-0 lanes
#pragma unroll
        for (uint32_t b = 0; b < 4; ++b) {
            const unsigned long long bb = __ballot((val >> b) & 1u);
            m &= ((lane >> b) & 1u) ? bb : ~bb;
        }
        if (lane < 16u && lane < p.log_vals && logrow) logrow[(size_t)sidx * p.log_vals + lane] = (unsigned char)__popcll(m);
    };

    // SYS: sweep number and position in the order of the current quad's first step (uniform), kept by increments
    uint64_t sys_sweep = 0;
    uint32_t sys_pos = 0;
    if constexpr (SYS) {
        const uint64_t s_first = (step0 >> 2) << 2;
        sys_sweep = s_first / vd.n_free;
        sys_pos = (uint32_t)(s_first - sys_sweep * vd.n_free);
    }
    auto sys_draw = [&](const uint32_t j) -> uint32_t { // the draw of step j of the current quad
        uint64_t sw = sys_sweep;
        uint32_t ps = sys_pos + j;
        while (ps >= vd.n_free) { ps -= vd.n_free; ++sw; }
        const u32x4 d = philox_sys_draw(sw, ps, gchain, p.seed_lo, p.seed_hi);
        const uint32_t w = (uint32_t)sw & 3u;
        return w == 0 ? d.x : w == 1 ? d.y : w == 2 ? d.z : d.w;
    };
    const PrioRotor prio; // (wave_prio.h: the blocks of a CU progress as equals)
    for (uint64_t q = step0 >> 2; (q << 2) < step1; ++q) {
        const uint32_t q_lo = (uint32_t)q, q_hi = (uint32_t)(q >> 32);
        prio.tick(q_lo, p.prio_shift);
        const u32x2v sv = plan[q - (step0 >> 2)]; // the quad's four variables, 16 bits each
        const uint32_t sc[4] = {sv[0] & 0xffffu, sv[0] >> 16, sv[1] & 0xffffu, sv[1] >> 16};
        u32x4 dr{0, 0, 0, 0};
        if constexpr (!SYS) dr = philox4x32_10(q_lo, q_hi, (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
        // step index inside the launch, as 32 bits: a step before step0 (first quad of a quad-unaligned launch) wraps to a huge
        // value, so ONE unsigned scalar compare against the launch length covers both ends (64-bit compares are VALU work)
        const uint32_t rel0 = (uint32_t)((q << 2) - step0);
        int j = 0;
#pragma unroll 1
        while (j < 4) {
            const uint32_t sidx = rel0 + (uint32_t)j;
            if (sidx >= p.n_steps) { ++j; continue; }
            if (LOG && sidx >= lim) { ++j; continue; } // the group has finished its AdvanceChain: nothing changes any more
            const StepHdr ha = header(j == 0 ? sc[0] : j == 1 ? sc[1] : j == 2 ? sc[2] : sc[3]); // (no register arrays indexed by j)
            const uint32_t dua = SYS ? sys_draw((uint32_t)j) : j == 0 ? dr.x : j == 1 ? dr.y : j == 2 ? dr.z : dr.w;
            // the next step runs together with this one when it exists and its variable is neither this one nor a neighbour
            bool pair = j < 3 && sidx + 1u < p.n_steps && (!LOG || sidx + 1u < lim);
            StepHdr hb = ha;
            if (pair) {
                hb = header(j == 0 ? sc[1] : j == 1 ? sc[2] : sc[3]);
                const uint32_t bit = hb.v & 63u;
                pair = (((bit < 32u ? ha.f_lo >> bit : ha.f_hi >> (bit - 32u)) & 1u) == 0u);
            }
            if (pair) {
                const uint32_t dub = SYS ? sys_draw((uint32_t)j + 1u) : j == 0 ? dr.y : j == 1 ? dr.z : dr.w;
                float w[2][4];
                gather(ha, w[0]);
                gather(hb, w[1]);
                const int C[2] = {(int)(ha.h & 0xffu), (int)(hb.h & 0xffu)};
                const uint32_t du[2] = {dua, dub};
                uint32_t val[2];
                pw4_cond<2>(w, C, du, blk, m1, m2, k1, k2, val);
                commit(ha, val[0]);
                commit(hb, val[1]);
                if constexpr (LOG)
                    if (counting) { log_step(val[0], sidx); log_step(val[1], sidx + 1u); }
                j += 2;
            } else {
                float w[1][4];
                gather(ha, w[0]);
                const int C[1] = {(int)(ha.h & 0xffu)};
                const uint32_t du[1] = {dua};
                uint32_t val[1];
                pw4_cond<1>(w, C, du, blk, m1, m2, k1, k2, val);
                commit(ha, val[0]);
                if constexpr (LOG)
                    if (counting) log_step(val[0], sidx);
                j += 1;
            }
        }
        if constexpr (SYS) {
            sys_pos += 4u;
            while (sys_pos >= vd.n_free) { sys_pos -= vd.n_free; ++sys_sweep; }
        }
    }

    __syncthreads();
    for (uint32_t v = blk; v < p.n_vars; v += 4u) p.state[(size_t)v * p.pitch + chain] = st[v * 64u];
    if (counting) {
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = lane; i < p.H; i += 64u) {
            const uint32_t c = hist[i];
            if (c) atomicAdd(&gc[i], (unsigned long long)c);
        }
    }
}

hipError_t launch_sweep_pw4(const SweepParams &p, uint32_t blocks, uint32_t lds_bytes, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, s, p);
        return hipGetLastError();
    };
    if (p.sys_order) return (p.log || p.limit) ? hipErrorInvalidValue : go(&k_sweep_pw4<false, true>);
    return (p.log || p.limit) ? go(&k_sweep_pw4<true>) : go(&k_sweep_pw4<false>);
}

} // namespace gr
