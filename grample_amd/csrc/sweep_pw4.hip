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

namespace gr {

namespace {

__device__ __forceinline__ uint32_t uni_p(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

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
template <int CTRL>
__device__ __forceinline__ double qperm_d(double x) {
    return __hiloint2double(qperm_i<CTRL>(__double2hiint(x)), qperm_i<CTRL>(__double2loint(x)));
}

// cond.h's combination of the 4 block sums of a chain, lane c holding q_c: inclusive prefix z_c with
//   z_1 = q1 + q0, z_2 = (q2 + q1) + q0, z_3 = (q3 + q2) + (q1 + q0)   (two Hillis-Steele steps)
// m1 / m2 = 1.0 on the lanes that take part in a step, else 0.0 (fma(x, 1.0, y) is x + y; fma(x, 0.0, y) is y).
__device__ __forceinline__ double quad_scan(const double q, const double m1, const double m2) {
    const double y = fma(qperm_d<kShr1>(q), m1, q);
    return fma(qperm_d<kShr2>(y), m2, y);
}

} // namespace

template <int GF>
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
    const uint32_t group_gid = uni_p((uint32_t)(gchain >> 6)); // the block's scan stream
    const uint64_t step0 = ((uint64_t)p.step0_hi << 32) | p.step0_lo;
    const uint64_t step1 = step0 + p.n_steps;
    const uint32_t *progs = p.vprogs + vd.vprog_base;
    const uint32_t *poff = p.vprog_off + vd.progoff_base;
    const uint32_t *freev = p.free_vars + vd.free_base;
    const unsigned char *tab = reinterpret_cast<const unsigned char *>(p.views + vd.views_base);
    const bool counting = p.counting != 0;
    const double m1 = blk >= 1u ? 1.0 : 0.0, m2 = blk >= 2u ? 1.0 : 0.0;
    const int k1 = blk >= 1u ? -1 : 0, k2 = blk >= 2u ? -1 : 0; // the same as integer masks

    for (uint64_t q = step0 >> 2; (q << 2) < step1; ++q) {
        const uint32_t q_lo = (uint32_t)q, q_hi = (uint32_t)(q >> 32);
        const u32x4 sc = philox4x32_10<true>(q_lo, q_hi, group_gid, DOM_SCAN, p.seed_lo, p.seed_hi);                       // scalar unit
        const u32x4 dr = philox4x32_10(q_lo, q_hi, (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const uint64_t s = (q << 2) + j;
            if (s < step0 || s >= step1) continue;
            const uint32_t su = j == 0 ? sc.x : j == 1 ? sc.y : j == 2 ? sc.z : sc.w; // (no register arrays indexed by j)
            const uint32_t du = j == 0 ? dr.x : j == 1 ? dr.y : j == 2 ? dr.z : dr.w;
            const uint32_t v = freev[uni_p(mul_hi_u32(uni_p(su), vd.n_free))]; // VarSample (sampler.go:135-174)
            const uint32_t *pg = progs + poff[v];
            const uint32_t h = pg[0];
            const int C = (int)(h & 0xffu);
            const uint32_t rowlen = (h >> 8) & 0xffu, n_slots = h >> 16; // slots = factors padded to a multiple of GF
            const uint32_t ho = pg[1];
            // byte offset of my block inside a row; a lane whose block lies past the variable's rows reads block 0 instead
            // (valid memory; its values are marked absent below)
            const uint32_t col = 4u * blk < rowlen ? 16u * blk : 0u;

            // ---- f32 log sums of my 4 values, factors in Model.Funcs order (gibbs-simple.go:186).  The padding slots of
            // the program read an all-zero row (w + 0.0f is w), so nothing here is guarded: per slot one state byte, one
            // multiply-add for the row offset, one 16-byte load, two packed adds.
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            const uint4 *ent = reinterpret_cast<const uint4 *>(pg + 4);
            for (uint32_t f0 = 0; f0 < n_slots; f0 += GF) {
                uint4 e[GF];
#pragma unroll
                for (int i = 0; i < GF; ++i) e[i] = ent[f0 + (uint32_t)i];
                uint32_t x[GF];
#pragma unroll
                for (int i = 0; i < GF; ++i) x[i] = (uint32_t)st[e[i].y];
                float4 r[GF];
#pragma unroll
                for (int i = 0; i < GF; ++i) r[i] = *reinterpret_cast<const float4 *>(tab + e[i].x + (__umul24(x[i], e[i].z) + col));
#pragma unroll
                for (int i = 0; i < GF; ++i) { w[0] += r[i].x; w[1] += r[i].y; w[2] += r[i].z; w[3] += r[i].w; }
            }
            bool live[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                live[i] = (int)(4u * blk) + i < C;
                w[i] = live[i] ? w[i] : -INFINITY; // absent values: exp -> +0.0, never floored, never drawn
            }
            // ---- cond.h, value-parallel
            float m = fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3]));
            m = fmaxf(m, qperm_f<kXor1>(m));
            m = fmaxf(m, qperm_f<kXor2>(m));
            double d[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = (double)__expf(w[i] - m);
            const double tot = qperm_d<kLast>(quad_scan(((d[0] + d[1]) + d[2]) + d[3], m1, m2));
            const double t0 = kFloorEps * tot, te = kFloorEps * t0;
            int guess = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) guess += (live[i] && d[i] < t0) ? 1 : 0;
            double e4[4];
            for (;;) {
                // floors in the blocks before mine = exclusive quad prefix of the guesses
                int s1 = guess + (k1 & qperm_i<kShr1>(guess));
                s1 = s1 + (k2 & qperm_i<kShr2>(s1));
                double k = (double)(s1 - guess);
                int n = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e4[i] = d[i];
                    n += (live[i] && e4[i] < floor_thr(t0, te, k)) ? 1 : 0;
                    floor_step(e4[i], k, t0, te, live[i]);
                }
                if (!__any(n != guess)) break;
                guess = n;
            }
            const double zq = quad_scan(((e4[0] + e4[1]) + e4[2]) + e4[3], m1, m2);
            const double r = (double)du * (qperm_d<kLast>(zq) * 0x1p-32); // r = U * total' (sampler.go:110-117), cond.h's pick()
            const double before = qperm_d<kShr1>(zq);                      // sum of the blocks before mine (unused by block 0)
            double cum = fma(before, m1, e4[0]);
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i > 0) cum += e4[i];
                cnt += ((int)(4u * blk) + i < C - 1 && r >= cum) ? 1 : 0;
            }
            cnt += qperm_i<kXor1>(cnt);
            const uint32_t val = (uint32_t)(cnt + qperm_i<kXor2>(cnt));
            // ---- g.last[v] = val (gibbs-simple.go:267) and Marginal[val] += 1 (chain.go:230-236)
            if (blk == 0u) {
                st[v * 64u] = (unsigned char)val;
                if (counting) atomicAdd(&hist[ho + val], 1u);
            }
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
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sweep_pw4<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_sweep_pw4<8>), dim3(blocks), dim3(256), lds_bytes, s, p);
    return hipGetLastError();
}

} // namespace gr
