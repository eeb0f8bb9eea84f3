// The conditional law of one Gibbs update on the device: from the Card summed log potentials of a
// variable to its integer draw thresholds.  ONE definition, shared by every kernel (direct sweep,
// memoisation, probe, and the value-parallel sweep in sweep_pw4.hip), so that all of them produce the
// same thresholds bit for bit.
//
// Reference (sampler/gibbs-simple.go:227-258 + sampler/sampler.go:110-123):
//   shift so min(w) = 1.5 when min < -8; w = exp(w); tot = sum w; then SEQUENTIALLY for i:
//     if w_i / tot < 1e-6 { d = tot * 1e-6; tot += d; w_i += d }
//   WeightedSample re-sums the weights, r = U * tot2, picks the first i with r <= cum_i.
// Here (round 3: f32 + integer instead of f64 -- on gfx950 an f64 add / fma / compare and
// the two DPP moves a double takes cost 4.2 cycles of VALU issue each, an f32 add / mul / fma 2.3 and an integer add 2.3:
// profiles/r03_issue_rate_ubench.txt):
//   d_i   = __expf(w_i - max w)                         shift by the max: identical after normalisation   (f32)
//   sums  : values are grouped in BLOCKS of 4 (block b = values 4b .. 4b+3, summed left to right);
//           the (at most 4) block sums q_b of a group of 16 values are combined as
//               prefix of block 1 = q0, of block 2 = q0 + q1, of block 3 = (q1 + q2) + q0,
//               total = (q2 + q3) + (q0 + q1)          (absent blocks count as +0.0, which is exact)
//           -- a fixed summation order that a lane-per-chain loop and a 4-lanes-per-chain scan both
//           reproduce exactly (f32: at most 64 terms in (0, 1], relative error < 5e-7)
//   floor : t0 = 1e-6 * total; SEQUENTIALLY for i, k = floors so far: thr = t0 * (1 + k * 1e-6); if d_i < thr { d_i += thr; k++ }
//           -- the reference's rule with thr standing for tot * 1e-6 (its delta): same predicate, same increments; the
//           running total grows linearly in k instead of geometrically (floor_thr: 1.2e-10 relative at Card 16)
//   masses: total' re-summed after the floor (WeightedSample re-sums, sampler.go:110-116); every value gets the INTEGER mass
//               n_i = floor(d_i * ((2^32 - 2^20) / total') + 0.5)
//           (relative error of a mass ~1e-6 from exp, sums and the division; +-0.5 units = 1.2e-10 of probability), and
//           from here on everything is exact integer arithmetic: P_i = n_0 + .. + n_i (inclusive prefix), N = P_{Card-1}
//           -- about 2^32 - 2^20 and always below 2^32, whatever the rounding of the masses did to their sum, because the draw
//           scales by N itself:
//   draw  : value = #{ i < Card-1 : mulhi(u, N) >= P_i },  u a 32-bit uniform: r = floor(u * N / 2^32) is uniform on [0, N)
//           up to one part in 2^32, and the reference's "first i with r <= cum_i" is this count.  One multiply per update
//           and one integer compare per value.  The memoised tables store T_i = ceil(P_i * 2^32 / N) = min{ u : mulhi(u, N)
//           >= P_i } (threshold_u32, exact), so "u >= T_i" there is the same predicate for every u.
//           A probability is thereby a multiple of about 2^-32 (2.3e-10): 1.2e-4 relative at the 1e-6 floor, as with the
//           f64 formulation of rounds 1 and 2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gr {

constexpr float kFloorEps = 1e-6f; // gibbs-simple.go:248 (probability floor), also the growth of the running total
constexpr float kMassScale = 4293918720.0f; // 2^32 - 2^20 (exact in f32): the masses of one conditional sum to about this -- and, whatever
                                            // their rounding (< 2e-6 relative + 32 units), to less than 2^32

// The floor threshold after k floors: 1e-6 x the running total.  The reference grows the total by its own 1e-6 at every
// floor, tot_k = tot_0 (1 + 1e-6)^k; here tot_k = tot_0 (1 + k 1e-6) -- the same to within k^2 / 2 * 1e-12 (1.2e-10 relative
// at Card 16), far inside the f32 exp error -- because this form is a pure function of k: a lane that owns a block of
// values needs the number of floors before it, not a k-step recurrence.  t0 = 1e-6 * tot_0, te = 1e-6 * t0.
__device__ __forceinline__ float floor_thr(const float t0, const float te, const float k) { return fmaf(k, te, t0); }

// one step of the sequential rule (gibbs-simple.go:246-258): value d against the threshold after k floors
__device__ __forceinline__ void floor_step(float &d, float &k, const float t0, const float te, const bool live) {
    const float thr = floor_thr(t0, te, k);
    const float fs = (live && d < thr) ? 1.0f : 0.0f;
    d = fmaf(fs, thr, d); // d += thr when floored (fs * thr is exact)
    k += fs;
}

// (2^32 - 2^20) / total' -- one definition (V_RCP_F32, 1 ulp, + one multiply) so that every kernel form scales alike
__device__ __forceinline__ float mass_scale(const float total) { return kMassScale * __builtin_amdgcn_rcpf(total); }
// the integer mass of a value (0 for an absent one: d = 0)
__device__ __forceinline__ uint32_t mass_of(const float d, const float scale) { return (uint32_t)fmaf(d, scale, 0.5f); }

// After the floor: the inclusive integer prefix sums P[i] (i < Card - 1) of the masses and their total N.
// FULL = the variable uses all MAXC values: same arithmetic, none of the per-value `i < C` guards.
template <int MAXC, bool FULL>
__device__ __forceinline__ void cumulative_impl(const float (&w)[MAXC], const int C, uint32_t (&P)[MAXC > 1 ? MAXC - 1 : 1], uint32_t &N) {
    constexpr int NB = (MAXC + 3) / 4; // blocks of 4 values
    constexpr int NG = (NB + 3) / 4;   // groups of 4 blocks
    float m = w[0];
#pragma unroll
    for (int i = 1; i < MAXC; ++i)
        if (FULL || i < C) m = fmaxf(m, w[i]);
    float d[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) d[i] = (FULL || i < C) ? __expf(w[i] - m) : 0.0f;

    // the total, in the blocked order
    auto total = [&]() -> float {
        float q[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            q[b] = d[4 * b];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (4 * b + k < MAXC) q[b] += d[4 * b + k];
        }
        float base = 0.0f; // total of the groups before the current one
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int b0 = 4 * g, nb = (NB - b0) < 4 ? (NB - b0) : 4;
            auto qb = [&](int j) -> float { return q[b0 + j < NB ? b0 + j : NB - 1]; }; // (index clamped: never read past nb)
            float z = qb(0); // inclusive prefix of the group's last block = the group's total
            if (nb == 2) z = qb(0) + qb(1);
            if (nb == 3) z = (qb(1) + qb(2)) + qb(0);
            if (nb == 4) z = (qb(2) + qb(3)) + (qb(0) + qb(1));
            base = g == 0 ? z : base + z;
        }
        return base;
    };

    const float t0 = kFloorEps * total(), te = kFloorEps * t0;
    float k = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) floor_step(d[i], k, t0, te, FULL || i < C); // every value must stay possible (gibbs-simple.go:246-258)

    const float scale = mass_scale(total()); // WeightedSample re-sums the weights (sampler.go:110-116)
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        c += mass_of(d[i], scale); // (an absent value adds 0)
        if (i < MAXC - 1 && (FULL || i < C - 1)) P[i] = c;
    }
    N = c;
}

template <int MAXC>
__device__ __forceinline__ void cumulative(const float (&w)[MAXC], const int C, uint32_t (&P)[MAXC > 1 ? MAXC - 1 : 1], uint32_t &N) {
    if (C == MAXC) cumulative_impl<MAXC, true>(w, C, P, N);
    else cumulative_impl<MAXC, false>(w, C, P, N);
}

// the categorical draw (sampler.go:117-123): how many prefix sums the scaled uniform has reached
template <int MAXC>
__device__ __forceinline__ uint32_t pick(const uint32_t (&P)[MAXC > 1 ? MAXC - 1 : 1], const uint32_t N, const int C, const uint32_t u) {
    const uint32_t r = __umulhi(u, N);
    uint32_t val = 0;
    if (C == MAXC) {
#pragma unroll
        for (int i = 0; i < MAXC - 1; ++i) val += (r >= P[i]) ? 1u : 0u;
    } else {
#pragma unroll
        for (int i = 0; i < MAXC - 1; ++i)
            if (i < C - 1) val += (r >= P[i]) ? 1u : 0u;
    }
    return val;
}

// The 32-bit threshold equivalent to one compare of pick(): the smallest u with mulhi(u, N) >= P, i.e. ceil(P * 2^32 / N).
// P < N for every i < Card - 1 (the values behind it keep a mass of at least 1e-6 of N ~ 4 294 units after the floor), so the
// quotient is below 2^32.  Used by the memoisation kernel and the probe -- "u >= threshold_u32(P_i, N)" is, for every
// 32-bit u, the predicate the direct kernels evaluate.
__device__ __forceinline__ uint32_t threshold_u32(const uint32_t P, const uint32_t N) {
    const unsigned long long num = ((unsigned long long)P << 32) + (unsigned long long)N - 1ull;
    return (uint32_t)(num / (unsigned long long)N);
}

} // namespace gr
