// The conditional law of one Gibbs update on the device: from the Card summed log potentials of a
// variable to its cumulative draw thresholds.  ONE definition, shared by every kernel (direct sweep,
// memoisation, probe, and the value-parallel sweep in sweep_pw4.hip), so that all of them produce the
// same thresholds bit for bit.
//
// Reference (sampler/gibbs-simple.go:227-258 + sampler/sampler.go:110-123):
//   shift so min(w) = 1.5 when min < -8; w = exp(w); tot = sum w; then SEQUENTIALLY for i:
//     if w_i / tot < 1e-6 { d = tot * 1e-6; tot += d; w_i += d }
//   WeightedSample re-sums the weights, r = U * tot2, picks the first i with r <= cum_i.
// Here:
//   d_i   = (double) __expf(w_i - max w)               shift by the max: identical after normalisation
//   sums  : values are grouped in BLOCKS of 4 (block b = values 4b .. 4b+3, summed left to right);
//           the (at most 4) block sums q_b of a group of 16 values are combined as
//               prefix of block 1 = q0, of block 2 = q0 + q1, of block 3 = (q1 + q2) + q0,
//               total = (q2 + q3) + (q0 + q1)          (absent blocks count as +0.0, which is exact)
//           -- a fixed summation order that a lane-per-chain loop and a 4-lanes-per-chain scan both
//           reproduce exactly (f64; any order is within 1e-16 of the reference's left-to-right sum)
//   floor : t0 = 1e-6 * total; SEQUENTIALLY for i, k = floors so far: thr = t0 * (1 + k * 1e-6); if d_i < thr { d_i += thr; k++ }
//           -- the reference's rule with thr standing for tot * 1e-6 (its delta): same predicate, same increments; the
//           running total grows linearly in k instead of geometrically (floor_thr: 1.2e-10 relative at Card 16)
//   draw  : value = #{ i < Card-1 : (double)u * s >= cum_i },  u a 32-bit uniform, s = total' * 2^-32 (exact),
//           total' and cum_i re-summed after the floor in the block order above (cum inside a block: prefix of the
//           block, then left to right).  The reference draws r = U * total' and takes the first i with r <= cum_i: the
//           same law with U quantised to 32 bits.  One multiply per update and one compare per value -- no division, no
//           per-value scaling.  The memoised tables store the equivalent integer thresholds
//           T_i = min{ u : (double)u * s >= cum_i } (threshold_u32, exact), so "u >= T_i" there is the same predicate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gr {

constexpr double kFloorEps = 1e-6; // gibbs-simple.go:248 (probability floor), also the growth of the running total

// The floor threshold after k floors: 1e-6 x the running total.  The reference grows the total by its own 1e-6 at every
// floor, tot_k = tot_0 (1 + 1e-6)^k; here tot_k = tot_0 (1 + k 1e-6) -- the same to within k^2 / 2 * 1e-12 (1.2e-10 relative
// at Card 16), far inside the f32 exp error -- because this form is a pure function of k: a lane that owns a block of
// values needs the number of floors before it, not a k-step recurrence.  t0 = 1e-6 * tot_0, te = 1e-6 * t0.
__device__ __forceinline__ double floor_thr(const double t0, const double te, const double k) { return fma(k, te, t0); }

// one step of the sequential rule (gibbs-simple.go:246-258): value d against the threshold after k floors
__device__ __forceinline__ void floor_step(double &d, double &k, const double t0, const double te, const bool live) {
    const double thr = floor_thr(t0, te, k);
    const double fs = (live && d < thr) ? 1.0 : 0.0;
    d = fma(fs, thr, d); // d += thr when floored (fs * thr is exact)
    k += fs;
}

// After the floor: the cumulative sums cum[i] (i < Card - 1) and s = total' * 2^-32.
// FULL = the variable uses all MAXC values: same arithmetic, none of the per-value `i < C` guards.
template <int MAXC, bool FULL>
__device__ __forceinline__ void cumulative_impl(const float (&w)[MAXC], const int C, double (&cum)[MAXC > 1 ? MAXC - 1 : 1], double &s) {
    constexpr int NB = (MAXC + 3) / 4; // blocks of 4 values
    constexpr int NG = (NB + 3) / 4;   // groups of 4 blocks
    float m = w[0];
#pragma unroll
    for (int i = 1; i < MAXC; ++i)
        if (FULL || i < C) m = fmaxf(m, w[i]);
    double d[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) d[i] = (FULL || i < C) ? (double)__expf(w[i] - m) : 0.0;

    // block sums, block prefixes P[b] (sum of the blocks before b; P[0] is unused) and the total
    double P[NB];
    auto sums = [&]() -> double {
        double q[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            q[b] = d[4 * b];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (4 * b + k < MAXC) q[b] += d[4 * b + k];
        }
        double base = 0.0; // total of the groups before the current one
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int b0 = 4 * g, nb = (NB - b0) < 4 ? (NB - b0) : 4;
            auto qb = [&](int j) -> double { return q[b0 + j < NB ? b0 + j : NB - 1]; }; // (index clamped: never read past nb)
            double z[4] = {qb(0), 0.0, 0.0, 0.0}; // inclusive prefix inside the group
            if (nb > 1) z[1] = qb(0) + qb(1);
            if (nb > 2) z[2] = (qb(1) + qb(2)) + qb(0);
            if (nb > 3) z[3] = (qb(2) + qb(3)) + z[1];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nb) {
                    if (g == 0) { if (j > 0) P[b0 + j] = z[j - 1]; }
                    else P[b0 + j] = j > 0 ? base + z[j - 1] : base;
                }
            base = g == 0 ? z[nb - 1] : base + z[nb - 1];
        }
        return base;
    };

    const double t0 = kFloorEps * sums(), te = kFloorEps * t0;
    double k = 0.0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) floor_step(d[i], k, t0, te, FULL || i < C); // every value must stay possible (gibbs-simple.go:246-258)

    s = sums() * 0x1p-32; // WeightedSample re-sums the weights (sampler.go:110-116); r = U * total' becomes (double)u * s
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        double c = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * b + k;
            if (i < MAXC - 1) {
                c = (k == 0) ? (b == 0 ? d[i] : P[b] + d[i]) : c + d[i];
                if (FULL || i < C - 1) cum[i] = c;
            }
        }
    }
}

template <int MAXC>
__device__ __forceinline__ void cumulative(const float (&w)[MAXC], const int C, double (&cum)[MAXC > 1 ? MAXC - 1 : 1], double &s) {
    if (C == MAXC) cumulative_impl<MAXC, true>(w, C, cum, s);
    else cumulative_impl<MAXC, false>(w, C, cum, s);
}

// the categorical draw (sampler.go:117-123): how many cumulative sums the scaled uniform has passed
template <int MAXC>
__device__ __forceinline__ uint32_t pick(const double (&cum)[MAXC > 1 ? MAXC - 1 : 1], const double s, const int C, const uint32_t u) {
    const double r = (double)u * s;
    uint32_t val = 0;
    if (C == MAXC) {
#pragma unroll
        for (int i = 0; i < MAXC - 1; ++i) val += (r >= cum[i]) ? 1u : 0u;
    } else {
#pragma unroll
        for (int i = 0; i < MAXC - 1; ++i)
            if (i < C - 1) val += (r >= cum[i]) ? 1u : 0u;
    }
    return val;
}

// The integer threshold equivalent to one compare of pick(): the smallest u with (double)u * s >= cum, found exactly
// (u * s is monotone in u; the quotient gives a candidate within one step of it).  Every remaining value keeps a mass
// of at least 1e-6 of the total after the floor, so such a u always exists below 2^32.  Used by the memoisation kernel
// and the probe -- "u >= threshold_u32(cum_i, s)" is, for every 32-bit u, the predicate the direct kernels evaluate.
__device__ __forceinline__ uint32_t threshold_u32(const double cum, const double s) {
    double t = fmin(ceil(cum / s), 4294967295.0);
    while (t > 0.0 && (t - 1.0) * s >= cum) t -= 1.0;
    while (t < 4294967295.0 && t * s < cum) t += 1.0;
    return (uint32_t)t;
}

} // namespace gr
