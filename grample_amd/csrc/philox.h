// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
// as 1, 2, 3", SC'11) -- counter-based generator used for every random choice
// on the device.  Replaces the reference's shared MT19937-64 stream
// (rand/rand.go:19-54): a draw is a pure function of (seed, chain, step), so
// results do not depend on launch geometry, chunking or the number of GPUs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GR_HD __host__ __device__ __forceinline__
#else
#define GR_HD inline
#endif

namespace gr {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u;
constexpr uint32_t PHILOX_M1 = 0xCD9E8D57u;
constexpr uint32_t PHILOX_W0 = 0x9E3779B9u;
constexpr uint32_t PHILOX_W1 = 0xBB67AE85u;

// counter word 3 selects the stream ("domain"); words 0..1 carry the 64-bit
// position inside the stream, word 2 the chain / wave / variant index.
enum : uint32_t {
    DOM_DRAW = 1,     // per chain, per 4 steps: the categorical draws
    DOM_SCAN = 2,     // per wavefront, per 4 steps: which variable to update
    DOM_INIT = 3,     // per chain, per variable: start state
    DOM_COLLAPSE = 4, // per variant, per try: Collapse(-1) selection
    DOM_SCAN_CHAIN = 5, // per chain, per 4 steps: which variable to update (strict mode GR_FLAG_SCAN_PER_CHAIN)
    DOM_DRAW_SYS = 6,   // systematic scan (GR_FLAG_SCAN_SYSTEMATIC): per chain, per POSITION in the sweep order, per 4 sweeps: the draws
};

struct u32x4 {
    uint32_t x, y, z, w;
};

// a ^ b ^ c: one V_BITOP3_B32 on gfx950 (truth table 0x96) instead of two V_XOR_B32 -- the compiler does not form it
// from the expression; a third of Philox's 60 vector instructions otherwise
GR_HD uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
    return a ^ b ^ c;
#endif
}

// UNIFORM = every argument is wave-uniform (the per-group scan stream): plain xors, so that the whole block stays on the
// scalar unit -- V_BITOP3 exists on the vector unit only and would drag a uniform Philox call onto it
template <bool UNIFORM = false>
GR_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        const uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = UNIFORM ? ((uint32_t)(p1 >> 32) ^ c1 ^ k0) : xor3((uint32_t)(p1 >> 32), c1, k0);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = UNIFORM ? ((uint32_t)(p0 >> 32) ^ c3 ^ k1) : xor3((uint32_t)(p0 >> 32), c3, k1);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// The draws of the systematic scan: update `pos` of sweep `sweep` of chain `gchain` uses word (sweep & 3) of this call, so that
// a lane that owns one position of the order (k_sweep_chroma) needs one Philox call per 4 sweeps and no exchange with other
// lanes.  Counter: (sweep / 4 [low 32 bits], pos [20 bits] | sweep / 4 [bits 32..43] << 20, chain [low 32], domain | chain [high] << 4).
GR_HD u32x4 philox_sys_draw(uint64_t sweep, uint32_t pos, uint64_t gchain, uint32_t k0, uint32_t k1) {
    const uint64_t sq = sweep >> 2;
    return philox4x32_10((uint32_t)sq, pos | ((uint32_t)(sq >> 32) << 20), (uint32_t)gchain, DOM_DRAW_SYS + ((uint32_t)(gchain >> 32) << 4), k0, k1);
}

// uniform integer in [0, n) from a 32-bit word: floor(u * n / 2^32).  The
// selection probabilities differ from 1/n by at most n/2^32 (2.3e-8 for n=100);
// a random-scan Gibbs chain has the same stationary law for any positive
// selection probabilities, so this never biases a marginal.
GR_HD uint32_t mul_hi_u32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

} // namespace gr
