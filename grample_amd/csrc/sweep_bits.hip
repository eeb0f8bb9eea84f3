// K1, step-parallel form with BIT-PACKED chain state (fast_form 11; VERDICT r02 "next #5" / SURVEY 8(f) N4).
//
// k_sweep_par4's geometry and process -- 256-lane blocks = one 64-chain scan group, lane = step slot j * 16 + chain c, the scan
// plan of k_scan_plan with its dependency levels, the same Philox words, thresholds, commit order and counting -- but the state of
// a variable for the 16 chains of a wavefront is ONE 16-bit mask in LDS (8 bytes per variable and block instead of 80): a step
// slot reads its neighbours' masks with row-uniform ds_read_u16, extracts bit (chain) of each, and writes the variable's new mask
// as the 16-bit slice of the compare's lane mask.  Bit-identical to every other form.
// Reference path, per update: VarSample sampler/sampler.go:135-174 (the scan plan), the gather of the blanket's state
// gibbs-simple.go:186-220 (here: bit (chain) of each neighbour's mask -> configuration index), stabilise / exp / floor :227-258 and
// WeightedSample sampler.go:90-130 (the memoised threshold of that configuration against the chain's draw), write-back :262-270
// (the mask slice), Marginal[val] += 1 chain.go:230-236 (the (zeros, ones) histogram pair).
//
// Built to MEASURE the form, not to replace the byte layout: the inner loop needs the same number of LDS instructions (a
// row-uniform u16 read costs the LDS pipe what a per-lane byte read does) and ~10 VALU instructions more per step for the bit
// extracts and the mask slice (scripts/ubench/state_form.hip: 1.06e12 against 1.26e12 updates/s in isolation); what it saves is
// LDS footprint, which the headline launch does not lack.  The sampler picks it by itself for the models whose STATE fills LDS (more than
// half of a CU's 160 KB as bytes: dv-rel_2, 1 200 variables -- two blocks per CU instead of one, 3.0e11 -> 4.6e11); fast_form 11 forces it.
//
// Records (64 bytes per selectable variable, built by gr_sampler::rebuild for this form with absolute LDS addresses -- the dynamic LDS
// array must start at LDS address 0, checked like in k_sweep_par4 -- and read from global memory one iteration ahead):
//   w0 = the variable's mask row (var * 8)    w1 = its (zeros, ones) histogram pair    w2 = its threshold table    w3 = neighbours
//   w[4 + i] = neighbour's mask row | (4 * multiplier) << 16   (unused slots: the variable's own row, multiplier 0)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_layout.h"
#include "kernels.h"
#include "philox.h"
#include "wave_prio.h"

namespace gr {

namespace {
typedef __attribute__((address_space(3))) uint32_t lds_u32b_t;
typedef __attribute__((address_space(3))) unsigned short lds_u16b_t;
__device__ __forceinline__ uint32_t lds_addr_b(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }
__device__ __forceinline__ uint32_t uni_bb(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
// (high half of w) * b + c in one instruction (k_sweep_par4's multiply-add on packed slot words)
__device__ __forceinline__ uint32_t mad_u16_hi_b(const uint32_t w, const uint32_t b, const uint32_t c) {
    uint32_t d;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(w), "v"(b), "v"(c));
    return d;
}
} // namespace

// TABG: the threshold pool stays in global memory (pools too large for LDS: record word 2 = byte offset into the group's pool)
template <int SLOTS, bool COUNT, bool TABG = false>
__global__ void __launch_bounds__(256) k_sweep_bits(const FastParams p) {
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, j = lane >> 4, c = lane & 15u;
    const uint32_t group = blockIdx.x, chain = group * 64u + wave * 16u + c;
    const uint32_t variant = (group * 64u) / p.chains_per_variant;
    const VariantDesc vd = p.variants[variant];
    const uint32_t base = lds_addr_b(lds);
    // LDS: masks [n_vars][4 wavefronts] u16 | histogram pairs [n_free] | threshold tables | records
    const uint32_t hist_off = (p.n_vars * 8u + 15u) & ~15u, tab_off = (hist_off + vd.n_free * 8u + 15u) & ~15u;
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.cpt + vd.cpt_base);
        uint4 *dst = reinterpret_cast<uint4 *>(lds + tab_off);
        if constexpr (!TABG)
            for (uint32_t i = tid; i < ((vd.cpt_len + 3u) >> 2); i += 256u) dst[i] = src[i];
        uint32_t *h = reinterpret_cast<uint32_t *>(lds + hist_off);
        for (uint32_t i = tid; i < 2u * vd.n_free; i += 256u) h[i] = 0;
        // pack: step slot j takes the variables v = j (mod 4); the compare's lane mask holds one variable per DPP row
        for (uint32_t v0 = 0; v0 < p.n_vars; v0 += 4u) {
            const uint32_t v = v0 + j;
            const bool on = v < p.n_vars;
            const uint32_t b = on ? (uint32_t)p.state[(size_t)v * p.pitch + chain] : 0u;
            const unsigned long long m = __ballot(b != 0u);
            const uint32_t half = j >= 2u ? (uint32_t)(m >> 32) : (uint32_t)m;
            if (on && c == 0u) *reinterpret_cast<unsigned short *>(lds + v * 8u + wave * 2u) = (unsigned short)(half >> (16u * (j & 1u)));
        }
    }
    __syncthreads();
    if (base != 0u) { // the records hold absolute LDS addresses (see k_sweep_par4): fail loudly
        if (threadIdx.x == 0) atomicOr(p.err, 1u);
        return;
    }
    const uint32_t w2 = wave * 2u; // my wavefront's mask inside a row
    const uint64_t gchain = (((uint64_t)p.chain_offset_hi << 32) | p.chain_offset_lo) + chain;
    const uint64_t q0 = (((uint64_t)p.step0_hi << 32) | p.step0_lo) >> 2;
    const uint32_t nq = p.n_steps >> 2; // multiple of 4
    const uint32_t *gplan = p.plan + (size_t)group * p.plan_pitch + j;
    const unsigned char *recs_g = reinterpret_cast<const unsigned char *>(p.recs_par);
    const unsigned char *tabg = reinterpret_cast<const unsigned char *>(p.cpt + vd.cpt_base);
    const bool tail = c == 15u, hi_half = j >= 2u;
    const uint32_t sl_shift = 16u * (j & 1u); // my step slot's 16 bits inside its half of a lane mask

    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    // the conditional of one step from the masks: threshold word of the configuration, compared with the draw
    // (h, sw: the step's record -- header and neighbour words -- read once per step with 16-byte LDS reads)
    auto sample = [&](const v4u &h, const v4u (&sw)[SLOTS / 4], const uint32_t u) -> bool {
        uint32_t m[SLOTS];
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) m[i] = *(lds_u16b_t *)(uintptr_t)((sw[i / 4][i % 4] & 0xffffu) + w2);
        uint32_t a = h.z; // table address
#pragma unroll
        for (int i = SLOTS - 1; i >= 0; --i) a = mad_u16_hi_b(sw[i / 4][i % 4], __builtin_amdgcn_ubfe(m[i], c, 1u), a); // bit (chain) of the mask x multiplier
        if constexpr (TABG) return u >= *reinterpret_cast<const uint32_t *>(tabg + a);
        else return u >= *(lds_u32b_t *)(uintptr_t)a;
    };

    // GR_FLAG_SCAN_SYSTEMATIC (p.sys_rank set): the plan follows the sweep order; my steps are 4 apart
    const bool sys = p.sys_rank != nullptr;
    uint64_t sys_sweep = 0;
    uint32_t sys_pos = 0;
    if (sys) {
        const uint64_t s_first = (q0 << 2) + j;
        sys_sweep = s_first / vd.n_free;
        sys_pos = (uint32_t)(s_first - sys_sweep * vd.n_free);
    }
    const PrioRotor prio;
    // plan words two iterations ahead, records one iteration ahead (global memory: the vector-memory path is otherwise idle)
    uint32_t pwa[4], pwn[4];
    v4u hn[4], swn[4][SLOTS / 4];
    const __amdgpu_buffer_rsrc_t rec_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(recs_g), 0, 0x7fffffff, 0x00020000);
    auto fetch = [&](const int k) { // (buffer loads: the record pool is the resource, the plan word's offset the vector offset -- no 64-bit address adds)
        const uint32_t off = pwn[k] & (kPlanOffMask & ~63u);
        hn[k] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, off, 0, 0));
#pragma unroll
        for (int i = 0; i < SLOTS / 4; ++i) swn[k][i] = __builtin_bit_cast(v4u, __builtin_amdgcn_raw_buffer_load_b128(rec_rsrc, off + 16u + 16u * (uint32_t)i, 0, 0));
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) { pwn[k] = gplan[(size_t)min((uint32_t)k, nq - 1u) * 4u]; pwa[k] = gplan[(size_t)min(4u + (uint32_t)k, nq - 1u) * 4u]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) fetch(k);
    for (uint32_t g0 = 0; g0 < nq; g0 += 4) {
        prio.tick(g0, p.prio_shift);
        uint32_t pwc[4];
        v4u hc[4], swc[4][SLOTS / 4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pwc[k] = pwn[k]; hc[k] = hn[k];
#pragma unroll
            for (int i = 0; i < SLOTS / 4; ++i) swc[k][i] = swn[k][i];
            pwn[k] = pwa[k];
            pwa[k] = gplan[(size_t)min(g0 + 8u + (uint32_t)k, nq - 1u) * 4u];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) fetch(k);
        uint32_t U[4];
        if (sys) { // GR_FLAG_SCAN_SYSTEMATIC: one Philox call per step, keyed by (sweep, position) -- see k_sweep_dir4
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32x4 d = philox_sys_draw(sys_sweep, sys_pos, gchain, p.seed_lo, p.seed_hi);
                const uint32_t w = (uint32_t)sys_sweep & 3u;
                U[k] = w == 0 ? d.x : w == 1 ? d.y : w == 2 ? d.z : d.w;
                sys_pos += 4u;
                while (sys_pos >= vd.n_free) { sys_pos -= vd.n_free; ++sys_sweep; }
            }
        } else {
            // draws: my Philox call is quad g0 + j; the 4 x 4 transpose of k_sweep_par4
            const uint64_t q = q0 + g0 + j;
            const u32x4 W = philox4x32_10((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)gchain, DOM_DRAW + ((uint32_t)(gchain >> 32) << 4), p.seed_lo, p.seed_hi);
            const auto a = __builtin_amdgcn_permlane32_swap(W.x, W.z, false, false);
            const auto b = __builtin_amdgcn_permlane32_swap(W.y, W.w, false, false);
            const auto cc = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
            const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
            U[0] = cc[0]; U[1] = cc[1]; U[2] = d[0]; U[3] = d[1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t pw = pwc[k];
            const v4u h = hc[k];
            const v4u (&sw)[SLOTS / 4] = swc[k];
            const uint32_t own = h.x + wave * 2u; // my variable's mask of my wavefront
            // commit: the step slot's 16 draws are a 16-bit slice of the compare's lane mask
            auto commit = [&](const unsigned long long drawn) {
                const uint32_t half = hi_half ? (uint32_t)(drawn >> 32) : (uint32_t)drawn;
                *(lds_u16b_t *)(uintptr_t)own = (unsigned short)(half >> sl_shift); // (the slot's 16 lanes store the same value)
            };
            bool val = sample(h, sw, U[k]);
            unsigned long long drawn = __ballot(val);
            if (pw < kPlanLevelBit) commit(drawn); // level 0: commit in the first pass
            if (__any(pw >= kPlanDepsBit)) {       // the quad has steps that do not commute: re-evaluate level by level
                const uint32_t lv = (pw >> 1) & 3u, maxlv = uni_bb(pw >> 3) & 3u;
                for (uint32_t level = 1; level <= maxlv; ++level) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (reads masks other step slots have just written: see k_sweep_par4)
                    const bool v2 = sample(h, sw, U[k]);
                    const unsigned long long mine = __ballot(lv == level);
                    drawn = (drawn & ~mine) | (__ballot(v2) & mine);
                    if (lv == level) commit(drawn);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if constexpr (COUNT) {
                const uint32_t half = hi_half ? (uint32_t)(drawn >> 32) : (uint32_t)drawn;
                const uint32_t s1 = (uint32_t)__builtin_popcount((half >> sl_shift) & 0xffffu);
                if (tail) {
                    const uint32_t hp = h.y;
                    __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>((uintptr_t)hp),
                                           ((unsigned long long)s1 << 32) | (unsigned long long)(16u - s1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t v0 = 0; v0 < p.n_vars; v0 += 4u) { // unpack
        const uint32_t v = v0 + j;
        if (v < p.n_vars) {
            const uint32_t m = *reinterpret_cast<const unsigned short *>(lds + v * 8u + wave * 2u);
            p.state[(size_t)v * p.pitch + chain] = (unsigned char)((m >> c) & 1u);
        }
    }
    if constexpr (COUNT) {
        const uint32_t *h = reinterpret_cast<const uint32_t *>(lds + hist_off);
        unsigned long long *gc = p.counts + (size_t)variant * p.H;
        for (uint32_t i = tid; i < vd.n_free; i += 256u) {
            const uint32_t v = p.free_vars[vd.free_base + i];
            const uint32_t hoff = p.cprogs[vd.cprog_base + p.cprog_off[vd.cprogoff_base + v] + 1u];
            if (h[2u * i]) atomicAdd(&gc[hoff], (unsigned long long)h[2u * i]);
            if (h[2u * i + 1u]) atomicAdd(&gc[hoff + 1u], (unsigned long long)h[2u * i + 1u]);
        }
    }
}

template <int SLOTS>
static hipError_t launch_bits_t(const FastParams &p, uint32_t blocks, uint32_t lds, hipStream_t s) {
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, p);
        return hipGetLastError();
    };
    if (p.tab_global) return p.counting ? go(&k_sweep_bits<SLOTS, true, true>) : go(&k_sweep_bits<SLOTS, false, true>);
    return p.counting ? go(&k_sweep_bits<SLOTS, true>) : go(&k_sweep_bits<SLOTS, false>);
}

// slots: most neighbours of a variable (<= 12); lds: bytes of the layout above for the largest group
hipError_t launch_sweep_bits(const FastParams &p, int slots, uint32_t blocks, uint32_t lds, hipStream_t s) {
    if (slots <= 4) return launch_bits_t<4>(p, blocks, lds, s);
    if (slots <= 8) return launch_bits_t<8>(p, blocks, lds, s);
    if (slots <= 12) return launch_bits_t<12>(p, blocks, lds, s);
    return hipErrorInvalidValue;
}

} // namespace gr
