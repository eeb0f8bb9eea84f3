// Micro-benchmark for VERDICT r02 "next #5" (bit-packed binary state on the headline kernel): the inner loop of the 4-lane
// memoised sweep -- gather the 4 neighbour states, build the configuration index, fetch the memoised threshold, compare with
// the draw, commit -- with the chain state held two ways, everything else equal:
//   BYTES  one byte per (variable, chain) in LDS, row stride 80 (k_sweep_par4's layout): ds_read_u8 at (row + chain),
//          index by multiply-adds, ds_write_b8
//   BITS   one 16-bit mask per (variable, wavefront) -- the 16 chains of a step slot -- : ds_read_u16 at a row-uniform
//          address, bit (chain) of each mask extracted and accumulated, the step's new mask = the 16-bit slice of the
//          compare's lane mask, ds_write_b16
// Same geometry as the kernel (256-lane blocks = 64 chains, lane = step slot * 16 + chain, 4 wavefronts per SIMD: 4 blocks
// per CU), same record fetch (one 16-byte + one 8-byte LDS read per step), same cheap hash in place of Philox (the draw
// stream is not what is compared), no dependency levels, no histogram.  The variables of a step come from a plan in LDS.
// Output: ns per step-quad and updates/s for both forms, and the VALU / LDS instruction counts of the two loops (from the
// disassembly: scripts/ubench/state_form.bin).
// Build: hipcc --offload-arch=gfx950 -O2 -o state_form.bin state_form.hip      Output: profiles/r03_state_form_ubench.txt
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

constexpr int kVars = 100, kPlan = 4096, kRowBytes = 80, kRowBits = 8; // (4 wavefronts x u16)

struct Rec { uint32_t n[4]; uint32_t tab, self; uint32_t pad[2]; }; // neighbour row offsets, threshold-table offset, own row offset

__device__ __forceinline__ uint32_t hash_draw(uint32_t s, uint32_t chain) {
    uint32_t u = s * 2654435761u ^ chain * 40503u;
    u ^= u >> 15; u *= 0x2c1b3c6du; u ^= u >> 12;
    return u;
}

// LDS: [0, st_bytes) state, then records (32 B each), then thresholds (16 u32 per variable), then the plan (u8 per step)
template <bool BITS>
__global__ void __launch_bounds__(256, 4) k_state_form(const Rec *recs, const uint32_t *thr, const unsigned char *plan, uint32_t n_iter, uint32_t *out) {
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr uint32_t st_bytes = BITS ? kVars * kRowBits : kVars * kRowBytes;
    constexpr uint32_t rec_off = (st_bytes + 15u) & ~15u, thr_off = rec_off + kVars * 32u, plan_off = thr_off + kVars * 64u;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6, j = lane >> 4, c = lane & 15u, cb = wave * 16u + c;
    for (uint32_t i = tid; i < kVars * 8u; i += 256) reinterpret_cast<uint32_t *>(lds + rec_off)[i] = reinterpret_cast<const uint32_t *>(recs)[i];
    for (uint32_t i = tid; i < kVars * 16u; i += 256) reinterpret_cast<uint32_t *>(lds + thr_off)[i] = thr[i];
    for (uint32_t i = tid; i < kPlan; i += 256) lds[plan_off + i] = plan[i];
    for (uint32_t i = tid; i < st_bytes; i += 256) lds[i] = (unsigned char)((i * 37u + blockIdx.x) >> 3 & (BITS ? 0xffu : 1u));
    __syncthreads();
    const uint32_t chain = blockIdx.x * 64u + cb;
    const uint32_t lane_base = BITS ? wave * 2u : cb; // what is added to a row offset
    uint32_t acc = 0;
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t s = (it * 4u + j) & (kPlan - 1u);
        const uint32_t v = lds[plan_off + s];
        const uint4 rn = *reinterpret_cast<const uint4 *>(lds + rec_off + v * 32u);
        const uint2 rt = *reinterpret_cast<const uint2 *>(lds + rec_off + v * 32u + 16u);
        const uint32_t u = hash_draw(it * 4u + j, chain);
        uint32_t idx;
        if constexpr (BITS) {
            const uint32_t m0 = *reinterpret_cast<const unsigned short *>(lds + rn.x + lane_base);
            const uint32_t m1 = *reinterpret_cast<const unsigned short *>(lds + rn.y + lane_base);
            const uint32_t m2 = *reinterpret_cast<const unsigned short *>(lds + rn.z + lane_base);
            const uint32_t m3 = *reinterpret_cast<const unsigned short *>(lds + rn.w + lane_base);
            idx = ((m0 >> c) & 1u) | (((m1 >> c) & 1u) << 1) | (((m2 >> c) & 1u) << 2) | (((m3 >> c) & 1u) << 3);
        } else {
            const uint32_t x0 = lds[rn.x + lane_base], x1 = lds[rn.y + lane_base], x2 = lds[rn.z + lane_base], x3 = lds[rn.w + lane_base];
            idx = x0 + 2u * x1 + 4u * x2 + 8u * x3;
        }
        const uint32_t T = *reinterpret_cast<const uint32_t *>(lds + rt.x + idx * 4u);
        const bool val = u >= T;
        if constexpr (BITS) {
            const unsigned long long b = __ballot(val);
            const uint32_t half = j >= 2u ? (uint32_t)(b >> 32) : (uint32_t)b;
            *reinterpret_cast<unsigned short *>(lds + rt.y + lane_base) = (unsigned short)(half >> (16u * (j & 1u))); // the 16 lanes of the slot write the same value
        } else {
            lds[rt.y + lane_base] = val ? 1 : 0;
        }
        acc += val ? 1u : 0u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    out[blockIdx.x * 256u + tid] = acc;
}

int main() {
    // a 10 x 10 grid's neighbours (torus), like Grids_13's blankets
    std::vector<Rec> recs[2];
    for (int bits = 0; bits < 2; ++bits) {
        const uint32_t row = bits ? kRowBits : kRowBytes;
        const uint32_t st_bytes = kVars * row, rec_off = (st_bytes + 15u) & ~15u, thr_off = rec_off + kVars * 32u;
        for (int v = 0; v < kVars; ++v) {
            const int x = v % 10, y = v / 10;
            const int nb[4] = {((x + 1) % 10) + y * 10, ((x + 9) % 10) + y * 10, x + ((y + 1) % 10) * 10, x + ((y + 9) % 10) * 10};
            Rec r{};
            for (int k = 0; k < 4; ++k) r.n[k] = (uint32_t)nb[k] * row;
            r.tab = thr_off + (uint32_t)v * 64u; r.self = (uint32_t)v * row;
            recs[bits].push_back(r);
        }
    }
    std::vector<uint32_t> thr(kVars * 16);
    for (size_t i = 0; i < thr.size(); ++i) thr[i] = 0x20000000u + (uint32_t)(i * 2654435761u >> 2);
    std::vector<unsigned char> plan(kPlan);
    uint32_t x = 12345;
    for (int i = 0; i < kPlan; ++i) { x = x * 1664525u + 1013904223u; plan[i] = (unsigned char)((x >> 8) % kVars); }
    Rec *d_recs; uint32_t *d_thr, *d_out; unsigned char *d_plan;
    const uint32_t blocks = 1024; // 65 536 chains: 4 blocks per CU, 4 wavefronts per SIMD
    CHK(hipMalloc(&d_recs, kVars * sizeof(Rec))); CHK(hipMalloc(&d_thr, thr.size() * 4)); CHK(hipMalloc(&d_plan, kPlan)); CHK(hipMalloc(&d_out, blocks * 256 * 4));
    CHK(hipMemcpy(d_thr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_plan, plan.data(), kPlan, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const uint32_t n_iter = 50000 / 4 * 4; // 200 000 steps per chain
    std::printf("# state_form: %u chains (%u blocks of 64), %u updates per chain, 4 lanes per chain, 4 wavefronts per SIMD\n", blocks * 64, blocks, n_iter * 4);
    for (int rep = 0; rep < 2; ++rep)
        for (int bits = 0; bits < 2; ++bits) {
            CHK(hipMemcpy(d_recs, recs[bits].data(), kVars * sizeof(Rec), hipMemcpyHostToDevice));
            const uint32_t st_bytes = kVars * (bits ? kRowBits : kRowBytes);
            const uint32_t lds = ((st_bytes + 15u) & ~15u) + kVars * 32u + kVars * 64u + kPlan;
            CHK(hipEventRecord(e0));
            if (bits) hipLaunchKernelGGL(k_state_form<true>, dim3(blocks), dim3(256), lds, 0, d_recs, d_thr, d_plan, n_iter, d_out);
            else hipLaunchKernelGGL(k_state_form<false>, dim3(blocks), dim3(256), lds, 0, d_recs, d_thr, d_plan, n_iter, d_out);
            CHK(hipGetLastError());
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
            const double upd = (double)blocks * 64.0 * n_iter * 4.0;
            if (rep) std::printf("%-5s state: %8.3f ms  %.3e updates/s  (LDS %u B per block)\n", bits ? "BITS" : "BYTES", ms, upd / (ms * 1e-3), lds);
        }
    return 0;
}
