// Micro-benchmark: issue cost of the integer instructions Philox is made of, per wavefront, at 1 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int OP>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters, unsigned long long *cyc) {
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77u, d = b + 99u;
    uint32_t e = a * 3u, f = b * 5u, g = c * 7u, h = d * 11u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { // v_mad_u64_u32 (what the compiler emits for the 32x32->64 products)
                uint64_t p0 = (uint64_t)a * 0xD2511F53u, p1 = (uint64_t)b * 0xCD9E8D57u, p2 = (uint64_t)c * 0xD2511F53u, p3 = (uint64_t)d * 0xCD9E8D57u;
                a = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; b = (uint32_t)(p1 >> 32) ^ (uint32_t)p2; c = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; d = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
            } else if (OP == 1) { // xor only (same dependency shape)
                a ^= b + 1u; b ^= c + 2u; c ^= d + 3u; d ^= a + 4u;
                a ^= e; b ^= f; c ^= g; d ^= h;
            } else if (OP == 2) { // v_mul_hi_u32
                a = __umulhi(a, 0xD2511F53u) ^ b; b = __umulhi(b, 0xCD9E8D57u) ^ c; c = __umulhi(c, 0xD2511F53u) ^ d; d = __umulhi(d, 0xCD9E8D57u) ^ a;
            } else if (OP == 3) { // v_mul_lo_u32
                a = a * 0xD2511F53u ^ b; b = b * 0xCD9E8D57u ^ c; c = c * 0xD2511F53u ^ d; d = d * 0xCD9E8D57u ^ a;
            } else if (OP == 4) { // v_mul_u32_u24 / mad_u24
                a = __umul24(a, 0x511F53u) ^ b; b = __umul24(b, 0x9E8D57u) ^ c; c = __umul24(c, 0x511F53u) ^ d; d = __umul24(d, 0x9E8D57u) ^ a;
            } else if (OP == 5) { // v_mul_hi_u32_u24
                a = (uint32_t)(((uint64_t)(a & 0xffffffu) * 0x511F53u) >> 32) ^ b; b = (uint32_t)(((uint64_t)(b & 0xffffffu) * 0x9E8D57u) >> 32) ^ c; c = (uint32_t)(((uint64_t)(c & 0xffffffu) * 0x511F53u) >> 32) ^ d; d = (uint32_t)(((uint64_t)(d & 0xffffffu) * 0x9E8D57u) >> 32) ^ a;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
static void run(const char *name, int ops_per_u, int waves_per_simd) {
    uint32_t *out; unsigned long long *cyc;
    const int blocks = 256 * waves_per_simd; // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(out, 1, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, 2, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double groups = (double)iters * 8; // groups of 4 independent (op + xor) pairs
    std::printf("%-18s waves/SIMD=%d: %.3f ms, %.1f ns per group of 4x(%s+xor) per wave => %.2f ns per instr-pair; clock counter %.1f ticks/group\n", name, waves_per_simd, ms,
                ms * 1e6 / groups, name, ms * 1e6 / groups / 4, (double)c / groups);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("mad_u64_u32", 1, w);
        run<1>("xor_add", 1, w);
        run<2>("mul_hi_u32", 1, w);
        run<3>("mul_lo_u32", 1, w);
        run<4>("mul_u32_u24", 1, w);
        run<5>("mul_hi_u32_u24", 1, w);
    }
    return 0;
}
