// Checks the slot-count sequence of k_sweep_par4 (mbcnt + add-with-carry + DPP subtract) against popcounts, on random masks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t *in, int *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const bool val = in[blockIdx.x * 64 + lane] & 1u;
    const unsigned long long drawn = __ballot(val);
    const uint32_t excl = __builtin_amdgcn_mbcnt_hi((uint32_t)(drawn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)drawn, 0u));
    uint32_t incl;
    asm("v_addc_co_u32_e64 %0, vcc, %1, 0, %2" : "=v"(incl) : "v"(excl), "s"(drawn) : "vcc");
    uint32_t a = incl, b = incl;
    asm("s_nop 1\n\tv_subrev_u32_dpp %0, %1, %1 row_bcast:15 row_mask:0xe bank_mask:0xf" : "+v"(a) : "v"(incl));
    asm("s_nop 1\n\tv_sub_u32_dpp %0, %1, %1 row_bcast:15 row_mask:0xe bank_mask:0xf" : "+v"(b) : "v"(incl));
    const int prev = __builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xe, 0xf, false);
    const uint32_t row = lane >> 4;
    int *o = out + (blockIdx.x * 64 + lane) * 6;
    o[0] = (int)excl; o[1] = (int)incl; o[2] = prev; o[3] = (int)a; o[4] = (int)b; o[5] = __popcll(drawn & (0xffffull << (16 * row)));
}
int main() {
    const int B = 4;
    uint32_t h[B * 64]; static int o[B * 64 * 6];
    uint32_t x = 12345;
    for (int i = 0; i < B * 64; ++i) { x = x * 1664525u + 1013904223u; h[i] = x >> 13; }
    uint32_t *di; int *dout;
    (void)hipMalloc(&di, sizeof h); (void)hipMalloc(&dout, sizeof o);
    (void)hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, di, dout);
    (void)hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i)
        if ((i & 15) >= 14 || (i & 15) == 0) printf("lane %2d: excl %2d incl %2d prev %2d subrev_dpp %3d sub_dpp %3d ref %2d\n", i, o[i*6], o[i*6+1], o[i*6+2], o[i*6+3], o[i*6+4], o[i*6+5]);
    return 0;
}
