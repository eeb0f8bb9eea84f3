// Rotating wavefront priority for the sweep kernels (device code only).  (No counterpart in the reference: its chains are goroutines the Go
// scheduler runs, sampler/chain.go:197-215; here they are wavefronts a SIMD's arbiter picks from.)
//
// A gfx950 SIMD's arbiter serves its OLDEST wavefront first.  The sweep kernels are long loops of equal work per block, and a
// CU holds a few blocks at once (one wavefront of each per SIMD): the block that arrived first runs ahead of the others all
// the way, leaves early, and the CU finishes the launch at 3, 2, then 1 wavefront per SIMD -- measured on the headline launch
// (Grids_13, 65 536 chains = exactly 4 blocks per CU, GR_DEV_BLOCK_TIMES): block times 3.05 .. 4.44 ms, mean 3.69; the launch
// lasts as long as the slowest block, 17 % of the wavefront slots' time is idle.  With the priority of every wavefront
// rotating -- priority (t + slot on the SIMD) mod 4, t advancing every few thousand instructions of the wavefront's own
// loop -- the blocks of a CU progress as equals: 3.68 .. 3.99 ms, and the launch is 10 % shorter (profiles/r03_block_times.txt).
#pragma once
#include <stdint.h>

namespace gr {

struct PrioRotor {
    uint32_t slot; // HW_ID.wave_id: which of the SIMD's wavefront slots this wavefront sits in (scalar)
    __device__ __forceinline__ PrioRotor() : slot((uint32_t)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3u) {}
    // call with a wave-uniform loop counter; the priority changes when its low `shift` bits are zero (all scalar work);
    // shift >= 32: never (the kernels take it from their parameters: the host picks the period per kernel family)
    __device__ __forceinline__ void tick(const uint32_t it, const uint32_t shift) const {
        if (shift < 32u && (it & ((1u << shift) - 1u)) == 0u) {
            const uint32_t pr = ((it >> shift) + slot) & 3u;
            if (pr == 0u) __builtin_amdgcn_s_setprio(0);
            else if (pr == 1u) __builtin_amdgcn_s_setprio(1);
            else if (pr == 2u) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
    }
};

} // namespace gr
