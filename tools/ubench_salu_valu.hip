// ubench_salu_valu.hip -- do scalar-unit-bound waves and vector-ALU-bound waves of one CU add up?  (round 5: the level generator is one
// serial program per env; compiled wave-uniform it runs on the CU's scalar unit, compiled per lane group on the SIMDs' vector ALUs -- if the
// two kinds of wave overlap, a refill split between both shapes would generate more levels per second than either.)
// One kernel, one wave per block, 16 resident waves per CU; every wave runs ITERS rounds of a dependent integer chain (an LCG + xorshift,
// like a draw + test) either on SGPRs (mode S: values made uniform with readfirstlane) or on VGPRs (mode V).  mode 0: all V, 1: all S,
// 2: half of the blocks S, half V, mixed on every CU.  Prints microseconds per launch; additive units => mode 2 ~= max(half of 0, half of 1).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_salu_valu tools/ubench_salu_valu.hip && /tmp/ubench_salu_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
__global__ __launch_bounds__(64, 4) void k_mix(int mode, int iters, unsigned* out) {
    const bool scalar = mode == 1 || (mode == 2 && (__popc(blockIdx.x) & 1));      // (parity of the index bits: every CU of every XCD gets both kinds -- blockIdx & 1
                                                                                    // would put all odd blocks on the odd XCDs)
    unsigned acc = 0;
    if (scalar) {
        unsigned s = __builtin_amdgcn_readfirstlane(blockIdx.x * 2654435761u + 12345u);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s = s * 1664525u + 1013904223u;
                s ^= s >> 11;
                s = __builtin_amdgcn_readfirstlane(s);
                if ((s & 1023u) == 7u) s += 3u;            // a rarely taken scalar branch
            }
        }
        acc = s;
    } else {
        unsigned v = (blockIdx.x * 64 + (threadIdx.x >> 5)) * 2654435761u + 12345u;      // two "lane groups" per wave with different values
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v = v * 1664525u + 1013904223u;
                v ^= v >> 11;
                if ((v & 1023u) == 7u) v += 3u;            // divergent between the groups now and then
            }
        }
        acc = v;
    }
    if (acc == 0x12345u) out[blockIdx.x] = acc;
}
int main() {
    unsigned* out; hipMalloc(&out, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 16, iters = 4096;
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<float> ms;
        for (int rep = 0; rep < 12; ++rep) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(64), 0, 0, mode, iters, out);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b);
            if (rep >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("{\"mode\": \"%s\", \"blocks\": %d, \"iters\": %d, \"us_per_launch\": %.1f}\n", mode == 0 ? "all vector" : mode == 1 ? "all scalar" : "half scalar, half vector", blocks, iters, ms[ms.size() / 2] * 1e3);
    }
    return 0;
}
