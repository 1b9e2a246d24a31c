// ubench_listatomic.hip -- what one list atomic per stepping wave costs, and what reading NEXT to it costs.  (experiment for round 5)
//
// k_step appends a wave's finished envs to a list with ONE returning atomicAdd per wave (bbai_engine.hip, "compact finished envs into
// the reset list"); on reset-heavy batches nearly every wave has one (PickupLoc 262 144 envs: ~2 700 of 4 096 waves per step).  Round 4
// found (profiles/r04/NOTES.md section 11) that a plain LOAD of the same cache line by every wave -- the window's count block,
// win_prefix -- queued behind those atomics and, loads returning in order, held up everything the wave loaded after it: k_step 0.094 ms
// against 0.057 with the counts on lines of their own (WIN_ENTRY).  What is left, by arithmetic only, is the atomic itself: ~2 700 x
// 11-16 ns on one address = 0.03-0.045 ms of a 0.057-ms kernel.  This tool isolates both effects with k_step's shape (one wave per
// block, a dependent load -> some arithmetic -> the atomic -> a dependent store):
//   counters = 1, 4, 16, 64   sub-lists: wave w adds to counter w % counters (each on its own 128-byte line)
//   reader = none | same | own    every wave first loads a word from the line of counter 0 (same) or from a line nobody writes (own),
//                                 and a second, independent word AFTER it (what in-order return delays)
//   frac = share of the waves that have something to append (the reset rate)
// Output: one JSON line per (counters, reader, frac): microseconds per launch (median of 50), waves, atomics per launch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_listatomic tools/ubench_listatomic.hip && /tmp/ubench_listatomic 4096
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// READER: 0 none, 1 the line of counter 0, 2 a line nobody writes
template <int READER>
__global__ __launch_bounds__(64) void k_list(int64_t n_envs, const uint32_t* __restrict__ state, uint32_t* counters /* [ncount][32] */, int ncount,
                                             uint32_t thresh /* a wave appends iff mix(block) < thresh */, const uint32_t* __restrict__ quiet,
                                             int32_t* __restrict__ list, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x;
    const int64_t env = (int64_t)blockIdx.x * 64 + lane;
    if (env >= n_envs) return;
    uint32_t acc = 0;
    if (READER == 1) acc += counters[1 + (lane & 7)];           // (words 1..8 of counter 0's line: never written, always zero)
    if (READER == 2) acc += quiet[lane & 7];
    uint32_t v = state[env];                                    // the step's own first load: issued AFTER the read above, returns after it
    v += acc;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) v = mix(v + k);                // a little dependent arithmetic (the step body)
    const bool mine = mix((uint32_t)blockIdx.x * 2654435761u) < thresh && (lane & 31) == 7;     // two finished envs per appending wave
    const unsigned long long bal = __ballot(mine);
    if (bal) {
        const int leader = __ffsll((long long)bal) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&counters[32 * (blockIdx.x % ncount)], (uint32_t)__popcll(bal));
        base = __shfl(base, leader);
        if (mine) list[(int64_t)(blockIdx.x % ncount) * n_envs + base + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)env;
    }
    out[env] = v;
}

int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 4096;
    const int64_t n = (int64_t)waves * 64;
    uint32_t *state, *counters, *quiet, *out;
    int32_t* list;
    HIP_OK(hipMalloc(&state, n * 4)); HIP_OK(hipMalloc(&out, n * 4)); HIP_OK(hipMalloc(&counters, 64 * 32 * 4)); HIP_OK(hipMalloc(&quiet, 256));
    HIP_OK(hipMalloc(&list, 64 * n * 4));
    HIP_OK(hipMemset(state, 1, n * 4)); HIP_OK(hipMemset(quiet, 0, 256));
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a)); HIP_OK(hipEventCreate(&b));
    const double fracs[] = {0.0, 0.1, 0.67, 1.0};
    const int ncounts[] = {1, 4, 16, 64};
    for (int reader = 0; reader < 3; ++reader)
        for (double frac : fracs)
            for (int nc : ncounts) {
                if (frac == 0.0 && nc != 1) continue;
                const uint32_t thresh = frac >= 1.0 ? 0xFFFFFFFFu : (uint32_t)(frac * 4294967296.0);
                std::vector<float> ms;
                for (int rep = 0; rep < 60; ++rep) {
                    HIP_OK(hipMemsetAsync(counters, 0, 64 * 32 * 4, 0));
                    HIP_OK(hipEventRecord(a, 0));
                    if (reader == 0) hipLaunchKernelGGL(k_list<0>, dim3(waves), dim3(64), 0, 0, n, state, counters, nc, thresh, quiet, list, out);
                    else if (reader == 1) hipLaunchKernelGGL(k_list<1>, dim3(waves), dim3(64), 0, 0, n, state, counters, nc, thresh, quiet, list, out);
                    else hipLaunchKernelGGL(k_list<2>, dim3(waves), dim3(64), 0, 0, n, state, counters, nc, thresh, quiet, list, out);
                    HIP_OK(hipEventRecord(b, 0));
                    HIP_OK(hipEventSynchronize(b));
                    float t = 0;
                    HIP_OK(hipEventElapsedTime(&t, a, b));
                    if (rep >= 10) ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                uint32_t h[64 * 32];
                HIP_OK(hipMemcpy(h, counters, sizeof(h), hipMemcpyDeviceToHost));
                uint64_t entries = 0;
                for (int c = 0; c < nc; ++c) entries += h[32 * c];
                printf("{\"waves\": %d, \"reader\": \"%s\", \"frac\": %.2f, \"counters\": %d, \"us_per_launch\": %.2f, \"us_min\": %.2f, \"atomics\": %llu}\n", waves,
                       reader == 0 ? "none" : reader == 1 ? "same line" : "own line", frac, nc, ms[ms.size() / 2] * 1e3, ms.front() * 1e3,
                       (unsigned long long)(entries / 2));
                fflush(stdout);
            }
    return 0;
}
