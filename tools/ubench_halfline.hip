// ubench_halfline.hip -- does a per-lane window gather out of 64-byte half-lines move less than out of 128-byte lines?  
//
// k_step<true, .> fetches its 7x7 window from the env's window plane: ONE 128-byte line (8 rows x 16 cells) per env-step for 49 useful bytes --
// the counters say 1.6 x the algorithmic bytes at 4.0-4.4 TB/s.  A plane of 64-byte half-lines (8 rows x 8 cells; origin classes per 2 cells in x
// and per 2 rows in y; ~2.5 x the plane memory) would hold any window in ONE half-line.  Whether that helps depends on what the memory system
// moves for a 64-byte-aligned 64-byte gather: a half line, or the whole 128-byte line anyway.  This measures exactly that, at k_step's shape:
// one lane = one env, 1 048 576 envs, every lane one pseudo-random line / half-line of ITS OWN region (the regions far apart: nothing is shared),
// the window's rows as k_step loads them (line: 7 x 12 bytes at a 16-byte pitch; half-line: 7 x 8 bytes at an 8-byte pitch), 16 bytes of
// coalesced SoA traffic next to it.  Variants:
//   line128     region = 52 lines of 128 B (BossLevel's window plane, 6.5 KB per env)
//   half64      region = 256 half-lines of 64 B (16 KB per env), 64-byte aligned
//   half64x2    the same gather but two adjacent half-lines per lane (128 B, line-aligned): what a 128-byte fetch granularity would make half64 cost
// One JSON line each: us per launch, GB/s of USEFUL window bytes (49 B per lane).  Run under rocprofv3 --pmc FETCH_SIZE for the bytes.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_halfline tools/ubench_halfline.hip && /tmp/ubench_halfline
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>      // 0 line128, 1 half64, 2 half64x2
__global__ __launch_bounds__(64) void k_gather(int64_t n, const uint8_t* __restrict__ plane, const uint32_t* __restrict__ soa, uint32_t* __restrict__ out, uint32_t salt) {
    const int64_t env = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (env >= n) return;
    const uint32_t r = mix((uint32_t)env * 2654435761u + salt);
    const uint32_t s = soa[env];
    uint32_t acc = s;
    if (MODE == 0) {
        const uint8_t* line = plane + env * (int64_t)(52 * 128) + (r % 52) * 128 + ((r >> 8) & 1) * 16 + ((r >> 9) & 1) * 4;
#pragma unroll
        for (int k = 0; k < 7; ++k) { const uint32_t* q = (const uint32_t*)(line + 16 * k); acc += q[0] ^ q[1] ^ q[2]; }
    } else {
        const uint8_t* half = plane + env * (int64_t)(256 * 64) + (r % 256) * 64;
#pragma unroll
        for (int k = 0; k < 7; ++k) { const uint32_t* q = (const uint32_t*)(half + 8 * k); acc += q[0] ^ q[1]; }
        if (MODE == 2) {
            const uint8_t* other = (const uint8_t*)((uintptr_t)half ^ 64);
#pragma unroll
            for (int k = 0; k < 7; ++k) { const uint32_t* q = (const uint32_t*)(other + 8 * k); acc += q[0] ^ q[1]; }
        }
    }
    out[env] = acc;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1048576;
    uint8_t* plane; uint32_t *soa, *out;
    const size_t bytes = (size_t)n * 256 * 64;
    if (hipMalloc(&plane, bytes) != hipSuccess) { printf("{\"error\": \"hipMalloc\"}\n"); return 1; }
    (void)hipMalloc(&soa, n * 4); (void)hipMalloc(&out, n * 4);
    (void)hipMemset(plane, 1, bytes); (void)hipMemset(soa, 2, n * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"line128", "half64", "half64x2"};
    const unsigned grid = (unsigned)((n + 63) / 64);
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 3; ++v) {
            float sum = 0, best = 1e9f; const int reps = 40, skip = 8;
            for (int r = 0; r < reps; ++r) {
                (void)hipEventRecord(e0, 0);
                const uint32_t salt = 17u + 977u * r;          // another line per launch: nothing stays in a cache between launches by luck
                if (v == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(64), 0, 0, n, plane, soa, out, salt);
                else if (v == 1) hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(64), 0, 0, n, plane, soa, out, salt);
                else hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(64), 0, 0, n, plane, soa, out, salt);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= skip) { sum += ms; if (ms < best) best = ms; }
            }
            const float avg = sum / (reps - skip);
            printf("{\"variant\": \"%s\", \"envs\": %lld, \"rep\": %d, \"us_avg\": %.2f, \"us_best\": %.2f, \"useful_window_GBs\": %.1f, \"if_128B_lines_GBs\": %.1f, \"if_64B_halves_GBs\": %.1f}\n",
                   names[v], (long long)n, rep, avg * 1e3, best * 1e3, n * 49.0 / (avg * 1e-3) / 1e9, n * 128.0 * (v == 2 ? 1 : 1) / (avg * 1e-3) / 1e9,
                   n * 64.0 * (v == 2 ? 2 : 1) / (avg * 1e-3) / 1e9);
        }
    return 0;
}
