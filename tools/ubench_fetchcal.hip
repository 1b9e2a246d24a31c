// ubench_fetchcal.hip -- what does rocprofv3's FETCH_SIZE count for a SPARSE gather on gfx950?  (calibration tool)
//
// MI355X_MICROARCH.md: FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reports exactly half the bytes of a wide coalesced stream
// (128-byte requests tallied at 64 B); "other access widths are uncalibrated".  k_step's traffic figure (round 2: 761 MB
// per launch = 3.09 x algorithmic) doubled FETCH_SIZE on the strength of a calibration on k_render's COALESCED reads.
// This tool runs access patterns whose distinct 64-byte and 128-byte pieces are known exactly, so that a
// `rocprofv3 --pmc FETCH_SIZE` pass over it tells which granularity a per-lane gather really costs:
//   stream16    every lane reads 16 contiguous bytes (the guide's calibrated case)
//   one_per_256 every lane reads ONE dword, lanes 256 B apart  -> 1 line (128 B) / 1 half-line (64 B) per lane
//   one_per_128 every lane reads ONE dword, lanes 128 B apart
//   one_per_64  every lane reads ONE dword, lanes  64 B apart  -> two lanes share a 128-B line
//   window      k_step's gather: 7 rows x 3 dwords at a 32-byte pitch out of a 1728-byte record, window origin
//               pseudo-random per env (distinct 128-B lines and 64-B pieces per env counted on the host: printed)
// Each kernel touches every byte at most once (no reuse, arrays far larger than L2 + the memory-side cache).
// Output: one JSON line per kernel with the exact byte counts at both granularities; the PMC pass supplies FETCH_SIZE.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_fetchcal tools/ubench_fetchcal.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__host__ __device__ inline uint32_t hmix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void cal_stream16(int64_t n, const u32x4* __restrict__ src, uint32_t* __restrict__ sink) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = src[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[0] = 1;
}
template <int STRIDE>
__global__ __launch_bounds__(256) void cal_one_per(int64_t n, const uint8_t* __restrict__ src, uint32_t* __restrict__ sink) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = *(const uint32_t*)(src + i * STRIDE + 4 * (mix((uint32_t)i) % (STRIDE / 4 < 16 ? STRIDE / 4 : 16)));
    if (v == 0x12345678u) sink[0] = 1;
}
constexpr int REC = 1728, ES = 32;
__host__ __device__ inline int window_origin(int64_t env) {
    const uint32_t r = hmix((uint32_t)env * 2654435761u + 17u);
    const int tx = (int)(r % 16), ty = (int)((r >> 8) % 16);            // top-left cell of the 7x7 window incl. margin
    return ty * ES + tx;
}
__global__ __launch_bounds__(256) void cal_window(int64_t n, const uint8_t* __restrict__ rec, uint32_t* __restrict__ sink) {
    const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (env >= n) return;
    const int a0 = window_origin(env);
    const uint32_t* q = (const uint32_t*)(rec + env * REC + (a0 & ~3));
    uint32_t s = 0;
#pragma unroll
    for (int r = 0; r < 7; ++r) s += q[r * (ES / 4)] + q[r * (ES / 4) + 1] + q[r * (ES / 4) + 2];
    if (s == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1048576;
    const size_t bytes = (size_t)n * 2048;
    uint8_t* buf; uint32_t* sink;
    (void)hipMalloc(&buf, bytes + 4096); (void)hipMalloc(&sink, 64);
    (void)hipMemset(buf, 1, bytes + 4096); (void)hipMemset(sink, 0, 64);
    (void)hipDeviceSynchronize();
    const unsigned grid = (unsigned)((n + 255) / 256);
    // exact distinct pieces of the window pattern
    double lines128 = 0, pieces64 = 0;
    for (int64_t e = 0; e < n; ++e) {
        const int64_t base = e * REC + (window_origin(e) & ~3);
        uint64_t seen128[8], seen64[16]; int n128 = 0, n64 = 0;
        for (int r = 0; r < 7; ++r)
            for (int d = 0; d < 3; ++d) {
                const int64_t a = base + r * ES + 4 * d;
                const uint64_t l = a >> 7, p = a >> 6;
                bool f = false; for (int k = 0; k < n128; ++k) f |= seen128[k] == l; if (!f) seen128[n128++] = l;
                f = false; for (int k = 0; k < n64; ++k) f |= seen64[k] == p; if (!f) seen64[n64++] = p;
            }
        lines128 += n128; pieces64 += n64;
    }
    // (adjacent envs share a line at record boundaries only when 1728-byte records straddle: counted per env, an upper bound within 2 %)
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        struct { const char* name; double b128, b64, useful; } rows[5] = {
            {"stream16", (double)n * 16, (double)n * 16, (double)n * 16},
            {"one_per_256", (double)n * 128, (double)n * 64, (double)n * 4},
            {"one_per_128", (double)n * 128, (double)n * 64, (double)n * 4},
            {"one_per_64", (double)n * 64, (double)n * 64, (double)n * 4},
            {"window", lines128 * 128, pieces64 * 64, (double)n * 84}};
        for (int v = 0; v < 5; ++v) {
            (void)hipEventRecord(e0, 0);
            switch (v) {
            case 0: hipLaunchKernelGGL(cal_stream16, dim3(grid), dim3(256), 0, 0, n, (const u32x4*)buf, sink); break;
            case 1: hipLaunchKernelGGL(cal_one_per<256>, dim3(grid), dim3(256), 0, 0, n, buf, sink); break;
            case 2: hipLaunchKernelGGL(cal_one_per<128>, dim3(grid), dim3(256), 0, 0, n, buf, sink); break;
            case 3: hipLaunchKernelGGL(cal_one_per<64>, dim3(grid), dim3(256), 0, 0, n, buf, sink); break;
            case 4: hipLaunchKernelGGL(cal_window, dim3(grid), dim3(256), 0, 0, n, buf, sink); break;
            }
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2)
                printf("{\"kernel\": \"%s\", \"lanes\": %lld, \"ms\": %.4f, \"bytes_if_128B_granules\": %.0f, \"bytes_if_64B_granules\": %.0f, "
                       "\"useful_bytes\": %.0f, \"GBs_at_128B\": %.0f}\n", rows[v].name, (long long)n, ms, rows[v].b128, rows[v].b64, rows[v].useful,
                       rows[v].b128 / ms / 1e6);
        }
    }
    return 0;
}
