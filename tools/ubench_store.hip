// Store-pattern microbenchmark (experiment tool, not part of the product): how does HBM write
// throughput on MI355X depend on the contiguous bytes each 256-thread block streams and on NT stores?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_store(u32x4* out, long vec_per_block, long total_vec) {
    long base = (long)blockIdx.x * vec_per_block;
    long end = base + vec_per_block < total_vec ? base + vec_per_block : total_vec;
    for (long q = base + threadIdx.x; q < end; q += 256) {
        u32x4 v = {(uint32_t)q, 1u, 2u, 3u};
        if (NT) __builtin_nontemporal_store(v, out + q); else out[q] = v;
    }
}
// grid-stride: chunk c of size vec_per_chunk handled by block (c % gridDim)
template <bool NT>
__global__ __launch_bounds__(256) void k_store_gs(u32x4* out, long vec_per_chunk, long total_vec) {
    long nchunks = (total_vec + vec_per_chunk - 1) / vec_per_chunk;
    for (long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        long base = c * vec_per_chunk;
        long end = base + vec_per_chunk < total_vec ? base + vec_per_chunk : total_vec;
        for (long q = base + threadIdx.x; q < end; q += 256) {
            u32x4 v = {(uint32_t)q, 1u, 2u, 3u};
            if (NT) __builtin_nontemporal_store(v, out + q); else out[q] = v;
        }
    }
}
int main() {
    const long bytes = 1048576L * 9408L, total_vec = bytes / 16;
    u32x4* buf; hipMalloc(&buf, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    long sizes[] = {4096, 16384, 75264, 150528, 602112, 2408448};
    for (int nt = 0; nt < 2; ++nt)
        for (long sz : sizes) {
            long vpb = sz / 16; long grid = (total_vec + vpb - 1) / vpb;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                for (int it = 0; it < 5; ++it) {
                    if (nt) hipLaunchKernelGGL(k_store<true>, dim3(grid), dim3(256), 0, 0, buf, vpb, total_vec);
                    else hipLaunchKernelGGL(k_store<false>, dim3(grid), dim3(256), 0, 0, buf, vpb, total_vec);
                }
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep) printf("direct   nt=%d bytes/block=%8ld grid=%8ld : %.3f ms  %.0f GB/s\n", nt, sz, grid, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
            }
        }
    for (int nt = 0; nt < 2; ++nt)
        for (long sz : {4096L, 16384L, 75264L})
            for (int g : {1024, 2048, 4096, 8192}) {
                long vpb = sz / 16;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(a);
                    for (int it = 0; it < 5; ++it) {
                        if (nt) hipLaunchKernelGGL(k_store_gs<true>, dim3(g), dim3(256), 0, 0, buf, vpb, total_vec);
                        else hipLaunchKernelGGL(k_store_gs<false>, dim3(g), dim3(256), 0, 0, buf, vpb, total_vec);
                    }
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (rep) printf("gridstr  nt=%d bytes/chunk=%8ld grid=%8d : %.3f ms  %.0f GB/s\n", nt, sz, g, ms / 5, bytes / (ms / 5 * 1e-3) / 1e9);
                }
            }
    return 0;
}
