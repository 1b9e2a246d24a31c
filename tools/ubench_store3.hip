// Store-pattern microbenchmark #3 (experiment tool): persistent blocks that grab 4-16 KB chunks in global order
// through an atomic ticket, vs the static grid-stride assignment.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <bool DYNAMIC>
__global__ __launch_bounds__(256) void k_persist(u32x4* out, long vec_per_chunk, long nchunks, long total_vec, unsigned long long* ticket) {
    __shared__ long s_c;
    long c = blockIdx.x;
    for (;;) {
        if (DYNAMIC) {
            if (threadIdx.x == 0) s_c = (long)atomicAdd(ticket, 1ull);
            __syncthreads();
            c = s_c;
            __syncthreads();
        }
        if (c >= nchunks) break;
        long base = c * vec_per_chunk;
        long end = base + vec_per_chunk < total_vec ? base + vec_per_chunk : total_vec;
        for (long q = base + threadIdx.x; q < end; q += 256) {
            u32x4 v = {(uint32_t)q, 1u, 2u, 3u};
            out[q] = v;
        }
        if (!DYNAMIC) c += gridDim.x;
    }
}
int main() {
    const long bytes = 1048576L * 9408L, total_vec = bytes / 16;
    u32x4* buf; (void)hipMalloc(&buf, bytes);
    unsigned long long* ticket; (void)hipMalloc(&ticket, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (long chunk : {4096L, 8192L, 16384L, 65536L})
        for (int grid : {1024, 2048, 4096})
            for (int dyn = 0; dyn < 2; ++dyn) {
                long vpc = chunk / 16, nchunks = (total_vec + vpc - 1) / vpc;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipMemset(ticket, 0, 8);
                    hipEventRecord(a);
                    if (dyn) hipLaunchKernelGGL(k_persist<true>, dim3(grid), dim3(256), 0, 0, buf, vpc, nchunks, total_vec, ticket);
                    else hipLaunchKernelGGL(k_persist<false>, dim3(grid), dim3(256), 0, 0, buf, vpc, nchunks, total_vec, ticket);
                    hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (rep && ms < best) best = ms;
                }
                printf("chunk=%6ld grid=%5d %s : %.3f ms (%4.0f GB/s)\n", chunk, grid, dyn ? "ticket " : "static ", best, bytes / best / 1e6);
            }
    return 0;
}
