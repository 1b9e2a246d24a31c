// Store-pattern microbenchmark #2 (experiment tool): looped store streams with different thread->address maps.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// MODE 0: block-interleaved (q += blockDim)   MODE 1: wave-contiguous (each wave owns a contiguous span)
template <int MODE, bool NT>
__global__ void k_loop(u32x4* out, long vec_per_block, long total_vec) {
    long base = (long)blockIdx.x * vec_per_block;
    long end = base + vec_per_block < total_vec ? base + vec_per_block : total_vec;
    if (MODE == 0) {
        for (long q = base + threadIdx.x; q < end; q += blockDim.x) {
            u32x4 v = {(uint32_t)q, 1u, 2u, 3u};
            if (NT) __builtin_nontemporal_store(v, out + q); else out[q] = v;
        }
    } else {
        int nw = blockDim.x / 64, w = threadIdx.x / 64, lane = threadIdx.x & 63;
        long per_wave = (vec_per_block + nw - 1) / nw;
        long wb = base + w * per_wave, we = wb + per_wave < end ? wb + per_wave : end;
        for (long q = wb + lane; q < we; q += 64) {
            u32x4 v = {(uint32_t)q, 1u, 2u, 3u};
            if (NT) __builtin_nontemporal_store(v, out + q); else out[q] = v;
        }
    }
}
template <int MODE, bool NT>
float run(u32x4* buf, long vpb, long total_vec, int block, hipEvent_t a, hipEvent_t b) {
    long grid = (total_vec + vpb - 1) / vpb;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        for (int it = 0; it < 4; ++it) hipLaunchKernelGGL((k_loop<MODE, NT>), dim3(grid), dim3(block), 0, 0, buf, vpb, total_vec);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms / 4 < best) best = ms / 4;
    }
    return best;
}
int main() {
    const long bytes = 1048576L * 9408L, total_vec = bytes / 16;
    u32x4* buf; (void)hipMalloc(&buf, bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int block : {64, 256, 1024})
        for (long per_thread : {1L, 2L, 4L, 18L, 147L}) {
            long vpb = per_thread * block;
            float t0 = run<0, false>(buf, vpb, total_vec, block, a, b);
            float t1 = run<1, false>(buf, vpb, total_vec, block, a, b);
            float t2 = run<0, true>(buf, vpb, total_vec, block, a, b);
            printf("block=%4d stores/thread=%3ld bytes/block=%8ld : interleaved %.3f ms (%4.0f GB/s) | wave-contig %.3f ms (%4.0f GB/s) | interleaved-NT %.3f ms (%4.0f GB/s)\n",
                   block, per_thread, vpb * 16, t0, bytes / t0 / 1e6, t1, bytes / t1 / 1e6, t2, bytes / t2 / 1e6);
        }
    return 0;
}
