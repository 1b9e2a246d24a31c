// ubench_render.hip -- launch-shape sweep for the pixel render (experiment tool, VERDICT r1 task 6).
//
// k_render is a pure store stream (9408 B/env out, 147 B/env in) that sat at 5.1-5.3 TB/s while a plain 9.9 GB fill on the
// same box reaches 6.9 TB/s.  This program renders 1 048 576 synthetic observations with several kernel SHAPES, checks every
// output byte of every shape against the first one, and prints ms / TB/s per shape:
//   block<G>   the shipped shape: 256-thread block, G envs per iteration behind two __syncthreads, grid-stride over groups
//              (grid = ngroups -> one group per short-lived block; grid = 1024..16384 -> looped)
//   wave<W>    wave-autonomous: the block only shares the LDS atlas; each of its W waves renders whole envs on its own
//              (49 lanes stage the env's tile ids in a wave-private LDS row, no block barrier in the loop), 9408 B = 588
//              x 16 B per env written as ten 1-KiB wave stores
//   tilein     either shape fed with 49 tile ids per env (what k_step could emit) instead of the 147-byte encoding + LUT
// and the same for non-temporal vs plain stores, plus a plain fill of the same buffer as the in-process ceiling.
//   queue / wqueue   persistent blocks that take their groups (envs) from an atomic ticket counter: `tools/ubench_render <n> queue`
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_render tools/ubench_render.hip && tools/ubench_render
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int VIEW = 7, OBS_BYTES = 147, PIX_BYTES = 9408, TILE_BYTES = 192, N_TILES = 58;
constexpr int CHUNKS_PER_ROW = 21, VEC_PER_ENV = PIX_BYTES / 16;

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_init(int64_t n, uint8_t* image, uint8_t* tiles, uint8_t* atlas, uint8_t* lut) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 512) lut[i] = mix((uint32_t)i + 99u) % N_TILES;
    if (i < N_TILES * TILE_BYTES) atlas[i] = (uint8_t)mix((uint32_t)i + 7u);
    if (i >= n * 49) return;
    const uint32_t r = mix((uint32_t)i);
    const int t = r & 7, c = (r >> 3) % 6, s = (r >> 8) % 3;
    image[3 * i] = t; image[3 * i + 1] = c; image[3 * i + 2] = s;
    const int cell = (int)(i % 49);
    tiles[i] = (uint8_t)(mix((uint32_t)((cell == 27 ? 256 : 0) + (t | (c << 3) | (s << 6))) + 99u) % N_TILES);
}

__device__ __forceinline__ uint64_t render_chunk(const uint8_t* s_atlas, const uint8_t* tiles49, int ch) {
    const int py = ch / CHUNKS_PER_ROW, cx = ch - py * CHUNKS_PER_ROW;
    const int ti = cx / 3, part = cx - ti * 3;
    const int tj = py >> 3, ty = py & 7;
    const int tile = tiles49[ti * VIEW + tj];
    return *(const uint64_t*)(s_atlas + tile * TILE_BYTES + ty * 24 + part * 8);
}
template <bool NT>
__device__ __forceinline__ void put(u32x4* p, uint64_t lo, uint64_t hi) {
    u32x4 v = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int GROUP, bool NT, bool TILEIN, int DRAIN = 0, int LDS_PAD = 0>
__global__ __launch_bounds__(256) void k_block(int64_t n, const uint8_t* __restrict__ image, const uint8_t* __restrict__ tiles,
                                               uint8_t* __restrict__ pixels, const uint8_t* __restrict__ atlas, const uint8_t* __restrict__ lut) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[N_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[512];
    __shared__ uint8_t s_tile[GROUP * 49 + 8];
    __shared__ uint8_t s_pad[LDS_PAD + 1];       // occupancy throttle (experiment)
    if (LDS_PAD && threadIdx.x == 0) s_pad[LDS_PAD] = 0;
    for (int k = threadIdx.x; k < N_TILES * TILE_BYTES / 8; k += 256) ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
    for (int k = threadIdx.x; k < 512; k += 256) s_lut[k] = lut[k];
    const int64_t ngroups = (n + GROUP - 1) / GROUP;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t env0 = grp * GROUP;
        const int ne = (int)(n - env0 < GROUP ? n - env0 : GROUP);
        __syncthreads();
        for (int c = threadIdx.x; c < ne * 49; c += 256) {
            if (TILEIN) s_tile[c] = tiles[env0 * 49 + c];
            else {
                const int e = c / 49, cell = c - e * 49;
                const uint8_t* o = image + (env0 + e) * OBS_BYTES + cell * 3;
                s_tile[c] = s_lut[(cell == 27 ? 256 : 0) + (o[0] | (o[1] << 3) | (o[2] << 6))];
            }
        }
        __syncthreads();
        u32x4* out = (u32x4*)(pixels + env0 * PIX_BYTES);
        for (int q = threadIdx.x; q < ne * VEC_PER_ENV; q += 256) {
            const int e = q / VEC_PER_ENV, k = q - e * VEC_PER_ENV;
            const uint8_t* t49 = s_tile + e * 49;
            put<NT>(out + q, render_chunk(s_atlas, t49, 2 * k), render_chunk(s_atlas, t49, 2 * k + 1));
            if (DRAIN == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // at most one store in flight per wave
            if (DRAIN == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (DRAIN == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
    }
}

template <int WAVES, bool NT, bool TILEIN>
__global__ __launch_bounds__(64 * WAVES) void k_wave(int64_t n, const uint8_t* __restrict__ image, const uint8_t* __restrict__ tiles,
                                                     uint8_t* __restrict__ pixels, const uint8_t* __restrict__ atlas, const uint8_t* __restrict__ lut) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[N_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[512];
    __shared__ uint8_t s_tile[WAVES][64];
    for (int k = threadIdx.x; k < N_TILES * TILE_BYTES / 8; k += 64 * WAVES) ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
    for (int k = threadIdx.x; k < 512; k += 64 * WAVES) s_lut[k] = lut[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint8_t* t49 = s_tile[wv];
    for (int64_t env = (int64_t)blockIdx.x * WAVES + wv; env < n; env += (int64_t)gridDim.x * WAVES) {
        if (lane < 49) {
            if (TILEIN) t49[lane] = tiles[env * 49 + lane];
            else {
                const uint8_t* o = image + env * OBS_BYTES + lane * 3;
                t49[lane] = s_lut[(lane == 27 ? 256 : 0) + (o[0] | (o[1] << 3) | (o[2] << 6))];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x4* out = (u32x4*)(pixels + env * PIX_BYTES);
#pragma unroll 2
        for (int k = lane; k < VEC_PER_ENV; k += 64) put<NT>(out + k, render_chunk(s_atlas, t49, 2 * k), render_chunk(s_atlas, t49, 2 * k + 1));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// one-shot<G, T, DIRECT>: ONE G-env group per block of T threads (the shipped k_render launch shape).  DIRECT = false: the
// 11 KB atlas is copied into the block's LDS first (shipped); DIRECT = true: only the G x 49 tile ids go through LDS and
// every lane reads its two 8-byte tile-row pieces straight from the global atlas (11 KB: resident in every CU's L1).
template <int GROUP, int T, bool DIRECT>
__global__ __launch_bounds__(T) void k_shot(int64_t n, const uint8_t* __restrict__ image, uint8_t* __restrict__ pixels,
                                            const uint8_t* __restrict__ atlas, const uint8_t* __restrict__ lut) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[DIRECT ? 16 : N_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[DIRECT ? 16 : 512];
    __shared__ uint8_t s_tile[GROUP * 49 + 8];
    if (!DIRECT) {
        for (int k = threadIdx.x; k < N_TILES * TILE_BYTES / 8; k += T) ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
        for (int k = threadIdx.x; k < 512; k += T) s_lut[k] = lut[k];
        __syncthreads();
    }
    const int64_t env0 = (int64_t)blockIdx.x * GROUP;
    const int ne = (int)(n - env0 < GROUP ? n - env0 : GROUP);
    for (int c = threadIdx.x; c < ne * 49; c += T) {
        const int e = c / 49, cell = c - e * 49;
        const uint8_t* o = image + (env0 + e) * OBS_BYTES + cell * 3;
        const int key = (cell == 27 ? 256 : 0) + (o[0] | (o[1] << 3) | (o[2] << 6));
        s_tile[c] = DIRECT ? lut[key] : s_lut[key];
    }
    __syncthreads();
    u32x4* out = (u32x4*)(pixels + env0 * PIX_BYTES);
    const uint8_t* src = DIRECT ? atlas : s_atlas;
    for (int q = threadIdx.x; q < ne * VEC_PER_ENV; q += T) {
        const int e = q / VEC_PER_ENV, k = q - e * VEC_PER_ENV;
        const uint8_t* t49 = s_tile + e * 49;
        put<true>(out + q, render_chunk(src, t49, 2 * k), render_chunk(src, t49, 2 * k + 1));
    }
}

// queue<G, T>: PERSISTENT blocks (the 11 KB atlas is loaded into LDS once per block, as in block<G>) that take their G-env
// groups from a global atomic counter instead of a fixed stride: blocks then advance through the output in (nearly) global
// order, like one-shot blocks being dispatched one after the other -- the order in which the pure store stream is fastest
// (profiles/r02/ubench_store.txt: 4-KiB one-shot blocks 6.97 TB/s, 75-KB blocks 5.8, grid-stride loops 5.5) -- without paying
// the atlas reload per group that keeps one-shot render blocks from being small.
// NC > 1: NC ticket counters, 256 bytes apart (one address serves ~88 M atomics/s -- 11.4 ns each, measured: queue<1,..> takes
// 11.9 ms for 1 048 576 tickets); counter c hands out the c-th NC-th of the groups to the blocks with blockIdx % NC == c.
template <int GROUP, int T, bool NT, int NC = 1>
__global__ __launch_bounds__(T) void k_queue(int64_t n, const uint8_t* __restrict__ image, uint8_t* __restrict__ pixels,
                                             const uint8_t* __restrict__ atlas, const uint8_t* __restrict__ lut, unsigned int* __restrict__ counter) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[N_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[512];
    __shared__ uint8_t s_tile[2][GROUP * 49 + 8];
    __shared__ unsigned int s_grp[2];
    for (int k = threadIdx.x; k < N_TILES * TILE_BYTES / 8; k += T) ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
    for (int k = threadIdx.x; k < 512; k += T) s_lut[k] = lut[k];
    const unsigned int all_groups = (unsigned int)((n + GROUP - 1) / GROUP);
    const unsigned int part = (all_groups + NC - 1) / NC, first = (blockIdx.x % NC) * part;
    const unsigned int ngroups = first + part < all_groups ? first + part : all_groups;      // this counter's groups: [first, ngroups)
    counter += 64 * (blockIdx.x % NC);
    int buf = 0;
    if (threadIdx.x == 0) s_grp[0] = first + atomicAdd(counter, 1u);
    __syncthreads();
    for (;;) {
        const unsigned int grp = s_grp[buf];
        if (grp >= ngroups) break;
        const int64_t env0 = (int64_t)grp * GROUP;
        const int ne = (int)(n - env0 < GROUP ? n - env0 : GROUP);
        for (int c = threadIdx.x; c < ne * 49; c += T) {
            const int e = c / 49, cell = c - e * 49;
            const uint8_t* o = image + (env0 + e) * OBS_BYTES + cell * 3;
            s_tile[buf][c] = s_lut[(cell == 27 ? 256 : 0) + (o[0] | (o[1] << 3) | (o[2] << 6))];
        }
        if (threadIdx.x == 0) s_grp[buf ^ 1] = first + atomicAdd(counter, 1u);      // the next group's ticket rides under this group's stores
        __syncthreads();                                                    // ONE barrier per group (tile rows and tickets are double-buffered)
        u32x4* out = (u32x4*)(pixels + env0 * PIX_BYTES);
        for (int q = threadIdx.x; q < ne * VEC_PER_ENV; q += T) {
            const int e = q / VEC_PER_ENV, k = q - e * VEC_PER_ENV;
            const uint8_t* t49 = s_tile[buf] + e * 49;
            put<NT>(out + q, render_chunk(s_atlas, t49, 2 * k), render_chunk(s_atlas, t49, 2 * k + 1));
        }
        buf ^= 1;
    }
}

// wqueue<W>: the same ticket queue per WAVE (the block only shares the atlas): a wave takes one env, 49 lanes stage its tile ids
// in a wave-private LDS row, and the wave writes the env's 9408 bytes as ten 1-KiB stores -- no block barrier after the atlas load.
template <int WAVES, bool NT>
__global__ __launch_bounds__(64 * WAVES) void k_wqueue(int64_t n, const uint8_t* __restrict__ image, uint8_t* __restrict__ pixels,
                                                       const uint8_t* __restrict__ atlas, const uint8_t* __restrict__ lut, unsigned int* __restrict__ counter) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[N_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[512];
    __shared__ uint8_t s_tile[WAVES][56];
    for (int k = threadIdx.x; k < N_TILES * TILE_BYTES / 8; k += 64 * WAVES) ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
    for (int k = threadIdx.x; k < 512; k += 64 * WAVES) s_lut[k] = lut[k];
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (;;) {
        unsigned int env = 0;
        if (lane == 0) env = atomicAdd(counter, 1u);
        env = __shfl(env, 0);
        if ((int64_t)env >= n) break;
        if (lane < 49) {
            const uint8_t* o = image + (int64_t)env * OBS_BYTES + lane * 3;
            s_tile[w][lane] = s_lut[(lane == 27 ? 256 : 0) + (o[0] | (o[1] << 3) | (o[2] << 6))];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        u32x4* out = (u32x4*)(pixels + (int64_t)env * PIX_BYTES);
        for (int k = lane; k < VEC_PER_ENV; k += 64)
            put<NT>(out + k, render_chunk(s_atlas, s_tile[w], 2 * k), render_chunk(s_atlas, s_tile[w], 2 * k + 1));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// gather<T>: the plain-fill shape with the render's data: one-shot blocks of T threads, one 16-byte store per thread, NO LDS and
// no barrier -- every lane looks its two tile ids up in a precomputed 49-byte tile plane (what k_step could emit) and reads
// its two 8-byte tile-row pieces from the global atlas (L1).  Blocks are not aligned to envs.
template <int T, bool NT>
__global__ __launch_bounds__(T) void k_gather(int64_t n, const uint8_t* __restrict__ tiles, uint8_t* __restrict__ pixels,
                                              const uint8_t* __restrict__ atlas) {
    const int64_t q = (int64_t)blockIdx.x * T + threadIdx.x;
    if (q >= n * VEC_PER_ENV) return;
    const int64_t e = q / VEC_PER_ENV;
    const int k = (int)(q - e * VEC_PER_ENV);
    const uint8_t* t49 = tiles + e * 49;
    put<NT>((u32x4*)pixels + q, render_chunk(atlas, t49, 2 * k), render_chunk(atlas, t49, 2 * k + 1));
}

__global__ void k_fill(u32x4* out, int64_t nvec) {       // one 4-KiB span per 256-thread block, no loop: the plain-fill shape
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q < nvec) { u32x4 v = {1u, 2u, 3u, (uint32_t)q}; out[q] = v; }
}
__global__ void k_diff(const u32x4* a, const u32x4* b, int64_t nvec, unsigned long long* bad) {
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += (int64_t)gridDim.x * blockDim.x) {
        u32x4 x = a[q], y = b[q];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) atomicAdd(bad, 1ull);
    }
}

static int64_t n;
static uint8_t *image, *tiles, *pix, *ref, *atlas, *lut;
static unsigned long long* bad;
static hipEvent_t e0, e1;
static bool have_ref = false;

template <typename F>
static void run(const char* name, int grid, F launch) {
    (void)hipMemsetAsync(pix, 0, (size_t)n * PIX_BYTES, 0);
    float sum = 0, best = 1e9f; const int reps = 7, skip = 2;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (r >= skip) { sum += ms; if (ms < best) best = ms; }
    }
    unsigned long long nbad = 0;
    if (!have_ref) { (void)hipMemcpy(ref, pix, (size_t)n * PIX_BYTES, hipMemcpyDeviceToDevice); have_ref = true; }
    else {
        (void)hipMemset(bad, 0, 8);
        hipLaunchKernelGGL(k_diff, dim3(4096), dim3(256), 0, 0, (const u32x4*)pix, (const u32x4*)ref, n * (int64_t)VEC_PER_ENV, bad);
        (void)hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost);
    }
    const float avg = sum / (reps - skip);
    printf("{\"shape\": \"%s\", \"grid\": %d, \"avg_ms\": %.4f, \"min_ms\": %.4f, \"TBs_at_avg\": %.3f, \"TBs_at_min\": %.3f, \"mismatching_vec\": %llu}\n",
           name, grid, avg, best, n * 9555.0 / avg / 1e9, n * 9555.0 / best / 1e9, nbad);
    fflush(stdout);
}

#define BLOCK_CASE(G, NT, TI, GRID) run("block<" #G "> nt=" #NT " tilein=" #TI, GRID, [&] { \
    hipLaunchKernelGGL((k_block<G, NT, TI>), dim3(GRID), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); })
#define WAVE_CASE(W, NT, TI, GRID) run("wave<" #W "> nt=" #NT " tilein=" #TI, GRID, [&] { \
    hipLaunchKernelGGL((k_wave<W, NT, TI>), dim3(GRID), dim3(64 * W), 0, 0, n, image, tiles, pix, atlas, lut); })

int main(int argc, char** argv) {
    n = argc > 1 ? atoll(argv[1]) : 1048576;
    (void)hipMalloc(&image, n * OBS_BYTES); (void)hipMalloc(&tiles, n * 49); (void)hipMalloc(&pix, (size_t)n * PIX_BYTES);
    (void)hipMalloc(&ref, (size_t)n * PIX_BYTES); (void)hipMalloc(&atlas, N_TILES * TILE_BYTES); (void)hipMalloc(&lut, 512); (void)hipMalloc(&bad, 8);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_init, dim3((unsigned)((n * 49 + 255) / 256)), dim3(256), 0, 0, n, image, tiles, atlas, lut);
    (void)hipDeviceSynchronize();
    const int g8 = (int)((n + 7) / 8);
    BLOCK_CASE(8, true, false, g8 / 8);           // the shipped launch (reference output)
    {   // plain fill ceiling on the same buffer, same process
        const int64_t nvec = n * (int64_t)VEC_PER_ENV;
        float sum = 0, best = 1e9f;
        for (int r = 0; r < 7; ++r) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, (u32x4*)pix, nvec);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        printf("{\"shape\": \"plain fill, one 4 KiB span per block\", \"avg_ms\": %.4f, \"min_ms\": %.4f, \"TBs_at_avg\": %.3f}\n", sum / 5, best,
               n * 9408.0 / (sum / 5) / 1e9);
    }
#define SHOT_CASE(G, T, D) run("one-shot<" #G "," #T "> " #D, (int)((n + G - 1) / G), [&] { \
    hipLaunchKernelGGL((k_shot<G, T, D>), dim3((unsigned)((n + G - 1) / G)), dim3(T), 0, 0, n, image, pix, atlas, lut); })
    if (argc > 2 && argv[2][0] == 'q') {   // tools/ubench_render <n> queue : persistent blocks fed by an atomic ticket counter
        unsigned int* counter = nullptr;
        (void)hipMalloc(&counter, 64 * 256);
        const int cus = 256;
#define QUEUEN_CASE(G, T, NC, BPC) run("queue<" #G "," #T "> nt counters=" #NC " blocks/CU=" #BPC, cus * BPC, [&] { \
    (void)hipMemsetAsync(counter, 0, 64 * 256, 0); \
    hipLaunchKernelGGL((k_queue<G, T, true, NC>), dim3(cus * BPC), dim3(T), 0, 0, n, image, pix, atlas, lut, counter); })
        if (argv[2][1] == '2') {          // tools/ubench_render <n> q2 : several ticket counters
            for (int rep = 0; rep < 2; ++rep) {
                SHOT_CASE(8, 1024, false); SHOT_CASE(2, 512, false);
                QUEUEN_CASE(8, 1024, 1, 2); QUEUEN_CASE(8, 1024, 8, 2); QUEUEN_CASE(4, 512, 8, 4); QUEUEN_CASE(2, 512, 8, 4); QUEUEN_CASE(2, 512, 32, 4);
                QUEUEN_CASE(1, 256, 32, 8); QUEUEN_CASE(2, 256, 32, 8); QUEUEN_CASE(4, 1024, 16, 2); QUEUEN_CASE(2, 512, 64, 4); QUEUEN_CASE(4, 512, 32, 4);
                QUEUEN_CASE(16, 1024, 1, 2); QUEUEN_CASE(16, 1024, 8, 2);
            }
            return 0;
        }
#define QUEUE_CASE(G, T, NT, BPC) run("queue<" #G "," #T "> nt=" #NT " blocks/CU=" #BPC, cus * BPC, [&] { \
    (void)hipMemsetAsync(counter, 0, 64 * 256, 0); \
    hipLaunchKernelGGL((k_queue<G, T, NT>), dim3(cus * BPC), dim3(T), 0, 0, n, image, pix, atlas, lut, counter); })
#define WQUEUE_CASE(W, NT, BPC) run("wqueue<" #W "> nt=" #NT " blocks/CU=" #BPC, cus * BPC, [&] { \
    (void)hipMemsetAsync(counter, 0, 4, 0); \
    hipLaunchKernelGGL((k_wqueue<W, NT>), dim3(cus * BPC), dim3(64 * W), 0, 0, n, image, pix, atlas, lut, counter); })
        for (int rep = 0; rep < 2; ++rep) {
            SHOT_CASE(8, 1024, false); SHOT_CASE(2, 512, false);
            QUEUE_CASE(1, 256, true, 8); QUEUE_CASE(2, 512, true, 4); QUEUE_CASE(1, 512, true, 4); QUEUE_CASE(2, 256, true, 8);
            QUEUE_CASE(4, 512, true, 4); QUEUE_CASE(2, 512, true, 2); QUEUE_CASE(2, 512, false, 4); QUEUE_CASE(8, 1024, true, 2);
            WQUEUE_CASE(4, true, 8); WQUEUE_CASE(8, true, 4); WQUEUE_CASE(4, false, 8); WQUEUE_CASE(16, true, 2);
        }
        return 0;
    }
    if (argc > 2) {          // tools/ubench_render <n> direct : the LDS-atlas one-shot shapes against atlas reads from L1
        for (int rep = 0; rep < 2; ++rep) {
            SHOT_CASE(8, 1024, false); SHOT_CASE(8, 1024, true);
            SHOT_CASE(2, 512, false);  SHOT_CASE(2, 512, true);
            SHOT_CASE(4, 512, true);   SHOT_CASE(4, 1024, true);
            SHOT_CASE(1, 256, true);   SHOT_CASE(2, 256, true);  SHOT_CASE(1, 512, true); SHOT_CASE(4, 256, true);
            SHOT_CASE(1, 256, false);  SHOT_CASE(16, 1024, true);
#define GATHER_CASE(T, NT) run("gather<" #T "> nt=" #NT " (tile plane in, no LDS)", (int)((n * VEC_PER_ENV + T - 1) / T), [&] { \
    hipLaunchKernelGGL((k_gather<T, NT>), dim3((unsigned)((n * VEC_PER_ENV + T - 1) / T)), dim3(T), 0, 0, n, tiles, pix, atlas); })
            GATHER_CASE(256, true); GATHER_CASE(512, true); GATHER_CASE(1024, true); GATHER_CASE(64, true); GATHER_CASE(256, false);
        }
        return 0;
    }
    BLOCK_CASE(8, false, false, g8 / 8);
    BLOCK_CASE(8, true, false, g8);               // one 8-env group per short-lived block
    BLOCK_CASE(8, true, false, 2048);
    BLOCK_CASE(8, true, false, 1024);
    BLOCK_CASE(4, true, false, (int)((n + 3) / 4));
    BLOCK_CASE(4, true, false, 2048);
    BLOCK_CASE(2, true, false, (int)((n + 1) / 2));
    BLOCK_CASE(2, true, false, 2048);
    BLOCK_CASE(16, true, false, 2048);
    BLOCK_CASE(8, true, true, g8 / 8);
    BLOCK_CASE(8, true, true, 2048);
    run("block<8> nt drain=1 (one store in flight per wave)", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 1>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> nt drain=2", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 2>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> nt drain=4", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 4>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> plain drain=1", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, false, false, 1>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> plain drain=2", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, false, false, 2>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> nt drain=1 one group per block", g8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 1>), dim3(g8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<2> nt drain=1 one group per block", (int)(n / 2), [&] { hipLaunchKernelGGL((k_block<2, true, false, 1>), dim3((unsigned)(n / 2)), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> nt 4 blocks/CU (LDS pad)", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 0, 20000>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> nt 2 blocks/CU (LDS pad)", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 0, 45000>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    run("block<8> nt 2 blocks/CU drain=1", g8 / 8, [&] { hipLaunchKernelGGL((k_block<8, true, false, 1, 45000>), dim3(g8 / 8), dim3(256), 0, 0, n, image, tiles, pix, atlas, lut); });
    WAVE_CASE(4, true, false, 2048);
    WAVE_CASE(4, false, false, 2048);
    WAVE_CASE(4, true, false, 1024);
    WAVE_CASE(4, true, false, 4096);
    WAVE_CASE(4, true, false, 16384);
    WAVE_CASE(4, true, false, 65536);
    WAVE_CASE(8, true, false, 1024);
    WAVE_CASE(8, true, false, 2048);
    WAVE_CASE(8, true, false, 512);
    WAVE_CASE(2, true, false, 4096);
    WAVE_CASE(4, true, true, 2048);
    WAVE_CASE(8, true, true, 1024);
    WAVE_CASE(4, false, true, 2048);
    return 0;
}
