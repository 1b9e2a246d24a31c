// ubench_gather.hip -- what CAN the k_step access pattern reach on this part?  (experiment tool, VERDICT r1 task 4a)
//
// k_step = lane-per-env: coalesced SoA traffic (hot 16 B R+W, stale 8 B R+W, vhead 4 B, vset 8 B, action 1 B, reward 4 B,
// done 1 B, dir 1 B, obs 147 B W) plus a PER-ENV GATHER out of a 1728-byte env record (BossLevel): the 7x7 window = 7 rows
// x 12 bytes at a 32-byte pitch (2-3 128-B lines), the front cell's id byte and the carried object's appearance byte.
// The variants below keep that traffic and strip the game logic, so each one is the ceiling of a kernel SHAPE:
//   soa        SoA loads/stores + LDS-staged 147-B obs copy-out only (no record access)
//   gather     per-lane window gather only (addresses from the env index), one dword out per lane
//   chain      round 1's k_step: hot -> front cell byte -> window rows per lane (position depends on it) -> appearance
//              byte (depends on the window), then the SoA stores + obs
//   flat       every record load issued as soon as the hot word is there (8-row speculative union window + id byte +
//              appearance byte), one wait, then the SoA stores + obs
//   flat2      flat with two envs per lane (memory-level parallelism x2, same traffic)
//   coopR      chain, but the wave fetches its 64 windows TOGETHER: lane g of pass i loads row (64 i + g) % 7 of env
//              (64 i + g) / 7 (7 passes of 64 x 12 B), parks it in that env's LDS row; each lane then reads its own window
//              from LDS.  Lanes that hit the same 64/128-B piece of a record are one request instead of 2-4.
//   coopU      coopR on the PRE-action window widened by one row / column in the facing direction (8 passes): it holds the
//              front cell AND the window after a forward move, so the per-lane front-cell load disappears as well
//   envtile    (VERDICT r2 task 2 i) appearance plane ENV-TILED: E[tile of 64 envs][cell][64] -- a wave's 64 lanes read byte
//              `lane` of 49 (+1 front) cell rows of 64 B; coalesced only when the lanes want the SAME cells, and the agents of
//              64 random envs never stand on the same cell: 50 half-lines per env
//   ztile      (task 2 ii) appearance plane in 16x8-cell tiles of 128 B (one line each): the window overlaps 1-4 tiles
//   vline      a REDUNDANT window plane: one 128-byte line (8 rows x 16 cells) for every window origin class
//              (x origin / 8, y origin / 2): 52 lines = 6.5 KB per env instead of 1 KB, the window of any pose lies in ONE
//              line; the front cell of the transition comes from a 2-byte SoA cache written by the previous step (the
//              window it was taken from is the one fetched then), so no other record line is touched on a plain step
// chain / flat / coop* run on two record layouts: pitch 32 (today: 22 cells + 2 x 5 margin) and pitch 24 (no side margin).
// Output: one JSON line per (variant, layout): ms per launch over 1 048 576 envs, GB/s against k_step's 235 algorithmic
// bytes per env-step.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip && tools/ubench_gather
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int OBS = 147, PAD = 148;
struct Layout { int ES, OFF_E, OFF_I, OFF_APP, stride; };       // OFF_E = offset of cell (0, 0)
static const Layout LAY32 = {32, 5 * 32 + 5, 1024, 1508, 1728};
static const Layout LAY24 = {24, 7 * 24, 864, 1348, 1536};      // 7 spare rows above/below the 22 grid rows

struct Hot { uint8_t ax, ay, dir, carry; uint16_t step, max_steps; uint32_t pre4; uint8_t vstate, frozen, last_locked, slot; };

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void k_init(int64_t n, Hot* hot, uint64_t* stale, uint32_t* vhead, uint64_t* vset, uint8_t* act, uint8_t* rec, int stride) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = mix((uint32_t)i * 2654435761u + 17u);
    Hot h; h.ax = 1 + r % 20; h.ay = 1 + (r >> 8) % 20; h.dir = (r >> 16) & 3; h.carry = (r >> 18) & 15; h.step = 0; h.max_steps = 576;
    h.pre4 = ~0u; h.vstate = 0; h.frozen = 0; h.last_locked = 255; h.slot = 0;
    hot[i] = h; stale[i] = 0; vhead[i] = 1u << 8; vset[i] = r; act[i] = r % 7;
    uint32_t* p = (uint32_t*)(rec + i * (int64_t)stride);
    for (int k = 0; k < stride / 4; ++k) p[k] = mix(r + k) & 0x01010101u;      // cells 0/1
}

// 147-byte obs of the block's envs: LDS rows of 148 B -> one contiguous dword-coalesced span (as k_step does it)
template <int BLOCK>
__device__ __forceinline__ void copy_out(const uint8_t* s_obs, uint8_t* image, int64_t env0, int64_t n) {
    const int64_t nb = n - env0 < BLOCK ? n - env0 : BLOCK;
    const int ndw = (int)nb * OBS >> 2;
    const uint32_t* s32 = (const uint32_t*)s_obs;
    uint32_t* out = (uint32_t*)(image + env0 * OBS);
    for (int d = threadIdx.x; d < ndw; d += BLOCK) {
        const int b = 4 * d, e = b / OBS, off = b - e * OBS;
        uint32_t v;
        if (off <= OBS - 4) { const int q = e * (PAD / 4) + (off >> 2); v = __builtin_amdgcn_alignbyte(s32[q + 1], s32[q], off & 3); }
        else { v = 0; for (int k = 0; k < 4; ++k) { const int bb = b + k, ee = bb / OBS; v |= (uint32_t)s_obs[ee * PAD + bb - ee * OBS] << (8 * k); } }
        out[d] = v;
    }
}

__device__ __forceinline__ void fake_obs(uint32_t* row, const uint32_t* w, int nw, uint32_t extra) {
#pragma unroll
    for (int k = 0; k < 37; ++k) row[k] = (w[k % nw] >> (k & 7)) ^ extra;
}

struct Args {
    int64_t n; Layout L; uint8_t* rec; Hot* hot; uint64_t* stale; const uint32_t* vhead; const uint64_t* vset;
    const uint8_t* act; uint8_t* image; uint8_t* dirs; float* rew; uint8_t* done; uint32_t* sink;
};

__device__ __forceinline__ int fdx(int d) { return (d == 0) - (d == 2); }
__device__ __forceinline__ int fdy(int d) { return (d == 1) - (d == 3); }

template <int VARIANT, int BLOCK>   // 0 soa, 2 chain, 3 flat, 4 coopR, 5 coopU
__global__ __launch_bounds__(BLOCK) void k_shape(Args a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_obs[BLOCK * PAD];
    const int64_t env0 = (int64_t)blockIdx.x * BLOCK, env = env0 + threadIdx.x;
    const bool active = env < a.n;
    const int ES = a.L.ES, es4 = ES >> 2, OFF_I = a.L.OFF_I, OFF_APP = a.L.OFF_APP;
    Hot h; uint64_t st = 0; uint32_t vh = 0; uint64_t vs = 0; int action = 0;
    const uint8_t* rec = a.rec + (active ? env : a.n - 1) * (int64_t)a.L.stride;
    if (active) { h = a.hot[env]; st = a.stale[env]; vh = a.vhead[env]; vs = a.vset[env]; action = a.act[env]; }
    else { h = a.hot[a.n - 1]; }
    uint32_t w[24];
    int nw = 4;
    w[0] = h.pre4; w[1] = vh; w[2] = (uint32_t)vs; w[3] = (uint32_t)st;
    uint32_t extra = 0;
    const int fx = h.ax + fdx(h.dir), fy = h.ay + fdy(h.dir);
    if (VARIANT == 2 && active) {
        const int fe = rec[a.L.OFF_E + fy * ES + fx];                           // round trip 1
        if (action == 2 && (fe & 1)) { h.ax = fx; h.ay = fy; }
        const int tx = h.ax + (h.dir == 0 ? 0 : h.dir == 2 ? -6 : -3), ty = h.ay + (h.dir == 1 ? 0 : h.dir == 3 ? -6 : -3);
        const int a0 = a.L.OFF_E + ty * ES + tx;
        const uint32_t* q = (const uint32_t*)(rec + (a0 & ~3));
#pragma unroll
        for (int r = 0; r < 7; ++r) { w[3 * r] = q[r * es4]; w[3 * r + 1] = q[r * es4 + 1]; w[3 * r + 2] = q[r * es4 + 2]; }   // round trip 2
        nw = 21;
        extra = rec[OFF_APP + ((w[10] >> 8) & 31)];                            // round trip 3
        if (fe & 2) extra ^= rec[OFF_I + fy * 22 + fx];
    }
    if (VARIANT == 3 && active) {
        // union of the windows of "stayed" and "moved forward": 8 rows (dir 1/3) or 8 columns (still 3 dwords)
        const int tx = h.ax + (h.dir == 0 ? 0 : h.dir == 2 ? -7 : -3), ty = h.ay + (h.dir == 1 ? 0 : h.dir == 3 ? -7 : -3);
        const int a0 = a.L.OFF_E + ty * ES + tx;
        const uint32_t* q = (const uint32_t*)(rec + (a0 & ~3));
#pragma unroll
        for (int r = 0; r < 8; ++r) { w[3 * r] = q[r * es4]; w[3 * r + 1] = q[r * es4 + 1]; w[3 * r + 2] = q[r * es4 + 2]; }
        nw = 24;
        const uint32_t idb = rec[OFF_I + fy * 22 + fx];
        extra = rec[OFF_APP + (h.carry & 31)] ^ idb;                            // all in flight together
        if (action == 2 && (w[12] & 1)) { h.ax = fx; h.ay = fy; }
    }
    if (VARIANT == 4 || VARIANT == 5) {
        constexpr int ROWS = VARIANT == 4 ? 7 : 8;
        const int lane = threadIdx.x & 63;
        uint8_t* wrows = s_obs + (threadIdx.x & ~63) * PAD;                     // this wave's 64 LDS rows
        const uint8_t* wrec = a.rec + (env0 + (threadIdx.x & ~63)) * (int64_t)a.L.stride;
        const int64_t last = a.n - 1 - (env0 + (threadIdx.x & ~63));           // clamp for a ragged last wave
        int fe = 0;
        if (VARIANT == 4) {
            fe = rec[a.L.OFF_E + fy * ES + fx];                                 // per-lane front cell (round trip 1)
            if (action == 2 && (fe & 1)) { h.ax = fx; h.ay = fy; }
        }
        int tx, ty;
        if (VARIANT == 4) { tx = h.ax + (h.dir == 0 ? 0 : h.dir == 2 ? -6 : -3); ty = h.ay + (h.dir == 1 ? 0 : h.dir == 3 ? -6 : -3); }
        else { tx = h.ax + (h.dir == 0 ? 0 : h.dir == 2 ? -7 : -3); ty = h.ay + (h.dir == 1 ? 0 : h.dir == 3 ? -7 : -3); }
        const int a0 = a.L.OFF_E + ty * ES + tx;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int g = i * 64 + lane, e = g / ROWS, r = g - e * ROWS;
            const int a0e = __shfl(a0, e);
            const int64_t ee = e < last ? e : last;
            const uint32_t* q = (const uint32_t*)(wrec + ee * (int64_t)a.L.stride + (a0e & ~3)) + r * es4;
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
            uint32_t* dst = (uint32_t*)(wrows + e * PAD + 80 + 8 * r);          // 8 rows x 8 B at the tail of the env's row
            dst[0] = __builtin_amdgcn_alignbyte(d1, d0, a0e & 3);
            dst[1] = __builtin_amdgcn_alignbyte(d2, d1, a0e & 3);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t* mine = (const uint32_t*)(s_obs + threadIdx.x * PAD + 80);
#pragma unroll
        for (int k = 0; k < 2 * ROWS; ++k) w[k] = mine[k];
        nw = 2 * ROWS;
        if (VARIANT == 5) {
            fe = w[6] & 0xFF;                                                   // the front cell is a byte of the staged window
            if (action == 2 && (fe & 1)) { h.ax = fx; h.ay = fy; }
        }
        if (active) {
            extra = rec[OFF_APP + ((w[5] >> 8) & 31)];
            if (fe & 2) extra ^= rec[OFF_I + fy * 22 + fx];
        }
        __builtin_amdgcn_wave_barrier();                                        // every lane has read its window: rows may be overwritten
    }
    if (active) {
        h.step++;
        a.hot[env] = h;
        a.stale[env] = st + 1;
        a.rew[env] = (float)extra;
        a.done[env] = (uint8_t)(extra & 1);
        a.dirs[env] = h.dir;
        fake_obs((uint32_t*)(s_obs + threadIdx.x * PAD), w, nw, extra);
    }
    __syncthreads();
    copy_out<BLOCK>(s_obs, a.image, env0, a.n);
}


// ---- round 3 layouts ------------------------------------------------------------------------------------------------
// LAYOUT 0 envtile, 1 ztile, 2 vline.  `plane` is the layout's own buffer; everything else as in k_shape.
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_layout(Args a, const uint8_t* __restrict__ plane, uint16_t* __restrict__ fcache) {
    constexpr int BLOCK = 256;
    __shared__ __attribute__((aligned(16))) uint8_t s_obs[BLOCK * PAD];
    const int64_t env0 = (int64_t)blockIdx.x * BLOCK, env = env0 + threadIdx.x;
    const bool active = env < a.n;
    Hot h; uint64_t st = 0; uint32_t vh = 0; uint64_t vs = 0; int action = 0;
    const int64_t ee = active ? env : a.n - 1;
    h = a.hot[ee];
    const uint8_t ax0 = h.ax, ay0 = h.ay;          // (the pose is NOT advanced across launches: these planes have no slack rows)
    if (active) { st = a.stale[env]; vh = a.vhead[env]; vs = a.vset[env]; action = a.act[env]; }
    uint32_t w[24];
    int nw = 4;
    w[0] = h.pre4; w[1] = vh; w[2] = (uint32_t)vs; w[3] = (uint32_t)st;
    uint32_t extra = 0;
    const int fx = h.ax + fdx(h.dir), fy = h.ay + fdy(h.dir);
    if (active) {
        int fe;
        if (LAYOUT == 0) {
            const uint8_t* tile = plane + (ee >> 6) * (int64_t)(1024 * 64) + (ee & 63);
            fe = tile[((fy + 5) * 32 + fx + 5) * 64];
            if (action == 2 && (fe & 1)) { h.ax = fx; h.ay = fy; }
            const int tx = h.ax + 5 + (h.dir == 0 ? 0 : h.dir == 2 ? -6 : -3), ty = h.ay + 5 + (h.dir == 1 ? 0 : h.dir == 3 ? -6 : -3);
            uint32_t acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    const int idx = r * 7 + q;
                    acc[idx >> 2] |= (uint32_t)tile[((ty + r) * 32 + tx + q) * 64] << (8 * (idx & 3));
                }
#pragma unroll
            for (int k = 0; k < 13; ++k) w[k] = acc[k];
            nw = 13;
        } else if (LAYOUT == 1) {
            const uint8_t* rec = plane + ee * (int64_t)1024;
            auto zaddr = [](int x, int y) { return ((y >> 3) * 2 + (x >> 4)) * 128 + (y & 7) * 16 + (x & 15); };
            fe = rec[zaddr(fx + 5, fy + 5)];
            if (action == 2 && (fe & 1)) { h.ax = fx; h.ay = fy; }
            const int tx = h.ax + 5 + (h.dir == 0 ? 0 : h.dir == 2 ? -6 : -3), ty = h.ay + 5 + (h.dir == 1 ? 0 : h.dir == 3 ? -6 : -3);
            const int x4 = tx & ~3;
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int d = 0; d < 3; ++d) w[3 * r + d] = *(const uint32_t*)(rec + zaddr(x4 + 4 * d, ty + r));
            nw = 21;
        } else {
            const uint32_t fc = fcache[env];                                      // coalesced: last step's front cell + carried appearance
            fe = fc & 0xFF;
            if (action == 2 && (fe & 1)) { h.ax = fx; h.ay = fy; }
            const int tx = h.ax + 5 + (h.dir == 0 ? 0 : h.dir == 2 ? -6 : -3), ty = h.ay + 5 + (h.dir == 1 ? 0 : h.dir == 3 ? -6 : -3);
            const int xo = tx >> 3, yo = ty >> 1;
            const uint8_t* line = plane + ee * (int64_t)(52 * 128) + (yo * 4 + xo) * 128;
            const int c4 = (tx - 8 * xo) & ~3, r0 = ty - 2 * yo;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const uint32_t* q = (const uint32_t*)(line + (r0 + r) * 16 + c4);
                w[3 * r] = q[0]; w[3 * r + 1] = q[1]; w[3 * r + 2] = (c4 + 8 < 16) ? q[2] : 0u;
            }
            nw = 21;
            extra = fc >> 8;
            fcache[env] = (uint16_t)((w[10] & 0xFF) | (fc & 0xFF00));            // next step's front cell out of this window
        }
        if (LAYOUT != 2) extra = (fe & 2) ? 1u : 0u;
        h.step++;
        h.ax = ax0; h.ay = ay0;
        a.hot[env] = h;
        a.stale[env] = st + 1;
        a.rew[env] = (float)extra;
        a.done[env] = (uint8_t)(extra & 1);
        a.dirs[env] = h.dir;
        fake_obs((uint32_t*)(s_obs + threadIdx.x * PAD), w, nw, extra);
    }
    __syncthreads();
    copy_out<BLOCK>(s_obs, a.image, env0, a.n);
}

constexpr int BLOCK = 256;
// two envs per lane: the block of 256 lanes owns 512 envs; every lane issues both envs' loads before using either
__global__ __launch_bounds__(BLOCK) void k_flat2(Args a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_obs[BLOCK * PAD];
    const int ES = a.L.ES, es4 = ES >> 2;
    const int64_t base = (int64_t)blockIdx.x * (2 * BLOCK);
    Hot h[2]; uint64_t st[2]; uint32_t vh[2]; uint64_t vs[2]; int action[2]; uint32_t w[2][24]; uint32_t extra[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t env = base + k * BLOCK + threadIdx.x;
        if (env < a.n) { h[k] = a.hot[env]; st[k] = a.stale[env]; vh[k] = a.vhead[env]; vs[k] = a.vset[env]; action[k] = a.act[env]; }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t env = base + k * BLOCK + threadIdx.x;
        if (env < a.n) {
            const uint8_t* rec = a.rec + env * (int64_t)a.L.stride;
            const int fx = h[k].ax + fdx(h[k].dir), fy = h[k].ay + fdy(h[k].dir);
            const int tx = h[k].ax + (h[k].dir == 0 ? 0 : h[k].dir == 2 ? -7 : -3), ty = h[k].ay + (h[k].dir == 1 ? 0 : h[k].dir == 3 ? -7 : -3);
            const int a0 = a.L.OFF_E + ty * ES + tx;
            const uint32_t* q = (const uint32_t*)(rec + (a0 & ~3));
#pragma unroll
            for (int r = 0; r < 8; ++r) { w[k][3 * r] = q[r * es4]; w[k][3 * r + 1] = q[r * es4 + 1]; w[k][3 * r + 2] = q[r * es4 + 2]; }
            extra[k] = rec[a.L.OFF_APP + (h[k].carry & 31)] ^ rec[a.L.OFF_I + fy * 22 + fx];
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t env0 = base + k * BLOCK, env = env0 + threadIdx.x;
        if (env < a.n) {
            if (action[k] == 2 && (w[k][12] & 1)) h[k].ax++;
            h[k].step++;
            a.hot[env] = h[k]; a.stale[env] = st[k] + 1; a.rew[env] = (float)extra[k]; a.done[env] = (uint8_t)(extra[k] & 1); a.dirs[env] = h[k].dir;
            w[k][0] ^= vh[k] ^ (uint32_t)vs[k];
            fake_obs((uint32_t*)(s_obs + threadIdx.x * PAD), w[k], 24, extra[k]);
        }
        __syncthreads();
        if (env0 < a.n) copy_out<BLOCK>(s_obs, a.image, env0, a.n);
        __syncthreads();
    }
}

// the gather alone: 7 rows x 3 dwords at the record stride, addresses from the env index only
__global__ __launch_bounds__(BLOCK) void k_gather(Args a) {
    const int64_t env = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (env >= a.n) return;
    const uint32_t r = mix((uint32_t)env);
    const int a0 = a.L.OFF_E + (r % 14) * a.L.ES + (r >> 8) % 14;
    const uint32_t* q = (const uint32_t*)(a.rec + env * (int64_t)a.L.stride + (a0 & ~3));
    const int es4 = a.L.ES >> 2;
    uint32_t s = 0;
#pragma unroll
    for (int rr = 0; rr < 7; ++rr) s += q[rr * es4] + q[rr * es4 + 1] + q[rr * es4 + 2];
    a.sink[env] = s;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1048576;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int lay = 0; lay < 2; ++lay) {
        Args a; a.n = n; a.L = lay == 0 ? LAY32 : LAY24;
        const int stride = a.L.stride;
        uint8_t* rec_base; (void)hipMalloc(&rec_base, (size_t)n * stride + 8192); a.rec = rec_base + 4096; /* speculative rows may reach past either end */ (void)hipMalloc(&a.hot, n * 16); (void)hipMalloc(&a.stale, n * 8);
        (void)hipMalloc((void**)&a.vhead, n * 4); (void)hipMalloc((void**)&a.vset, n * 8); (void)hipMalloc((void**)&a.act, n);
        (void)hipMalloc(&a.image, n * OBS + 64); (void)hipMalloc(&a.dirs, n); (void)hipMalloc(&a.rew, n * 4); (void)hipMalloc(&a.done, n);
        (void)hipMalloc(&a.sink, n * 4);
        hipLaunchKernelGGL(k_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, a.hot, a.stale, (uint32_t*)a.vhead, (uint64_t*)a.vset,
                           (uint8_t*)a.act, a.rec, stride);
        (void)hipDeviceSynchronize();
        const unsigned grid = (unsigned)((n + BLOCK - 1) / BLOCK);
        const char* names[] = {"soa", "gather", "chain", "flat", "flat2", "coopR", "coopU", "chain/128-thread blocks", "chain/64-thread blocks",
                               "coopR/128-thread blocks", "coopR/64-thread blocks", "soa/64-thread blocks"};
        for (int v = 0; v < 12; ++v) {
            if (lay != 0 && (v == 0 || v == 11)) continue;      // soa does not touch the records
            float best = 1e9f, sum = 0; const int reps = 24, skip = 4;
            for (int r = 0; r < reps; ++r) {
                (void)hipEventRecord(e0, 0);
                switch (v) {
                case 0: hipLaunchKernelGGL((k_shape<0, 256>), dim3(grid), dim3(BLOCK), 0, 0, a); break;
                case 7: hipLaunchKernelGGL((k_shape<2, 128>), dim3(grid * 2), dim3(128), 0, 0, a); break;
                case 8: hipLaunchKernelGGL((k_shape<2, 64>), dim3(grid * 4), dim3(64), 0, 0, a); break;
                case 9: hipLaunchKernelGGL((k_shape<4, 128>), dim3(grid * 2), dim3(128), 0, 0, a); break;
                case 10: hipLaunchKernelGGL((k_shape<4, 64>), dim3(grid * 4), dim3(64), 0, 0, a); break;
                case 11: hipLaunchKernelGGL((k_shape<0, 64>), dim3(grid * 4), dim3(64), 0, 0, a); break;
                case 1: hipLaunchKernelGGL(k_gather, dim3(grid), dim3(BLOCK), 0, 0, a); break;
                case 2: hipLaunchKernelGGL((k_shape<2, 256>), dim3(grid), dim3(BLOCK), 0, 0, a); break;
                case 3: hipLaunchKernelGGL((k_shape<3, 256>), dim3(grid), dim3(BLOCK), 0, 0, a); break;
                case 4: hipLaunchKernelGGL(k_flat2, dim3((grid + 1) / 2), dim3(BLOCK), 0, 0, a); break;
                case 5: hipLaunchKernelGGL((k_shape<4, 256>), dim3(grid), dim3(BLOCK), 0, 0, a); break;
                case 6: hipLaunchKernelGGL((k_shape<5, 256>), dim3(grid), dim3(BLOCK), 0, 0, a); break;
                }
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= skip) { sum += ms; if (ms < best) best = ms; }
            }
            const float avg = sum / (reps - skip);
            const double alg = v == 1 ? 7.0 * 12 + 4 : 235.0;
            printf("{\"variant\": \"%s\", \"envs\": %lld, \"row_pitch\": %d, \"rec_stride\": %d, \"avg_ms\": %.4f, \"min_ms\": %.4f, "
                   "\"alg_bytes_per_env\": %.0f, \"alg_GBs_at_avg\": %.0f, \"frac_of_8TBs\": %.3f}\n", names[v], (long long)n, a.L.ES, stride, avg, best,
                   alg, n * alg / avg / 1e6, n * alg / avg / 1e6 / 8000.0);
            fflush(stdout);
        }
        (void)hipFree(rec_base); (void)hipFree(a.hot); (void)hipFree(a.stale); (void)hipFree((void*)a.vhead); (void)hipFree((void*)a.vset);
        (void)hipFree((void*)a.act); (void)hipFree(a.image); (void)hipFree(a.dirs); (void)hipFree(a.rew); (void)hipFree(a.done); (void)hipFree(a.sink);
    }
    {   // round 3 layouts: their own planes (cells 0/1 like the records above), the same SoA arrays
        Args a; a.n = n; a.L = LAY32;
        (void)hipMalloc(&a.hot, n * 16); (void)hipMalloc(&a.stale, n * 8);
        (void)hipMalloc((void**)&a.vhead, n * 4); (void)hipMalloc((void**)&a.vset, n * 8); (void)hipMalloc((void**)&a.act, n);
        (void)hipMalloc(&a.image, n * OBS + 64); (void)hipMalloc(&a.dirs, n); (void)hipMalloc(&a.rew, n * 4); (void)hipMalloc(&a.done, n);
        (void)hipMalloc(&a.sink, n * 4);
        uint8_t* scratch_rec; (void)hipMalloc(&scratch_rec, (size_t)n * 64);
        hipLaunchKernelGGL(k_init, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, n, a.hot, a.stale, (uint32_t*)a.vhead, (uint64_t*)a.vset,
                           (uint8_t*)a.act, scratch_rec, 64);
        const size_t plane_bytes[3] = {(size_t)((n + 63) / 64) * 1024 * 64, (size_t)n * 1024, (size_t)n * 52 * 128};
        const char* names[3] = {"envtile: E[tile][cell][64]", "ztile: 16x8-cell tiles (128 B)", "vline: one line per window origin class (x/8, y/2) + cached front cell"};
        uint16_t* fcache; (void)hipMalloc(&fcache, n * 2); (void)hipMemset(fcache, 1, n * 2);
        const unsigned grid = (unsigned)((n + 255) / 256);
        for (int v = 0; v < 3; ++v) {
            uint8_t* plane; (void)hipMalloc(&plane, plane_bytes[v] + 4096); (void)hipMemset(plane, 1, plane_bytes[v] + 4096);
            (void)hipDeviceSynchronize();
            float best = 1e9f, sum = 0; const int reps = 24, skip = 4;
            for (int r = 0; r < reps; ++r) {
                (void)hipEventRecord(e0, 0);
                if (v == 0) hipLaunchKernelGGL((k_layout<0>), dim3(grid), dim3(256), 0, 0, a, plane, fcache);
                else if (v == 1) hipLaunchKernelGGL((k_layout<1>), dim3(grid), dim3(256), 0, 0, a, plane, fcache);
                else hipLaunchKernelGGL((k_layout<2>), dim3(grid), dim3(256), 0, 0, a, plane, fcache);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (r >= skip) { sum += ms; if (ms < best) best = ms; }
            }
            const float avg = sum / (reps - skip);
            printf("{\"variant\": \"%s\", \"envs\": %lld, \"plane_bytes_per_env\": %.0f, \"avg_ms\": %.4f, \"min_ms\": %.4f, "
                   "\"alg_bytes_per_env\": 235, \"alg_GBs_at_avg\": %.0f, \"frac_of_8TBs\": %.3f}\n", names[v], (long long)n, (double)plane_bytes[v] / n, avg, best,
                   n * 235.0 / avg / 1e6, n * 235.0 / avg / 1e6 / 8000.0);
            fflush(stdout);
            (void)hipFree(plane);
        }
    }
    return 0;
}
