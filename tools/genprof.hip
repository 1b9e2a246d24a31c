// tools/genprof.hip -- phase profile of the level generator at the product's execution model (one env per group of 32 lanes, two envs
// per wave, working set in LDS): average shader-clock cycles per phase / attempts / RNG draws per generated level, and the per-level
// latency percentiles.  The hooks are bbai_gen.hpp's own (Gen::tick / Gen::count, compiled in for a context with kProfile).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/genprof tools/genprof.hip && /tmp/genprof GoToLocal      (lease.sh genprof)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../babyai_amd/csrc/bbai_types.hpp"
#include "../babyai_amd/csrc/bbai_gen.hpp"
#include "../babyai_amd/csrc/bbai_seed.hpp"
using namespace bbai;
constexpr int G = 32;
struct ProfCtx {                                   // bbai_engine.hip GroupCtx<32> + the profiling hooks
    static constexpr int kLanes = G;
    static constexpr bool kProfile = true;
    __device__ __forceinline__ int lane() const { return (int)threadIdx.x & (G - 1); }
    __device__ __forceinline__ int nlanes() const { return G; }
    __device__ __forceinline__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __device__ __forceinline__ uint32_t shfl(uint32_t v, int src) const { return __shfl(v, src, G); }
    __device__ __forceinline__ uint32_t shfl_up1(uint32_t v) const { uint32_t t = __shfl_up(v, 1, G); return lane() == 0 ? 0u : t; }
    __device__ __forceinline__ uint32_t shfl_down1(uint32_t v) const { uint32_t t = __shfl_down(v, 1, G); return lane() == G - 1 ? 0u : t; }
    __device__ __forceinline__ bool any(bool p) const {
        const unsigned long long b = __ballot(p);
        return ((b >> ((int)threadIdx.x & ~(G - 1) & 63)) & 0xFFFFFFFFull) != 0ull;
    }
    __device__ __forceinline__ unsigned long long now() const { return (unsigned long long)clock64(); }
};
template <int KIND>
__global__ __launch_bounds__(64, 4) void k_prof(LevelCfg c, int n, int rounds, uint32_t* mts, unsigned long long* out, unsigned long long* per_level) {
    __shared__ GenWork ws[2];
    __shared__ uint32_t s_mt[2][MT_N + MT_CH];
    const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
    GenWork& w = ws[grp];
    const ProfCtx ctx;
    for (int env = blockIdx.x * 2 + grp; env < n; env += gridDim.x * 2) {
        uint32_t* mt = mts + (size_t)env * MT_N;
        ctx.sync();
        for (int k = lane; k < MT_N; k += G) s_mt[grp][k] = mt[k];
        ctx.sync();
        int mti = MT_N, last = -1;
        for (int r = 0; r < rounds; ++r) {
            const unsigned long long t0 = clock64();
            Gen<ProfCtx> g(ctx, c, w, s_mt[grp], s_mt[grp] + MT_N, mti, last);
            g.template generate_kind<KIND>();
            mti = g.mti; last = g.last_locked;
            const unsigned long long t1 = clock64();
            if (lane == 0) {
                for (int k = 0; k < PH_N; ++k) atomicAdd(&out[k], g.prof[k]);
                atomicAdd(&out[PH_N], t1 - t0);
                per_level[(size_t)env * rounds + r] = t1 - t0;
            }
        }
    }
}
int main(int argc, char** argv) {
    const char* name = argc > 1 ? argv[1] : "GoToLocal";
    LevelCfg c; memset(&c, 0, sizeof(c));
    c.room_size = 8; c.num_rows = 3; c.num_cols = 3; c.num_dists = 18;
    int kind = K_GOTO;
    if (!strcmp(name, "BossLevel")) { kind = c.kind = K_LEVELGEN; c.locked_room_prob = 0.5; c.locations = 1; c.unblocking = 1; c.implicit_unlock = 1;
        c.n_action_kinds = 4; for (int i = 0; i < 4; ++i) c.action_kinds[i] = i; c.n_instr_kinds = 3; for (int i = 0; i < 3; ++i) c.instr_kinds[i] = i; }
    else if (!strcmp(name, "PickupLoc")) { kind = c.kind = K_LEVELGEN; c.num_rows = c.num_cols = 1; c.num_dists = 8; c.locked_room_prob = 0; c.locations = 1; c.unblocking = 0;
        c.implicit_unlock = 1; c.n_action_kinds = 1; c.action_kinds[0] = AK_PICKUP; c.n_instr_kinds = 1; c.instr_kinds[0] = IK_ACTION; }
    else if (!strcmp(name, "GoTo")) { c.kind = K_GOTO; c.connect = 1; c.check_reach = 1; c.instr = L_GOTO; c.target = TG_DIST; }
    else if (!strcmp(name, "GoToLocal")) { c.kind = K_GOTO; c.num_rows = c.num_cols = 1; c.num_dists = 8; c.check_reach = 1; c.instr = L_GOTO; c.target = TG_DIST; }
    else { printf("unknown level\n"); return 1; }
    if (fill_layout(c) != 0) { printf("layout failed\n"); return 1; }
    const int n = argc > 2 ? atoi(argv[2]) : 32768, rounds = 4;
    std::vector<uint32_t> mt((size_t)n * MT_N);
    for (int i = 0; i < n; ++i) seed_env(1000 + i, mt.data() + (size_t)i * MT_N);
    uint32_t* dmt; unsigned long long *dout, *dper;
    hipMalloc(&dmt, mt.size() * 4); hipMalloc(&dout, 8 * 32); hipMalloc(&dper, (size_t)n * rounds * 8);
    hipMemcpy(dmt, mt.data(), mt.size() * 4, hipMemcpyHostToDevice); hipMemset(dout, 0, 8 * 32);
    const int blocks = std::min(n / 2, 8192);
    if (kind == K_LEVELGEN) hipLaunchKernelGGL(k_prof<K_LEVELGEN>, dim3(blocks), dim3(64), 0, 0, c, n, rounds, dmt, dout, dper);
    else hipLaunchKernelGGL(k_prof<K_GOTO>, dim3(blocks), dim3(64), 0, 0, c, n, rounds, dmt, dout, dper);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    unsigned long long out[32]; hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> per((size_t)n * rounds); hipMemcpy(per.data(), dper, per.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"build", "lock", "connect", "dists", "agent", "reach", "instr", "validate", "ATTEMPTS", "DRAWS", "TWISTS"};
    const double L = (double)n * rounds;
    printf("%s: %d levels, %d lanes per env, %d waves resident-capable; shader-clock cycles per level (one group's view: its wave shares a SIMD with up to 3 others)\n", name, (int)L, G, blocks);
    for (int k = 0; k < PH_N; ++k) printf("  %-9s %10.1f\n", names[k], out[k] / L);
    printf("  total     %10.1f cycles per level\n", out[PH_N] / L);
    std::sort(per.begin(), per.end());
    printf("  per-level cycles: p50 %llu  p90 %llu  p99 %llu  max %llu\n", per[per.size() / 2], per[per.size() * 9 / 10], per[per.size() * 99 / 100], per.back());
    return 0;
}
