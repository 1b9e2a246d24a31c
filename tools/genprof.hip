// Generator phase profiler (experiment tool): runs the wave-per-env level generator of bbai_gen.hpp with a
// profiling context and prints average cycles per phase / attempts / RNG draws per generated level.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/genprof tools/genprof.hip && ./tools/genprof BossLevel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../babyai_amd/csrc/bbai_types.hpp"
#include "../babyai_amd/csrc/bbai_gen.hpp"
#include "../babyai_amd/csrc/bbai_seed.hpp"
using namespace bbai;
struct ProfCtx {
    static constexpr bool kProfile = true;
    static constexpr int kLanes = 64;
    __device__ uint32_t shfl(uint32_t v, int src) const { return __shfl(v, src); }
    __device__ int lane() const { return (int)threadIdx.x; }
    __device__ int nlanes() const { return 64; }
    __device__ void sync() const { __syncthreads(); }
    __device__ uint32_t shfl_up1(uint32_t v) const { uint32_t t = __shfl_up(v, 1); return threadIdx.x == 0 ? 0u : t; }
    __device__ uint32_t shfl_down1(uint32_t v) const { uint32_t t = __shfl_down(v, 1); return threadIdx.x == 63 ? 0u : t; }
    __device__ bool any(bool p) const { return __ballot(p) != 0ull; }
    __device__ unsigned long long now() const { return wall_clock64(); }
};
__global__ __launch_bounds__(64) void k_prof(LevelCfg c, int n, int rounds, uint32_t* mts, unsigned long long* out, unsigned long long* per_level) {
    __shared__ GenWork w;
    for (int env = blockIdx.x; env < n; env += gridDim.x) {
        uint32_t* mt = mts + (size_t)env * MT_N;
        __syncthreads();
        for (int k = threadIdx.x; k < MT_N; k += 64) w.mt[k] = mt[k];
        __syncthreads();
        int mti = MT_N, last = -1;
        for (int r = 0; r < rounds; ++r) {
            unsigned long long t0 = wall_clock64();
            Gen<ProfCtx> g(ProfCtx(), c, w, mti, last);
            g.generate();
            mti = g.mti; last = g.last_locked;
            unsigned long long t1 = wall_clock64();
            if (threadIdx.x == 0) {
                for (int k = 0; k < PH_N; ++k) atomicAdd(&out[k], g.prof[k]);
                atomicAdd(&out[PH_N], t1 - t0);
                per_level[(size_t)env * rounds + r] = t1 - t0;
            }
        }
    }
}
int main(int argc, char** argv) {
    const char* name = argc > 1 ? argv[1] : "BossLevel";
    LevelCfg c; memset(&c, 0, sizeof(c));
    c.room_size = 8; c.num_rows = 3; c.num_cols = 3; c.num_dists = 18;
    if (!strcmp(name, "BossLevel")) { c.kind = K_LEVELGEN; c.locked_room_prob = 0.5; c.locations = 1; c.unblocking = 1; c.implicit_unlock = 1;
        c.n_action_kinds = 4; for (int i = 0; i < 4; ++i) c.action_kinds[i] = i; c.n_instr_kinds = 3; for (int i = 0; i < 3; ++i) c.instr_kinds[i] = i; }
    else if (!strcmp(name, "GoTo")) { c.kind = K_GOTO; c.connect = 1; c.check_reach = 1; c.instr = L_GOTO; c.target = TG_DIST; }
    else if (!strcmp(name, "GoToLocal")) { c.kind = K_GOTO; c.num_rows = c.num_cols = 1; c.num_dists = 8; c.check_reach = 1; c.instr = L_GOTO; c.target = TG_DIST; }
    else { printf("unknown level\n"); return 1; }
    fill_layout(c);
    const int n = 4096, rounds = 8;
    std::vector<uint32_t> mt((size_t)n * MT_N);
    for (int i = 0; i < n; ++i) seed_env(1000 + i, mt.data() + (size_t)i * MT_N);
    uint32_t* dmt; unsigned long long *dout, *dper;
    hipMalloc(&dmt, mt.size() * 4); hipMalloc(&dout, 8 * 32); hipMalloc(&dper, (size_t)n * rounds * 8);
    hipMemcpy(dmt, mt.data(), mt.size() * 4, hipMemcpyHostToDevice); hipMemset(dout, 0, 8 * 32);
    hipLaunchKernelGGL(k_prof, dim3(n), dim3(64), 0, 0, c, n, rounds, dmt, dout, dper);
    hipDeviceSynchronize();
    unsigned long long out[32]; hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost);
    std::vector<unsigned long long> per((size_t)n * rounds); hipMemcpy(per.data(), dper, per.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"build", "lock", "connect", "dists", "agent", "reach", "instr", "validate", "ATTEMPTS", "DRAWS", "TWISTS"};
    double L = (double)n * rounds;
    printf("%s: %d levels; wall_clock64 ticks are 100 MHz (10 ns)\n", name, (int)L);
    for (int k = 0; k < PH_N; ++k) printf("  %-9s %10.1f per level\n", names[k], out[k] / L);
    printf("  total     %10.1f ticks per level = %.1f us\n", out[PH_N] / L, out[PH_N] / L * 0.01);
    std::sort(per.begin(), per.end());
    printf("  per-level us: p50 %.1f  p90 %.1f  p99 %.1f  max %.1f\n", per[per.size() / 2] * 0.01, per[per.size() * 9 / 10] * 0.01,
           per[per.size() * 99 / 100] * 0.01, per.back() * 0.01);
    return 0;
}
