// tools/genl_check.hip -- the lane = level generator (bbai_genl.hpp) on the DEVICE against the same header on the HOST: one lane per env, K levels
// per env, every record / pose / RNG position compared.  (The host form is pinned to bbai_gen.hpp and the oracle by tests/test_hostsim_genl.py;
// this closes the remaining gap: what the device compiler makes of the same source.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/genl_check tools/genl_check.hip && /tmp/genl_check SynthS5R2 4096 8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#define BBAI_GENL_TRACE(m, tag, val) (m).trace(tag, val)
#include "../babyai_amd/csrc/bbai_types.hpp"
#include "../babyai_amd/csrc/bbai_gen.hpp"
#include "../babyai_amd/csrc/bbai_genl.hpp"
#include "../babyai_amd/csrc/bbai_seed.hpp"
using namespace bbai;
constexpr int TR = 9000;
struct DevMem : LaneRng<DevMem> {
    uint32_t* lds;
    int32_t* tr; int ntr;
    __device__ __forceinline__ void trace(int tag, int val) { if (tr && ntr + 3 <= TR) { tr[ntr] = tag; tr[ntr + 1] = val; tr[ntr + 2] = position() + 10000 * par; ntr += 3; } }
    __device__ __forceinline__ uint32_t next_u32() { const uint32_t y = LaneRng<DevMem>::next_u32(); trace(50, (int)(y & 0xFFFF)); return y; }
    __device__ __forceinline__ void topup() { if (__ballot(low()) != 0ull && avail() < LANE_FIFO) refill(); }
    __device__ __forceinline__ uint32_t ld(int k) const { return lds[k << 6]; }
    __device__ __forceinline__ void st(int k, uint32_t v) { lds[k << 6] = v; }
};
struct HostMem : LaneRng<HostMem> {
    uint32_t* w;
    int32_t* tr; int ntr;
    void trace(int tag, int val) { if (tr && ntr + 3 <= TR) { tr[ntr] = tag; tr[ntr + 1] = val; tr[ntr + 2] = position() + 10000 * par; ntr += 3; } }
    uint32_t next_u32() { const uint32_t y = LaneRng<HostMem>::next_u32(); trace(50, (int)(y & 0xFFFF)); return y; }
    void topup() { if (low()) refill(); }
    uint32_t ld(int k) const { return w[k]; }
    void st(int k, uint32_t v) { w[k] = v; }
};
struct Out { int32_t pos, par, nobj, attempts; Hot hot; };
template <int KIND>
__global__ __launch_bounds__(64) void k_check(LevelCfg c, int n, int levels, uint32_t* mts, uint32_t* mtt, uint8_t* recs, Out* outs, const uint8_t* tmpl, int lane_words, int32_t* trace, int trace_env) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    const int env = blockIdx.x * 64 + threadIdx.x;
    if (env >= n) return;
    DevMem mem;
    mem.lds = s_dyn + threadIdx.x; mem.mts_env = mts + (size_t)env * MT_N; mem.mtt_env = mtt + (size_t)env * 2 * MT_N; mem.fifo0 = lane_layout(c).fifo; mem.start(MT_N, 0);
    mem.tr = env == trace_env ? trace : nullptr; mem.ntr = 0;
    int last_locked = -1;
    for (int l = 0; l < levels; ++l) {
        int attempts = 0;
        for (;;) {
            GenL<DevMem> g(mem, c, last_locked);
            const bool ok = g.template attempt<KIND>();
            last_locked = g.last_locked;
            ++attempts;
            if (!ok) continue;
            g.write_record(recs + ((size_t)env * levels + l) * c.rec_bytes, tmpl);
            Out o; o.pos = mem.position(); o.par = mem.par; o.nobj = g.nobj; o.attempts = attempts;
            memset(&o.hot, 0, sizeof(Hot));
            o.hot.ax = g.ax; o.hot.ay = g.ay; o.hot.dir = g.adir; o.hot.max_steps = (uint16_t)g.max_steps();
            o.hot.last_locked = last_locked < 0 ? NONE8 : (uint8_t)last_locked;
            outs[(size_t)env * levels + l] = o;
            break;
        }
    }
}
template <int KIND>
static void host_ref(const LevelCfg& c, int levels, uint32_t* mt, uint8_t* recs, Out* outs, const uint8_t* tmpl, int32_t* trace = nullptr) {
    static HostMem mem;
    static uint32_t words[160];
    std::vector<uint32_t> mtt(2 * MT_N, 0);
    mem.w = words;
    memset(words, 0xA5, sizeof(words));
    mem.mts_env = mt; mem.mtt_env = mtt.data(); mem.fifo0 = lane_layout(c).fifo; mem.start(MT_N, 0);
    mem.tr = trace; mem.ntr = 0;
    int last_locked = -1;
    alignas(16) static uint8_t arec[4096];
    for (int l = 0; l < levels; ++l) {
        int attempts = 0;
        for (;;) {
            GenL<HostMem> g(mem, c, last_locked);
            const bool ok = g.template attempt<KIND>();
            last_locked = g.last_locked;
            ++attempts;
            if (!ok) continue;
            g.write_record(arec, tmpl);
            memcpy(recs + (size_t)l * c.rec_bytes, arec, c.rec_bytes);
            Out o; o.pos = mem.position(); o.par = mem.par; o.nobj = g.nobj; o.attempts = attempts;
            memset(&o.hot, 0, sizeof(Hot));
            o.hot.ax = g.ax; o.hot.ay = g.ay; o.hot.dir = g.adir; o.hot.max_steps = (uint16_t)g.max_steps();
            o.hot.last_locked = last_locked < 0 ? NONE8 : (uint8_t)last_locked;
            outs[l] = o;
            break;
        }
    }
}
int main(int argc, char** argv) {
    const char* name = argc > 1 ? argv[1] : "SynthS5R2";
    LevelCfg c; memset(&c, 0, sizeof(c));
    c.room_size = 8; c.num_rows = 3; c.num_cols = 3; c.num_dists = 18; c.kind = K_LEVELGEN; c.locked_room_prob = 0.5; c.locations = 1; c.unblocking = 1; c.implicit_unlock = 1;
    c.n_action_kinds = 4; for (int i = 0; i < 4; ++i) c.action_kinds[i] = i; c.n_instr_kinds = 3; for (int i = 0; i < 3; ++i) c.instr_kinds[i] = i;
    if (!strcmp(name, "BossLevel")) {}
    else if (!strcmp(name, "SynthS5R2")) { c.room_size = 5; c.num_rows = c.num_cols = 2; c.num_dists = 7; c.n_instr_kinds = 1; c.locations = 0; c.implicit_unlock = 0; }
    else if (!strcmp(name, "MiniBossLevel")) { c.room_size = 5; c.num_rows = c.num_cols = 2; c.num_dists = 7; c.locked_room_prob = 0.25; }
    else if (!strcmp(name, "PickupLoc")) { c.num_rows = c.num_cols = 1; c.num_dists = 8; c.locked_room_prob = 0; c.unblocking = 0; c.n_action_kinds = 1; c.action_kinds[0] = AK_PICKUP; c.n_instr_kinds = 1; }
    else if (!strcmp(name, "GoTo")) { c.kind = K_GOTO; c.connect = 1; c.check_reach = 1; c.instr = L_GOTO; c.target = TG_DIST; c.locked_room_prob = 0; c.locations = 0; c.unblocking = 0; c.implicit_unlock = 0; c.n_action_kinds = 0; c.n_instr_kinds = 0; }
    else if (!strcmp(name, "GoToLocal")) { c.kind = K_GOTO; c.num_rows = c.num_cols = 1; c.num_dists = 8; c.check_reach = 1; c.instr = L_GOTO; c.target = TG_DIST; c.locked_room_prob = 0; c.locations = 0; c.unblocking = 0; c.implicit_unlock = 0; c.n_action_kinds = 0; c.n_instr_kinds = 0; }
    else { printf("unknown level\n"); return 1; }
    if (getenv("GC_IMPLICIT")) c.implicit_unlock = atoi(getenv("GC_IMPLICIT"));
    if (getenv("GC_LOC")) c.locations = atoi(getenv("GC_LOC"));
    if (getenv("GC_PROB")) c.locked_room_prob = atof(getenv("GC_PROB"));
    if (getenv("GC_NINSTR")) c.n_instr_kinds = atoi(getenv("GC_NINSTR"));
    if (getenv("GC_UNB")) c.unblocking = atoi(getenv("GC_UNB"));
    if (getenv("GC_NACT")) c.n_action_kinds = atoi(getenv("GC_NACT"));
    if (fill_layout(c) != 0) { printf("layout failed\n"); return 1; }
    const int n = argc > 2 ? atoi(argv[2]) : 1024, levels = argc > 3 ? atoi(argv[3]) : 6;
    std::vector<uint32_t> mt((size_t)n * MT_N);
    for (int i = 0; i < n; ++i) seed_env(40 + i, mt.data() + (size_t)i * MT_N);
    std::vector<uint8_t> tmpl(lane_template_bytes(c));
    lane_build_template(c, tmpl.data());
    uint32_t *dmt, *dtt; uint8_t *drec, *dtmpl; Out* dout;
    hipMalloc(&dmt, mt.size() * 4); hipMalloc(&dtt, mt.size() * 8); hipMalloc(&drec, (size_t)n * levels * c.rec_bytes); hipMalloc(&dout, (size_t)n * levels * sizeof(Out));
    hipMalloc(&dtmpl, tmpl.size());
    hipMemcpy(dmt, mt.data(), mt.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dtmpl, tmpl.data(), tmpl.size(), hipMemcpyHostToDevice);
    hipMemset(drec, 0xEE, (size_t)n * levels * c.rec_bytes);
    const int lw = lane_layout(c).words;
    const int trace_env = getenv("GC_TRACE") ? atoi(getenv("GC_TRACE")) : -1;
    int32_t* dtrace; hipMalloc(&dtrace, TR * 4); hipMemset(dtrace, 0, TR * 4);
    if (c.kind == K_LEVELGEN) hipLaunchKernelGGL(k_check<K_LEVELGEN>, dim3((n + 63) / 64), dim3(64), lw * 256, 0, c, n, levels, dmt, dtt, drec, dout, dtmpl, lw, dtrace, trace_env);
    else hipLaunchKernelGGL(k_check<K_GOTO>, dim3((n + 63) / 64), dim3(64), lw * 256, 0, c, n, levels, dmt, dtt, drec, dout, dtmpl, lw, dtrace, trace_env);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<uint8_t> recs((size_t)n * levels * c.rec_bytes); std::vector<Out> outs((size_t)n * levels);
    hipMemcpy(recs.data(), drec, recs.size(), hipMemcpyDeviceToHost); hipMemcpy(outs.data(), dout, outs.size() * sizeof(Out), hipMemcpyDeviceToHost);
    long bad = 0;
    std::vector<uint8_t> hrec((size_t)levels * c.rec_bytes); std::vector<Out> hout(levels);
    for (int i = 0; i < n; ++i) {
        if (c.kind == K_LEVELGEN) host_ref<K_LEVELGEN>(c, levels, mt.data() + (size_t)i * MT_N, hrec.data(), hout.data(), tmpl.data());
        else host_ref<K_GOTO>(c, levels, mt.data() + (size_t)i * MT_N, hrec.data(), hout.data(), tmpl.data());
        for (int l = 0; l < levels; ++l) {
            const Out& a = outs[(size_t)i * levels + l]; const Out& b = hout[l];
            const bool same = !memcmp(&a, &b, sizeof(Out)) && !memcmp(&recs[((size_t)i * levels + l) * c.rec_bytes], &hrec[(size_t)l * c.rec_bytes], c.rec_bytes);
            if (!same) {
                if (bad < 8) printf("MISMATCH env %d level %d: device pos %d par %d nobj %d attempts %d | host pos %d par %d nobj %d attempts %d\n", i, l, a.pos, a.par, a.nobj, a.attempts, b.pos, b.par, b.nobj, b.attempts);
                ++bad;
                break;
            }
        }
    }
    if (trace_env >= 0) {
        std::vector<int32_t> dt(TR), ht(TR, 0);
        hipMemcpy(dt.data(), dtrace, TR * 4, hipMemcpyDeviceToHost);
        std::vector<uint32_t> mt2(MT_N); seed_env(40 + trace_env, mt2.data());
        if (c.kind == K_LEVELGEN) host_ref<K_LEVELGEN>(c, levels, mt2.data(), hrec.data(), hout.data(), tmpl.data(), ht.data());
        else host_ref<K_GOTO>(c, levels, mt2.data(), hrec.data(), hout.data(), tmpl.data(), ht.data());
        for (int k = 0; k + 3 <= TR; k += 3) {
            const bool same = dt[k] == ht[k] && dt[k + 1] == ht[k + 1] && dt[k + 2] == ht[k + 2];
            printf("  %s tag %2d  dev val %d pos %d | host tag %2d val %d pos %d\n", same ? "  " : "!!", dt[k], dt[k + 1], dt[k + 2], ht[k], ht[k + 1], ht[k + 2]);
            if (!same) break;
        }
    }
    printf("%s: %d envs x %d levels, %ld envs differ\n", name, n, levels, bad);
    return bad ? 2 : 0;
}
