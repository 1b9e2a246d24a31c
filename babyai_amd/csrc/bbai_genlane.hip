// bbai_genlane.hip -- k_pregen_lane, the look-ahead level generator with ONE LANE = ONE LEVEL (bbai_genl.hpp), in a translation unit of
// its own because it is compiled with `-mllvm -disable-machine-cse`.
//
// Why.  ROCm 7.2's backend miscompiles this kernel at -O2 / -O3: with machine-CSE on, LevelGen.rand_obj takes the `rand_bool()` draw of
// `if (cfg.locations && rand_bool())` in SOME trips of its loop although cfg.locations is 0 -- one extra draw, the env's stream shifted for
// good (SynthS5R2: 930 of 2 048 envs wrong within eight levels; levels with locations = 1 or without PutNext unaffected).  Found with
// tools/genl_check.hip (this header on the device against the same header on the host), pinned by `-mllvm -opt-bisect-limit`: the first bad
// pass execution is "machine-cse on k_check" (profiles/r06/NOTES.md section 2).  -O1, or -O3 without that pass, generate every level of every
// covered kind exactly as the host does (tests/test_gpu_lane_generator.py keeps it that way: the product kernel against the lane-group
// kernel on every covered level).  The rest of the engine keeps the pass.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "bbai_types.hpp"
#include "bbai_kernels.hpp"
#include "bbai_gen.hpp"
#include "bbai_genl.hpp"
#include "bbai_step.hpp"
#include "bbai_view.hpp"

using namespace bbai;

// ---- k_pregen_lane: the look-ahead generator with ONE LANE = ONE LEVEL (bbai_genl.hpp) -----------------------------------------------
// A lane takes an entry of the refill's work list (the same list k_pregen walks; `dense`: every env), generates the env's pending levels one
// ATTEMPT of the rejection loop per trip of the wave's main loop, writes an accepted level out by itself (template copy + scatter) and goes
// on to its next level / entry while its neighbours retry.  Working set: lane_layout's words in LDS, interleaved over the wave's lanes.
// Draws come out of the env's tempered generations in memory (`mtt`, see bbai_genl.hpp); the top of the main loop is where the wave is
// converged and where it twists, all 64 lanes on one env's 624 words at a time, the state of every lane whose position has reached the
// latest generation.
struct LaneMemDev : LaneRng<LaneMemDev> {
    uint32_t* lds;                   // word k of this lane: lds[64 k]
    __device__ __forceinline__ uint32_t ld(int k) const { return lds[k << 6]; }
    __device__ __forceinline__ void st(int k, uint32_t v) { lds[k << 6] = v; }
    // GenL's hint "converged, about to draw": if any lane of the wave is running low, every lane with room fetches -- their loads travel together
    __device__ __forceinline__ void topup() { if (__ballot(low()) != 0ull && avail() < LANE_FIFO) refill(); }
};
struct StorePacker {                 // encode_cells_to's sink: the 37 dwords of an observation straight to (dword-aligned) memory
    uint32_t* p;
    __device__ __forceinline__ void put(int j, uint32_t d) { p[j] = d; }
    __device__ __forceinline__ void finish() {}
};
constexpr int LANE_PL_PITCH = 72;    // a lane's 8 x 8 plane in LDS (first observation of the small rooms), as k_step<.., CP> parks it
template <int KIND, bool OBS>
__global__ __launch_bounds__(64, 2) void k_pregen_lane(LevelCfg c, int64_t n, uint8_t* __restrict__ next_recs, Hot* __restrict__ next_hots,
                                                        uint32_t* __restrict__ mts, uint32_t* __restrict__ mtt, uint8_t* __restrict__ mtpar,
                                                        int32_t* __restrict__ mtis, const int32_t* __restrict__ gen_list,
                                                        const uint32_t* __restrict__ gen_count /* NULL: dense */, int depth,
                                                        uint8_t* __restrict__ pending, const uint8_t* __restrict__ first_slot,
                                                        unsigned long long* __restrict__ gen_failures, uint8_t* __restrict__ next_obs,
                                                        const uint8_t* __restrict__ tmpl, int lane_words) {
    extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
    // [lane_words][64] the lanes' words | one area for the cooperative twist's state ([MT_N + MT_CH] words, top of the main loop) AND, (OBS),
    // the lanes' 8 x 8 planes ([64][LANE_PL_PITCH] bytes, write-out): never live together
    uint32_t* const s_tw = s_dyn + lane_words * 64;
    uint8_t* const s_pl = (uint8_t*)s_tw;
    __shared__ uint32_t s_start[SHARDS + 1];
    const int lane = (int)threadIdx.x;
    const GroupCtx<64> wave;
    int64_t count = n;
    const int64_t cap = gen_sublist_cap(n);
    if (gen_count) {
        uint32_t cc = gen_count[threadIdx.x * GEN_COUNT_U32], incl = cc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += t; }
        s_start[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) s_start[0] = 0;
        __syncthreads();
        count = (int64_t)s_start[SHARDS];
    }
    if ((int64_t)blockIdx.x * 64 >= count) return;
    const int64_t stride = (int64_t)gridDim.x * 64;
    int64_t it = (int64_t)blockIdx.x * 64 + lane;
    LaneMemDev mem;
    mem.lds = s_dyn + lane; mem.mts_env = mts; mem.mtt_env = mtt; mem.fifo0 = lane_layout(c).fifo; mem.start(0, 0);
    bool have = false;
    int64_t env = 0;
    int cnt = 0, done_levels = 0, slot = 0, last_locked = -1, attempts = 0;
    for (;;) {
        if (!have) {
            while (it < count) {
                int64_t cand = it;
                if (gen_count) {
                    int j = 0;
#pragma unroll
                    for (int o = SHARDS / 2; o; o >>= 1) if ((int64_t)s_start[j + o] <= it) j += o;
                    cand = (int64_t)gen_list[(int64_t)j * cap + (it - (int64_t)s_start[j])];
                }
                it += stride;
                const int pc = pending[cand];
                if (pc == 0) continue;
                env = cand; cnt = pc; have = true;
                break;
            }
            if (have) {
                mem.start(mtis[env], mtpar[env]);
                mem.mts_env = mts + env * MT_N;
                mem.mtt_env = mtt + env * (2 * MT_N);
                slot = first_slot[env];
                const int prev = slot == 0 ? depth - 1 : slot - 1;
                last_locked = next_hots[ring_at(prev, env, depth)].last_locked;
                last_locked = last_locked == NONE8 ? -1 : last_locked;
                done_levels = 0; attempts = 0;
            }
        }
        if (__ballot(have) == 0ull) break;
        // Twists, where the wave is converged: every lane that has entered the latest generation gets the next one (and so at least MT_N
        // draws ahead of it).  The 64 lanes share an env's 624 words: load, the lane-group generator's own twist in LDS, raw state back,
        // tempered outputs over the half that held the generation before the previous one.
        {
            unsigned long long need = __ballot(have && mem.position() >= 0);
            if (need) {
                while (need) {
                    const int f = __ffsll((long long)need) - 1;
                    need &= need - 1;
                    const int64_t ef = ((int64_t)__shfl((int)(env >> 32), f) << 32) | (uint32_t)__shfl((int)env, f);
                    const int pf = __shfl(mem.par, f);
                    uint32_t* mt = mts + ef * MT_N;
                    uint32_t* tt = mtt + ef * (2 * MT_N) + (pf ^ 1) * MT_N;
                    uint32_t v[10];
#pragma unroll
                    for (int q = 0; q < 10; ++q) { const int k = lane + 64 * q; v[q] = mt[k < MT_N ? k : MT_N - 1]; }
                    wave.sync();
#pragma unroll
                    for (int q = 0; q < 10; ++q) { const int k = lane + 64 * q; if (k < MT_N) s_tw[k] = v[q]; }
                    mt_twist(wave, s_tw);
#pragma unroll
                    for (int q = 0; q < 10; ++q) {
                        const int k = lane + 64 * q;
                        if (k < MT_N) { const uint32_t x = s_tw[k]; mt[k] = x; tt[k] = mt_temper(x); }
                    }
                    if (lane == f) mem.twisted();
                }
                // the lanes read what their neighbours just stored: stores landed (release), this CU's vector cache forgets the old half (acquire)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        if (!have) continue;
        GenL<LaneMemDev> g(mem, c, last_locked);
        const bool ok = g.template attempt<KIND>();
        last_locked = g.last_locked;
        const bool gave_up = !ok && ++attempts >= Gen<GroupCtx<64>>::MAX_ATTEMPTS;
        if (!ok && !gave_up) continue;
        // write-out
        const int64_t at = ring_at(slot, env, depth);
        g.write_record(next_recs + at * (int64_t)c.rec_bytes, tmpl);
        if constexpr (OBS) {
            // in-place layout (small single rooms): the level's first observation through k_step's own pipeline -- the 8 x 8 plane in LDS,
            // window rows, rotation / occlusion / masking in registers, the encoding -- and the level's C plane row
            uint8_t* ob = next_obs + at * OBS_SLOT;
            uint8_t* pl = s_pl + lane * LANE_PL_PITCH;
            const uint2* tp = (const uint2*)(tmpl + c.rec_bytes);
#pragma unroll
            for (int k = 0; k < 8; ++k) *(uint2*)(pl + 8 * k) = tp[k];
            for (int o = 0; o < g.nobj; ++o) {
                const uint32_t w = g.obj(o);
                pl[8 * GenL<LaneMemDev>::o_y(w) + GenL<LaneMemDev>::o_x(w)] = (uint8_t)GenL<LaneMemDev>::o_app(w);
            }
            uint32_t wl[VIEW], wh[VIEW], cp[13];
            int fe2;
            window_rows_cpl(pl, c.H, g.ax, g.ay, g.adir, wl, wh);
            view_rows_perm(wl, wh, g.adir, (uint32_t)E_EMPTY, -1, cp, fe2);
            encode_cells_to(cp, StorePacker{(uint32_t*)ob});
            uint2* row = (uint2*)(ob + CPL_OFF);
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = *(const uint2*)(pl + 8 * k);
            uint32_t* ids = (uint32_t*)(ob + CPL_OFF + CPL_PLANE);
            for (int q = 0; q < CPL_MAX_IDS; q += 4) {
                uint32_t v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    uint32_t id = 0xFFu;
                    if (q + b < g.nobj) { const uint32_t w = g.obj(q + b); id = (uint32_t)(GenL<LaneMemDev>::o_y(w) << 3 | GenL<LaneMemDev>::o_x(w)); }
                    v |= id << (8 * b);
                }
                ids[q >> 2] = v;
            }
        }
        {
            Hot h;
            h.ax = (uint8_t)g.ax; h.ay = (uint8_t)g.ay; h.dir = (uint8_t)g.adir; h.carry = NONE8;
            h.step = 0; h.max_steps = (uint16_t)g.max_steps();
            h.pre4 = 0xFFFFFFFFu;
            h.vstate = 0; h.frozen = 0;
            if (gave_up) {
                h.frozen = 2;
                atomicAdd(gen_failures, 1ull);
            }
            h.last_locked = last_locked < 0 ? NONE8 : (uint8_t)last_locked;
            h.slot = 0;
            next_hots[at] = h;
        }
        slot = slot + 1 == depth ? 0 : slot + 1;
        attempts = 0;
        if (++done_levels == cnt) {
            mtis[env] = mem.position();
            mtpar[env] = (uint8_t)mem.par;
            pending[env] = 0;
            have = false;
        }
    }
}

// host side of the launch (bbai_engine.hip launch_pregen_lane fills the arguments)
void bbai::bbai_lane_launch(const LaneLaunch& a) {
    const dim3 g(a.blocks), b(64);
    const size_t lds = (size_t)a.lane_words * 64 * 4 + std::max<size_t>((MT_N + MT_CH) * 4, a.next_obs ? 64 * LANE_PL_PITCH : 0);
#define LANE_LAUNCH(KK, OO) hipLaunchKernelGGL((k_pregen_lane<KK, OO>), g, b, lds, a.stream, a.cfg, a.n, a.next_rec, a.next_hot, a.mt, a.mtt, a.mtpar, a.mti, \
                                              a.gen_list, a.gen_count, a.depth, a.pending, a.first_slot, a.fails, a.next_obs, a.tmpl, a.lane_words)
    if (a.cfg.kind == K_LEVELGEN) { if (a.next_obs) LANE_LAUNCH(K_LEVELGEN, true); else LANE_LAUNCH(K_LEVELGEN, false); }
    else { if (a.next_obs) LANE_LAUNCH(K_GOTO, true); else LANE_LAUNCH(K_GOTO, false); }
#undef LANE_LAUNCH
}
