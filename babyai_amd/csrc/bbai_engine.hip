// bbai_engine.hip -- HIP kernels + C ABI of the batched BabyAI engine (gfx950 / MI355X).
//
// Kernels (all integer / byte work, HBM- and latency-bound; no MFMA by design):
//   k_step<VP, FUSE>  lane = env, one wave per block.  Coalesced SoA loads of the 16-byte hot state, action, stale set and verifier program;
//                  per-lane transition + verifier on the env's record; the 7x7 window is fetched as 7 rows x 3 dwords (one 128-byte line
//                  of the window plane, VP) and rotated, occluded and masked in REGISTERS (bbai_view.hpp: byte permutes, SWAR opacity,
//                  dot-product row masks); the 147-byte encodings of the block's 64 envs are staged in LDS at the output pitch and leave as one
//                  contiguous 16-byte-per-lane span.  Finished envs: FUSE 1 -- the stepping wave consumes their look-ahead slots itself
//                  (consume_env); FUSE 3 (in-place layout) -- every finished lane moves its own env on to its next ring slot
//                  (advance_load / advance_finish); FUSE 0 -- compacted into a reset list for k_consume.  The fused paths leave NO
//                  returning atomic and no list behind: one fire-and-forget add to a sharded total per wave, per-env bytes for the refill.
//   k_pregen<F, G, OBS>  (F = level family) one env per group of G lanes: the NEXT levels of an env's MT19937 stream, working set in LDS
//                  (bbai_gen.hpp), into the env's look-ahead ring (OBS: + the level's first observation).  step() draws no randomness, so an
//                  env's level sequence is a pure function of its seed: generation runs ahead of need on a second HIP stream, one launch per
//                  window of B consume-ticks over the list k_compact builds from the window's `pending` bytes.
//   k_compact / k_mark / k_gate   the windows' turnover: the refill's work list (look-ahead stream), the refill's completion count, and
//                  the step stream's wait for "every env is sure to keep a window's worth of ready levels" (see NWIN below).
//   k_consume      wave = env over the reset list (unfused steps, reset()): look-ahead slot -> live state, SoA verifier view, first observation.
//   k_tokens       lane = env: mission text as fixed-vocabulary token ids of the envs that started a new episode.
//   k_render_q / k_render   RGBImgPartialObsWrapper as a pure tile-atlas gather: atlas + per-cell tile ids in LDS, 16 bytes per lane per
//                  store, a wave writes 1 KiB of contiguous pixels; persistent blocks fed by ONE ticket counter from 262 144 envs
//                  (k_render_q), one-shot (512, 2) blocks below.
//   k_bot<W>       lane = env: one decision of the reference's GOFAI expert (babyai/bot.py) per env, W = occupancy target
//                  (bbai_bot.hpp); only launched by bbai_bot_act / bbai_bot_rollout.
//
// Reference semantics: see bbai_step.hpp / bbai_gen.hpp / bbai_bot.hpp headers for file:line citations.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#if defined(BBAI_BOT_PROF)
__device__ unsigned long long g_bot_prof[32];        // experiment builds: phase timers of the expert (bbai_bot.hpp)
#endif
#include "../../include/bbai.h"
#include "bbai_types.hpp"
#include "bbai_kernels.hpp"
#include "bbai_gen.hpp"
#include "bbai_genl.hpp"
#include "bbai_step.hpp"
#include "bbai_view.hpp"
#include "bbai_bot.hpp"
#include "bbai_seed.hpp"

using namespace bbai;

static_assert(sizeof(bbai_level_cfg) == sizeof(LevelCfg), "C ABI cfg mirrors bbai::LevelCfg");

static thread_local char g_err[512] = "";
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            snprintf(g_err, sizeof(g_err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,       \
                     hipGetErrorString(e_));                                                  \
            return BBAI_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

#define ARG_FAIL(msg)                                                                        \
    do {                                                                                      \
        snprintf(g_err, sizeof(g_err), "%s: %s", __func__, msg);                              \
        return BBAI_ERR_ARG;                                                                  \
    } while (0)

// Entry points run on the handle's device but must leave the caller's current device alone (torch tracks the
// HIP current device of the thread; a library that changes it redirects the caller's later allocations).
struct DeviceGuard {
    int prev = -1, err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define ON_DEVICE(dev)                                                                        \
    DeviceGuard guard_(dev);                                                                  \
    HIP_TRY((hipError_t)guard_.err)

constexpr int PROF_RING = 256;          // event pairs per profiled kernel (bbai_profile)
constexpr int MAX_SIDES = 8;            // look-ahead streams a window's refill can be split over
#ifndef LOOKAHEAD_STREAMS_DEFAULT
#define LOOKAHEAD_STREAMS_DEFAULT 1
#endif
constexpr int MAX_PERIOD = 96;          // refill period B (ticks per look-ahead refill); ring depth D = 2B (+ 1 in place): slot numbers and per-window counts stay bytes
constexpr int NWIN = 34;                // window buffers (see "the windows' bookkeeping" below): at most 33 refills outstanding, whatever B

struct bbai_env {
    LevelCfg cfg;
    int64_t n;
    int device;
    uint8_t* rec;         // [n][rec_bytes]
    Hot* hot;             // [n]
    uint64_t* stale;      // [n]
    uint32_t* mt;         // [n][624]
    int32_t* mti;         // [n]
    uint32_t* vhead;      // [n]     verifier program head (SoA, see VProg)
    uint64_t* vset;       // [8][n]  obj_set bitmasks, k = 2*leaf + slot
    // Look-ahead ring of depth D = 2B: every env owns D pre-generated levels; a finished env consumes slot hot.slot.
    // Consume-ticks are grouped into windows of B ticks; the slots consumed during window w are refilled by ONE
    // k_pregen launch at the end of the window (so the generator's slowest level is paid once per B ticks), and
    // that refill only has to land before window w+2 starts (worst case an env consumes one slot per tick).
    int depth, period;    // D, B
    uint8_t* next_rec;    // [D][n][rec_bytes]
    Hot* next_hot;        // [D][n]
    uint8_t* next_obs;    // [D][n][OBS_SLOT] in-place layout only: the first observation of every look-ahead level, written by the generator
    uint8_t* pending;     // [NWIN][n]  per window buffer: slots consumed by env in the window (0 = not in the window)
    uint8_t* first_slot;  // [NWIN][n]  first slot the env freed in the window
    uint32_t* win_meta;   // [NWIN][META_U32] one 128-byte line per window buffer (k_gate; the consume paths' atomicMax)
    unsigned long long* totals;   // [SHARDS][SHARD_U64] resets so far, sharded over cache lines (sum = bbai_reset_count)
    unsigned long long* flow;     // [FLOW_WORDS] refilled windows, gate time-outs, generator give-ups
    int32_t* gen_list;    // [SHARDS][gen_sublist_cap(n)] the refill's work list (look-ahead stream only): k_compact -> k_pregen
    uint32_t* gen_count;  // [SHARDS][GEN_COUNT_U32] its sub-list lengths
    int32_t* reset_list;  // [n]     unfused consume only: envs finished by the current step (k_step<.., 0> -> k_consume / k_tokens)
    uint8_t* reset_slot;  // [n]     ... and the look-ahead slot each of them consumes (spares k_consume one dependent round trip)
    uint32_t* counters;   // [2][16] [p][0] = reset list length; ping-pong by step parity so that k_consume can zero the
                          //         other one for the next step (no memset launch on the step path)
    int step_parity;
    bool next_counter_clean;
    uint8_t* tokens;      // optional caller-owned [n][72] mission token buffer kept current on resets
    hipStream_t side;     // look-ahead generation stream (= sides[0]: bbai_seed's first fill, the probe)
    // Round 6: a window's refill is split over n_sides look-ahead streams, each with a CONTIGUOUS range of 64-env blocks and a work list of its own.
    // An env's levels are generated in stream order on ITS stream (its MT19937 state never has two writers); the streams run side by side, so
    // refill w + 1 of one range starts while refill w of another is still at its slowest level.  k_mark runs on a stream of its own behind an
    // event of every part (`refilled` stays "every env of windows < r has its levels").
    hipStream_t sides[MAX_SIDES];
    hipStream_t mark_stream;
    hipEvent_t ev_part[MAX_SIDES];
    int side_prio;
    int n_sides;          // BBAI_LOOKAHEAD_STREAMS / option "lookahead_streams"
    int32_t* gen_lists[MAX_SIDES];      // sides >= 1 (side 0: gen_list / gen_count below)
    uint32_t* gen_counts[MAX_SIDES];
    hipStream_t split;    // bbai_step_render: the second half-batch's k_step runs here, under the first half's render
    hipEvent_t ev_split0, ev_splitB;
    int step_render_split;   // BBAI_STEP_RENDER_SPLIT / option "step_render_split": 1 = split (from STEP_RENDER_SPLIT_MIN envs), 0 = never, -1 = default
    hipEvent_t ev_consumed;
    hipEvent_t ev_refill[NWIN];     // recorded behind every window's refill; only waited for in strict mode (below) -- k_gate reads flow[FLOW_REFILLED]
    int gate_strict;      // BBAI_GATE_STRICT / option "gate_strict": 1 = rounds 1-4's rule as well: the stream waits (an event) for the refill of window
                          // x - 2 before window x starts, so k_gate never has to wait (A/B runs; a fallback should a profiler serialise the two streams)
    // The relaxed gate is a device-side spin on a value the look-ahead stream's k_mark stores: it relies on kernels of the two streams making
    // progress CONCURRENTLY, which HIP does not promise (streams can share a hardware queue; a profiler can serialise launches).  So:
    //   * the first window a caller's stream opens PROBES it (probe_stream: a waiting kernel on the caller's stream, the releasing one on the
    //     look-ahead stream behind it); a stream that fails runs under the strict rule (event waits, k_gate never has to wait) from then on;
    //   * a gate that gives up all the same sets a sticky word in pinned host memory, which every entry point reads before it enqueues anything:
    //     the handle refuses to step on (BBAI_ERR_STATE) instead of consuming slots that were never refilled.
    volatile uint32_t* host_flags;    // pinned, mapped: [0] a window gate timed out (sticky)  [1] the last probe's verdict (1 concurrent, 2 not)
    uint32_t* host_flags_dev;         // the device's address of the same words
    hipStream_t probed[8];            // caller streams already probed ...
    bool probed_ok[8];                // ... and what the probe said
    int n_probed;
    int gate_probe;                   // BBAI_GATE_PROBE (default 1): 0 = trust the streams (no probe), 2 = treat every probe as failed (tests)
    int gate_forced;                  // the current caller stream failed the probe: strict rule
    hipStream_t last_stream;   // the caller's stream of the previous call (compared, never used): a handle follows ONE stream
    bool have_stream;          // at a time; a call on another stream waits for ev_switch = end of the previous call
    hipEvent_t ev_switch;      // (enter_call / leave_call)
    bool call_events;
    int render_tpb;       // BBAI_RENDER_TPB: 256 / 512 / 1024 threads per render block; anything else = by batch size
    int render_group;     // BBAI_RENDER_GROUP: 2, 4 or 8 envs per one-shot render block; anything else = by batch size (bbai_render)
    int pregen_cap;       // BBAI_PREGEN_BLOCKS: upper bound on look-ahead lane groups per launch (experiments)
    int pregen_per_group; // BBAI_PREGEN_PER_GROUP / option "pregen_per_group": single-room levels: list entries per working lane group of a refill (default 32 = one per tick of the longest window)
    int pregen_min;       // BBAI_PREGEN_MIN / option "pregen_min": single-room levels: lane groups that work on a refill at least (k_pregen: entries / 32 otherwise); 0 = the whole grid, as mazes always get
    int pregen_group;     // BBAI_PREGEN_GROUP: lanes per env in k_pregen: 32 (default: two envs per wave), 16 or 64
    // the lane = level generator (bbai_genl.hpp, k_pregen_lane): every LevelGen parameterisation and the single-instruction levels without a lock-first prologue
    unsigned long long* tap_mask; uint32_t* tap_rank0; int32_t* tap_perm; int64_t* tap_ids; int64_t tap_count;     // bbai_step_tap_set: the envs bbai_step_tapped logs
    uint32_t* mtt;        // [n][2][MT_N] tempered outputs of the latest and the previous MT19937 generation of every env (NULL: kind not covered)
    uint8_t* mtpar;       // [n] which half holds the latest generation
    uint8_t* lane_tmpl;   // the kind's record template + C plane (lane_build_template)
    int lane_words;       // LDS words per lane (lane_layout)
    int pregen_lane;      // BBAI_PREGEN_LANE / option "pregen_lane": 1 (default where covered) = k_pregen_lane generates, 0 = the lane-group kernel k_pregen
    int lane_blocks;      // BBAI_LANE_BLOCKS / option "lane_blocks": upper bound on its waves per launch
    int step_prio;        // BBAI_STEP_PRIO: s_setprio level of the step-path kernels' waves (they share CUs with k_pregen)
    // optional per-kernel timing (bbai_profile): HIP event pairs on the launch stream around k_step / k_consume / k_render
    bool prof_on;
    struct ProfSlot { hipEvent_t a, b; bool used; } prof[3][PROF_RING];
    int prof_pos[3];
    double prof_ms[3];
    int64_t prof_n[3];
    int64_t prof_step_ticks;   // steps the bracketed k_step launches took (a bbai_rollout launch takes up to a window's worth): option "profile_step_ticks"
    int rollout_multi;    // BBAI_ROLLOUT_MULTI / option "rollout_multi": bbai_rollout steps up to a look-ahead window per k_step launch (1, default) or one step per launch (0)
    int64_t tick;         // number of consume_and_refill calls so far
    uint8_t* vplane;      // [n][v_bytes] window plane (bbai_types.hpp): one 128-byte line per window-origin class; BBAI_VPLANE=0: none
    uint16_t* fcache;     // [n] appearance of the front cell (low byte) and of the carried object (high byte) after the last step
    uint8_t* cplane;      // [n][cpl_bytes] C plane rows of the small single rooms (bbai_types.hpp cpl_ok; in-place layout only); BBAI_CPLANE=0: none
    uint8_t* lsm;         // [n] done-action verifier mode only (BABYAI_DONE_ACTIONS / bbai_set_done_actions): bit k = leaf k's
                          //     lastStepMatch (babyai/levels/verifier.py:213-230); NULL = the normal mode
    unsigned int* render_tickets;   // [64][64] ticket counters of k_render_q + its departure counter; zero between launches (the kernel leaves them so)
    int render_queue;     // BBAI_RENDER_QUEUE / option "render_queue": -1 = by batch size (default), 0 = one-shot blocks only, m = queue shape m (render_launch)
    int render_queue_bpc; // option "render_queue_bpc": persistent render blocks per CU (0 = 1024 threads' worth: ONE 1024-thread block per CU --
                          // profiles/r04/render_queue_ab_1M_b.jsonl: k_render 1.50 ms against 1.61 with two)
    int render_queue_blocks;   // option "render_queue_blocks": their total number (0 = by render_queue_bpc)
    int render_pace;      // option "render_pace" (experiment): 1/16 ns of wall clock per render ticket (then with two counters), 0 (default) = as fast as the
                          // one counter serves them
    int n_cus;            // compute units of the device
    int done_action_enum; // option "done_action_enum": done-action mode only -- bbai_step's `done` actions count as the enum member (verifier.py:543-545)
    int consume_fused;    // BBAI_CONSUME_FUSED / option "consume_fused": -1 = by batch size, 0 = k_consume launch, 1 = inside k_step
    int inplace;          // BBAI_INPLACE: the in-place layout (live_slot below): an env's live record IS the look-ahead slot its episode was generated into
    uint8_t* atlas;       // [n_tiles][192]
    uint8_t* lut;         // [2][256]
    int n_tiles;
    bool seeded, live;
    uint8_t* bot_state;   // [n][bot_state_bytes(bot_stack)] the expert's per-env plan (bbai_bot_act; allocated on first use)
    int bot_stack;        // subgoal stack capacity per env (BBAI_BOT_STACK, default 48)
    uint16_t* bot_work;   // [bot_threads][BOT_WORK_WORDS] BFS scratch per resident thread
    uint32_t* bot_rows;   // [bot_threads][2 * MAX_W] row masks of the second (through-blockers) search
    int bot_group;        // lanes per env of the expert kernel: 0 = lane = env (k_bot), 16 = one 16-lane group per env (k_botg); BBAI_BOT_GROUP / option
    int bot_eager;        // BBAI_BOT_EAGER (default 1): expand the first search tree at the top of every decision
    int64_t bot_threads;
    uint64_t* bot_stats;  // [2] decisions that ended in a dead bot: by the reference's rules / by our capacity limits
};

// ------------------------------------------------------------------------------------------
// k_step
// ------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// ---- the windows' bookkeeping: no lists, no same-address atomics on the step path -----------------------------------------------
// Rounds 1-4 compacted the finished envs of every tick into a window list for the refill (one RETURNING atomic per stepping wave
// on ONE address: ~2 700 of them per step at 262 144 reset-heavy envs, served at ~11 ns each -- half of that k_step's time) and made
// the step stream wait, at the start of window w + 2, for the refill of window w (an env MIGHT finish on every tick).  Now:
//   * what a stepping wave leaves behind is one fire-and-forget add to a sharded total (SHARDS cache lines) and per-env bytes; the
//     refill's work list is built where it costs nothing: k_compact, on the look-ahead stream in front of k_pregen, turns the window's
//     `pending` bytes (n bytes per B ticks) into SHARDS dense sub-lists (one returning atomic per 64 envs that hold a finished one,
//     spread over SHARDS counters), which k_pregen walks as ONE list through a prefix of the sub-counts -- the same perfectly
//     balanced entry-per-group distribution as before.  (Letting the generator's groups scan the bytes themselves was measured
//     first -- lease r05a: a group then finds 0 to 4 envs where its neighbour finds one, a wave lives as long as its unluckiest
//     group, and the mazes' steps slowed by 10-30 % under the generator's idle lanes.)
//   * every window records M = the most often ONE env finished in it (1 unless short episodes repeat inside a window: the
//     rare atomicMax in the consume paths); an env's unrefilled slots are <= the sum of M over the windows whose refill has not
//     landed, so window x may start as soon as that sum is <= B (every env then still has B ready levels, and a window consumes
//     at most B) -- k_gate, one wave on the step stream at every window start, waits for exactly that instead of for "refill
//     w - 2 has landed".  A reset storm (a million maze envs timing out on the same tick: 37 ms of generator time) then runs
//     UNDER the following windows instead of stopping the step stream, as long as no env finishes B more times meanwhile.
//   * NWIN = 34 window buffers (pending / first_slot / meta): up to 33 refills can be outstanding (B + 1 of them for B <= 32).
// tests/test_ring_protocol.py models the rule (sufficient, and the ring depths stay tight).
__device__ __forceinline__ void count_resets(unsigned long long* __restrict__ totals, unsigned int k, unsigned int blk) {
    atomicAdd(&totals[(blk & (SHARDS - 1)) * SHARD_U64], (unsigned long long)k);        // (result unused: a no-return atomic)
}
// envs (= threads) per k_step block.  The kernel is bound by its chain of dependent memory round trips, not by bytes or
// instructions, and a block is what waits at its barriers for its slowest wave: ONE wave per block (64) measured against
// 128 / 256 in round 3 (profiles/r03/step_variants_ab.jsonl: BossLevel encoded 1 048 576 envs k_step 0.130 -> 0.124 -> 0.111 ms,
// PickupLoc 262 144 0.071 -> 0.061 -> 0.051, GoTo 131 072 0.0212 -> 0.0192 -> 0.0184, GoToLocal 65 536 0.0244 -> 0.0216 -> 0.0214).
// The in-wave consume (FUSE) RELIES on it: the LDS traffic of a block is ordered by the wave's program order alone.
constexpr int STEP_BLOCK = 64;
#ifndef BBAI_STEP_WAVES
#define BBAI_STEP_WAVES 1          // minimum waves per SIMD the register allocation of k_step has to allow (the compiler's own figures:
#endif                             // babyai_amd/kernel_resources.json, quoted in DESIGN.md section 4; the block's 9.4 KB of LDS stop at 17 blocks per CU)
// BBAI_PREFETCH_ID=1 (experiment): the id-plane entry of the front cell fetched WITH the window.  Measured slower everywhere
// (step_variants_ab.jsonl: BossLevel encoded 1M k_step 0.130 -> 0.148 ms, GoTo 131 072 0.021 -> 0.028): one more line per
// env-step costs more than the verifier's occasional extra round trip.  Off.
#ifndef BBAI_PREFETCH_ID
#define BBAI_PREFETCH_ID 0
#endif

// BBAI_VIEW_LDS=1 (A/B builds): rounds 2-4's view -- the window parked in LDS and read back cell by cell (view_cells / encode_view below).  The shipped
// path is bbai_view.hpp: the same view as byte permutes on packed registers (k_step: -~900 of ~2 600 vector instructions, -63 LDS operations per env-step).
#ifndef BBAI_VIEW_LDS
#define BBAI_VIEW_LDS 0
#endif
// Observation with the 7x7 window staged in LDS (rounds 2-4's k_step path).  49 scattered byte loads per lane keep the
// texture-address unit busy for most of k_step (tools/step_ab.py ablation), so the window is fetched in WORLD
// orientation as 7 rows x 3 aligned dwords, byte-aligned with v_alignbyte, parked in 56 dword-aligned bytes inside the
// lane's own LDS obs row (`scr`, bbai_step.hpp row_scratch), and read back in VIEW orientation (rotation = per-direction
// address arithmetic on ds_read_u8).  All of a lane's reads precede its writes and lanes only touch bytes of their own
// row, so no barrier is needed here.
// The window's rows come from `q` (first aligned dword of row 0), `rstride` dwords apart, `off` = byte offset of the
// window's first column inside that dword: the record's appearance plane (rstride = ES / 4) or the env's V-plane line
// (rstride = 4).  `ce` = appearance of what the agent carries (E_EMPTY: nothing).  `fe2` receives the appearance of the
// cell in front of the agent (view cell (3, 5)) for the verifier and the next step's transition.
// Two halves, so that the verifier (which only needs fe2) can run between them while nothing of the 37-dword encoding is
// live yet: view_cells fetches and rotates the window (cp = the 49 cells, vis = visibility rows), encode_view writes the
// encoding from them.
// window_fetch issues the loads (7 rows x 3 dwords: one dwordx3 each); view_cells consumes them.  k_step puts the rare
// object actions (pickup / drop / toggle: dependent record loads and stores) BETWEEN the two, so their memory round trips
// overlap the window's instead of preceding it.  Such an action changes exactly one cell of the window that was fetched
// before it ran -- the one in front of the agent, view cell (3, 5): `nfe` >= 0 is its new appearance, patched in LDS.
__device__ __forceinline__ void window_fetch(const uint32_t* __restrict__ q, int rstride, uint32_t* wd) {
#pragma unroll
    for (int r = 0; r < VIEW; ++r) { wd[3 * r] = q[r * rstride]; wd[3 * r + 1] = q[r * rstride + 1]; wd[3 * r + 2] = q[r * rstride + 2]; }
}
__device__ __forceinline__ void view_cells(const uint32_t* wd, int off, int dir, uint32_t ce, int nfe,
                                           uint8_t* __restrict__ scr /* this lane's 56 bytes of LDS scratch */, uint32_t* cp, uint32_t* vis, int& fe2) {
    uint32_t* win = (uint32_t*)scr;                          // 7 rows x 8 bytes, dword aligned
#pragma unroll
    for (int r = 0; r < VIEW; ++r) {
        win[2 * r] = __builtin_amdgcn_alignbyte(wd[3 * r + 1], wd[3 * r], off);
        win[2 * r + 1] = __builtin_amdgcn_alignbyte(wd[3 * r + 2], wd[3 * r + 1], off);
    }
    // view (vi, vj) -> window byte: dir3 (vj, vi), dir0 (vi, 6-vj), dir1 (6-vj, 6-vi), dir2 (6-vi, vj); row pitch 8
    const int k0 = dir == 0 ? 6 : dir == 1 ? 54 : dir == 2 ? 48 : 0;
    const int kvi = dir == 0 ? 8 : dir == 1 ? -1 : dir == 2 ? -8 : 1;
    const int kvj = dir == 0 ? -1 : dir == 1 ? -8 : dir == 2 ? 1 : 8;
    uint8_t* wb = scr + k0;
    if (nfe >= 0) wb[kvi * 3 + kvj * 5] = (uint8_t)nfe;      // (same lane: LDS operations of a lane stay in order)
#pragma unroll
    for (int k = 0; k < 13; ++k) cp[k] = 0;                      // the 49 cells, 4 per dword, view order [vi][vj]
    uint32_t opq[VIEW] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int vi = 0; vi < VIEW; ++vi)
#pragma unroll
        for (int vj = 0; vj < VIEW; ++vj) {
            const int idx = vi * VIEW + vj;
            const uint32_t e = wb[kvi * vi + kvj * vj];
            cp[idx >> 2] |= e << (8 * (idx & 3));
            opq[vj] |= (e_opaque((int)e) ? 1u : 0u) << vi;
        }
    process_vis_rows(opq, vis);
    fe2 = (int)((cp[(3 * VIEW + 5) >> 2] >> (8 * ((3 * VIEW + 5) & 3))) & 0xFFu);
    {   // the agent's own cell (3,6) shows what it carries
        constexpr int idx = 3 * VIEW + 6;
        cp[idx >> 2] = (cp[idx >> 2] & ~(0xFFu << (8 * (idx & 3)))) | (ce << (8 * (idx & 3)));
    }
}
// Four cells at a time: a dword of (visibility-masked) appearance bytes e0..e3 becomes the 12 encoding bytes
// t0 c0 s0 t1 | c1 s1 t2 c2 | s2 t3 c3 s3 (type = e & 7, colour = (e >> 3) & 7, state = e >> 6) with three field extractions on
// the whole dword and six byte permutes (v_perm_b32: selector bytes 0-3 pick from the second operand, 4-7 from the first,
// 0x0C is zero) -- 11 instructions per four cells instead of ~55 shifting every channel byte into place on its own.
__device__ __forceinline__ void encode_view(const uint32_t* cp, const uint32_t* vis, RowPacker o) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        // the cells of this dword that are visible: byte b <- bit (idx / 7) of vis[idx % 7], idx = 4k + b
        uint32_t m = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int idx = 4 * k + b;
            if (idx < VIEW * VIEW) m |= (uint32_t)__builtin_amdgcn_sbfe((int)vis[idx % VIEW], idx / VIEW, 1) & (0xFFu << (8 * b));   // v_bfe_i32: 0 / ~0
        }
        const uint32_t x = cp[k] & m;
        const uint32_t t = x & 0x07070707u, c = (x >> 3) & 0x07070707u, st = (x >> 6) & 0x03030303u;
        if (k < 12) {
            o.put(3 * k, __builtin_amdgcn_perm(__builtin_amdgcn_perm(t, c, 0x050C0004u), st, 0x07000504u));
            o.put(3 * k + 1, __builtin_amdgcn_perm(__builtin_amdgcn_perm(c, st, 0x060C0105u), t, 0x07020504u));
            o.put(3 * k + 2, __builtin_amdgcn_perm(__builtin_amdgcn_perm(st, t, 0x070C0306u), c, 0x07030504u));
        } else {
            o.put(36, (t & 0xFFu) | ((c & 0xFFu) << 8) | ((st & 0xFFu) << 16));   // cell 48: three bytes, the row's last dword
        }
    }
    o.finish();
}

// Wave-cooperative observation of ONE env (used where a wave owns an env: consume_env): lane l < 49 owns view cell
// (vi, vj) = (l % 7, l / 7); the opacity mask of the whole view is one ballot; every lane runs the 7-row
// visibility sweep on it and writes its own three bytes.  In two halves so that the caller can put other memory traffic
// between the cell load and its use: observe_fetch returns the lane's cell, observe_emit does the rest.
__device__ __forceinline__ int observe_fetch(const LevelCfg& c, const uint8_t* __restrict__ rec, const Hot& h, int lane) {
    const int vi = lane % VIEW, vj = lane / VIEW;
    int e = E_EMPTY;
    if (lane < VIEW * VIEW) {
        int x, y;
        view_to_world(h.ax, h.ay, h.dir, vi, vj, x, y);
        e = rec[e_index(c, x, y)];
    }
    return e;
}
__device__ __forceinline__ void observe_emit(const LevelCfg& c, const uint8_t* __restrict__ rec, const Hot& h, int e,
                                             uint8_t* __restrict__ dst, int lane) {
    const int vi = lane % VIEW, vj = lane / VIEW;
    const unsigned long long opaque = __ballot(lane < VIEW * VIEW && e_opaque(e));
    uint32_t opq[VIEW], vis[VIEW];
#pragma unroll
    for (int r = 0; r < VIEW; ++r) opq[r] = (uint32_t)(opaque >> (VIEW * r)) & 0x7Fu;
    process_vis_rows(opq, vis);
    if (lane < VIEW * VIEW) {
        if (vi == 3 && vj == 6) e = h.carry != NONE8 ? rec[c.off_app + h.carry] : (int)E_EMPTY;
        uint32_t row = 0;
#pragma unroll
        for (int r = 0; r < VIEW; ++r) row = (vj == r) ? vis[r] : row;
        const bool v = row >> vi & 1;
        uint8_t* o = dst + (vi * VIEW + vj) * 3;
        o[0] = v ? e_type(e) : 0; o[1] = v ? e_color(e) : 0; o[2] = v ? e_state(e) : 0;
    }
}

// V-plane helpers (bbai_types.hpp "window plane").  Patch one cell into every line that holds it.
__device__ __forceinline__ void v_patch(const LevelCfg& c, uint8_t* __restrict__ vrow, int x, int y, int val) {
    const int xm = x + MARGIN, ym = y + MARGIN, nxo = v_nxo(c), nyo = v_nyo(c);
    const int yo_lo = ym >= 6 ? (ym - 6) >> 1 : 0, yo_hi = (ym >> 1) < nyo - 1 ? (ym >> 1) : nyo - 1;
    for (int xo = (xm >> 3) - 1; xo <= (xm >> 3); ++xo) {
        if (xo < 0 || xo >= nxo) continue;
        for (int yo = yo_lo; yo <= yo_hi; ++yo) vrow[(yo * nxo + xo) * VLINE + (ym - 2 * yo) * 16 + (xm - 8 * xo)] = (uint8_t)val;
    }
}
// One 16-byte row segment of a V-plane line out of an appearance plane (`E`, row pitch ES); `sc` = plane index of a cell
// to show as empty (the start-carry object, which leaves the grid right after the first observation), or -1.
__device__ __forceinline__ u32x4 v_segment(const LevelCfg& c, const uint8_t* __restrict__ E, int line, int r, int sc) {
    const int nxo = v_nxo(c);
    const int yo = line / nxo, xo = line - yo * nxo;
    const int prow = 2 * yo + r, pcol = 8 * xo;
    const int base = prow * c.ES + pcol;
    uint32_t w[4];
    // branch-free: a dword outside the plane is read at offset 0 and replaced by zero, so the four loads (and those of the
    // caller's other segments) are in flight together -- as conditional loads each one was its own round trip
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const bool ok = prow < c.EH && pcol + 4 * d < c.ES;
        uint32_t v = *(const uint32_t*)(E + (ok ? base + 4 * d : 0));
        const int k = sc - (base + 4 * d);
        if (k >= 0 && k < 4) v = (v & ~(0xFFu << (8 * k))) | ((uint32_t)E_EMPTY << (8 * k));
        w[d] = ok ? v : 0u;
    }
    u32x4 out = {w[0], w[1], w[2], w[3]};
    return out;
}

// An env's C plane row (bbai_types.hpp) from its record by ONE wave: lane l = plane cell (l & 7, l >> 3); lanes < cpl_ids = the id bytes (an
// object stands on the grid iff the id plane holds it at its recorded position).  consume_env (reset()) and k_sync_cpl (imports).
__device__ __forceinline__ void cpl_build_wave(const LevelCfg& c, const uint8_t* __restrict__ rec, uint8_t* __restrict__ row, int lane) {
    const int x = lane & 7, y = lane >> 3;
    const int e = (x < c.W && y < c.H) ? (int)rec[e_index(c, x, y)] : (int)E_WALL;
    int v = 0xFF;
    if (lane < c.maxo) {
        const int ox = rec[c.off_pos + 2 * lane], oy = rec[c.off_pos + 2 * lane + 1];
        if (ox < c.W && oy < c.H && rec[c.off_I + i_index(c, ox, oy)] == lane + 2) v = oy << 3 | ox;
    }
    row[lane] = (uint8_t)e;
    if (lane < cpl_ids(c)) row[CPL_PLANE + lane] = (uint8_t)v;
}

// ---- the in-place layout (bbai_env::inplace, chosen at bbai_create) ------------------------------------------------------------
// Classic layout: every env has a live record of its own (rec[env]); a finished env's next level is COPIED out of its look-ahead
// slot (1.3 - 1.7 KB + the window plane), by a k_consume launch behind every step or by the stepping wave.  On reset-heavy small
// shards (single rooms: 2 % of the envs finish on every step) that second dependent launch is 40 % of a step (profiles/r04/NOTES.md
// section 2), and doing its work inside the stepping waves costs more than the launch.  In-place layout: the live record of an env
// IS the ring slot its episode was generated into -- the slot BEFORE hot.slot -- and a finished env just moves on to the next
// slot: nothing is copied, the stepping LANE loads the new pose and program (one round trip), emits the first observation with the
// step's own window pipeline and swaps the SoA state.  The ring is one slot deeper (2B + 1: the live one + the 2B look-ahead
// levels of the classic ring); the slot an episode leaves is the one the window's refill regenerates.  rec[] stays allocated as the
// staging area of export / import / checkpoints.  No window plane in this layout (the window comes out of the record's appearance
// plane: measures equal on the shards this is for).
// (ring_at -- the ENV-MAJOR look-ahead ring's addressing -- lives in bbai_kernels.hpp)
constexpr int OBS_BLOCK = 160;          // bytes of a next_obs slot that hold the first observation (147 used; sixteen-byte loads); OBS_SLOT / CPL_OFF: bbai_types.hpp
__device__ __forceinline__ int live_slot(int next_slot, int depth) { return (next_slot ? next_slot : depth) - 1; }
__device__ __forceinline__ uint8_t* live_rec(const LevelCfg& c, int64_t n, int64_t env, uint8_t* recs, uint8_t* ring, int depth, int next_slot) {
    (void)n;
    return ring ? ring + ring_at(live_slot(next_slot, depth), env, depth) * (int64_t)c.rec_bytes : recs + env * (int64_t)c.rec_bytes;
}

// look-ahead slot -> live state of ONE env by ONE wave (k_consume: wave = env over the reset list; k_step<.., FUSE>: the wave that
// stepped the env): coalesced record copy, SoA verifier view, first observation of the new episode (to `obs_dst`: the caller's
// image row, or the block's LDS row in k_step), window plane + front cache, window bookkeeping for the batched refill.
// `win_meta` = the meta line of the tick's window (its M is raised when an env finishes for the second time inside one window).
// The job is a handful of kilobytes per env, so what it costs is its chain of dependent memory round trips (a reset-heavy small
// shard pays it on every step): everything that depends on nothing but the slot is LOADED FIRST, in batches that are all in
// flight together (pose, program, the record's 16-byte vectors, the window plane's row segments), the one load that needs the
// new pose (the view cell) goes out as soon as the pose is there, and the stores follow.  Round 3's form (load - store pairs
// in loops) was ten round trips long.
__device__ __forceinline__ void consume_env(const LevelCfg& c, int64_t n, int64_t env, int slot, int lane, uint8_t* recs,
                                            Hot* __restrict__ hots, uint64_t* __restrict__ stales, uint8_t* next_recs /* in-place: the start-carry patch goes into the slot */,
                                            const Hot* __restrict__ next_hots, uint32_t* __restrict__ vheads, uint64_t* __restrict__ vsets,
                                            int depth, uint8_t* __restrict__ pending, uint8_t* __restrict__ first_slot,
                                            uint32_t* __restrict__ win_meta, uint8_t* __restrict__ obs_dst, uint8_t* __restrict__ dirs,
                                            uint8_t* __restrict__ vplane /* or NULL */,
                                            uint16_t* __restrict__ fcache, uint8_t* __restrict__ lsm_arr /* or NULL */,
                                            bool inplace = false /* the slot BECOMES the live record: no copy; the slot the episode leaves is what gets refilled */,
                                            uint8_t* __restrict__ cplane = nullptr /* in-place small rooms: the env's C plane row is rebuilt from the slot */) {
    const int nvec = c.rec_bytes >> 4;
    uint8_t* nrec = next_recs + ring_at(slot, env, depth) * (int64_t)c.rec_bytes;
    Hot h = next_hots[ring_at(slot, env, depth)];
    const Prog* p = (const Prog*)(nrec + c.off_prog);
    const int start_carry = p->start_carry;
    const uint64_t pset = lane < 8 ? p->set[lane >> 1][lane & 1] : 0ull;
    const uint32_t vh = vhead_pack(*p);
    const int pend = lane == 0 ? (int)pending[env] : 0;
    h.slot = (uint8_t)(slot + 1 == depth ? 0 : slot + 1);
    // the view cell of the new pose (the one load that needs the pose)
    const int e_view = observe_fetch(c, nrec, h, lane);
    uint32_t fe0 = nrec[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];       // (the front cell for the cache: with the view cells, not behind everything)
    // record: slot -> live copy
    if (!inplace) {
        const u32x4* src = (const u32x4*)nrec;
        u32x4* dst = (u32x4*)(recs + env * (int64_t)c.rec_bytes);
        constexpr int CPB = 2;
        for (int k0 = lane; k0 < nvec; k0 += 64 * CPB) {
            u32x4 buf[CPB];
#pragma unroll
            for (int j = 0; j < CPB; ++j) buf[j] = src[k0 + 64 * j < nvec ? k0 + 64 * j : nvec - 1];
            asm volatile("" : "+v"(buf[0]), "+v"(buf[1]));       // (both loads in flight before the first store: the scheduler otherwise pairs them load - store - load - store)
#pragma unroll
            for (int j = 0; j < CPB; ++j) if (k0 + 64 * j < nvec) dst[k0 + 64 * j] = buf[j];
        }
    }
    // the new episode's window plane, straight from the slot (L2 hits next to the copy above.  Parking the plane in LDS was
    // measured and dropped in round 3: any LDS at all makes k_consume's blocks queue behind the generator's waves for it)
    uint8_t* vrow = vplane ? vplane + env * (int64_t)v_bytes(c) : nullptr;
    if (vplane) {
        const int nseg = v_nxo(c) * v_nyo(c) * 8;
        constexpr int SGB = 4;
        for (int s0 = lane; s0 < nseg; s0 += 64 * SGB) {
            u32x4 seg[SGB];
#pragma unroll
            for (int j = 0; j < SGB; ++j) { const int sg = s0 + 64 * j < nseg ? s0 + 64 * j : nseg - 1; seg[j] = v_segment(c, nrec, sg >> 3, sg & 7, -1); }
#pragma unroll
            for (int j = 0; j < SGB; ++j) { const int sg = s0 + 64 * j; if (sg < nseg) *(u32x4*)(vrow + (sg >> 3) * VLINE + (sg & 7) * 16) = seg[j]; }
        }
    }
    uint8_t* crow = cplane ? cplane + env * (int64_t)cpl_bytes(c) : nullptr;
    if (crow) cpl_build_wave(c, nrec, crow, lane);
    // the verifier's SoA view of the new program
    if (lane < 8) vsets[(int64_t)lane * n + env] = pset;
    if (lane == 8) vheads[env] = vh;
    // first observation of the new episode, straight from the slot (identical bytes to the live copy)
    observe_emit(c, nrec, h, e_view, obs_dst, lane);
    if (lane == 0) {
        uint64_t stale0 = 0;
        uint32_t ce0 = E_EMPTY;
        // PutNext*Carrying: the first observation above still shows the object on the grid (the reference builds
        // it before handing the object to the agent, bonus_levels.py:821-829); now move it into the agent's hands.  In the
        // window plane (and the front cache) its cell is empty from the start.
        if (start_carry != NONE8) {
            const int sx = nrec[c.off_pos + 2 * start_carry], sy = nrec[c.off_pos + 2 * start_carry + 1];
            if (vplane) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the segment stores of every lane have landed; the patch goes over them
                v_patch(c, vrow, sx, sy, E_EMPTY);
            }
            if (crow) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the row bytes of the other lanes)
                crow[8 * sy + sx] = (uint8_t)E_EMPTY;
                crow[CPL_PLANE + start_carry] = 0xFF;
            }
            if (e_index(c, sx, sy) == e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))) fe0 = E_EMPTY;
            ce0 = nrec[c.off_app + start_carry];
            apply_start_carry(c, inplace ? nrec : recs + env * (int64_t)c.rec_bytes, h, stale0, start_carry);
        }
        if (vplane || crow) fcache[env] = (uint16_t)(fe0 | (ce0 << 8));
        hots[env] = h;
        stales[env] = stale0;
        if (lsm_arr) lsm_arr[env] = 0;                  // fresh instruction objects: lastStepMatch = False (verifier.py:213-214)
        dirs[env] = h.dir;
        // window bookkeeping for the batched refill: first consumption in this window registers the env
        if (pend == 0) first_slot[env] = (uint8_t)(inplace ? live_slot(slot, depth) : slot);
        else atomicMax(win_meta, (uint32_t)(pend + 1));       // (rare: the env finished before in this window)
        pending[env] = (uint8_t)(pend + 1);
    }
}

// In-place layout: a finished env moves on to its next look-ahead slot, done by the env's OWN lane inside k_step (all the finished
// lanes of a wave side by side: no per-env loop, no tail).  Everything it needs depends on the slot alone -- pose, program, window
// bookkeeping and the new episode's first observation, which the generator wrote next to the level (computing it here, with the
// step's own window pipeline, doubled the vector work of every wave that carries a finished env: measured, profiles/r04/
// inplace_own_lane_observation_ab.jsonl) -- so it is ONE round trip, and it is issued the moment the lane knows its episode is over
// (advance_load, right behind the step's own stores); advance_finish swaps the SoA state of the env and puts the observation into
// the lane's LDS row.  Nothing here waits for another wave: the window keeps no list (see NWIN above).
template <bool CP>
struct AdvanceRegs {
    u32x4 hv, tail /* Prog bytes 96..111: kind[4], root, n_a, n_b, strict, start_carry */, o[OBS_BLOCK / 16];
    u32x4 cp[CP ? (CPL_PLANE + CPL_MAX_IDS) / 16 : 1];      // the next level's C plane row
    uint64_t ps[8];
    uint32_t pend;
};
template <bool CP>
__device__ __forceinline__ void advance_load(const LevelCfg& c, int64_t env, int next /* hot.slot: the slot that becomes live */, int depth,
                                             const uint8_t* ring, const Hot* __restrict__ next_hots, const uint8_t* __restrict__ next_obs,
                                             const uint8_t* __restrict__ pending, AdvanceRegs<CP>& r) {
    const int64_t at = ring_at(next, env, depth);
    const uint8_t* nrec = ring + at * (int64_t)c.rec_bytes;
    r.hv = *(const u32x4*)(next_hots + at);
    const Prog* p = (const Prog*)(nrec + c.off_prog);
#pragma unroll
    for (int k = 0; k < 8; ++k) r.ps[k] = p->set[k >> 1][k & 1];
    r.tail = *(const u32x4*)((const uint8_t*)p + 96);
    r.pend = pending[env];
    const u32x4* ob = (const u32x4*)(next_obs + at * OBS_SLOT);
#pragma unroll
    for (int k = 0; k < OBS_BLOCK / 16; ++k) r.o[k] = ob[k];
    if constexpr (CP) {
        const u32x4* cr = (const u32x4*)(next_obs + at * OBS_SLOT + CPL_OFF);
#pragma unroll
        for (int k = 0; k < (CPL_PLANE + CPL_MAX_IDS) / 16; ++k) r.cp[k] = cr[k];      // (the slot holds 96 bytes whatever the level's id count)
    }
}
template <bool CP>
__device__ __forceinline__ void advance_finish(const LevelCfg& c, int64_t n, int64_t env, int lane, int next, int depth, uint8_t* ring, const AdvanceRegs<CP>& r,
                                               Hot* __restrict__ hots, uint64_t* __restrict__ stales, uint32_t* __restrict__ vheads, uint64_t* __restrict__ vsets,
                                               uint8_t* __restrict__ pending, uint8_t* __restrict__ first_slot, uint32_t* __restrict__ win_meta,
                                               uint8_t* __restrict__ s_rows, uint8_t* __restrict__ dirs, uint8_t* __restrict__ lsm_arr,
                                               uint8_t* __restrict__ cplane, uint16_t* __restrict__ fcache) {
    static_assert(sizeof(Prog) == 112 && offsetof(Prog, kind) == 96 && offsetof(Prog, start_carry) == 104, "Prog tail");
    Hot h;
    __builtin_memcpy(&h, &r.hv, sizeof(h));
    h.slot = (uint8_t)(next + 1 == depth ? 0 : next + 1);
    Prog pt;                                    // (only the tail fields are read below)
    __builtin_memcpy((uint8_t*)&pt + 96, &r.tail, 16);
    const uint32_t vh = vhead_pack(pt);
    const int start_carry = pt.start_carry;
    {
        RowPacker rp(s_rows, lane);
#pragma unroll
        for (int j = 0; j < 37; ++j) rp.put(j, r.o[j >> 2][j & 3]);
        rp.finish();
    }
    uint64_t stale0 = 0;
    uint32_t ce0 = E_EMPTY;
    if constexpr (CP) {
        u32x4* crow = (u32x4*)(cplane + env * (int64_t)cpl_bytes(c));
        const int nv = cpl_bytes(c) >> 4;
#pragma unroll
        for (int k = 0; k < (CPL_PLANE + CPL_MAX_IDS) / 16; ++k) if (k < nv) crow[k] = r.cp[k];
    }
    // PutNext*Carrying (consume_env): the first observation shows the object on the grid; now it is in the agent's hands
    if (start_carry != NONE8) {
        uint8_t* nrec = ring + ring_at(next, env, depth) * (int64_t)c.rec_bytes;
        if constexpr (CP) {
            const int sx = nrec[c.off_pos + 2 * start_carry], sy = nrec[c.off_pos + 2 * start_carry + 1];
            ce0 = nrec[c.off_app + start_carry];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the row above has landed; the patch goes over it
            uint8_t* crow = cplane + env * (int64_t)cpl_bytes(c);
            crow[8 * sy + sx] = (uint8_t)E_EMPTY;
            crow[CPL_PLANE + start_carry] = 0xFF;
        }
        apply_start_carry(c, nrec, h, stale0, start_carry);
    }
    if constexpr (CP) fcache[env] = (uint16_t)(E_EMPTY | (ce0 << 8));
#pragma unroll
    for (int k = 0; k < 8; ++k) vsets[(int64_t)k * n + env] = r.ps[k];
    vheads[env] = vh;
    hots[env] = h;
    stales[env] = stale0;
    if (lsm_arr) lsm_arr[env] = 0;
    dirs[env] = h.dir;
    if (r.pend == 0) first_slot[env] = (uint8_t)live_slot(next, depth);      // the slot this env's finished episode lived in: free for the refill
    else atomicMax(win_meta, r.pend + 1u);
    pending[env] = (uint8_t)(r.pend + 1);
}

// VP: the window comes from the env's V-plane line (ONE 128-byte line per step) and the transition's inputs -- the
// appearance of the front cell and of the carried object -- from the 2-byte cache the previous step left (`fcache`), so a
// plain move / turn touches no other record line; without VP both come out of the record (round 2's path: 2-3 lines for
// the window + the lines of the front cell's id and the carried object's appearance).
// FUSE: a wave whose envs finished consumes their look-ahead slots ITSELF (consume_env for every set bit of the wave's ballot,
// the new episode's first observation straight into the block's LDS rows), instead of listing them for a k_consume launch.
// `fuse` carries what k_consume's arguments carried.
// The step's own tap (bbai_step_tapped): the listed envs' outputs of THIS step into caller-owned log rows, written by the stepping lanes
// themselves -- what a bbai_tap_ids launch behind the step would copy, without the launch (k_tap is 3 us + a dependent-launch gap: a quarter
// of a 65 536-env step).  mask[block] bit l = env 64 block + l is listed; its log row = perm[rank0[block] + listed envs below it in the block].
struct TapArgs {
    const unsigned long long* mask; const uint32_t* rank0; const int32_t* perm;
    uint8_t* image_out; uint8_t* dir_out; double* rew_out; uint8_t* done_out;
    int64_t count;        // listed envs = log rows per tick (a launch of several ticks moves on by one row set per tick)
};
struct FuseArgs {
    uint8_t* next_recs; const Hot* next_hots; const uint8_t* next_obs; int depth;
    uint8_t* pending; uint8_t* first_slot; uint32_t* win_meta; unsigned long long* totals;
};
// step_body: ONE tick of a 64-env block (the whole of k_step; k_step_ticks calls it once per tick).  `s_obs`: the block's LDS rows.
template <bool VP, int FUSE /* 0: finished envs listed for k_consume; 1: consumed by the stepping wave (consume_env); 3: in-place layout (advance_load / advance_finish) */,
          bool CP = false /* in-place small single rooms: pose-independent C plane row instead of the record's planes (bbai_types.hpp) */>
__device__ __forceinline__ void step_body(const LevelCfg& c, int64_t n, uint8_t* __restrict__ recs,
                                                     Hot* __restrict__ hots, uint64_t* __restrict__ stales,
                                                     uint32_t* vheads, uint64_t* vsets /* read by every lane, WRITTEN for the envs the wave moves on (FUSE): no restrict */,
                                                     const uint8_t* __restrict__ actions, uint8_t* image /* read (frozen envs re-emit) AND written: no restrict */,
                                                     uint8_t* __restrict__ dirs, float* __restrict__ rewards,
                                                     double* __restrict__ rewards64, uint8_t* __restrict__ dones, int auto_reset,
                                                     int32_t* __restrict__ reset_list, uint8_t* __restrict__ reset_slot, uint32_t* __restrict__ counters,
                                                     int prio, uint8_t* __restrict__ vplane, uint16_t* __restrict__ fcache,
                                                     uint8_t* __restrict__ lsm_arr /* NULL, or the done-action mode's per-env bits */,
                                                     int enum_done /* done-action mode: this step's `done` actions are the enum member (bbai_step.hpp verify_side) */,
                                                     FuseArgs fuse, int64_t block0 /* first 64-env block of this launch (bbai_step_render steps the batch in two halves) */,
                                                     uint8_t* __restrict__ cplane /* CP: [n][cpl_bytes] */, const TapArgs& tap /* mask == NULL: none */,
                                                     uint8_t* const s_obs, const int lane /* threadIdx.x */, const int blk_x /* blockIdx.x */) {
    static_assert(!CP || (FUSE == 3 && !VP), "the C plane belongs to the in-place layout");
    uint8_t* const s_rows = s_obs + ROWS_FRONT;
    if (prio) __builtin_amdgcn_s_setprio(3);            // the look-ahead generator's waves share the CUs: issue ours first
    const int64_t env0 = ((int64_t)blk_x + block0) * STEP_BLOCK;
    const int64_t env = env0 + lane;
    const bool active = env < n;
    bool want_reset = false;
    bool frozen_copy = false; // CP: a frozen lane's row copy, deferred until every lane has read its parked plane (below)
    int my_slot = 0;
    AdvanceRegs<CP> adv;      // (in-place layout: the finished lanes' next-slot loads)
    if (active) {
        // everything the step needs from the SoA arrays in ONE memory round trip, before the frozen test (the loads the
        // branch would otherwise delay are a second round trip on every step's critical path)
        u32x4 hv = *(const u32x4*)(hots + env);
        uint64_t stale = stales[env];
        VProg vp; vp.bind(vheads[env], vsets + env, n);
        int action = actions[env];
        uint32_t fc = (VP || CP) ? (uint32_t)fcache[env] : 0u;
        Lsm lsm = {lsm_arr ? (uint32_t)lsm_arr[env] : 0u, lsm_arr != nullptr};
        // CP: the env's whole grid + object positions come with the SoA state -- nothing below depends on a second memory round trip
        u32x4 pv[4] = {}, iv[2] = {};
        uint8_t* crow = nullptr;
        if constexpr (CP) {
            crow = cplane + env * (int64_t)cpl_bytes(c);
            const u32x4* cr = (const u32x4*)crow;
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = cr[k];
            iv[0] = cr[4];
            const u32x4 none = {~0u, ~0u, ~0u, ~0u};
            iv[1] = none;
            if (cpl_ids(c) > 16) iv[1] = cr[5];
            asm volatile("" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(iv[0]), "+v"(iv[1]));
        }
        // (the empty asm pins the loaded values here: the compiler would otherwise sink the loads into the branch)
        asm volatile("" : "+v"(hv), "+v"(stale), "+v"(vp.head), "+v"(vp.set00), "+v"(action), "+v"(fc));
        Hot h;
        __builtin_memcpy(&h, &hv, sizeof(h));
        my_slot = h.slot;
        uint8_t* rec = FUSE == 3 ? fuse.next_recs + ring_at(live_slot(h.slot, fuse.depth), env, fuse.depth) * (int64_t)c.rec_bytes : recs + env * (int64_t)c.rec_bytes;
        if (!h.frozen) {
            double reward = 0.0;
            const EnvRef r = env_ref(c, rec, vp);
            uint8_t* vrow = VP ? vplane + env * (int64_t)v_bytes(c) : nullptr;
            int fe, ce;
            // CP: the lane's plane parked in LDS (8 rows x 8 bytes at a 72-byte lane pitch, inside the block's obs-row area: every lane's reads of
            // it precede, in the one wave's program order, every lane's row writes at the end of the step), for the per-lane row / cell addressing
            uint8_t* const pl = s_obs + lane * 72;
            if constexpr (CP) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    *(uint2*)(pl + 16 * k) = make_uint2(pv[k][0], pv[k][1]);
                    *(uint2*)(pl + 16 * k + 8) = make_uint2(pv[k][2], pv[k][3]);
                }
                fe = pl[8 * (h.ay + dir_dy(h.dir)) + h.ax + dir_dx(h.dir)];
                ce = (int)(fc >> 8);
            } else if (VP) {
                fe = (int)(fc & 0xFFu); ce = (int)(fc >> 8);
            } else {
                fe = r.E[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
                ce = h.carry != NONE8 ? r.app[h.carry] : (int)E_EMPTY;
            }
            if (action != A_RESET_ENV) apply_pose(h, action, fe);
            // the 7x7 window of the pose after the action.  Grid.slice extents (get_view_exts): its top-left world cell
            const int dir = h.dir;
            const int txm = h.ax + MARGIN + (dir == 0 ? 0 : dir == 2 ? -6 : -3);
            const int tym = h.ay + MARGIN + (dir == 1 ? 0 : dir == 3 ? -6 : -3);
            uint32_t wd[3 * VIEW];
            uint32_t wl[VIEW], wh[VIEW];
            if constexpr (CP) {
                window_rows_cpl(pl, c.H, h.ax, h.ay, dir, wl, wh);
            } else if (VP) {
                const uint8_t* line = vrow + ((tym >> 1) * v_nxo(c) + (txm >> 3)) * VLINE + (tym & 1) * 16 + (txm & 4);
                window_fetch((const uint32_t*)line, 4, wd);
            } else {
                window_fetch((const uint32_t*)(rec + ((tym * c.ES + txm) & ~3)), c.ES >> 2, wd);     // (ES is a multiple of 4)
            }
            // the id-plane entry of the front cell, fetched WITH the window: the verifier's common question ("is the object
            // in front of me one of the described ones") and the object actions then need no further memory round trip
            // (BBAI_PREFETCH_ID=0: read lazily, as rounds 1-2 did -- one more line per env-step, one round trip less)
            int idf = -1;
#if BBAI_PREFETCH_ID
            idf = r.I[i_index(c, h.ax + dir_dx(dir), h.ay + dir_dy(dir))];
#endif
            const int fpos = (h.ay + dir_dy(dir)) << 3 | (h.ax + dir_dx(dir));      // CP: the front cell in C plane coordinates
            if constexpr (CP) {
                const uint32_t idw[8] = {iv[0][0], iv[0][1], iv[0][2], iv[0][3], iv[1][0], iv[1][1], iv[1][2], iv[1][3]};
                idf = cid_lookup(idw, 8, fpos);                 // (what r.I would say about an object there; 0 = none)
            }
            // pickup / drop / toggle, while the window is on its way
            int nfe = -1;
            if (action != A_RESET_ENV) {
                int nid = -1;
                nfe = apply_objects(c, r, h, stale, action, fe, ce, idf, &nid);
                if (VP && nfe >= 0) v_patch(c, vrow, h.ax + dir_dx(dir), h.ay + dir_dy(dir), nfe);
                if constexpr (CP) {          // the env's C plane row follows the record: the cell, and who stands (or no longer stands) on it
                    if (nfe >= 0) crow[fpos] = (uint8_t)nfe;
                    if (nid >= 0) {
                        if (idf >= 2) crow[CPL_PLANE + idf - 2] = 0xFF;              // picked up / an opened box
                        if (nid >= 2) crow[CPL_PLANE + nid - 2] = (uint8_t)fpos;     // dropped / a box's content
                    }
                }
                if (idf >= 0 && nid >= 0) idf = nid;
            }
            int fe2;
            uint32_t cp[13];
#if BBAI_VIEW_LDS
            uint32_t vis[VIEW];
            view_cells(wd, txm & 3, dir, (uint32_t)ce, nfe, s_rows + row_scratch(lane), cp, vis, fe2);
#else
            if constexpr (CP) view_rows_perm(wl, wh, dir, (uint32_t)ce, nfe, cp, fe2);
            else view_cells_perm(wd, txm & 3, dir, (uint32_t)ce, nfe, cp, fe2);       // (bbai_view.hpp: rotation, occlusion and masking in registers)
#endif
            // "env.reset() for THIS env, now" (A_RESET_ENV, bbai_step.hpp): the episode ends with done = 1, reward = 0
            const bool done = action == A_RESET_ENV ? true : finish_step(c, r, h, stale, action, fe2, reward, lsm, idf, enum_done != 0);
            if (lsm_arr) lsm_arr[env] = (uint8_t)lsm.bits;
            if (done && !auto_reset) h.frozen = 1;
            want_reset = done && auto_reset;
            hots[env] = h;
            stales[env] = stale;
            if (VP || CP) fcache[env] = (uint16_t)((uint32_t)fe2 | ((uint32_t)ce << 8));
            rewards[env] = (float)reward;
            if (rewards64) rewards64[env] = reward;        // the reference's Python float, bit for bit (levelgen.py:59-61)
            dones[env] = done ? 1 : 0;
            dirs[env] = h.dir;
#if BBAI_VIEW_LDS
            encode_view(cp, vis, RowPacker(s_rows, lane));
#else
            encode_cells(cp, RowPacker(s_rows, lane));
#endif
        }
        // frozen envs keep re-emitting their last outputs: copy them through LDS unchanged
        else {
            if (h.frozen == 2 && auto_reset) {      // level the generator gave up on (last-resort guard): skip to the next one
                rewards[env] = 0.0f;
                if (rewards64) rewards64[env] = 0.0;
                dones[env] = 1;
                want_reset = true;
            }
            if constexpr (CP) {
                frozen_copy = true;
            } else {
                const uint8_t* src = image + env * OBS_BYTES;
                for (int b = 0; b < OBS_BYTES; ++b) s_rows[lane * OBS_BYTES + b] = src[b];
            }
        }
        if constexpr (FUSE == 3) { if (want_reset) advance_load<CP>(c, env, my_slot, fuse.depth, fuse.next_recs, fuse.next_hots, fuse.next_obs, fuse.pending, adv); }
    }
    if constexpr (CP) {
        // The stepping lanes parked their planes INSIDE the obs-row area (at a 72-byte pitch: over other lanes' rows).  Their own rows are written
        // after every plane read by the wave's program order; a frozen lane's row copy sits in the other arm of a branch, which the compiler may
        // emit FIRST -- the parked planes then went over rows already copied (caught by test_manyenvs_freeze).  So it waits here, behind a
        // convergent fence that no arm of that branch can cross.
        __builtin_amdgcn_wave_barrier();
        if (frozen_copy) {
            const uint8_t* src = image + env * OBS_BYTES;
            for (int b = 0; b < OBS_BYTES; ++b) s_rows[lane * OBS_BYTES + b] = src[b];
        }
    }
    // finished envs.  Unfused: compacted into the reset list for k_consume (one returning atomic per wave).  Fused / in-place: counted
    // (one fire-and-forget add to this block's shard of the total) and moved on by this wave itself.
    {
        unsigned long long bal = __ballot(want_reset);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            if constexpr (FUSE == 0) {
                uint32_t basei = 0;
                if (lane == leader) basei = atomicAdd(&counters[0], (uint32_t)__popcll(bal));
                basei = __shfl(basei, leader);
                if (want_reset) {
                    const uint32_t at = basei + __popcll(bal & ((1ull << lane) - 1));
                    reset_list[at] = (int32_t)env;
                    reset_slot[at] = (uint8_t)my_slot;
                }
            } else if constexpr (FUSE == 3) {
                // in-place layout: every finished lane moves its own env on (its stores to its own SoA entries stay in program order)
                if (lane == leader) count_resets(fuse.totals, (unsigned int)__popcll(bal), (unsigned int)blk_x);
                if (want_reset)
                    advance_finish<CP>(c, n, env, lane, my_slot, fuse.depth, fuse.next_recs, adv, hots, stales, vheads, vsets, fuse.pending, fuse.first_slot,
                                       fuse.win_meta, s_rows, dirs, lsm_arr, cplane, fcache);
            } else {
                // Everything this wave stored to the records, window planes and SoA entries of these envs must have landed
                // before other lanes overwrite them (a terminal pickup patches the record the consume is about to replace).
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == leader) count_resets(fuse.totals, (unsigned int)__popcll(bal), (unsigned int)blk_x);
                while (bal) {
                    const int src = __ffsll((long long)bal) - 1;
                    bal &= bal - 1;
                    const int slot = __shfl(my_slot, src);
                    // (the new episode's first observation goes over the finished env's row; LDS traffic of the one wave stays in program order)
                    consume_env(c, n, env0 + src, slot, lane, recs, hots, stales, fuse.next_recs, fuse.next_hots, vheads, vsets,
                                fuse.depth, fuse.pending, fuse.first_slot, fuse.win_meta, s_rows + src * OBS_BYTES, dirs,
                                VP ? vplane : nullptr, fcache, lsm_arr);
                }
            }
        }
    }
    __syncthreads();
    // the block's contiguous obs span leaves as it lies in LDS: 16 bytes per lane per store (64 x 147 B = 588 x 16 B; the
    // span of every full block starts 16-byte aligned in the output).  The last, partial block ends with a byte tail.
    const int64_t nb = n - env0 < STEP_BLOCK ? n - env0 : STEP_BLOCK;      // envs in this block
    const int total = (int)nb * OBS_BYTES;
    uint8_t* out = image + env0 * OBS_BYTES;
    {
        // (a caller's buffer that is not 16-byte aligned -- a row of a [T][n][147] history with odd n -- gets dwords or bytes)
        const int al = (int)((uintptr_t)out & 15);
        int done_bytes = 0;
        if (al == 0) {
            const int nvec = total >> 4;
            const u32x4* s128 = (const u32x4*)s_rows;
            for (int v = lane; v < nvec; v += STEP_BLOCK) ((u32x4*)out)[v] = s128[v];     // (non-temporal here: measured, no effect -- profiles/r03/NOTES.md)
            done_bytes = nvec << 4;
        } else if ((al & 3) == 0) {
            const int ndw = total >> 2;
            const uint32_t* s32 = (const uint32_t*)s_rows;
            for (int d = lane; d < ndw; d += STEP_BLOCK) ((uint32_t*)out)[d] = s32[d];
            done_bytes = ndw << 2;
        }
        for (int b = done_bytes + lane; b < total; b += STEP_BLOCK) out[b] = s_rows[b];
    }
    // the step's own tap: a listed env's row out of LDS (for an env that finished: already its new episode's first observation), its direction /
    // reward / done as this wave stored them (agent-scope loads: the direction of a consumed env was stored by another lane)
    if (tap.mask) {
        const int64_t blk = (int64_t)blk_x + block0;
        const unsigned long long tm = tap.mask[blk];
        if (active && (tm >> lane & 1ull)) {
            const int64_t row = (int64_t)tap.perm[tap.rank0[blk] + (uint32_t)__popcll(tm & ((1ull << lane) - 1ull))];
            uint8_t* o = tap.image_out + row * OBS_BYTES;
            const uint8_t* srow = s_rows + lane * OBS_BYTES;
            for (int b = 0; b < OBS_BYTES; ++b) o[b] = srow[b];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            tap.dir_out[row] = __hip_atomic_load(dirs + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tap.done_out[row] = __hip_atomic_load(dones + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tap.rew_out[row] = __hip_atomic_load(rewards64 + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
// What a step launch is given: the kernels' one argument (the kernarg segment IS this struct).
struct StepArgs {
    LevelCfg c; int64_t n; uint8_t* recs; Hot* hots; uint64_t* stales; uint32_t* vheads; uint64_t* vsets; const uint8_t* actions; uint8_t* image; uint8_t* dirs;
    float* rewards; double* rewards64; uint8_t* dones; int auto_reset; int32_t* reset_list; uint8_t* reset_slot; uint32_t* counters; int prio; uint8_t* vplane;
    uint16_t* fcache; uint8_t* lsm_arr; int enum_done; FuseArgs fuse; int64_t block0; uint8_t* cplane; TapArgs tap;
    int ticks;            // k_step_ticks: steps this launch takes; tick t reads actions + t n and logs into the tap rows t * tap.count further on
};
template <bool VP, int FUSE, bool CP = false>
__global__ __launch_bounds__(STEP_BLOCK, BBAI_STEP_WAVES) void k_step(StepArgs a) {
    // the block's observation rows at the OUTPUT pitch of 147 bytes (bbai_step.hpp RowPacker), 16 bytes of front padding
    __shared__ __attribute__((aligned(16))) uint8_t s_obs[ROWS_FRONT + STEP_BLOCK * OBS_BYTES + 16];
    step_body<VP, FUSE, CP>(a.c, a.n, a.recs, a.hots, a.stales, a.vheads, a.vsets, a.actions, a.image, a.dirs, a.rewards, a.rewards64, a.dones, a.auto_reset, a.reset_list,
                            a.reset_slot, a.counters, a.prio, a.vplane, a.fcache, a.lsm_arr, a.enum_done, a.fuse, a.block0, a.cplane, a.tap, s_obs, (int)threadIdx.x, (int)blockIdx.x);
}
// Several ticks in one launch (bbai_rollout, open-loop actions): an env's step touches only its own state, its block's LDS rows and -- for a
// finished env -- look-ahead slots the window gate in front of the launch has vouched for, so a block walks through its ticks on its own, with
// no launch boundary (and no dependent-launch gap: 4-5 us, a third of a 65 536-env step) in between.  Everything a tick reads of the previous one
// was stored by THIS wave: its vector-memory operations stay in program order.
// Every tick reads its arguments from the kernarg segment AGAIN, through a pointer the compiler cannot see through: left to itself it hoists
// what the ticks share (fifty LevelCfg words, thirty pointers and all that derives from them) out of the loop and keeps it in registers across
// the body -- 251 VGPRs against 93, two waves per SIMD against five.
typedef const StepArgs __attribute__((address_space(4))) * StepArgsPtr;
// ... and the register budget is the four waves per SIMD the one-tick kernels of the default paths have (115-117 VGPRs): the constants the
// loop optimiser still parks in registers in front of the loop are rematerialised or, a handful, spilled (2-7 VGPRs: kernel_resources.json).
template <bool VP, int FUSE, bool CP = false>
__global__ __launch_bounds__(STEP_BLOCK, 4) void k_step_ticks(StepArgs a_) {
    __shared__ __attribute__((aligned(16))) uint8_t s_obs[ROWS_FRONT + STEP_BLOCK * OBS_BYTES + 16];
    const int ticks = a_.ticks;
    for (int tick = 0; tick < ticks; ++tick) {
        StepArgsPtr ap = (StepArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
        int t = tick, lane = (int)threadIdx.x, blk_x = (int)blockIdx.x;        // (the lane and block arithmetic likewise: re-derived per tick)
        asm volatile("" : "+s"(ap), "+s"(t), "+v"(lane), "+s"(blk_x));
        const StepArgs a = *(const StepArgs*)ap;          // (InferAddressSpaces turns these back into scalar loads of the constant segment)
        TapArgs tap = a.tap;
        tap.image_out += (int64_t)t * tap.count * OBS_BYTES; tap.dir_out += (int64_t)t * tap.count; tap.rew_out += (int64_t)t * tap.count; tap.done_out += (int64_t)t * tap.count;
        step_body<VP, FUSE, CP>(a.c, a.n, a.recs, a.hots, a.stales, a.vheads, a.vsets, a.actions + (int64_t)t * a.n, a.image, a.dirs, a.rewards, a.rewards64, a.dones, a.auto_reset,
                                a.reset_list, a.reset_slot, a.counters, a.prio, a.vplane, a.fcache, a.lsm_arr, a.enum_done, a.fuse, a.block0, a.cplane, tap, s_obs, lane, blk_x);
        __syncthreads();        // (one wave per block: orders this tick's LDS reads before the next one's writes for the compiler)
    }
}

// ------------------------------------------------------------------------------------------
// k_pregen / k_consume : look-ahead level generation (one wavefront generates one env's levels)
// ------------------------------------------------------------------------------------------
// One env per group of G lanes, 64 / G envs per wavefront (bbai_gen.hpp "Execution model").  sync() orders the group's LDS
// accesses: it is reached under divergent control flow (the groups of a wave are in different places of the generator),
// so it is a wave-local fence, never a workgroup barrier -- the workgroup is one wave.
// (GroupCtx<G>: bbai_kernels.hpp)

// (Lane = level -- GroupCtx<1>: the same templates with a one-lane context, working set in per-lane global memory, MT19937 state advanced in
// place -- was built and measured in round 5 (profiles/r05/NOTES.md): 86-95 VGPRs, but every access to the working set becomes a global
// round trip: bulk fill 2 x SLOWER (PickupLoc 7.4 -> 14.2 ns per level, GoTo 35 -> 82), in the step loop 2-7 x.  Removed.)
// The look-ahead generator.  A workgroup is ONE wave carrying 64 / G envs; every group walks its share of the window's `pending`
// bytes on its own: fetch an env that has levels pending, load its MT19937 state into the
// group's LDS block, then one ATTEMPT of the generator's rejection loop per trip of the main loop (Gen::attempt) -- a
// group whose attempt was accepted writes the level out and goes on to its next level / env while its neighbours retry,
// so the wave only idles lanes inside an attempt, never across attempts.
// Work list: the window's finished envs as SHARDS dense sub-lists (k_compact); entry k of their concatenation is found through the
// prefix of the sub-counts (64 words in LDS, a six-step search per entry: once per level, i.e. per ~50-300 us of work); groups stride
// over the entries, so every group gets the same number of envs to within one.  `dense` (bbai_seed's first fill): every env, no list.
// Minimum waves per SIMD the generator's register allocation has to allow.  4 (<= 128 VGPRs) instead of the 3 the compiler
// settles on by itself (131-135 VGPRs at two envs per wave): PickupLoc 262 144 envs 0.0939 -> 0.0877 ms per step, the GoTo family
// already fits (profiles/r04/pregen_waves_per_simd_ab.jsonl).  The bonus family would spill (167 VGPRs) and four envs per wave
// need 200: those keep 2.
#ifndef BBAI_PREGEN_WAVES
#define BBAI_PREGEN_WAVES 4
#endif
template <int KIND, int G, bool OBS /* in-place layout: the level's first observation is written next to it */>
__global__ __launch_bounds__(64, (KIND == K_BONUS || G == 16) ? 2 : BBAI_PREGEN_WAVES) void k_pregen(LevelCfg c, int64_t n, uint8_t* __restrict__ next_recs,
                                                  Hot* __restrict__ next_hots, uint32_t* __restrict__ mts,
                                                  int32_t* __restrict__ mtis,
                                                  const int32_t* __restrict__ gen_list, const uint32_t* __restrict__ gen_count /* NULL: dense -- every env, the whole grid works */,
                                                  int depth,
                                                  uint8_t* __restrict__ pending, const uint8_t* __restrict__ first_slot,
                                                  unsigned long long* __restrict__ gen_failures, int min_groups, int per_group /* list entries a working group should get */,
                                                  uint8_t* __restrict__ next_obs /* in-place layout: [D][n][OBS_SLOT], else NULL */) {
    constexpr int NG = 64 / G;
    typedef GroupCtx<G> Ctx;
    const Ctx ctx;
    __shared__ GenWork ws[NG];
    __shared__ uint32_t s_mt[NG][MT_N + MT_CH];      // the env's MT19937 state + the generator's chunk of tempered outputs (bbai_gen.hpp MT_CH)
    GenWork& w = ws[threadIdx.x / G];
    const int lane = ctx.lane();
    // the refill list: prefix of the sub-list lengths (one word per lane, a wave scan, parked in LDS for the groups' searches)
    __shared__ uint32_t s_start[SHARDS + 1];
    int64_t count = n;
    const int64_t cap = gen_sublist_cap(n);
    if (gen_count) {
        uint32_t c = gen_count[threadIdx.x * GEN_COUNT_U32], incl = c;            // (SHARDS == 64 == the block)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += t; }
        s_start[threadIdx.x + 1] = incl;
        if (threadIdx.x == 0) s_start[0] = 0;
        __syncthreads();
        count = (int64_t)s_start[SHARDS];
    }
    // How many lane groups WORK on a window's refill (single rooms): the grid is sized for the worst case (every env finished on every
    // tick), the list usually holds a fraction of that, and every resident generator wave holds registers and LDS that the step kernels'
    // workgroups queue for -- but a refill that takes as long as a window paces the whole step stream (k_gate).  active = entries /
    // per_group, at least `min_groups`, at most the grid; surplus blocks leave at once.  Round 4 (k_step + k_consume, 25 + 15 us per
    // step at 65 536 envs) found entries / 32 and >= 2 048 groups best.  Round 5, with k_step at 16 us, no second launch and the
    // stream free to run ahead of the refills, the SAME sweep says: more groups, shorter refills (profiles/r05/pregen_sizing_sweep.jsonl,
    // ms per step): GoToLocal 65 536 envs 0.0319 at 2 048 groups, 0.0205 at 4 096, 0.0179 at 6 144, 0.0199 at 8 192, 0.0206 with the whole grid;
    // GoToLocal 32 768: 0.0181 / 0.0122 / 0.0127 at 2 048 / 4 096 / 8 192; PickupLoc 262 144: 0.0755 at entries / 32, 0.067 at / 16, 0.059-0.063
    // at / 12 ... / 4; GoToLocal 262 144: 0.0706 / 0.0644 / 0.0618 / 0.0655 at / 32, 16, 8, 4; PickupLoc 524 288: 0.134 / 0.121 / 0.124 / 0.132.
    // Shipped: entries / 12, at least 6 144 groups.
    int64_t stride = (int64_t)gridDim.x * NG;
    if (gen_count && min_groups > 0) {
        int64_t active = count / per_group;
        active = active < min_groups ? min_groups : active;
        active = (active + NG - 1) / NG * NG;                   // whole blocks: every group of a block that stays has its own residue
        stride = active < stride ? active : stride;
    }
    int64_t it = (int64_t)blockIdx.x * NG + threadIdx.x / G;
    if ((int64_t)blockIdx.x * NG >= stride) return;             // (whole blocks only: the groups of a wave stay together)
    // the group's current env
    bool have = false;
    int64_t env = 0;
    int cnt = 0, done_levels = 0, slot = 0, mti = 0, last_locked = -1, attempts = 0;
    bool dirty = false;           // the env's state words were regenerated (a twist) since they were loaded: only then do they go back
    for (;;) {
        if (!have) {
            while (it < count) {
                int64_t cand = it;
                if (gen_count) {
                    // entry `it` of the concatenated sub-lists: the sub-list j with s_start[j] <= it < s_start[j + 1]
                    int j = 0;
#pragma unroll
                    for (int o = SHARDS / 2; o; o >>= 1) if ((int64_t)s_start[j + o] <= it) j += o;
                    cand = (int64_t)gen_list[(int64_t)j * cap + (it - (int64_t)s_start[j])];
                }
                it += stride;
                const int pc = pending[cand];            // levels to generate for this env (consecutive ring slots)
                if (pc == 0) continue;                   // (dense: env was not consumed in this window)
                env = cand; cnt = pc; have = true;
                break;
            }
            if (have) {
                // the env's generator state: all of its loads in flight together (MT19937 words, position, first slot) -- as a
                // load - store loop this was five dependent round trips before the first draw
                const uint32_t* mt = mts + env * MT_N;
                constexpr int MTQ = (MT_N + G - 1) / G;
                uint32_t mtw[MTQ];
#pragma unroll
                for (int q = 0; q < MTQ; ++q) { const int k = lane + q * G; mtw[q] = mt[k < MT_N ? k : MT_N - 1]; }
                mti = mtis[env];
                slot = first_slot[env];
                ctx.sync();
#pragma unroll
                for (int q = 0; q < MTQ; ++q) { const int k = lane + q * G; if (k < MT_N) s_mt[threadIdx.x / G][k] = mtw[q]; }
                ctx.sync();
                const int prev = slot == 0 ? depth - 1 : slot - 1;          // holds the level generated just before
                last_locked = next_hots[ring_at(prev, env, depth)].last_locked;   // LevelGen.locked_room survives episodes
                last_locked = last_locked == NONE8 ? -1 : last_locked;
                done_levels = 0; attempts = 0; dirty = false;
            }
        }
        if (__ballot(have) == 0ull) break;               // every group of the wave has run out of work
        if (!have) continue;
        Gen<Ctx> g(ctx, c, w, s_mt[threadIdx.x / G], s_mt[threadIdx.x / G] + MT_N, mti, last_locked);
        bool ok = g.template attempt<KIND>();
        mti = g.mti;
        dirty |= g.twisted;
        last_locked = g.last_locked;
        // last-resort guard (Gen::MAX_ATTEMPTS): never seen; keeps an impossible level from hanging the device
        const bool gave_up = !ok && ++attempts >= Gen<Ctx>::MAX_ATTEMPTS;
        if (!ok && !gave_up) continue;
        const int max_steps = g.finish();
        // write-out: record planes, tables, program
        uint8_t* rec = next_recs + ring_at(slot, env, depth) * (int64_t)c.rec_bytes;
        {
            const uint32_t* src = (const uint32_t*)w.E;
            uint32_t* dst = (uint32_t*)rec;
            const int ndw = (c.ES * c.EH) >> 2;
            for (int k = lane; k < ndw; k += G) dst[k] = src[k];
        }
        {
            const int cells = c.W * c.H, ndw = (cells + 3) >> 2;          // off_I is a dword multiple, the plane is padded to one
            const uint32_t* src = (const uint32_t*)w.I;
            uint32_t* dst = (uint32_t*)(rec + c.off_I);
            for (int k = lane; k < ndw; k += G) {
                uint32_t v = src[k];
                if (4 * k + 4 > cells) v &= 0xFFFFFFFFu >> (8 * (4 * k + 4 - cells));      // bytes past the plane stay zero
                dst[k] = v;
            }
        }
        for (int k = lane; k < c.maxo; k += G) {
            bool used = k < g.nobj;
            rec[c.off_app + k] = used ? w.app[k] : 0;
            rec[c.off_pos + 2 * k] = used ? w.px[k] : 0;
            rec[c.off_pos + 2 * k + 1] = used ? w.py[k] : 0;
            rec[c.off_cont + k] = used ? w.cont[k] : NONE8;
        }
        {
            const uint32_t* src = (const uint32_t*)&w.prog;
            uint32_t* dst = (uint32_t*)(rec + c.off_prog);
            for (int k = lane; k < (int)(sizeof(Prog) / 4); k += G) dst[k] = src[k];
        }
        if constexpr (OBS) {
            // In-place layout: the level's first observation (gen_obs at the start pose: MiniGridEnv.reset), from the appearance plane
            // in LDS.  Lane l of the group takes view cells l, l + G, ...: cell = vi + 7 vj; the opacity mask of the view is the
            // group's share of a ballot per round; every lane runs the 7-row visibility sweep and writes its cells' three bytes
            // (the layout observe_emit writes: cell (vi, vj) at byte (7 vi + vj) * 3; the agent's own cell shows what it carries: nothing yet).
            uint8_t* ob = next_obs + ring_at(slot, env, depth) * OBS_SLOT;
            constexpr int R = (VIEW * VIEW + G - 1) / G;
            constexpr unsigned long long GM = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
            int ec[R];
            unsigned long long opaque = 0;
            ctx.sync();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int cell = r * G + lane;
                int e = E_EMPTY;
                if (cell < VIEW * VIEW) {
                    int x, y;
                    view_to_world(g.ax, g.ay, g.adir, cell % VIEW, cell / VIEW, x, y);
                    e = w.E[(y + MARGIN) * c.ES + (x + MARGIN)];
                }
                ec[r] = e;
                const unsigned long long bal = __ballot(cell < VIEW * VIEW && e_opaque(e));
                opaque |= ((bal >> ((int)threadIdx.x & ~(G - 1) & 63)) & GM) << (r * G);
            }
            uint32_t opq[VIEW], vis[VIEW];
#pragma unroll
            for (int r = 0; r < VIEW; ++r) opq[r] = (uint32_t)(opaque >> (VIEW * r)) & 0x7Fu;
            process_vis_rows(opq, vis);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int cell = r * G + lane;
                if (cell < VIEW * VIEW) {
                    const int vi = cell % VIEW, vj = cell / VIEW;
                    const int e = (vi == 3 && vj == 6) ? (int)E_EMPTY : ec[r];
                    uint32_t row = 0;
#pragma unroll
                    for (int q = 0; q < VIEW; ++q) row = (vj == q) ? vis[q] : row;
                    const bool v = row >> vi & 1;
                    uint8_t* o = ob + (vi * VIEW + vj) * 3;
                    o[0] = v ? e_type(e) : 0; o[1] = v ? e_color(e) : 0; o[2] = v ? e_state(e) : 0;
                }
            }
            // ... and, for the small single rooms, the level's C plane row (bbai_types.hpp): the grid at pitch 8 + where every object stands
            if (cpl_ok(c)) {
                uint8_t* row = ob + CPL_OFF;
                for (int d = lane; d < CPL_PLANE / 4; d += G) {
                    const int y = d >> 1, x0 = (d & 1) * 4;
                    uint32_t v = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int x = x0 + b;
                        const uint32_t e = (x < c.W && y < c.H) ? (uint32_t)w.E[(y + MARGIN) * c.ES + (x + MARGIN)] : (uint32_t)E_WALL;
                        v |= e << (8 * b);
                    }
                    ((uint32_t*)row)[d] = v;
                }
                for (int k = lane; k < CPL_MAX_IDS; k += G)
                    row[CPL_PLANE + k] = (k < g.nobj && w.px[k] != NONE8) ? (uint8_t)(w.py[k] << 3 | w.px[k]) : (uint8_t)0xFF;
            }
        }
        if (lane == 0) {
            Hot h;
            h.ax = (uint8_t)g.ax; h.ay = (uint8_t)g.ay; h.dir = (uint8_t)g.adir; h.carry = NONE8;
            h.step = 0; h.max_steps = (uint16_t)max_steps;
            h.pre4 = 0xFFFFFFFFu;
            h.vstate = 0; h.frozen = 0;
            if (gave_up) {
                h.frozen = 2;
                atomicAdd(gen_failures, 1ull);
            }
            h.last_locked = last_locked < 0 ? NONE8 : (uint8_t)last_locked;
            h.slot = 0;
            next_hots[ring_at(slot, env, depth)] = h;
        }
        slot = slot + 1 == depth ? 0 : slot + 1;
        attempts = 0;
        if (++done_levels == cnt) {                      // this env's levels are done: MT state back, buffer entry free
            // draws only advance the index: the 624 state words change at a twist alone (every 624 draws -- one single-room level in seven)
            if (dirty) {
                uint32_t* mt = mts + env * MT_N;
                ctx.sync();
                for (int k = lane; k < MT_N; k += G) mt[k] = s_mt[threadIdx.x / G][k];
            }
            if (lane == 0) {
                mtis[env] = mti;
                pending[env] = 0;                        // buffer entry is free for a later window
            }
            have = false;
        }
    }
}

// Derived / canonical forms of the lane generator's RNG state.
//   k_mt_sync:  after anything that wrote (mts, mtis) in the canonical form (imports, checkpoint loads, the lane-group generator): the latest
//               generation's tempered outputs into half 0, parity 0.
//   k_mt_canon: before anything that reads the canonical form (checkpoint saves, the lane-group generator): an env whose position lies in
//               the PREVIOUS generation gets that generation's raw words back (un-tempered from its half) and position + 624.
__global__ __launch_bounds__(256) void k_mt_sync(int64_t n, const uint32_t* __restrict__ mts, uint32_t* __restrict__ mtt, uint8_t* __restrict__ mtpar) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * MT_N) return;
    const int64_t env = i / MT_N;
    const int k = (int)(i - env * MT_N);
    mtt[env * (2 * MT_N) + k] = mt_temper(mts[i]);
    if (k == 0) mtpar[env] = 0;
}
__global__ __launch_bounds__(256) void k_mt_canon(int64_t n, uint32_t* __restrict__ mts, uint32_t* __restrict__ mtt, uint8_t* __restrict__ mtpar, int32_t* __restrict__ mtis) {
    // one wave per env (the position is read by every lane before lane 0 rewrites it: the wave runs in lockstep up to the barrier)
    const int64_t env = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (env >= n) return;
    const int pos = mtis[env];
    const int par = mtpar[env];
    __builtin_amdgcn_wave_barrier();
    if (pos >= 0) return;
    const uint32_t* prev = mtt + env * (2 * MT_N) + (par ^ 1) * MT_N;
    for (int k = lane; k < MT_N; k += 64) mts[env * MT_N + k] = mt_untemper(prev[k]);
    if (lane == 0) { mtis[env] = pos + MT_N; mtpar[env] = (uint8_t)(par ^ 1); }
}

// look-ahead slot -> live state for the envs that finished (or all, on reset()): one wave copies one record
__global__ __launch_bounds__(256) void k_consume(LevelCfg c, int64_t n, uint8_t* recs, Hot* __restrict__ hots,
                                                 uint64_t* __restrict__ stales, uint8_t* next_recs,
                                                 const Hot* __restrict__ next_hots, uint32_t* __restrict__ vheads,
                                                 uint64_t* __restrict__ vsets, const int32_t* __restrict__ reset_list,
                                                 const uint8_t* __restrict__ reset_slot, const uint32_t* __restrict__ counter, int all,
                                                 unsigned long long* __restrict__ totals, int depth,
                                                 uint8_t* __restrict__ pending, uint8_t* __restrict__ first_slot,
                                                 uint32_t* __restrict__ win_meta,
                                                 uint8_t* __restrict__ image, uint8_t* __restrict__ dirs,
                                                 uint32_t* __restrict__ other_counter, int prio,
                                                 uint8_t* __restrict__ vplane /* or NULL */, uint16_t* __restrict__ fcache,
                                                 uint8_t* __restrict__ lsm_arr /* or NULL */, int inplace, uint8_t* __restrict__ cplane /* or NULL */) {
    if (prio) __builtin_amdgcn_s_setprio(3);
    const int64_t count = all ? n : (int64_t)counter[0];
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t it = wave; it < count; it += nwaves) {
        const int64_t env = all ? it : (int64_t)reset_list[it];
        const int slot = all ? (int)hots[env].slot : (int)reset_slot[it];       // (k_step listed it next to the env: no round trip through the env's state)
        consume_env(c, n, env, slot, lane, recs, hots, stales, next_recs, next_hots, vheads, vsets, depth, pending, first_slot,
                    win_meta, image + env * OBS_BYTES, dirs, vplane, fcache, lsm_arr, inplace != 0, cplane);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&totals[0], (unsigned long long)count);
        other_counter[0] = 0;       // the next step's k_step appends to the other ping-pong counter from zero
    }
}

// ---- window turnover (NWIN above) ------------------------------------------------------------------------------------------------
// k_compact, look-ahead stream, in front of the window's k_pregen: the envs whose `pending` byte is set, as SHARDS dense sub-lists.  Wave
// w covers envs [64 w, 64 w + 64) and appends to sub-list w % SHARDS: one returning atomic per wave that found any, spread over SHARDS
// counters (1 048 576 envs, every one pending: 256 per counter) -- off the step path, a few microseconds per window.
__global__ __launch_bounds__(256) void k_compact(int64_t n, const uint8_t* __restrict__ pending, int32_t* __restrict__ gen_list, uint32_t* __restrict__ gen_count,
                                                 int64_t wave0, int64_t waves /* this look-ahead stream's 64-env blocks: [wave0, wave0 + waves) */) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = wave0 + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t env = wave * 64 + lane;
    const bool mine = wave < wave0 + waves && env < n && pending[env] != 0;
    const unsigned long long bal = __ballot(mine);
    if (!bal) return;
    const int j = (int)(wave % SHARDS);
    const int leader = __ffsll((long long)bal) - 1;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&gen_count[j * GEN_COUNT_U32], (uint32_t)__popcll(bal));
    base = __shfl(base, leader);
    if (mine) gen_list[(int64_t)j * gen_sublist_cap(n) + base + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)env;
}
// k_mark, look-ahead stream, behind the refill of window w: `refilled` = w + 1.  (A kernel of its own: the refill's stores are visible to
// whoever sees this value because that kernel has ENDED -- no fence inside the generator's waves.)
__global__ void k_mark(unsigned long long* __restrict__ flow, unsigned long long refilled) {
    __hip_atomic_store(&flow[FLOW_REFILLED], refilled, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// k_gate, step stream, in front of the first tick of window x (which uses buffer x % NWIN): waits until
//   (a) the windows r .. x - 1 whose refill has not landed (r = `refilled`) are at most NWIN - 1 (buffer x % NWIN is free again), and
//   (b) the sum of their M (meta[0]; 1 unless an env finished repeatedly inside one window) is <= B: every env then has at least
//       2B - B = B ready levels, and window x consumes at most B per env;
// then clears the meta line of window x.  With r = x - 1 (rounds 1-4 waited for exactly that) both hold trivially, so the wait ends at the
// latest when refill x - 2 lands; every refill it can wait for was enqueued before it.  One wave; polls with s_sleep.  A wait beyond
// ~10 s of the constant 100-MHz clock gives up (counted in flow[FLOW_GATE_TIMEOUTS], read back as option "gate_timeouts": the handle's
// results are void then -- it means a lost refill, never seen) instead of hanging the device.
__global__ __launch_bounds__(64) void k_gate(unsigned long long* __restrict__ flow, uint32_t* __restrict__ metas, unsigned long long x, int period,
                                              uint32_t* __restrict__ host_fault /* pinned host word: sticky, read by every entry point */) {
    const int lane = (int)threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long r = __hip_atomic_load(&flow[FLOW_REFILLED], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long open = x > r ? x - r : 0ull;          // windows r .. x - 1
        uint32_t m = 0;
        if ((unsigned long long)lane < open && open < (unsigned long long)NWIN) {
            m = metas[(size_t)((r + lane) % NWIN) * META_U32];
            m = m < 1u ? 1u : m;
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) m += __shfl_xor(m, o);
        if (open < (unsigned long long)NWIN && m <= (uint32_t)period) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 1000000000ull) {
            if (lane == 0) {
                atomicAdd(&flow[FLOW_GATE_TIMEOUTS], 1ull);
                __hip_atomic_store(host_fault, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (lane == 0) metas[(size_t)(x % NWIN) * META_U32] = 0;
}

// probe_stream's two kernels: the waiter (caller's stream) polls a flag for at most ~20 ms of the 100-MHz clock, the setter (look-ahead stream, enqueued
// BEHIND it) raises it.  Verdict into pinned host memory: 1 = the setter ran while the waiter was resident (the streams are concurrent), 2 = it did not.
__global__ void k_probe_wait(unsigned long long* __restrict__ flow, uint32_t* __restrict__ host_verdict) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t v = 2;
    for (;;) {
        if (__hip_atomic_load(&flow[FLOW_PROBE], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0ull) { v = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) break;
        __builtin_amdgcn_s_sleep(16);
    }
    __hip_atomic_store(host_verdict, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_probe_set(unsigned long long* __restrict__ flow, unsigned long long v) {
    __hip_atomic_store(&flow[FLOW_PROBE], v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// In-place layout: rec[] is the staging area of export / import / checkpoints.  dir 0: live slots -> rec[first ..], dir 1: rec[first ..] -> live
// slots; one wave per env.
__global__ __launch_bounds__(256) void k_live_copy(LevelCfg c, int64_t n, int64_t first, int64_t count, uint8_t* __restrict__ recs,
                                                   uint8_t* __restrict__ ring, const Hot* __restrict__ hots, int depth, int dir) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int nvec = c.rec_bytes >> 4;
    for (int64_t it = wave; it < count; it += nwaves) {
        const int64_t env = first + it;
        u32x4* stage = (u32x4*)(recs + env * (int64_t)c.rec_bytes);
        u32x4* live = (u32x4*)(ring + ring_at(live_slot(hots[env].slot, depth), env, depth) * (int64_t)c.rec_bytes);
        for (int k = lane; k < nvec; k += 64) { if (dir) live[k] = stage[k]; else stage[k] = live[k]; }
    }
}
// ... and an imported hot state keeps the env's place in its ring (hot.slot): the slot says where the live record IS
__global__ void k_import_hot(int64_t first, int64_t count, const Hot* __restrict__ staged, Hot* __restrict__ hots) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Hot h = staged[i];
    h.slot = hots[first + i].slot;
    hots[first + i] = h;
}

// rebuild the SoA verifier view from the records (after bbai_import_state)
__global__ void k_sync_prog(LevelCfg c, int64_t n, int64_t first, int64_t count, const uint8_t* __restrict__ recs,
                            uint32_t* __restrict__ vheads, uint64_t* __restrict__ vsets) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int64_t env = first + i;
    const Prog* p = (const Prog*)(recs + env * (int64_t)c.rec_bytes + c.off_prog);
    for (int k = 0; k < 8; ++k) vsets[(int64_t)k * n + env] = p->set[k >> 1][k & 1];
    vheads[env] = vhead_pack(*p);
}

// rebuild the window plane and the front-cell cache from the live records (after bbai_import_state / checkpoint_load):
// one wave per env
__global__ __launch_bounds__(256) void k_sync_view(LevelCfg c, int64_t first, int64_t count, const uint8_t* __restrict__ recs,
                                                   const Hot* __restrict__ hots, uint8_t* __restrict__ vplane, uint16_t* __restrict__ fcache) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int nseg = v_nxo(c) * v_nyo(c) * 8;
    for (int64_t it = wave; it < count; it += nwaves) {
        const int64_t env = first + it;
        const uint8_t* rec = recs + env * (int64_t)c.rec_bytes;
        uint8_t* vrow = vplane + env * (int64_t)v_bytes(c);
        for (int sg = lane; sg < nseg; sg += 64) *(u32x4*)(vrow + (sg >> 3) * VLINE + (sg & 7) * 16) = v_segment(c, rec, sg >> 3, sg & 7, -1);
        if (lane == 0) {
            const Hot h = hots[env];
            const uint32_t fe = rec[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
            const uint32_t ce = h.carry != NONE8 ? rec[c.off_app + h.carry] : (uint32_t)E_EMPTY;
            fcache[env] = (uint16_t)(fe | (ce << 8));
        }
    }
}

// ... and the C plane rows + the carried object's appearance of the small single rooms (in-place layout): one wave per env, from the staged records
__global__ __launch_bounds__(256) void k_sync_cpl(LevelCfg c, int64_t first, int64_t count, const uint8_t* __restrict__ recs,
                                                  const Hot* __restrict__ hots, uint8_t* __restrict__ cplane, uint16_t* __restrict__ fcache) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t it = wave; it < count; it += nwaves) {
        const int64_t env = first + it;
        const uint8_t* rec = recs + env * (int64_t)c.rec_bytes;
        cpl_build_wave(c, rec, cplane + env * (int64_t)cpl_bytes(c), lane);
        if (lane == 0) {
            const Hot h = hots[env];
            const uint32_t fe = rec[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
            const uint32_t ce = h.carry != NONE8 ? rec[c.off_app + h.carry] : (uint32_t)E_EMPTY;
            fcache[env] = (uint16_t)(fe | (ce << 8));
        }
    }
}

// The reference's expert for every env (babyai/bot.py Bot.replan): lane = env, grid-stride over the batch with one BFS
// scratch block per resident thread.  A new episode (step_count == 0) starts a fresh Bot.
template <int WAVES_PER_SIMD>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_bot(LevelCfg c, int64_t n, const uint8_t* __restrict__ recs, const uint8_t* __restrict__ ring /* in-place layout: the live records are ring slots; else NULL */,
                                            int depth, const Hot* __restrict__ hots,
                                            const uint64_t* __restrict__ stales, uint8_t* __restrict__ states, int stack_cap,
                                            uint16_t* __restrict__ works, uint32_t* __restrict__ slow_rows, int eager, const uint8_t* __restrict__ prev_actions,
                                            uint8_t* __restrict__ out, unsigned long long* __restrict__ stats,
                                            int dead_action /* what a bot that gave up emits: BOT_DEAD, or A_RESET_ENV in a rollout */,
                                            uint8_t* __restrict__ gave_up /* or NULL: [n] 1 where the bot gave up at this decision */) {
    // the searches' hot row masks (expandable / queued / seen), [row][lane] in LDS: every lane on its own bank
    extern __shared__ uint32_t s_rows[];              // [R_FAST][H][64] row masks, then the queue ring uint16 [BOT_RING][64]
    uint16_t* s_ring = (uint16_t*)(s_rows + R_FAST * c.H * 64);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    BotWork w;
    w.eager = eager;
    w.ring = s_ring + threadIdx.x; w.ring_stride = 64; w.ring_size = BOT_RING;
    w.rows_fast = s_rows + threadIdx.x; w.rstride_fast = 64; w.rows_h = c.H;
    w.rows_slow = slow_rows + tid * ((R_ALL - R_FAST) * MAX_W); w.rstride_slow = 1;
    // per-thread contiguous scratch: measured faster than lane-interleaving it (BossLevel 262144 envs 12.8 vs 17.6 ms per
    // decision batch) -- the lanes' searches diverge at once, so an interleaved line holds one useful 2-byte element
    w.cells = c.W * c.H;                                    // 64 cells (512 B of scratch) for an 8x8 room, 484 for a 3x3 maze
    w.base = works + tid * (int64_t)(4 * w.cells);
    w.stride = 1;
    for (int64_t i = tid; i < n; i += nthreads) {
        const Hot h = hots[i];
        if (h.frozen) { out[i] = A_DONE; if (gave_up) gave_up[i] = 0; continue; }
        BotState& st = *(BotState*)(states + i * (int64_t)bot_state_bytes(stack_cap));
        const bool first = h.step == 0 || st.next_step != h.step;       // (bot_decide applies the same rule)
        const int taken = (prev_actions && !first) ? prev_actions[i] : -1;
        const bool was_dead = !first && st.dead;
        const int a = bot_decide(c, live_rec(c, n, i, (uint8_t*)recs, (uint8_t*)ring, depth, h.slot), h, stales[i], st, stack_cap, w, first, taken);
        out[i] = (uint8_t)(a == BOT_DEAD ? dead_action : a);
        if (gave_up) gave_up[i] = a == BOT_DEAD ? 1 : 0;
        if (a == BOT_DEAD && !was_dead) atomicAdd(&stats[st.dead == DEAD_CAPACITY ? 1 : 0], 1ull);
    }
}

// BBAI_BOT_GROUP_BUILD=1 (experiment builds only; the shipped library has no k_botg): the expert as one 16-lane group per env.  Built, verified on
// the host emulation of lane groups (tests/test_hostsim_bot.py, which stays) and on the device, measured 1.9-2.4 x SLOWER than lane = env
// (profiles/r05/NOTES.md section 11); a 4-waves-per-SIMD build of it produced different decisions, never explained -- 250 KB of dead-by-default
// code with an open question attached does not belong in the product.
#ifndef BBAI_BOT_GROUP_BUILD
#define BBAI_BOT_GROUP_BUILD 0
#endif
#if BBAI_BOT_GROUP_BUILD

// The expert as ONE LANE GROUP PER ENV (G lanes, 64 / G envs per wave; bbai_bot.hpp "Execution model"): the subgoal machine runs
// group-uniform, the view, the mask rows, the neighbours of a popped position and the acceptance / key scans are split over the lanes.
// Per group in LDS: the four row-mask arrays [R_ALL][H], search 1's predecessor + queue arrays [2][W * H] uint16 (the eager first
// search never leaves LDS), and the env's BotState for the length of the decision (the subgoal stack stays in global memory: a
// decision touches its top).  Search 2's arrays (only when search 1 failed) are per-resident-group global scratch.
__host__ __device__ inline int botg_group_words(const LevelCfg& c) {
    return R_ALL * c.H + (2 * c.W * c.H * 2 + 3) / 4 + (int)sizeof(BotState) / 4;
}
template <int G, int WAVES_PER_SIMD>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_botg(LevelCfg c, int64_t n, const uint8_t* __restrict__ recs, const uint8_t* __restrict__ ring, int depth,
                                             const Hot* __restrict__ hots, const uint64_t* __restrict__ stales, uint8_t* __restrict__ states, int stack_cap,
                                             uint16_t* __restrict__ works, int eager, const uint8_t* __restrict__ prev_actions, uint8_t* __restrict__ out,
                                             unsigned long long* __restrict__ stats, int dead_action, uint8_t* __restrict__ gave_up) {
    extern __shared__ uint32_t s_botg[];
    constexpr int NG = 64 / G, SW = (int)sizeof(BotState) / 4;
    const GroupCtx<G> ctx;
    const int lane = ctx.lane();
    const int cells = c.W * c.H;
    uint32_t* blk = s_botg + ((int)threadIdx.x / G) * botg_group_words(c);
    const int64_t group = (int64_t)blockIdx.x * NG + (int)threadIdx.x / G, ngroups = (int64_t)gridDim.x * NG;
    BotWork w;
    w.eager = eager;
    w.ring = nullptr; w.ring_stride = 0; w.ring_size = 0;
    w.rows_fast = blk; w.rstride_fast = 1; w.rows_h = c.H; w.fast_n = R_ALL;
    w.rows_slow = nullptr; w.rstride_slow = 0;
    w.cells = cells;
    w.near_q = (uint16_t*)(blk + R_ALL * c.H);
    w.base = works + group * (int64_t)(4 * cells);          // (arrays 2, 3: search 2)
    w.stride = 1;
    uint32_t* sst = blk + R_ALL * c.H + (2 * cells * 2 + 3) / 4;
    BotState& st = *(BotState*)sst;
    const size_t sbytes = bot_state_bytes(stack_cap);
    for (int64_t i = group; i < n; i += ngroups) {
        const Hot h = hots[i];
        if (h.frozen) {
            if (lane == 0) { out[i] = A_DONE; if (gave_up) gave_up[i] = 0; }
            continue;
        }
        uint32_t* gst = (uint32_t*)(states + i * (int64_t)sbytes);
        for (int k = lane; k < SW; k += G) sst[k] = gst[k];
        ctx.sync();
        const bool first = h.step == 0 || st.next_step != h.step;
        const int taken = (prev_actions && !first) ? prev_actions[i] : -1;
        const bool was_dead = !first && st.dead;
        const int a = bot_decide(ctx, c, live_rec(c, n, i, (uint8_t*)recs, (uint8_t*)ring, depth, h.slot), h, stales[i], st, (Subgoal*)(gst + SW), stack_cap, w, first, taken);
        ctx.sync();
        if (lane == 0) {
            out[i] = (uint8_t)(a == BOT_DEAD ? dead_action : a);
            if (gave_up) gave_up[i] = a == BOT_DEAD ? 1 : 0;
            if (a == BOT_DEAD && !was_dead) atomicAdd(&stats[st.dead == DEAD_CAPACITY ? 1 : 0], 1ull);
        }
        for (int k = lane; k < SW; k += G) gst[k] = sst[k];
        ctx.sync();
    }
}
#endif

// env.seed(s) for every env: lane = env.  Each lane writes its own 624-word state (2496-byte pitch): a wave's 64 open
// lines stay in L2 until they are full, so HBM sees each state line once.
__global__ __launch_bounds__(64) void k_seed(int64_t n, const uint64_t* __restrict__ seeds, uint32_t* __restrict__ mts, int32_t* __restrict__ mtis) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    seed_env(seeds[i], mts + i * MT_N);
    mtis[i] = MT_N;                               // output index 624: the first draw twists (RandomState.seed leaves pos = N)
}

__global__ void k_init_hot(int64_t n, Hot* __restrict__ hots, Hot* __restrict__ next_hots, uint64_t* __restrict__ stales, int depth) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        Hot h;
        memset(&h, 0, sizeof(h));
        h.carry = NONE8; h.frozen = 1; h.last_locked = NONE8;
        h.pre4 = 0xFFFFFFFFu;
        hots[i] = h;
        for (int d = 0; d < depth; ++d) next_hots[ring_at(d, i, depth)] = h;
        stales[i] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// k_render : encoded obs -> 56x56x3 pixels through the tile atlas
// ------------------------------------------------------------------------------------------
constexpr int RENDER_QUEUE_DEFAULT = 1;         // queue shape of render_launch used from RENDER_QUEUE_MIN_ENVS up
constexpr int RENDER_QUEUE_PACED = 3;           // ... with time-paced tickets: (1024, 8) blocks, two interleaved counters
constexpr int64_t RENDER_QUEUE_MIN_ENVS = 262144;

constexpr int CHUNKS_PER_ROW = PIX * 3 / 8;       // 21 eight-byte chunks per pixel row
constexpr int VEC_PER_ENV = PIX_BYTES / 16;       // 588 sixteen-byte stores per env

// One 8-byte piece of the env's pixel image: chunk id -> (tile, row in tile, third of the row).
__device__ __forceinline__ uint64_t render_chunk(const uint8_t* s_atlas, const uint8_t* tiles49, int ch) {
    const int py = ch / CHUNKS_PER_ROW, cx = ch - py * CHUNKS_PER_ROW;
    const int ti = cx / 3, part = cx - ti * 3;        // tile column (view x), 8-byte third of the tile row
    const int tj = py >> 3, ty = py & 7;              // tile row (view y), row inside the tile
    const int tile = tiles49[ti * VIEW + tj];
    return *(const uint64_t*)(s_atlas + tile * TILE_BYTES + ty * 24 + part * 8);
}

// (Round 3's alternative input -- a fused tile plane, one masked appearance byte per cell, left behind by k_step -- was measured
// once more with the ticket queue in round 4 (profiles/r04/render_queue_counters_1M_lease_d.jsonl: k_render 1.503 vs 1.505 ms, k_step
// + 0.02 ms) and removed.)
template <int RENDER_GROUP, int RENDER_BLOCK>     // envs per block iteration (between two barriers); threads per block
__global__ __launch_bounds__(RENDER_BLOCK) void k_render(int64_t n, const uint8_t* __restrict__ image,
                                                         uint8_t* __restrict__ pixels, const uint8_t* __restrict__ atlas,
                                                         const uint8_t* __restrict__ lut, int n_tiles) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[MAX_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[512];
    __shared__ uint8_t s_tile[RENDER_GROUP * VIEW * VIEW + 8];
    for (int k = threadIdx.x; k < n_tiles * TILE_BYTES / 8; k += RENDER_BLOCK)
        ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
    for (int k = threadIdx.x; k < 512; k += RENDER_BLOCK) s_lut[k] = lut[k];
    const int64_t ngroups = (n + RENDER_GROUP - 1) / RENDER_GROUP;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t env0 = grp * RENDER_GROUP;
        const int ne = (int)(n - env0 < RENDER_GROUP ? n - env0 : RENDER_GROUP);
        __syncthreads();                               // atlas loaded / previous group's tiles consumed
        // encoded cell -> atlas tile, once per cell (49 per env)
        for (int c = threadIdx.x; c < ne * VIEW * VIEW; c += RENDER_BLOCK) {
            const int e = c / (VIEW * VIEW), cell = c - e * (VIEW * VIEW);
            const uint8_t* o = image + (env0 + e) * OBS_BYTES + cell * 3;
            const int key = o[0] | (o[1] << 3) | (o[2] << 6);
            s_tile[c] = s_lut[(cell == 3 * VIEW + 6 ? 256 : 0) + key];
        }
        __syncthreads();
        // 16 bytes per lane per store: a wave writes 1 KiB of contiguous pixels
        u32x4* out = (u32x4*)(pixels + env0 * PIX_BYTES);
        for (int q = threadIdx.x; q < ne * VEC_PER_ENV; q += RENDER_BLOCK) {
            const int e = q / VEC_PER_ENV, k = q - e * VEC_PER_ENV;
            const uint8_t* t49 = s_tile + e * (VIEW * VIEW);
            const uint64_t lo = render_chunk(s_atlas, t49, 2 * k);
            const uint64_t hi = render_chunk(s_atlas, t49, 2 * k + 1);
            u32x4 v = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
            __builtin_nontemporal_store(v, out + q);   // streaming output: keep it out of L2/MALL
        }
    }
}

// k_render_q: the same render from PERSISTENT blocks (the atlas is loaded into LDS once per block) that take their work from an
// atomic ticket counter, so that the chip's stores advance through the output as ONE compact, evenly paced window -- the order
// in which the pure store stream is fastest (render_launch has the measurements).
//   * a ticket = K consecutive G-env groups; the next ticket is taken while the first group of the current one is being
//     staged, so its latency rides under the stores;
//   * NC counters, 256 bytes apart, INTERLEAVED: ticket t of counter c is super-group t * NC + c, a block uses counter
//     blockIdx % NC (= its XCD for NC = 8).  Shipped: NC = 1, K = 1 -- more counters or bigger tickets relieve the ticket rate
//     (~88 M/s per address) and measure SLOWER: the counters drift apart, the window widens;
//   * the counters clean up after themselves: the last block to leave (a departure counter) zeroes them for the next launch,
//     so the step path carries no memset.
// Tile rows and tickets are double-buffered: one barrier per group.
// (The counters are the handle's: two renders of one handle never overlap -- every entry point orders a call on another stream behind the
// handle's previous call, enter_call -- and a launch that was aborted by a device fault leaves a handle that is unusable anyway.)
template <int RENDER_GROUP, int RENDER_BLOCK, int NC, int K>
__global__ __launch_bounds__(RENDER_BLOCK) void k_render_q(int64_t n, const uint8_t* __restrict__ image,
                                                           uint8_t* __restrict__ pixels, const uint8_t* __restrict__ atlas,
                                                           const uint8_t* __restrict__ lut, int n_tiles, unsigned int* __restrict__ counters,
                                                           int pace_x16 /* experiment: 0, or 1/16 ns of wall clock per ticket (render_launch) */) {
    __shared__ __attribute__((aligned(16))) uint8_t s_atlas[MAX_TILES * TILE_BYTES];
    __shared__ uint8_t s_lut[512];
    __shared__ uint8_t s_tile[2][RENDER_GROUP * VIEW * VIEW + 8];
    __shared__ unsigned int s_ticket[2];
    __shared__ unsigned long long s_origin;
    for (int k = threadIdx.x; k < n_tiles * TILE_BYTES / 8; k += RENDER_BLOCK)
        ((uint64_t*)s_atlas)[k] = ((const uint64_t*)atlas)[k];
    for (int k = threadIdx.x; k < 512; k += RENDER_BLOCK) s_lut[k] = lut[k];
    const int64_t all_groups = (n + RENDER_GROUP - 1) / RENDER_GROUP;
    const int64_t n_super = (all_groups + K - 1) / K;
    const unsigned int cidx = blockIdx.x % NC;
    unsigned int* counter = counters + 64 * cidx;
    int buf = 0, tp = 0;
    if (threadIdx.x == 0) {
        const unsigned int t0 = atomicAdd(counter, 1u);
        s_ticket[0] = t0;
        // time-paced tickets (experiment): ticket sg is not started before origin + sg x pace; the 100-MHz constant clock in 1/16 ns
        s_origin = __builtin_amdgcn_s_memrealtime() * 160ull - ((unsigned long long)t0 * NC + cidx) * (unsigned long long)pace_x16;
    }
    __syncthreads();
    for (;;) {
        const int64_t sg = (int64_t)s_ticket[tp] * NC + cidx;
        if (sg >= n_super) break;
        if (pace_x16) {
            const unsigned long long due = s_origin + (unsigned long long)sg * (unsigned long long)pace_x16;
            while (__builtin_amdgcn_s_memrealtime() * 160ull < due) __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int kk = 0; kk < K; ++kk) {
            const int64_t grp = sg * K + kk;
            if (grp >= all_groups) break;
            const int64_t env0 = grp * RENDER_GROUP;
            const int ne = (int)(n - env0 < RENDER_GROUP ? n - env0 : RENDER_GROUP);
            for (int c = threadIdx.x; c < ne * VIEW * VIEW; c += RENDER_BLOCK) {
                const int e = c / (VIEW * VIEW), cell = c - e * (VIEW * VIEW);
                const uint8_t* o = image + (env0 + e) * OBS_BYTES + cell * 3;
                const int key = o[0] | (o[1] << 3) | (o[2] << 6);
                s_tile[buf][c] = s_lut[(cell == 3 * VIEW + 6 ? 256 : 0) + key];
            }
            if (kk == 0 && threadIdx.x == 0) s_ticket[tp ^ 1] = atomicAdd(counter, 1u);      // the next ticket rides under this one's stores
            __syncthreads();
            u32x4* out = (u32x4*)(pixels + env0 * PIX_BYTES);
            for (int q = threadIdx.x; q < ne * VEC_PER_ENV; q += RENDER_BLOCK) {
                const int e = q / VEC_PER_ENV, k = q - e * VEC_PER_ENV;
                const uint8_t* t49 = s_tile[buf] + e * (VIEW * VIEW);
                const uint64_t lo = render_chunk(s_atlas, t49, 2 * k);
                const uint64_t hi = render_chunk(s_atlas, t49, 2 * k + 1);
                u32x4 v = {(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
                __builtin_nontemporal_store(v, out + q);
            }
            buf ^= 1;
        }
        tp ^= 1;
    }
    // every block has taken its last ticket before it arrives here (the failing ticket was read through LDS behind a barrier);
    // the last one to arrive leaves all counters at zero for the next launch
    if (threadIdx.x == 0) {
        unsigned int* departed = counters + 64 * NC;
        if (atomicAdd(departed, 1u) == gridDim.x - 1) {
            for (int c = 0; c <= NC; ++c) atomicExch(counters + 64 * c, 0u);
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_tokens : mission strings as fixed-vocabulary token ids, produced on the device from the compiled
// instruction program (grammar: babyai/levels/verifier.py:64-94,248-249,287-288,318-319,366-367,439-440,
// 480-481,526-527).  Vocabulary ids = babyai_amd/missions.py VOCAB (1..32, 0 = padding).
// ------------------------------------------------------------------------------------------
constexpr int TOK_MAX = 72;      // longest sentence: two And-pairs of put-next clauses with locations
struct TokOut {
    uint8_t* p; int n;
    __device__ __forceinline__ void put(int id) { if (n < TOK_MAX) p[n++] = (uint8_t)id; }
};
__device__ __forceinline__ void tok_desc(TokOut& o, DescInfo d) {
    o.put(d.count > 1 ? 9 : 8);                         // a / the
    if (d.color != 7) o.put(11 + d.color);              // red green blue purple yellow grey
    o.put(d.type == 0 ? 10 : 24 - d.type);              // object | box ball key door
    if (d.loc == LOC_FRONT) { o.put(21); o.put(22); o.put(23); o.put(24); }      // in front of you
    else if (d.loc == LOC_BEHIND) { o.put(25); o.put(24); }                     // behind you
    else if (d.loc == LOC_LEFT) { o.put(26); o.put(27); o.put(28); }            // on your left
    else if (d.loc == LOC_RIGHT) { o.put(26); o.put(27); o.put(29); }           // on your right
}
__device__ __forceinline__ void tok_side(TokOut& o, const Prog* p, int base, int n) {
    for (int q = 0; q < n; ++q) {
        if (q) o.put(30);                                                        // and
        const int kind = p->kind[base + q];
        if (kind == L_GOTO) { o.put(1); o.put(2); }                              // go to
        else if (kind == L_PICKUP) { o.put(3); o.put(4); }                       // pick up
        else if (kind == L_OPEN) o.put(5);                                       // open
        else o.put(6);                                                           // put
        tok_desc(o, p->desc[base + q][0]);
        if (kind == L_PUTNEXT) { o.put(7); o.put(2); tok_desc(o, p->desc[base + q][1]); }   // next to
    }
}
__global__ __launch_bounds__(64) void k_tokens(LevelCfg c, int64_t n, const uint8_t* __restrict__ recs, const uint8_t* __restrict__ ring /* in-place layout, else NULL */, int depth,
                                               const Hot* __restrict__ hots, uint8_t* __restrict__ tokens,
                                               const int32_t* __restrict__ reset_list, const uint32_t* __restrict__ counter,
                                               int mode /* 0: the reset list (unfused consume); 1: every env; 2: the envs whose `dones` byte is set -- a fused / in-place
                                                           auto-reset step keeps no list, and there done == "a new episode started" */,
                                               const uint8_t* __restrict__ dones) {
    const int64_t count = mode ? n : (int64_t)counter[0];
    for (int64_t it = (int64_t)blockIdx.x * 64 + threadIdx.x; it < count; it += (int64_t)gridDim.x * 64) {
        const int64_t env = mode ? it : (int64_t)reset_list[it];
        if (mode == 2 && !dones[env]) continue;
        const Prog* p = (const Prog*)(live_rec(c, n, env, (uint8_t*)recs, (uint8_t*)ring, depth, ring ? hots[env].slot : 0) + c.off_prog);
        TokOut o; o.p = tokens + env * TOK_MAX; o.n = 0;
        tok_side(o, p, 0, p->n_a);
        if (p->root == R_BEFORE) { o.put(31); tok_side(o, p, 2, p->n_b); }                   // , then
        else if (p->root == R_AFTER) { o.put(32); o.put(24); tok_side(o, p, 2, p->n_b); }    // after you
        while (o.n < TOK_MAX) o.p[o.n++] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// k_tap : copy the outputs of `count` envs (and the pixels of the first `pix_count` of them) into log rows -- the parity
// tap of bench.py as ONE launch inside the timed region (five small tensor copies cost more than a 65 536-env step).
// ids == NULL: the first `count` envs; else env ids[k] -> log row k (any order, anywhere in the batch).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tap(int64_t count, int64_t pix_count, const int64_t* __restrict__ ids, const uint8_t* __restrict__ image,
                                             const uint8_t* __restrict__ dirs, const double* __restrict__ rew64, const uint8_t* __restrict__ dones,
                                             const uint8_t* __restrict__ pixels, uint8_t* __restrict__ image_out, uint8_t* __restrict__ dirs_out,
                                             double* __restrict__ rew64_out, uint8_t* __restrict__ dones_out, uint8_t* __restrict__ pixels_out) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < count * OBS_BYTES; i += nth) {
        const int64_t k = i / OBS_BYTES, b = i - k * OBS_BYTES;
        image_out[i] = image[(ids ? ids[k] : k) * OBS_BYTES + b];
    }
    for (int64_t i = tid; i < count; i += nth) {
        const int64_t e = ids ? ids[i] : i;
        dirs_out[i] = dirs[e]; dones_out[i] = dones[e]; rew64_out[i] = rew64[e];
    }
    constexpr int VEC = PIX_BYTES / 16;
    const u32x4* src = (const u32x4*)pixels;
    u32x4* dst = (u32x4*)pixels_out;
    for (int64_t i = tid; i < pix_count * VEC; i += nth) {
        const int64_t k = i / VEC, v = i - k * VEC;
        dst[i] = src[(ids ? ids[k] : k) * VEC + v];
    }
}

// ------------------------------------------------------------------------------------------
// k_gae : generalised advantage estimation of a rollout, lane = env (babyai/rl/algos/base.py:196-202 as ONE reverse
// scan per env instead of T passes of five tensor ops).  All buffers are env-major [P][T], the layout the reference
// flattens its experiences to (base.py:207-232), so nothing is transposed afterwards.  float32 arithmetic in the
// reference's operation order (python scalars multiply as float32; the file is built with -ffp-contract=off):
//   delta = (r + (d * next_value) * next_mask) - v ;  adv = delta + ((d * lambda) * next_adv) * next_mask
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gae(int64_t P, int T, const float* __restrict__ rewards, const float* __restrict__ values,
                                            const float* __restrict__ masks, const float* __restrict__ last_mask,
                                            const float* __restrict__ last_value, float d, float dl, float* __restrict__ adv,
                                            float* __restrict__ ret) {
    const int64_t p = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (p >= P) return;
    const float* r = rewards + p * T; const float* v = values + p * T; const float* m = masks + p * T;
    float next_value = last_value[p], next_mask = last_mask[p], next_adv = 0.0f;
    for (int i = T - 1; i >= 0; --i) {
        const float vi = v[i];
        const float delta = (r[i] + (d * next_value) * next_mask) - vi;
        const float a = delta + (dl * next_adv) * next_mask;
        adv[p * T + i] = a;
        ret[p * T + i] = vi + a;
        next_value = vi; next_mask = m[i]; next_adv = a;
    }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int bbai_version(void) { return 200; }
const char* bbai_last_error(void) { return g_err; }

int bbai_fill_layout(bbai_level_cfg* cfg) {
    if (!cfg) ARG_FAIL("null cfg");
    LevelCfg c;
    memcpy(&c, cfg, sizeof(c));
    if (fill_layout(c) != 0) { snprintf(g_err, sizeof(g_err), "unsupported level geometry"); return BBAI_ERR_ARG; }
    memcpy(cfg, &c, sizeof(c));
    return BBAI_OK;
}

static int create_finish(bbai_env* e);

static int validate_cfg(const LevelCfg& c) {
    if (c.kind != K_GOTO && c.kind != K_LEVELGEN && c.kind != K_BONUS) return -1;
    if (c.kind == K_BONUS && (c.script < 1 || c.script >= BS_COUNT)) return -1;
    if (c.num_dists < 0) return -1;
    if (c.kind == K_GOTO) {
        if (!c.redball && c.num_dists < 1) return -1;
        if (c.instr < L_GOTO || c.instr > L_PUTNEXT || c.target < TG_REDBALL || c.target > TG_LOCKED_ROOM_OBJ) return -1;
        if (c.instr == L_PUTNEXT && (c.target != TG_TWO_DISTS || c.num_dists < 2)) return -1;
        if ((c.target == TG_LOCKED_DOOR || c.target == TG_LOCKED_ROOM_OBJ) != (c.lock != 0)) return -1;
        if (c.lock && c.num_rows * c.num_cols < 2) return -1;
    }
    if (c.kind == K_LEVELGEN) {
        if (c.n_action_kinds < 1 || c.n_action_kinds > 4 || c.n_instr_kinds < 1 || c.n_instr_kinds > 3) return -1;
        for (int i = 0; i < c.n_action_kinds; ++i) if (c.action_kinds[i] < 0 || c.action_kinds[i] > 3) return -1;
        for (int i = 0; i < c.n_instr_kinds; ++i) if (c.instr_kinds[i] < 0 || c.instr_kinds[i] > 2) return -1;
    }
    // max_steps must fit uint16: 8 navigations * S^2 * rows * cols
    if (8 * c.room_size * c.room_size * c.num_rows * c.num_cols > 65535) return -1;
    return 0;
}

static int inplace_by_default(const LevelCfg& c, int64_t n_envs) {
    // Measured, ms per step classic -> in-place, settings alternated in one process (tools/ab.py).  Round 4 (profiles/r04/inplace_*.jsonl): GoToLocal 32 768
    // envs 0.0253 -> 0.0183, 65 536 0.0341 -> 0.0322, PickupLoc 262 144 0.0780 -> 0.0733; mazes lose at every size (no window plane in this layout:
    // BossLevel 1 048 576 0.111 -> 0.133).  Round 5, after the list atomic left the step path (profiles/r05/inplace_size_sweep.jsonl): GoToLocal 262 144
    // 0.0832 -> 0.0720, GoToLocal 1 048 576 0.333 -> 0.305, PickupLoc 524 288 0.140 -> 0.127, PickupLoc 1 048 576 0.321 -> 0.331 (the one point where the
    // classic layout is ahead, by 3 %).  So: every single-room batch (the ring's own 64-GiB cap -- bbai_create -- bounds it; a ring that does not fit
    // falls back to a shorter look-ahead period, never to another layout).
    (void)n_envs;
    return c.num_rows * c.num_cols > 1 ? 0 : 1;
}

int bbai_create(const bbai_level_cfg* cfg, int64_t n_envs, int device, bbai_env** out) {
    if (!cfg || !out || n_envs <= 0 || n_envs > (1ll << 30)) { snprintf(g_err, sizeof(g_err), "bad argument"); return BBAI_ERR_ARG; }
    LevelCfg c;
    memcpy(&c, cfg, sizeof(c));
    LevelCfg chk = c;
    if (fill_layout(chk) != 0 || memcmp(&chk, &c, sizeof(c)) != 0 || validate_cfg(c) != 0) {
        snprintf(g_err, sizeof(g_err), "level configuration rejected (layout not filled or out of range)");
        return BBAI_ERR_ARG;
    }
    ON_DEVICE(device);
    bbai_env* e = new bbai_env();
    memset(e, 0, sizeof(*e));
    e->cfg = c; e->n = n_envs; e->device = device;
    { const char* ev = getenv("BBAI_BOT_GROUP"); if (ev) e->bot_group = atoi(ev); }
    hipError_t err = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (err == hipSuccess) err = hipMalloc(p, bytes); };
    alloc((void**)&e->rec, (size_t)n_envs * c.rec_bytes);
    alloc((void**)&e->hot, (size_t)n_envs * sizeof(Hot));
    alloc((void**)&e->stale, (size_t)n_envs * 8);
    alloc((void**)&e->mt, (size_t)n_envs * MT_N * 4);
    alloc((void**)&e->mti, (size_t)n_envs * 4);
    alloc((void**)&e->vhead, (size_t)n_envs * 4);
    alloc((void**)&e->vset, (size_t)n_envs * 8 * 8);
    alloc((void**)&e->pending, NWIN * (size_t)n_envs);
    alloc((void**)&e->first_slot, NWIN * (size_t)n_envs);
    alloc((void**)&e->win_meta, NWIN * META_U32 * 4);
    alloc((void**)&e->totals, SHARDS * SHARD_U64 * 8);
    alloc((void**)&e->flow, FLOW_WORDS * 8);
    alloc((void**)&e->gen_list, (size_t)SHARDS * (size_t)gen_sublist_cap(n_envs) * 4);
    alloc((void**)&e->gen_count, SHARDS * GEN_COUNT_U32 * 4);
    e->gen_lists[0] = e->gen_list; e->gen_counts[0] = e->gen_count;         // (further look-ahead streams' lists: set_sides)
    {
        // BBAI_INPLACE: 1 / 0 force the in-place layout (live_slot above) on / off; default: by level family and batch size
        const char* iv = getenv("BBAI_INPLACE");
        e->inplace = iv ? (atoi(iv) != 0) : inplace_by_default(c, n_envs);
    }
    {
        const char* vv = getenv("BBAI_VPLANE");              // 0: round 2's record-only step path (A/B runs)
        if (!(vv && atoi(vv) == 0) && !e->inplace) {
            alloc((void**)&e->vplane, (size_t)n_envs * v_bytes(c));
            alloc((void**)&e->fcache, (size_t)n_envs * 2);
        }
        const char* cv = getenv("BBAI_CPLANE");              // 0: the in-place step path of rounds 4-5 (window out of the record; A/B runs)
        if (e->inplace && cpl_ok(c) && !(cv && atoi(cv) == 0)) {
            alloc((void**)&e->cplane, (size_t)n_envs * cpl_bytes(c));
            alloc((void**)&e->fcache, (size_t)n_envs * 2);
        }
    }
    if (lane_gen_ok(c) && (!e->inplace || cpl_ok(c))) {
        // Default: the single rooms.  Measured (profiles/r06/NOTES.md): bulk fills 2.2 - 2.9 ns per level against 6 - 7 (PickupLoc 262 144: 0.057 -> 0.043 ms
        // per step); the mazes' bulk rates are 17 (GoTo) and 31 (BossLevel) ns against 30 and 20, and a maze refill of a few thousand levels
        // lasts milliseconds as ONE level per lane where the lane-group kernel needs a third of that -- the mazes keep k_pregen.
        const char* lv = getenv("BBAI_PREGEN_LANE");
        e->pregen_lane = lv ? (atoi(lv) != 0) : (c.num_rows * c.num_cols == 1);
        const char* lb = getenv("BBAI_LANE_BLOCKS");
        e->lane_blocks = lb ? std::max(1, atoi(lb)) : 16384;
        e->lane_words = lane_layout(c).words;
        alloc((void**)&e->mtt, (size_t)n_envs * 2 * MT_N * 4);
        alloc((void**)&e->mtpar, (size_t)n_envs);
        alloc((void**)&e->lane_tmpl, (size_t)lane_template_bytes(c));
        if (err == hipSuccess) {
            std::vector<uint8_t> t((size_t)lane_template_bytes(c));
            lane_build_template(c, t.data());
            err = hipMemcpy(e->lane_tmpl, t.data(), t.size(), hipMemcpyHostToDevice);
            if (err == hipSuccess) err = hipMemset(e->mtpar, 0, (size_t)n_envs);
            if (err == hipSuccess) err = hipMemset(e->mtt, 0, (size_t)n_envs * 2 * MT_N * 4);
        }
    }
    {
        // Refill period B (ticks per look-ahead refill, BBAI_LOOKAHEAD); ring depth D = 2B.  One k_pregen launch per
        // window lasts as long as its slowest level (hundreds of microseconds to milliseconds: rejection sampling has a
        // heavy tail) and has to land within one window, so B ticks of the step path must outlast it or the step stream
        // waits: default = the longest period of 96 (mazes), 64, 32, 16, 8, 4, 2 whose ring fits the cap.  (Up to 32 until round 6.  A refill launch lasts as
        // long as its slowest WAVE -- 0.4 ms for 64 single-room levels per lane-generator wave, 0.6-0.9 ms for a lane group's maze -- almost whatever its
        // size, and bbai_rollout's k_step takes 8 us per tick at 65 536 envs: 32 ticks no longer outlast a refill.  GoToLocal 65 536 envs 0.0146 -> 0.0103 ms
        // per step at 64, GoTo 131 072 0.0344 -> 0.0244 at 64 -> 0.0214 at 96: profiles/r06/NOTES.md section 9.)  The cap is BBAI_RING_GIB
        // (default 64 GiB -- the part has 288 GB; round 4: 1 048 576 GoTo envs waited for refills at period 4, the 16-GiB cap of
        // rounds 1-3) but never more than a quarter of the memory that is FREE right now (several handles or ranks
        // on one device, smaller parts), and an allocation that fails all the same is retried with half the period: a
        // shorter period only costs speed, never correctness.  (131072 GoTo envs -> 32; 1M BossLevel envs -> 16.)
        const char* ev = getenv("BBAI_LOOKAHEAD");
        const char* gv = getenv("BBAI_RING_GIB");
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)64 << 30;
        const size_t slot_bytes = (size_t)n_envs * c.rec_bytes;
        const size_t cap = std::min((size_t)(gv ? std::max(1, atoi(gv)) : 64) << 30, free_b / 4);
        int b = 2;
        if (ev) b = atoi(ev);
        else for (int cand : {96, 64, 32, 16, 8, 4, 2}) if (cand <= (c.num_rows * c.num_cols > 1 ? 96 : 64))      // (single rooms: 96 measured equal to 64)
             if (slot_bytes * (2 * (size_t)cand + (size_t)e->inplace) <= cap) { b = cand; break; }
        b = b < 1 ? 1 : (b > MAX_PERIOD ? MAX_PERIOD : b);
        for (; err == hipSuccess; b >>= 1) {
            e->period = b;
            e->depth = 2 * b + e->inplace;         // (in-place: the live slot + the 2B look-ahead levels)
            const size_t D = (size_t)e->depth;
            hipError_t r1 = hipMalloc((void**)&e->next_rec, D * slot_bytes);
            hipError_t r2 = r1 == hipSuccess ? hipMalloc((void**)&e->next_hot, D * (size_t)n_envs * sizeof(Hot)) : r1;
            hipError_t r3 = r2;
            if (r3 == hipSuccess && e->inplace) r3 = hipMalloc((void**)&e->next_obs, D * (size_t)n_envs * OBS_SLOT);
            if (r3 == hipSuccess) break;
            (void)hipGetLastError();                       // clear the sticky out-of-memory error before retrying
            if (e->next_rec) { (void)hipFree(e->next_rec); e->next_rec = nullptr; }
            if (e->next_hot) { (void)hipFree(e->next_hot); e->next_hot = nullptr; }
            if (e->next_obs) { (void)hipFree(e->next_obs); e->next_obs = nullptr; }
            if (b == 1 || ev) err = r3;                    // an explicit BBAI_LOOKAHEAD is a request, not a hint
        }
    }
    {
        // babyai/levels/verifier.py:17 `use_done_actions = os.environ.get('BABYAI_DONE_ACTIONS', False)`: any non-empty value
        const char* dv = getenv("BABYAI_DONE_ACTIONS");
        if (dv && dv[0]) alloc((void**)&e->lsm, (size_t)n_envs);
    }
    alloc((void**)&e->reset_list, (size_t)n_envs * 4);
    alloc((void**)&e->reset_slot, (size_t)n_envs);
    alloc((void**)&e->counters, 128);
    alloc((void**)&e->atlas, MAX_TILES * TILE_BYTES);
    alloc((void**)&e->render_tickets, 64 * 64 * 4);
    alloc((void**)&e->lut, 512);
    if (err == hipSuccess) {
        void* hf = nullptr;
        err = hipHostMalloc(&hf, 64, hipHostMallocMapped);
        if (err == hipSuccess) {
            memset(hf, 0, 64);
            e->host_flags = (volatile uint32_t*)hf;
            void* dv = nullptr;
            err = hipHostGetDevicePointer(&dv, hf, 0);
            e->host_flags_dev = (uint32_t*)dv;
        }
    }
    if (err != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "hipMalloc failed: %s", hipGetErrorString(err));
        bbai_destroy(e);
        return BBAI_ERR_NOMEM;
    }
    int rc = create_finish(e);
    if (rc != BBAI_OK) { bbai_destroy(e); return rc; }      // g_err holds the failing call
    *out = e;
    return BBAI_OK;
}

}  // extern "C"

// The look-ahead streams beyond the first, and the stream k_mark needs once there are several, exist only while asked for: every HIP stream
// is a hardware queue, and a dozen of them made EVERY dispatch of a handle that used two wait its turn (k_step 10 -> 50 us whenever a refill ran,
// profiles/r06/NOTES.md section 9).
static int set_sides(bbai_env* e, int want) {
    want = std::max(1, std::min(MAX_SIDES, want));
    for (int k = 1; k < want; ++k) {
        if (!e->sides[k]) HIP_TRY(hipStreamCreateWithPriority(&e->sides[k], hipStreamNonBlocking, e->side_prio));
        if (!e->gen_lists[k]) HIP_TRY(hipMalloc((void**)&e->gen_lists[k], (size_t)SHARDS * (size_t)gen_sublist_cap(e->n) * 4));
        if (!e->gen_counts[k]) HIP_TRY(hipMalloc((void**)&e->gen_counts[k], SHARDS * GEN_COUNT_U32 * 4));
    }
    if (want > 1) {
        if (!e->mark_stream) HIP_TRY(hipStreamCreateWithPriority(&e->mark_stream, hipStreamNonBlocking, e->side_prio));
        for (int k = 0; k < want; ++k) if (!e->ev_part[k]) HIP_TRY(hipEventCreateWithFlags(&e->ev_part[k], hipEventDisableTiming));
    }
    for (int k = want; k < MAX_SIDES; ++k) if (k >= 1 && e->sides[k]) { HIP_TRY(hipStreamDestroy(e->sides[k])); e->sides[k] = nullptr; }
    if (want == 1 && e->mark_stream) { HIP_TRY(hipStreamDestroy(e->mark_stream)); e->mark_stream = nullptr; }
    e->n_sides = want;
    e->n_probed = 0;
    return BBAI_OK;
}
static int create_finish(bbai_env* e) {
    const LevelCfg& c = e->cfg;
    const int64_t n_envs = e->n;
    const size_t D = (size_t)e->depth;
    HIP_TRY(hipMemset(e->rec, 0, (size_t)n_envs * c.rec_bytes));
    HIP_TRY(hipMemset(e->hot, 0, (size_t)n_envs * sizeof(Hot)));       // (slot 0 before any seed: an in-place handle that is only imported into keeps its live records in slot depth - 1)
    HIP_TRY(hipMemset(e->next_rec, 0, D * (size_t)n_envs * c.rec_bytes));
    if (e->next_obs) HIP_TRY(hipMemset(e->next_obs, 0, D * (size_t)n_envs * OBS_SLOT));
    HIP_TRY(hipMemset(e->pending, 0, NWIN * (size_t)n_envs));
    HIP_TRY(hipMemset(e->first_slot, 0, NWIN * (size_t)n_envs));
    HIP_TRY(hipMemset(e->win_meta, 0, NWIN * META_U32 * 4));
    HIP_TRY(hipMemset(e->totals, 0, SHARDS * SHARD_U64 * 8));
    HIP_TRY(hipMemset(e->flow, 0, FLOW_WORDS * 8));
    HIP_TRY(hipMemset(e->vhead, 0, (size_t)n_envs * 4));
    HIP_TRY(hipMemset(e->vset, 0, (size_t)n_envs * 64));
    HIP_TRY(hipMemset(e->counters, 0, 128));
    HIP_TRY(hipMemset(e->render_tickets, 0, 64 * 64 * 4));
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) cus = 0;
        e->n_cus = cus;
    }
    if (e->lsm) HIP_TRY(hipMemset(e->lsm, 0, (size_t)n_envs));
    if (e->vplane) {
        HIP_TRY(hipMemset(e->vplane, 0, (size_t)n_envs * v_bytes(c)));
        HIP_TRY(hipMemset(e->fcache, 0, (size_t)n_envs * 2));
    }
    if (e->cplane) {
        HIP_TRY(hipMemset(e->cplane, 0, (size_t)n_envs * cpl_bytes(c)));
        HIP_TRY(hipMemset(e->fcache, 0, (size_t)n_envs * 2));
    }
    {
        // The look-ahead stream.  (Confining it to a subset of the CUs with a CU mask was measured in round 3 --
        // profiles/r03/pregen_cus_ab.jsonl: 32 / 64 / 96 / 128 CUs change no step time by more than 1 %, and a masked stream is a
        // BLOCKING stream: with a caller on the NULL stream it serialises generation and stepping, 46 -> 72 us per step at
        // 65 536 GoToLocal envs.  Not kept.  Round 6 tried again for callers on their own stream: on this pool's runtime
        // hipExtStreamCreateWithCUMask changes NOTHING -- a bulk fill confined to 32 of 256 CUs runs at the unconfined rate
        // (profiles/r06/NOTES.md section 6).)
        int lo = 0, hi = 0;     // look-ahead generation should get wave slots as soon as any free up
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char* pv = getenv("BBAI_PREGEN_PRIORITY");     // 1 (default): highest priority, 0: default priority
        e->side_prio = (pv && atoi(pv) == 0) ? lo : hi;
        HIP_TRY(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, e->side_prio));
        e->sides[0] = e->side;
        const char* ls = getenv("BBAI_LOOKAHEAD_STREAMS");
        { int rc = set_sides(e, ls ? atoi(ls) : LOOKAHEAD_STREAMS_DEFAULT); if (rc != BBAI_OK) return rc; }
        HIP_TRY(hipEventCreateWithFlags(&e->ev_consumed, hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&e->split, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_split0, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_splitB, hipEventDisableTiming));
        for (int k = 0; k < NWIN; ++k) HIP_TRY(hipEventCreateWithFlags(&e->ev_refill[k], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_switch, hipEventDisableTiming));
    }
    {
        const char* cv = getenv("BBAI_CALL_EVENTS");
        e->call_events = cv && atoi(cv) != 0;
        const char* ev = getenv("BBAI_PREGEN_BLOCKS");
        // lane groups of a refill launch at most: 16 384 = two generator waves per SIMD of the part, the other half of every SIMD stays with the step
        // kernels (profiles/r05/pregen_residency_cap_ab.jsonl, ms per step at 32 768 / 16 384 / 8 192 / 4 096 groups: GoTo 131 072 envs 0.0438 / 0.0390 /
        // 0.0395 / 0.0478; BossLevel encoded 1 048 576 0.110 / 0.104 / 0.104; BossLevel pixels 131 072: no difference).  bbai_seed's first fill
        // (nothing else runs) takes the whole part: 32 768.
        e->pregen_cap = ev ? std::max(64, atoi(ev)) : 16384;
        const char* mv = getenv("BBAI_PREGEN_MIN");
        e->pregen_min = mv ? std::max(0, atoi(mv)) : 6144;
        const char* pp = getenv("BBAI_PREGEN_PER_GROUP");
        e->pregen_per_group = pp ? std::max(1, atoi(pp)) : 12;
        const char* pg = getenv("BBAI_PREGEN_GROUP");
        e->pregen_group = pg ? atoi(pg) : 32;
        const char* sp = getenv("BBAI_STEP_PRIO");
        e->step_prio = sp ? atoi(sp) : 1;        // (never slower, GoTo 131 072 envs -2 %: profiles/r03/step_prio_ab.jsonl)
        const char* rv = getenv("BBAI_RENDER_GROUP");
        e->render_group = rv ? atoi(rv) : 0;
        const char* qv = getenv("BBAI_RENDER_QUEUE");
        e->render_queue = qv ? atoi(qv) : -1;
        const char* pv2 = getenv("BBAI_RENDER_PACE");
        e->render_pace = pv2 ? std::max(0, atoi(pv2)) : 0;
        const char* ss = getenv("BBAI_STEP_RENDER_SPLIT");
        e->step_render_split = ss ? atoi(ss) : -1;
        const char* rm = getenv("BBAI_ROLLOUT_MULTI");
        e->rollout_multi = rm ? atoi(rm) : 1;
        const char* gs = getenv("BBAI_GATE_STRICT");
        e->gate_strict = gs ? atoi(gs) != 0 : 0;
        const char* gp = getenv("BBAI_GATE_PROBE");
        e->gate_probe = gp ? atoi(gp) : 1;
        const char* cf = getenv("BBAI_CONSUME_FUSED");
        e->consume_fused = cf ? atoi(cf) : -1;
        const char* tv = getenv("BBAI_RENDER_TPB");
        e->render_tpb = tv ? atoi(tv) : 0;
    }
    return BBAI_OK;
}

extern "C" {

void bbai_destroy(bbai_env* e) {
    if (!e) return;
    DeviceGuard guard_(e->device);
    (void)hipDeviceSynchronize();
    if (e->side) (void)hipStreamDestroy(e->side);
    for (int k = 1; k < MAX_SIDES; ++k) if (e->sides[k]) (void)hipStreamDestroy(e->sides[k]);
    if (e->mark_stream) (void)hipStreamDestroy(e->mark_stream);
    for (int k = 0; k < MAX_SIDES; ++k) if (e->ev_part[k]) (void)hipEventDestroy(e->ev_part[k]);
    for (int k = 1; k < MAX_SIDES; ++k) { if (e->gen_lists[k]) (void)hipFree(e->gen_lists[k]); if (e->gen_counts[k]) (void)hipFree(e->gen_counts[k]); }
    if (e->split) (void)hipStreamDestroy(e->split);
    if (e->ev_split0) (void)hipEventDestroy(e->ev_split0);
    if (e->ev_splitB) (void)hipEventDestroy(e->ev_splitB);
    if (e->ev_consumed) (void)hipEventDestroy(e->ev_consumed);
    if (e->ev_switch) (void)hipEventDestroy(e->ev_switch);
    for (int k = 0; k < NWIN; ++k) if (e->ev_refill[k]) (void)hipEventDestroy(e->ev_refill[k]);
    for (int k = 0; k < 3; ++k) for (int i = 0; i < PROF_RING; ++i) if (e->prof[k][i].a) { (void)hipEventDestroy(e->prof[k][i].a); (void)hipEventDestroy(e->prof[k][i].b); }
    void* bot_ptrs[] = {e->bot_state, e->bot_work, e->bot_stats, e->bot_rows};
    for (void* p : bot_ptrs) if (p) (void)hipFree(p);
    void* ptrs[] = {e->rec, e->hot, e->stale, e->mt, e->mti, e->vhead, e->vset, e->next_rec, e->next_hot, e->pending, e->first_slot, e->win_meta, e->totals, e->flow, e->gen_list, e->gen_count, e->reset_list, e->counters,
                    e->atlas, e->lut, e->vplane, e->fcache, e->lsm, e->render_tickets, e->reset_slot, e->next_obs, e->cplane, e->mtt, e->mtpar, e->lane_tmpl, e->tap_mask, e->tap_rank0, e->tap_perm, e->tap_ids};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (e->host_flags) (void)hipHostFree((void*)e->host_flags);
    delete e;
}

}  // extern "C"

// k_pregen is instantiated per level family so that a launch carries only that family's mission code, and per group
// width G (envs per wave = 64 / G; BBAI_PREGEN_GROUP, default 32: two envs per wave)
template <int G>
static void launch_pregen_g(const bbai_env* e, unsigned groups, bool listed /* false: dense -- every env, the whole grid works */, uint8_t* pending, const uint8_t* first_slot, int side) {
    unsigned long long* fails = e->flow + FLOW_GEN_FAILURES;
    const dim3 g((groups + 64 / G - 1) / (64 / G)), b(64);
    // Demand-sized groups only where a level is cheap (single rooms, <= 60 us per group): a maze level costs a group ~300 us,
    // and GoTo at 131 072 envs stalls the step stream with 4 entries per group (0.0534 vs 0.0385 ms per step,
    // profiles/r04/pregen_min_ab.jsonl) -- mazes keep the whole grid.
    const int min_groups = e->cfg.num_rows * e->cfg.num_cols > 1 ? 0 : e->pregen_min;
#define PREGEN_LAUNCH(KK, OO) hipLaunchKernelGGL((k_pregen<KK, G, OO>), g, b, 0, e->sides[side], e->cfg, e->n, e->next_rec, e->next_hot, e->mt, e->mti, e->gen_lists[side], \
                                                listed ? e->gen_counts[side] : nullptr, e->depth, pending, first_slot, fails, min_groups, e->pregen_per_group, e->next_obs)
    if (e->cfg.kind == K_LEVELGEN) { if (e->next_obs) PREGEN_LAUNCH(K_LEVELGEN, true); else PREGEN_LAUNCH(K_LEVELGEN, false); }
    else if (e->cfg.kind == K_BONUS) { if (e->next_obs) PREGEN_LAUNCH(K_BONUS, true); else PREGEN_LAUNCH(K_BONUS, false); }
    else { if (e->next_obs) PREGEN_LAUNCH(K_GOTO, true); else PREGEN_LAUNCH(K_GOTO, false); }
#undef PREGEN_LAUNCH
}
// k_pregen_lane lives in a translation unit of its own (bbai_genlane.hip: compiled without machine-CSE, see there)
static void launch_pregen_lane(const bbai_env* e, int64_t entries_hint, bool listed, uint8_t* pending, const uint8_t* first_slot, int side) {
    LaneLaunch a;
    a.cfg = e->cfg; a.n = e->n; a.next_rec = e->next_rec; a.next_hot = e->next_hot; a.mt = e->mt; a.mtt = e->mtt; a.mtpar = e->mtpar; a.mti = e->mti;
    a.gen_list = e->gen_lists[side]; a.gen_count = listed ? e->gen_counts[side] : nullptr; a.depth = e->depth; a.pending = pending; a.first_slot = first_slot;
    a.fails = e->flow + FLOW_GEN_FAILURES; a.next_obs = e->next_obs; a.tmpl = e->lane_tmpl; a.lane_words = e->lane_words;
    a.blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((entries_hint + 63) / 64, e->lane_blocks));
    a.stream = e->sides[side];
    bbai_lane_launch(a);
}
static void launch_pregen(const bbai_env* e, unsigned groups, bool listed, uint8_t* pending, const uint8_t* first_slot, int64_t entries_hint, int side = 0) {
    if (e->pregen_lane) { launch_pregen_lane(e, entries_hint, listed, pending, first_slot, side); return; }
    // Measured (profiles/r03/gen_rate_by_group_width.jsonl, pregen_group_width_in_bench.jsonl): levels per second of a bulk
    // fill 64 -> 32 -> 16 lanes per env: BossLevel 1 : 1.16 : 1.18, GoTo 1 : 1.13 : 1.17, PickupLoc 1 : 1.20 : 1.27,
    // GoToLocal 1 : 1.25 : 1.36; inside the step loop 32 is never behind 64 (GoToLocal 65 536 envs -3 %, PickupLoc 262 144
    // -6 %, GoTo 131 072 +-0) while 16 costs the step kernels of GoTo 131 072 9 % (fewer, fatter generator waves next to
    // them: 200 VGPRs and 20 KB of LDS each).
    if (e->pregen_group == 64) launch_pregen_g<64>(e, groups, listed, pending, first_slot, side);
    else if (e->pregen_group == 16) launch_pregen_g<16>(e, groups, listed, pending, first_slot, side);
    else launch_pregen_g<32>(e, groups, listed, pending, first_slot, side);
}

extern "C" {

static unsigned pregen_grid(const bbai_env* e, int64_t count_hint) {
    // one lane group per env, capped at pregen_cap groups in flight (the rest is reached by the groups' strides)
    int64_t g = std::min<int64_t>(count_hint, e->pregen_cap);
    return (unsigned)std::max<int64_t>(g, 1);
}

// A handle's launches are ordered by ONE caller stream at a time (plus the private look-ahead stream, which is tied to
// it by events).  A call that arrives on a DIFFERENT stream than the previous one is ordered behind everything the
// handle enqueued before.  Two ways to get the event that stream waits for (bbai_set_call_events):
//   off (default)  the event is recorded on the PREVIOUS stream at the moment of the switch -- free on the step path, but
//                  the previous stream must still exist then;
//   on             every call ends by recording the handle's completion event on its own stream and a switch only waits
//                  for it: the previous stream is never touched again (it may have been destroyed).  Measured cost
//                  (profiles/r03/call_events_ab.jsonl): +3.4 us per step at 65 536 GoToLocal envs (46.9 vs 43.5 us), +7 us
//                  per step + render at 131 072 pixel envs -- which is why it is opt-in.
static int enter_call(bbai_env* e, hipStream_t s) {
    if (e->host_flags && e->host_flags[0]) {         // (a plain read of pinned host memory: nothing is synchronised)
        snprintf(g_err, sizeof(g_err), "a window gate of this handle timed out waiting for a look-ahead refill (k_gate): its state is void -- "
                                       "re-seed it (bbai_seed) or destroy it");
        return BBAI_ERR_STATE;
    }
    if (e->have_stream && e->last_stream != s) {
        if (!e->call_events) HIP_TRY(hipEventRecord(e->ev_switch, e->last_stream));
        HIP_TRY(hipStreamWaitEvent(s, e->ev_switch, 0));
    }
    e->last_stream = s;
    e->have_stream = true;
    return BBAI_OK;
}
static int leave_call(bbai_env* e, hipStream_t s) {
    if (e->call_events) HIP_TRY(hipEventRecord(e->ev_switch, s));
    return BBAI_OK;
}
// enter_call / leave_call as a scope: once a call has entered, EVERY exit path -- also the error returns after work was
// already enqueued -- records the handle's completion event (call_events mode), so that a later call on another stream
// never waits for less than the handle really enqueued.  `return scope.leave();` on the success path reports a failing
// event record; the destructor covers the others.
struct CallScope {
    bbai_env* e; hipStream_t s; int rc; bool open;
    CallScope(bbai_env* e_, hipStream_t s_) : e(e_), s(s_), rc(enter_call(e_, s_)), open(false) { open = rc == BBAI_OK; }
    int leave() { open = false; return leave_call(e, s); }
    ~CallScope() { if (open) (void)leave_call(e, s); }
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;
};

// bbai_profile: bracket a launch with an event pair (ring of PROF_RING pairs per kernel; a pair is folded into the sums
// when its slot comes round again, i.e. long after it completed, or when the totals are read)
static void prof_fold(bbai_env* e, int k, int i) {
    bbai_env::ProfSlot& p = e->prof[k][i];
    if (!p.used) return;
    float ms = 0;
    if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { e->prof_ms[k] += ms; e->prof_n[k]++; }
    p.used = false;
}
struct ProfScope {
    bbai_env* e; int k; hipStream_t s; int i;
    ProfScope(bbai_env* e_, int k_, hipStream_t s_) : e(e_), k(k_), s(s_), i(-1) {
        if (!e->prof_on) return;
        i = e->prof_pos[k];
        e->prof_pos[k] = (i + 1) % PROF_RING;
        prof_fold(e, k, i);
        if (!e->prof[k][i].a) { (void)hipEventCreate(&e->prof[k][i].a); (void)hipEventCreate(&e->prof[k][i].b); }
        (void)hipEventRecord(e->prof[k][i].a, s);
    }
    ~ProfScope() {
        if (i < 0) return;
        (void)hipEventRecord(e->prof[k][i].b, s);
        e->prof[k][i].used = true;
    }
};

// A consume-tick (one reset() or one auto-resetting step) in three parts: window_begin -- at the first tick of a window k_gate holds
// the stream until every env is sure to find B ready levels (NWIN above); the consume itself -- k_consume over the reset list,
// or, fused, inside k_step (which therefore has to be launched AFTER window_begin); window_end -- the mission tokens of the new
// episodes and, at the last tick of a window, the window's close and ONE refill launch on the look-ahead stream for everything it consumed.
struct TickPos { int wb, pos; };
static TickPos tick_pos(const bbai_env* e) {
    const int64_t w = e->tick / e->period;           // window of this consume-tick
    return {(int)(w % NWIN), (int)(e->tick % e->period)};      // its buffer (pending / first_slot / meta), its place in the window
}
// Do kernels of stream `s` and of the look-ahead stream run concurrently?  Asked once per caller stream, at the first window it opens (a handle
// follows one stream at a time; up to 8 are remembered, a ninth is simply probed again).  Costs one stream synchronisation.
static int probe_stream(bbai_env* e, hipStream_t s, bool* ok) {
    for (int k = 0; k < e->n_probed; ++k) if (e->probed[k] == s) { *ok = e->probed_ok[k]; return BBAI_OK; }
    bool verdict = true;
    if (e->gate_probe == 2) verdict = false;
    else if (e->gate_probe != 0) {
        e->host_flags[1] = 0;
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, s, e->flow, 0ull);
        hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, s, e->flow, e->host_flags_dev + 1);
        // (the releasing kernel sits behind an event of EVERY look-ahead stream, on the stream k_mark uses: all of them have to run beside the caller's)
        for (int k = 0; k < e->n_sides && e->n_sides > 1; ++k) {
            hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, e->sides[k], e->flow, 0ull);
            HIP_TRY(hipEventRecord(e->ev_part[k], e->sides[k]));
            HIP_TRY(hipStreamWaitEvent(e->mark_stream, e->ev_part[k], 0));
        }
        hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, e->n_sides > 1 ? e->mark_stream : e->side, e->flow, 1ull);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        verdict = e->host_flags[1] == 1;
    }
    const int at = e->n_probed < 8 ? e->n_probed++ : 7;
    e->probed[at] = s; e->probed_ok[at] = verdict;
    *ok = verdict;
    return BBAI_OK;
}
static int window_begin(bbai_env* e, hipStream_t s) {
    if (e->tick % e->period == 0) {
        const int64_t x = e->tick / e->period;
        bool strict = e->gate_strict != 0;
        if (!strict) {
            bool ok = true;
            { int rc = probe_stream(e, s, &ok); if (rc != BBAI_OK) return rc; }
            e->gate_forced = ok ? 0 : 1;
            strict = !ok;
        }
        if (strict && x >= 2) HIP_TRY(hipStreamWaitEvent(s, e->ev_refill[(x - 2) % NWIN], 0));
        hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, s, e->flow, e->win_meta, (unsigned long long)(e->tick / e->period), e->period, e->host_flags_dev);
        HIP_TRY(hipGetLastError());
    }
    return BBAI_OK;
}
static int window_end(bbai_env* e, hipStream_t s, int tokens_mode /* k_tokens: 0 list, 1 every env, 2 by `dones` */, const uint8_t* dones,
                      int ticks = 1 /* consume-ticks the launch in front took (bbai_rollout: up to the window's last; no token buffer then) */) {
    const int B = e->period;
    const TickPos tp = tick_pos(e);
    const int wb = tp.wb, pos = tp.pos + ticks - 1;         // (the launch's LAST tick: all of them lie in this window)
    if (e->tokens) {
        const int64_t hint = tokens_mode ? e->n : std::max<int64_t>(e->n / 64, 64);
        hipLaunchKernelGGL(k_tokens, dim3((unsigned)std::min<int64_t>((hint + 63) / 64, 4096)), dim3(64), 0, s, e->cfg, e->n, e->rec, e->inplace ? e->next_rec : nullptr, e->depth, e->hot,
                           e->tokens, e->reset_list, e->counters + 16 * e->step_parity, tokens_mode, dones);
    }
    if (tokens_mode == 0) {                  // (unfused: the ping-pong list counters)
        e->step_parity ^= 1;
        e->next_counter_clean = true;
    }
    HIP_TRY(hipGetLastError());
    if (pos == B - 1) {
        // Window end, on the look-ahead stream: the window's work list (k_compact), one refill launch for everything consumed in it, the mark behind it.
        const int64_t w = e->tick / B;
        HIP_TRY(hipEventRecord(e->ev_consumed, s));
        const int64_t nb = (e->n + 63) / 64;
        const int S = (int)std::max<int64_t>(1, std::min<int64_t>(e->n_sides, nb / 4));          // (a stream's range: at least 256 envs; the streams beyond S idle)
        for (int k = 0; k < S; ++k) {
            const int64_t w0 = nb * k / S, w1 = nb * (k + 1) / S, ne = std::min<int64_t>(e->n, w1 * 64) - w0 * 64;
            hipStream_t ss = e->sides[k];
            HIP_TRY(hipStreamWaitEvent(ss, e->ev_consumed, 0));
            HIP_TRY(hipMemsetAsync(e->gen_counts[k], 0, SHARDS * GEN_COUNT_U32 * 4, ss));
            hipLaunchKernelGGL(k_compact, dim3((unsigned)((w1 - w0 + 3) / 4)), dim3(256), 0, ss, e->n, e->pending + (size_t)wb * e->n, e->gen_lists[k], e->gen_counts[k], w0, w1 - w0);
            const int64_t rh = std::max<int64_t>((int64_t)B * (ne / 64), 64);
            launch_pregen(e, (unsigned)std::max<int64_t>(1, std::min<int64_t>(rh, e->pregen_cap / S)), true, e->pending + (size_t)wb * e->n, e->first_slot + (size_t)wb * e->n,
                          std::min<int64_t>(ne, 2 * rh), k);
            if (e->n_sides > 1) {
                HIP_TRY(hipEventRecord(e->ev_part[k], ss));
                HIP_TRY(hipStreamWaitEvent(e->mark_stream, e->ev_part[k], 0));
            }
        }
        hipStream_t ms = e->n_sides > 1 ? e->mark_stream : e->side;
        hipLaunchKernelGGL(k_mark, dim3(1), dim3(1), 0, ms, e->flow, (unsigned long long)(w + 1));
        HIP_TRY(hipEventRecord(e->ev_refill[wb], ms));
        HIP_TRY(hipGetLastError());
    }
    e->tick += ticks;
    return BBAI_OK;
}
// main stream: slots -> live state (+ first obs) by k_consume; side stream: refill the consumed slots.
static int consume_and_refill(bbai_env* e, hipStream_t s, uint8_t* image, uint8_t* dirs, int all) {
    { int rc = window_begin(e, s); if (rc != BBAI_OK) return rc; }
    const TickPos tp = tick_pos(e);
    const int wb = tp.wb;
    const int64_t hint = all ? e->n : std::max<int64_t>(e->n / 64, 64);
    {
    ProfScope prof_(e, 1, s);
    hipLaunchKernelGGL(k_consume, dim3((unsigned)std::min<int64_t>((hint + 3) / 4, 8192)), dim3(256), 0, s, e->cfg, e->n, e->rec,
                       e->hot, e->stale, e->next_rec, e->next_hot, e->vhead, e->vset, e->reset_list, e->reset_slot, e->counters + 16 * e->step_parity, all,
                       e->totals, e->depth, e->pending + (size_t)wb * e->n, e->first_slot + (size_t)wb * e->n,
                       e->win_meta + (size_t)wb * META_U32, image, dirs,
                       e->counters + 16 * (e->step_parity ^ 1), e->step_prio, e->vplane, e->fcache, e->lsm, e->inplace, e->cplane);
    }
    return window_end(e, s, all ? 1 : 0, nullptr);
}

int bbai_seed(bbai_env* e, const uint64_t* seeds, int64_t n) {
    if (!e || !seeds || n != e->n) { snprintf(g_err, sizeof(g_err), "seed: need exactly n_envs seeds"); return BBAI_ERR_ARG; }
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());             // nothing of an earlier run may still be in flight on either stream
    // 8 bytes per env cross PCIe; sha512 + init_by_array run per lane (k_seed).  Staged in a buffer of its own (parked in the
    // verifier's obj_set area, a re-seed of a live handle followed by a hot-only import left every env's verifier sets zeroed).
    uint64_t* seeds_dev = nullptr;
    HIP_TRY(hipMalloc((void**)&seeds_dev, (size_t)n * 8));
    struct Free { void* p; ~Free() { (void)hipFree(p); } } free_seeds{seeds_dev};
    HIP_TRY(hipMemcpy(seeds_dev, seeds, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_seed, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, e->side, n, seeds_dev, e->mt, e->mti);
    hipLaunchKernelGGL(k_init_hot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, e->side, n, e->hot, e->next_hot, e->stale, e->depth);
    // fill every env's look-ahead ring with the first D levels of its stream (slot order == stream order).  EVERY
    // window buffer starts clean: a re-seed may land in the middle of a window that was using any of them.
    HIP_TRY(hipMemsetAsync(e->pending, 0, NWIN * (size_t)n, e->side));
    HIP_TRY(hipMemsetAsync(e->first_slot, 0, NWIN * (size_t)n, e->side));
    HIP_TRY(hipMemsetAsync(e->pending, e->depth - e->inplace, (size_t)n, e->side));      // (in-place: slot depth - 1 is the live one -- empty until the first reset)
    if (e->mtpar) HIP_TRY(hipMemsetAsync(e->mtpar, 0, (size_t)n, e->side));      // (position 624 of generation 0: the first draw's twist fills the other half)
    launch_pregen(e, (unsigned)std::max<int64_t>(1, std::min<int64_t>(n, std::max(e->pregen_cap, 32768))), false, e->pending, e->first_slot, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemsetAsync(e->win_meta, 0, NWIN * META_U32 * 4, e->side));
    HIP_TRY(hipMemsetAsync(e->counters, 0, 128, e->side));
    {   // the refill count starts over (window numbering restarts with tick 0); the reset total, the give-up and the time-out counts keep running
        HIP_TRY(hipDeviceSynchronize());
        const unsigned long long zero = 0;
        HIP_TRY(hipMemcpy(e->flow + FLOW_REFILLED, &zero, 8, hipMemcpyHostToDevice));
    }
    for (int k = 0; k < NWIN; ++k) HIP_TRY(hipEventRecord(e->ev_refill[k], e->side));
    e->tick = 0;
    e->step_parity = 0;
    e->next_counter_clean = true;
    HIP_TRY(hipDeviceSynchronize());
    e->host_flags[0] = 0;            // every ring slot was just regenerated: a handle voided by a gate time-out is whole again
    e->seeded = true;
    e->live = false;
    return BBAI_OK;
}

int bbai_reset(bbai_env* e, uint8_t* image, uint8_t* dirs, void* stream) {
    if (!e || !image || !dirs) ARG_FAIL("null handle or output buffer");
    if (!e->seeded) { snprintf(g_err, sizeof(g_err), "reset before seed"); return BBAI_ERR_STATE; }
    ON_DEVICE(e->device);
    hipStream_t s = (hipStream_t)stream;
    CallScope call(e, s);
    if (call.rc != BBAI_OK) return call.rc;
    int rc = consume_and_refill(e, s, image, dirs, 1);
    if (rc != BBAI_OK) return rc;
    e->live = true;
    return call.leave();
}

// k_step (+ the consume / refill of the envs it finished) on stream s; the caller has entered the call
static bool use_fused_consume(const bbai_env* e) {
    // option "consume_fused" / BBAI_CONSUME_FUSED: 1 = the stepping wave consumes its finished envs itself, 0 = k_consume launch,
    // -1 (default) = by level family.
    if (e->consume_fused >= 0) return e->consume_fused != 0;
    // Measured (profiles/r04/consume_fused_ab.jsonl, ms per step unfused -> fused): the mazes win -- GoTo 131 072 envs 0.046 -> 0.041,
    // GoTo 1 048 576 0.245 -> 0.211, BossLevel encoded 1 048 576 0.137 -> 0.125, pixels 1.718 -> 1.710 -- their episodes last hundreds of steps,
    // few waves have a finished env and one launch per step disappears; the single rooms lose -- GoToLocal 65 536 0.035 -> 0.054, PickupLoc
    // 262 144 0.085 -> 0.143 -- 2 % of the envs finish on every step, so most waves carry a reset or two and do them one after the
    // other behind their own step, where k_consume's waves do them all at once.
    return e->cfg.num_rows * e->cfg.num_cols > 1;
}
// One step in three parts, so that bbai_step_render can put the kernel of the batch's second half on another stream:
//   step_prepare  the window gate (fused: the slots this step's waves consume must be there) + what the kernel needs to know about the window
//   step_kernel   k_step over the 64-env blocks [block0, block0 + nblocks) on stream `ks`
//   step_finish   k_consume (unfused) / mission tokens / the window's close + refill -- on the caller's stream, behind EVERY k_step of the step
struct StepPlan { FuseArgs fa; bool fused; uint32_t* counter; TapArgs tap = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0}; };
static int step_prepare(bbai_env* e, int auto_reset, hipStream_t s, StepPlan& p) {
    p.fused = auto_reset && (e->inplace || use_fused_consume(e));
    p.counter = e->counters + 16 * e->step_parity;
    memset(&p.fa, 0, sizeof(p.fa));
    FuseArgs& fa = p.fa;
    if (p.fused) {
        { int rc = window_begin(e, s); if (rc != BBAI_OK) return rc; }        // the slots this step's waves consume have landed
        const TickPos tp = tick_pos(e);
        fa.next_recs = e->next_rec; fa.next_hots = e->next_hot; fa.next_obs = e->next_obs; fa.depth = e->depth;
        fa.pending = e->pending + (size_t)tp.wb * e->n; fa.first_slot = e->first_slot + (size_t)tp.wb * e->n;
        fa.win_meta = e->win_meta + (size_t)tp.wb * META_U32;
        fa.totals = e->totals;
        e->next_counter_clean = false;      // (a later unfused step clears its ping-pong counter itself)
    } else if (e->inplace) {                // (no auto-reset: the kernel still finds the live records through the ring)
        fa.next_recs = e->next_rec; fa.depth = e->depth;
        if (!e->next_counter_clean) HIP_TRY(hipMemsetAsync(p.counter, 0, 4, s));
        e->next_counter_clean = false;
    } else {
        if (!e->next_counter_clean) HIP_TRY(hipMemsetAsync(p.counter, 0, 4, s));   // (k_consume of the previous step zeroes it)
        e->next_counter_clean = false;
    }
    return BBAI_OK;
}
static int step_kernel(bbai_env* e, const StepPlan& p, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64,
                       uint8_t* dones, int auto_reset, hipStream_t ks, int enum_done, int64_t block0, int64_t nblocks, int ticks = 1) {
    ProfScope prof_(e, 0, ks);
    if (e->prof_on) e->prof_step_ticks += ticks;
    StepArgs a;
    a.c = e->cfg; a.n = e->n; a.recs = e->rec; a.hots = e->hot; a.stales = e->stale; a.vheads = e->vhead; a.vsets = e->vset; a.actions = actions; a.image = image; a.dirs = dirs;
    a.rewards = rewards; a.rewards64 = rewards64; a.dones = dones; a.auto_reset = auto_reset; a.reset_list = e->reset_list; a.reset_slot = e->reset_slot; a.counters = p.counter;
    a.prio = e->step_prio; a.vplane = e->vplane; a.fcache = e->fcache; a.lsm_arr = e->lsm; a.enum_done = enum_done; a.fuse = p.fa; a.block0 = block0; a.cplane = e->cplane;
    a.tap = p.tap; a.ticks = ticks;
#define STEP_LAUNCH(VV, FF, CC) do { if (ticks > 1) hipLaunchKernelGGL((k_step_ticks<VV, FF, CC>), dim3((unsigned)nblocks), dim3(STEP_BLOCK), 0, ks, a); \
                                     else hipLaunchKernelGGL((k_step<VV, FF, CC>), dim3((unsigned)nblocks), dim3(STEP_BLOCK), 0, ks, a); } while (0)
    if (e->inplace && e->cplane) STEP_LAUNCH(false, 3, true);
    else if (e->inplace) STEP_LAUNCH(false, 3, false);
    else if (p.fused) { if (e->vplane) STEP_LAUNCH(true, 1, false); else STEP_LAUNCH(false, 1, false); }
    else { if (e->vplane) STEP_LAUNCH(true, 0, false); else STEP_LAUNCH(false, 0, false); }
#undef STEP_LAUNCH
    HIP_TRY(hipGetLastError());
    return BBAI_OK;
}
static int step_finish(bbai_env* e, const StepPlan& p, uint8_t* image, uint8_t* dirs, const uint8_t* dones, int auto_reset, hipStream_t s) {
    // the number of finished envs is only known on the device: fixed grids, device-side count
    if (auto_reset) return p.fused ? window_end(e, s, 2, dones) : consume_and_refill(e, s, image, dirs, 0);
    return BBAI_OK;
}
static int64_t step_blocks(const bbai_env* e) { return (e->n + STEP_BLOCK - 1) / STEP_BLOCK; }
struct TapRows { uint8_t* image_out; uint8_t* dir_out; double* rew_out; uint8_t* done_out; };
static int step_launch(bbai_env* e, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64,
                       uint8_t* dones, int auto_reset, hipStream_t s, int enum_done, const TapRows* rows = nullptr) {
    StepPlan p;
    { int rc = step_prepare(e, auto_reset, s, p); if (rc != BBAI_OK) return rc; }
    // Inside the stepping waves wherever the step leaves the final outputs behind (fused consume, in-place layout, no auto-reset);
    // an unfused auto-resetting step writes the new episodes' first observations in k_consume: the tap is then a launch behind it.
    const bool in_kernel = rows && (p.fused || e->inplace || !auto_reset);
    if (in_kernel) p.tap = TapArgs{e->tap_mask, e->tap_rank0, e->tap_perm, rows->image_out, rows->dir_out, rows->rew_out, rows->done_out, e->tap_count};
    { int rc = step_kernel(e, p, actions, image, dirs, rewards, rewards64, dones, auto_reset, s, enum_done, 0, step_blocks(e)); if (rc != BBAI_OK) return rc; }
    { int rc = step_finish(e, p, image, dirs, dones, auto_reset, s); if (rc != BBAI_OK) return rc; }
    if (rows && !in_kernel) {
        hipLaunchKernelGGL(k_tap, dim3((unsigned)std::min<int64_t>((e->tap_count * OBS_BYTES + 255) / 256, 2048)), dim3(256), 0, s, e->tap_count, (int64_t)0, e->tap_ids, image, dirs,
                           rewards64, dones, (const uint8_t*)nullptr, rows->image_out, rows->dir_out, rows->rew_out, rows->done_out, (uint8_t*)nullptr);
        HIP_TRY(hipGetLastError());
    }
    return BBAI_OK;
}

int bbai_step(bbai_env* e, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64,
              uint8_t* dones, int auto_reset, void* stream) {
    if (!e || !actions || !image || !dirs || !rewards || !dones) ARG_FAIL("null handle or buffer");
    if (!e->live) { snprintf(g_err, sizeof(g_err), "step before reset"); return BBAI_ERR_STATE; }
    if (auto_reset && !e->seeded) {     // live through import_state only: there is no level stream to reset from
        snprintf(g_err, sizeof(g_err), "auto-reset step before seed");
        return BBAI_ERR_STATE;
    }
    ON_DEVICE(e->device);
    hipStream_t s = (hipStream_t)stream;
    CallScope call(e, s);
    if (call.rc != BBAI_OK) return call.rc;
    { int rc = step_launch(e, actions, image, dirs, rewards, rewards64, dones, auto_reset, s, e->done_action_enum); if (rc != BBAI_OK) return rc; }
    return call.leave();
}

int bbai_step_tap_set(bbai_env* e, const int64_t* ids, int64_t count) {
    if (!e || count < 0 || (count && !ids)) ARG_FAIL("null handle or id list");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    void* old[] = {e->tap_mask, e->tap_rank0, e->tap_perm, e->tap_ids};
    for (void* q : old) if (q) (void)hipFree(q);
    e->tap_mask = nullptr; e->tap_rank0 = nullptr; e->tap_perm = nullptr; e->tap_ids = nullptr; e->tap_count = 0;
    if (count == 0) return BBAI_OK;
    const int64_t nb = (e->n + STEP_BLOCK - 1) / STEP_BLOCK;
    std::vector<unsigned long long> mask((size_t)nb, 0ull);
    std::vector<uint32_t> rank0((size_t)nb, 0u);
    std::vector<int32_t> perm((size_t)count);
    std::vector<std::pair<int64_t, int32_t>> order((size_t)count);
    for (int64_t k = 0; k < count; ++k) {
        if (ids[k] < 0 || ids[k] >= e->n) ARG_FAIL("env id out of range");
        if (mask[(size_t)(ids[k] / STEP_BLOCK)] >> (ids[k] % STEP_BLOCK) & 1ull) ARG_FAIL("an env is listed twice");
        mask[(size_t)(ids[k] / STEP_BLOCK)] |= 1ull << (ids[k] % STEP_BLOCK);
        order[(size_t)k] = {ids[k], (int32_t)k};
    }
    std::sort(order.begin(), order.end());
    for (int64_t k = 0; k < count; ++k) perm[(size_t)k] = order[(size_t)k].second;      // k-th listed env in ascending order -> its log row
    uint32_t run = 0;
    for (int64_t b = 0; b < nb; ++b) { rank0[(size_t)b] = run; run += (uint32_t)__builtin_popcountll(mask[(size_t)b]); }
    HIP_TRY(hipMalloc((void**)&e->tap_mask, (size_t)nb * 8));
    HIP_TRY(hipMalloc((void**)&e->tap_rank0, (size_t)nb * 4));
    HIP_TRY(hipMalloc((void**)&e->tap_perm, (size_t)count * 4));
    HIP_TRY(hipMalloc((void**)&e->tap_ids, (size_t)count * 8));
    HIP_TRY(hipMemcpy(e->tap_mask, mask.data(), (size_t)nb * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->tap_rank0, rank0.data(), (size_t)nb * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->tap_perm, perm.data(), (size_t)count * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->tap_ids, ids, (size_t)count * 8, hipMemcpyHostToDevice));
    e->tap_count = count;
    return BBAI_OK;
}

int bbai_step_tapped(bbai_env* e, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64, uint8_t* dones,
                     int auto_reset, uint8_t* image_out, uint8_t* dir_out, double* reward64_out, uint8_t* done_out, void* stream) {
    if (!e || !actions || !image || !dirs || !rewards || !rewards64 || !dones || !image_out || !dir_out || !reward64_out || !done_out) ARG_FAIL("null handle or buffer");
    if (!e->tap_count) { snprintf(g_err, sizeof(g_err), "bbai_step_tapped before bbai_step_tap_set"); return BBAI_ERR_STATE; }
    if (!e->live) { snprintf(g_err, sizeof(g_err), "step before reset"); return BBAI_ERR_STATE; }
    if (auto_reset && !e->seeded) { snprintf(g_err, sizeof(g_err), "auto-reset step before seed"); return BBAI_ERR_STATE; }
    ON_DEVICE(e->device);
    hipStream_t s = (hipStream_t)stream;
    CallScope call(e, s);
    if (call.rc != BBAI_OK) return call.rc;
    const TapRows rows = {image_out, dir_out, reward64_out, done_out};
    { int rc = step_launch(e, actions, image, dirs, rewards, rewards64, dones, auto_reset, s, e->done_action_enum, &rows); if (rc != BBAI_OK) return rc; }
    return call.leave();
}

int bbai_set_atlas(bbai_env* e, const uint8_t* tiles, int n_tiles, const uint8_t* lut) {
    if (!e || !tiles || !lut || n_tiles < 1 || n_tiles > MAX_TILES) ARG_FAIL("null pointer or tile count out of range");
    ON_DEVICE(e->device);
    HIP_TRY(hipMemcpy(e->atlas, tiles, (size_t)n_tiles * TILE_BYTES, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->lut, lut, 512, hipMemcpyHostToDevice));
    e->n_tiles = n_tiles;
    return BBAI_OK;
}

}  // extern "C"

// Time-paced tickets (option "render_pace", off by default).  k_render_q at the rate of ONE ticket counter (11.4 ns per 8-env ticket
// = 6.6 TB/s) beats every unpaced shape because the counter PACES the chip's stores; a wall-clock gate over two counters can set the
// pace anywhere.  Measured in the step loop on seven boxes (profiles/r04/render_pace_*.jsonl, NOTES section 1): slower than the optimum
// the launch is pace-bound (131 072 tickets x pace), faster it falls off a shallow cliff; at the optimum it beat the one counter by
// 1.5-2.3 % on three boxes (k_render 1.466-1.482 vs 1.50 ms: 0.85 of peak) and LOST 0.5-1.5 % at every pace on two others; the optimum
// moved from 10.8 to 11.4 ns between boxes and between two processes on one box, an idle chip puts it elsewhere than the loop does
// (a first-render tuner), and a perturb-and-observe controller on the launches' own durations paid more for its event pairs and its
// hovering than it gained.  The one counter's 1.50 ms is the same on all of them: it ships; the gate stays as a knob.
static int render_launch(bbai_env* e, const uint8_t* input, uint8_t* pixels, void* stream, int64_t n_render = -1 /* envs input / pixels hold (default: the batch); the shape follows the BATCH size */) {
    CallScope call(e, (hipStream_t)stream);
    if (call.rc != BBAI_OK) return call.rc;
    const int64_t nr = n_render < 0 ? e->n : n_render;
    {
    ProfScope prof_(e, 2, (hipStream_t)stream);
    // From 262 144 envs up: k_render_q -- ONE persistent 1024-thread block per CU, 8-env groups handed out by ONE ticket
    // counter.  Everything below was measured inside the step loop, settings alternated in one process (tools/ab.py),
    // 1 048 576 BossLevel envs, k_render ms per launch (profiles/r04/render_queue_*.jsonl; four leases = four boxes):
    //   one-shot (1024, 8) blocks (rounds 2-3)                         1.64 - 1.72
    //   queue, two blocks per CU, one counter (round 3's lead)         1.60 - 1.61
    //   queue, ONE block per CU (256 blocks), one counter              1.50            <- shipped: 6.67 TB/s, 0.83 of peak
    //   ... 192 / 224 / 320 / 512 blocks                               1.66 / 1.53 / 1.51 / 1.64;   128: 2.14
    //   ... two / eight interleaved counters                           1.56 / 1.69 - 1.77
    //   ... 2 groups per ticket, 12- / 16-env groups                   1.60 / 1.55 - 1.58 / 1.61
    //   ... 512- / 256-thread blocks (8-env groups)                    1.61 - 1.63 / 1.57 (512 blocks)
    // One counter serves ~88 M tickets/s (131 072 tickets = 1.49 ms): the shipped shape runs AT the ticket rate, and every
    // way of relieving it (more counters, bigger tickets) is slower -- the single counter is what keeps the chip's stores one
    // compact, evenly paced window.  By batch size (queue vs the one-shot shape of that size, ms per step): 131 072 envs
    // 0.221 vs 0.214, 262 144 0.414 vs 0.418, 393 216 0.614 vs 0.639, 524 288 0.809 vs 0.848 -- below 262 144 the whole
    // encoding is still in the memory-side cache and short-lived (512, 2) blocks win.
    // BBAI_RENDER_QUEUE / option "render_queue": -1 = by batch size (default), 0 = never, m > 0 = queue shape m of the table below.
    const bool big = e->n >= 786432;
    int qm = e->render_queue;
    int pace = e->render_pace;
    if (qm < 0) {
        qm = e->n >= RENDER_QUEUE_MIN_ENVS ? RENDER_QUEUE_DEFAULT : 0;
        if (qm == RENDER_QUEUE_DEFAULT && pace > 0) qm = RENDER_QUEUE_PACED;        // (a pace needs tickets served faster than it admits them: two counters)
    }
    if (pace < 0) pace = 0;
    if (qm > 0) {
        const int cus = e->n_cus > 0 ? e->n_cus : 256;
#define RENDER_Q(GG, TT, NC, KK) do { \
            const int64_t tickets = ((nr + GG - 1) / GG + KK - 1) / KK; \
            const int64_t want = e->render_queue_blocks > 0 ? e->render_queue_blocks : (e->render_queue_bpc > 0 ? (int64_t)cus * e->render_queue_bpc : (int64_t)cus * 1024 / TT); \
            const unsigned blocks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, tickets)); \
            hipLaunchKernelGGL((k_render_q<GG, TT, NC, KK>), dim3(blocks), dim3(TT), 0, (hipStream_t)stream, nr, input, pixels, \
                               e->atlas, e->lut, e->n_tiles, e->render_tickets, pace); } while (0)
        switch (qm) {          // (shapes other than 1 stay for measurements: tests/test_gpu_parity.py checks every one byte for byte)
        default:
        case 1: RENDER_Q(8, 1024, 1, 1); break;       // shipped
        case 2: RENDER_Q(8, 1024, 8, 1); break;       // eight / two interleaved counters
        case 3: RENDER_Q(8, 1024, 2, 1); break;
        case 4: RENDER_Q(8, 1024, 1, 2); break;       // two groups per ticket
        case 5: RENDER_Q(12, 1024, 1, 1); break;      // bigger groups
        case 6: RENDER_Q(16, 1024, 1, 1); break;
        case 7: RENDER_Q(8, 512, 1, 1); break;        // smaller blocks
        case 8: RENDER_Q(8, 256, 1, 1); break;
        case 9: RENDER_Q(2, 512, 8, 1); break;        // small groups need more counters (524 288 tickets)
        case 10: RENDER_Q(9, 1024, 1, 1); break;      // just above the shipped group: a little more work per ticket
        case 11: RENDER_Q(10, 1024, 1, 1); break;
        }
#undef RENDER_Q
        HIP_TRY(hipGetLastError());
        return call.leave();
    }
    // Below: ONE G-env group per one-shot block of T threads, (512, 2) (round 2: profiles/r02/render_shape_*.jsonl; (1024, 8) from
    // 786 432 envs when the queue is switched off).  BBAI_RENDER_GROUP / BBAI_RENDER_TPB override (experiments).
    int G = e->render_group, T = e->render_tpb;
    if (G != 2 && G != 4 && G != 8) G = big ? 8 : 2;
    if (T != 256 && T != 512 && T != 1024) T = big ? 1024 : 512;
    const dim3 grid((unsigned)((nr + G - 1) / G));
#define RENDER_LAUNCH(GG, TT) hipLaunchKernelGGL((k_render<GG, TT>), grid, dim3(TT), 0, (hipStream_t)stream, nr, input, pixels, e->atlas, e->lut, e->n_tiles)
#define RENDER_G(GG) do { if (T == 1024) RENDER_LAUNCH(GG, 1024); else if (T == 512) RENDER_LAUNCH(GG, 512); else RENDER_LAUNCH(GG, 256); } while (0)
    if (G == 2) RENDER_G(2); else if (G == 4) RENDER_G(4); else RENDER_G(8);
#undef RENDER_G
#undef RENDER_LAUNCH
    }
    HIP_TRY(hipGetLastError());
    return call.leave();
}

extern "C" {

int bbai_render(bbai_env* e, const uint8_t* image, uint8_t* pixels, void* stream) {
    if (!e || !image || !pixels) ARG_FAIL("null handle or buffer");
    if (e->n_tiles <= 0) { snprintf(g_err, sizeof(g_err), "render before set_atlas"); return BBAI_ERR_STATE; }
    ON_DEVICE(e->device);
    return render_launch(e, image, pixels, stream);
}

}  // extern "C"

// step + render of a pixel batch as ONE call.  The render is a pure store stream at the chip's fill rate (k_render_q: 1.50 ms per
// 1 048 576 envs) and k_step (0.10 ms) runs in FRONT of it with the store pipes idle -- 6 % of the headline step.  Split (option
// "step_render_split" = 1): the batch is stepped in two halves; the second half's k_step goes to a stream of its own and runs UNDER the
// first half's render (the render's blocks hold 56 VGPRs and 20 KB of LDS per CU: room for k_step's waves next to them):
//     caller's stream:  gate . k_step(A) . k_render(A) ........ [wait B] . tokens / window close . k_render(B)
//     split stream:             k_step(B) ........
// Same kernels, same bytes (tests/test_gpu_parity.py::test_step_render_split_*).  MEASURED SLOWER at every size, settings alternated in
// one process (profiles/r05/step_render_split_ab.jsonl, ms per step unsplit / split): 1 048 576 envs 1.599 / 1.609, 524 288 0.806 / 0.808,
// 262 144 0.413 / 0.423, 131 072 0.210 / 0.222 -- the two half renders together cost more than the one (0.7565 x 2 vs 1.5007 ms: the
// store stream's pacing restarts) and the joins are not free; k_step under the render gains less than that.  Hence off by default; the
// entry point stays: it is the wrapped env's step() as one call.  Only with the fused / in-place consume (a k_consume launch would need
// both halves).
constexpr int64_t STEP_RENDER_SPLIT_MIN = 262144;
#ifndef STEP_RENDER_SPLIT_DEFAULT
#define STEP_RENDER_SPLIT_DEFAULT 0
#endif
static int step_render_launch(bbai_env* e, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64, uint8_t* dones,
                              int auto_reset, uint8_t* pixels, hipStream_t s, int enum_done) {
    StepPlan p;
    {
        CallScope call(e, s);
        if (call.rc != BBAI_OK) return call.rc;
        { int rc = step_prepare(e, auto_reset, s, p); if (rc != BBAI_OK) return rc; }
        const int want = e->step_render_split < 0 ? (STEP_RENDER_SPLIT_DEFAULT && e->n >= STEP_RENDER_SPLIT_MIN) : e->step_render_split;      // (an explicit 1: any size)
        const int64_t nb = step_blocks(e), hb = nb / 2;
        const bool split = want && pixels && (p.fused || !auto_reset) && hb > 0;
        if (!split) {
            { int rc = step_kernel(e, p, actions, image, dirs, rewards, rewards64, dones, auto_reset, s, enum_done, 0, nb); if (rc != BBAI_OK) return rc; }
            { int rc = step_finish(e, p, image, dirs, dones, auto_reset, s); if (rc != BBAI_OK) return rc; }
            { int rc = call.leave(); if (rc != BBAI_OK) return rc; }
            return pixels ? render_launch(e, image, pixels, s) : BBAI_OK;
        }
        const int64_t na = hb * STEP_BLOCK;                   // envs of the first half (a multiple of every render group size)
        // the second half's k_step, on the split stream, behind everything the caller's stream holds so far (the gate included)
        HIP_TRY(hipEventRecord(e->ev_split0, s));
        HIP_TRY(hipStreamWaitEvent(e->split, e->ev_split0, 0));
        { int rc = step_kernel(e, p, actions, image, dirs, rewards, rewards64, dones, auto_reset, e->split, enum_done, hb, nb - hb); if (rc != BBAI_OK) return rc; }
        HIP_TRY(hipEventRecord(e->ev_splitB, e->split));
        { int rc = step_kernel(e, p, actions, image, dirs, rewards, rewards64, dones, auto_reset, s, enum_done, 0, hb); if (rc != BBAI_OK) return rc; }
        { int rc = call.leave(); if (rc != BBAI_OK) return rc; }
        { int rc = render_launch(e, image, pixels, s, na); if (rc != BBAI_OK) return rc; }
        HIP_TRY(hipStreamWaitEvent(s, e->ev_splitB, 0));
        {
            CallScope call2(e, s);
            if (call2.rc != BBAI_OK) return call2.rc;
            { int rc = step_finish(e, p, image, dirs, dones, auto_reset, s); if (rc != BBAI_OK) return rc; }
            { int rc = call2.leave(); if (rc != BBAI_OK) return rc; }
        }
        return render_launch(e, image + na * OBS_BYTES, pixels + na * (int64_t)PIX_BYTES, s, e->n - na);
    }
}

extern "C" {

int bbai_step_render(bbai_env* e, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64,
                     uint8_t* dones, int auto_reset, uint8_t* pixels, void* stream) {
    if (!e || !actions || !image || !dirs || !rewards || !dones || !pixels) ARG_FAIL("null handle or buffer");
    if (!e->live) { snprintf(g_err, sizeof(g_err), "step before reset"); return BBAI_ERR_STATE; }
    if (auto_reset && !e->seeded) { snprintf(g_err, sizeof(g_err), "auto-reset step before seed"); return BBAI_ERR_STATE; }
    if (e->n_tiles <= 0) { snprintf(g_err, sizeof(g_err), "render before set_atlas"); return BBAI_ERR_STATE; }
    ON_DEVICE(e->device);
    return step_render_launch(e, actions, image, dirs, rewards, rewards64, dones, auto_reset, pixels, (hipStream_t)stream, e->done_action_enum);
}

// Register (or clear with NULL) a caller-owned uint8[n][72] device buffer that the engine keeps filled with the
// mission token ids of every env's current episode (rewritten whenever an env is reset).
int bbai_set_token_buffer(bbai_env* e, uint8_t* tokens_dev) {
    if (!e) ARG_FAIL("null handle");
    e->tokens = tokens_dev;
    if (tokens_dev && e->live) {        // episodes already running: fill every row now
        ON_DEVICE(e->device);
        HIP_TRY(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_tokens, dim3((unsigned)std::min<int64_t>((e->n + 63) / 64, 4096)), dim3(64), 0, 0, e->cfg, e->n, e->rec, e->inplace ? e->next_rec : nullptr, e->depth, e->hot,
                           tokens_dev, e->reset_list, e->counters, 1, nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    return BBAI_OK;
}

// in-place layout: rec[first ..] <- the live slots (dir 0) or the live slots <- rec[first ..] (dir 1); classic layout: nothing to do
static int live_copy(bbai_env* e, int64_t first, int64_t count, int dir) {
    if (!e->inplace || count <= 0) return BBAI_OK;
    hipLaunchKernelGGL(k_live_copy, dim3((unsigned)std::min<int64_t>((count + 3) / 4, 16384)), dim3(256), 0, 0, e->cfg, e->n, first, count, e->rec,
                       e->next_rec, e->hot, e->depth, dir);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return BBAI_OK;
}

int bbai_export_state(bbai_env* e, int64_t first, int64_t count, uint8_t* rec, uint8_t* hot, uint64_t* stale) {
    if (!e || first < 0 || count < 0 || first + count > e->n) ARG_FAIL("null handle or env range out of bounds");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    if (rec && count > 0) { int rc = live_copy(e, first, count, 0); if (rc != BBAI_OK) return rc; }
    if (rec) HIP_TRY(hipMemcpy(rec, e->rec + first * e->cfg.rec_bytes, (size_t)count * e->cfg.rec_bytes, hipMemcpyDeviceToHost));
    if (hot) HIP_TRY(hipMemcpy(hot, e->hot + first, (size_t)count * sizeof(Hot), hipMemcpyDeviceToHost));
    if (stale) HIP_TRY(hipMemcpy(stale, e->stale + first, (size_t)count * 8, hipMemcpyDeviceToHost));
    return BBAI_OK;
}

// window plane + front-cell cache follow the records (they are derived state: not part of exports or checkpoints)
static int sync_view(bbai_env* e, int64_t first, int64_t count) {
    if (e->cplane) {         // (in-place layout: e->rec is the staging area every import / checkpoint load has just filled with the live records)
        hipLaunchKernelGGL(k_sync_cpl, dim3((unsigned)std::min<int64_t>((count + 3) / 4, 16384)), dim3(256), 0, 0, e->cfg, first, count, e->rec, e->hot,
                           e->cplane, e->fcache);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    if (!e->vplane) return BBAI_OK;
    hipLaunchKernelGGL(k_sync_view, dim3((unsigned)std::min<int64_t>((count + 3) / 4, 16384)), dim3(256), 0, 0, e->cfg, first, count, e->rec, e->hot,
                       e->vplane, e->fcache);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return BBAI_OK;
}

int bbai_import_state(bbai_env* e, int64_t first, int64_t count, const uint8_t* rec, const uint8_t* hot, const uint64_t* stale) {
    if (!e || first < 0 || count < 0 || first + count > e->n) ARG_FAIL("null handle or env range out of bounds");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    if (rec) HIP_TRY(hipMemcpy(e->rec + first * e->cfg.rec_bytes, rec, (size_t)count * e->cfg.rec_bytes, hipMemcpyHostToDevice));
    if (hot && count > 0) {
        // (the imported hot state keeps this env's place in ITS ring in both layouts: the ring belongs to the handle -- in-place, hot.slot
        // says where the live record is; classic, an exporter's slot may not even exist here: it comes from another ring depth)
        Hot* staged = nullptr;
        HIP_TRY(hipMalloc((void**)&staged, (size_t)count * sizeof(Hot)));
        hipError_t r = hipMemcpy(staged, hot, (size_t)count * sizeof(Hot), hipMemcpyHostToDevice);
        if (r == hipSuccess) { hipLaunchKernelGGL(k_import_hot, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, first, count, staged, e->hot); r = hipGetLastError(); }
        if (r == hipSuccess) r = hipDeviceSynchronize();
        (void)hipFree(staged);
        HIP_TRY(r);
    }
    if (stale) HIP_TRY(hipMemcpy(e->stale + first, stale, (size_t)count * 8, hipMemcpyHostToDevice));
    if (rec && count > 0) { int rc = live_copy(e, first, count, 1); if (rc != BBAI_OK) return rc; }
    if (e->lsm && count > 0) HIP_TRY(hipMemset(e->lsm + first, 0, (size_t)count));     // (not part of the exported state: lastStepMatch = False)
    if (rec && count > 0) {
        hipLaunchKernelGGL(k_sync_prog, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, e->cfg, e->n, first, count, e->rec,
                           e->vhead, e->vset);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    // (a hot / stale-only import: the C plane rows are rebuilt from the staging records, which must then hold the LIVE ones)
    if (!rec && e->cplane && count > 0) { int rc = live_copy(e, first, count, 0); if (rc != BBAI_OK) return rc; }
    if (count > 0) { int rc = sync_view(e, first, count); if (rc != BBAI_OK) return rc; }
    e->live = true;
    return BBAI_OK;
}

// ---- checkpoint: EVERYTHING an auto-resetting batch needs to continue bit-identically in another handle --------------
// live records / hot / stale, the MT19937 streams, the look-ahead ring and its window bookkeeping, counters, and the
// expert's plans when the expert has been used.  Blob = header + the device arrays in a fixed order.
struct CkptHeader {
    uint64_t magic; int32_t version, period; int64_t n; LevelCfg cfg; int32_t depth, step_parity, next_counter_clean, seeded, live;
    int32_t has_lsm, pad0, pad1; int32_t bot_stack; int64_t tick; int64_t bot_threads;
};
constexpr int CKPT_VERSION = 4;        // 4: 256-byte next_obs slots (the C plane rows of the small rooms); 3: listless windows (NWIN buffers, meta lines, sharded totals, flow words); 2: env-major look-ahead ring
struct Seg { void* p; size_t bytes; };
// the blob's segments for a handle shaped (period, depth) with / without expert state and done-action bits
static int ckpt_segments(const bbai_env* e, int depth, bool with_bot, int bot_stack, bool with_lsm, Seg* out) {
    const size_t n = (size_t)e->n, D = (size_t)depth, rb = (size_t)e->cfg.rec_bytes;
    int k = 0;
    out[k++] = {e->rec, n * rb}; out[k++] = {e->hot, n * sizeof(Hot)}; out[k++] = {e->stale, n * 8};
    out[k++] = {e->mt, n * MT_N * 4}; out[k++] = {e->mti, n * 4}; out[k++] = {e->vhead, n * 4}; out[k++] = {e->vset, n * 64};
    out[k++] = {e->next_rec, D * n * rb}; out[k++] = {e->next_hot, D * n * sizeof(Hot)};
    out[k++] = {e->pending, NWIN * n}; out[k++] = {e->first_slot, NWIN * n};
    out[k++] = {e->win_meta, NWIN * META_U32 * 4}; out[k++] = {e->totals, SHARDS * SHARD_U64 * 8}; out[k++] = {e->flow, FLOW_WORDS * 8};
    out[k++] = {e->reset_list, n * 4}; out[k++] = {e->counters, 128};
    if (with_bot) { out[k++] = {e->bot_state, n * bot_state_bytes(bot_stack)}; out[k++] = {e->bot_stats, 16}; }
    if (with_lsm) out[k++] = {e->lsm, n};
    if (e->inplace) out[k++] = {e->next_obs, D * n * OBS_SLOT};
    return k;
}

// the lane generator's RNG state <-> the canonical (mts, mtis in [0, 624]) form: both streams idle
static int mt_canon(bbai_env* e) {
    if (!e->mtt) return BBAI_OK;
    hipLaunchKernelGGL(k_mt_canon, dim3((unsigned)((e->n + 3) / 4)), dim3(256), 0, 0, e->n, e->mt, e->mtt, e->mtpar, e->mti);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return BBAI_OK;
}
static int mt_sync(bbai_env* e) {
    if (!e->mtt) return BBAI_OK;
    const int64_t total = e->n * MT_N;
    hipLaunchKernelGGL(k_mt_sync, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, e->n, e->mt, e->mtt, e->mtpar);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return BBAI_OK;
}

int64_t bbai_checkpoint_bytes(bbai_env* e) {
    if (!e) return -1;
    Seg seg[24];
    const int k = ckpt_segments(e, e->depth, e->bot_state != nullptr, e->bot_stack, e->lsm != nullptr, seg);
    size_t total = sizeof(CkptHeader);
    for (int i = 0; i < k; ++i) total += seg[i].bytes;
    return (int64_t)total;
}

int bbai_checkpoint_save(bbai_env* e, void* host_buf, int64_t bytes) {
    if (!e || !host_buf || bytes != bbai_checkpoint_bytes(e)) ARG_FAIL("null pointer or buffer size != bbai_checkpoint_bytes()");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());             // both streams idle: the ring and the windows' bookkeeping are at rest, every refill has landed
    { int rc = live_copy(e, 0, e->n, 0); if (rc != BBAI_OK) return rc; }       // (in-place layout: the blob's record segment = the live slots)
    { int rc = mt_canon(e); if (rc != BBAI_OK) return rc; }                    // (the blob holds the canonical MT19937 form: every position in [0, 624])
    { int rc = mt_sync(e); if (rc != BBAI_OK) return rc; }
    CkptHeader h;
    memset(&h, 0, sizeof(h));
    h.magic = 0x42424149434b5054ull; h.version = CKPT_VERSION; h.period = e->period; h.n = e->n; h.cfg = e->cfg; h.depth = e->depth;
    h.step_parity = e->step_parity; h.next_counter_clean = e->next_counter_clean; h.seeded = e->seeded; h.live = e->live;
    h.has_lsm = e->lsm ? 1 : 0;
    h.bot_stack = e->bot_state ? e->bot_stack : 0; h.tick = e->tick; h.bot_threads = e->bot_threads;
    uint8_t* dst = (uint8_t*)host_buf;
    memcpy(dst, &h, sizeof(h)); dst += sizeof(h);
    Seg seg[24];
    const int k = ckpt_segments(e, e->depth, e->bot_state != nullptr, e->bot_stack, e->lsm != nullptr, seg);
    for (int i = 0; i < k; ++i) { HIP_TRY(hipMemcpy(dst, seg[i].p, seg[i].bytes, hipMemcpyDeviceToHost)); dst += seg[i].bytes; }
    return BBAI_OK;
}

static int bot_alloc(bbai_env* e, int cap);

int bbai_checkpoint_load(bbai_env* e, const void* host_buf, int64_t bytes) {
    if (!e || !host_buf || bytes < (int64_t)sizeof(CkptHeader)) ARG_FAIL("null pointer or short buffer");
    CkptHeader h;
    memcpy(&h, host_buf, sizeof(h));
    if (h.magic != 0x42424149434b5054ull) ARG_FAIL("not a bbai checkpoint");
    if (h.version != CKPT_VERSION) {
        snprintf(g_err, sizeof(g_err), "bbai_checkpoint_load: unsupported checkpoint version %d (this library reads version %d)", h.version, CKPT_VERSION);
        return BBAI_ERR_ARG;
    }
    if (h.n != e->n || memcmp(&h.cfg, &e->cfg, sizeof(LevelCfg)) != 0 || h.period < 1 || h.period > MAX_PERIOD || h.depth != 2 * h.period + e->inplace)
        ARG_FAIL("checkpoint was taken from a different level / batch size / state layout (BBAI_INPLACE)");
    // EVERYTHING is validated before the handle is touched (a refused blob leaves the handle exactly as it was): the expert's stack
    // capacity, the done-action mode, and the blob's size for the shape its header announces.
    if (h.bot_stack && e->bot_state && e->bot_stack != h.bot_stack) ARG_FAIL("the handle's expert uses a different stack capacity (BBAI_BOT_STACK)");
    if ((h.has_lsm != 0) != (e->lsm != nullptr)) ARG_FAIL("checkpoint and handle differ in the done-action mode (bbai_set_done_actions)");
    {
        Seg seg[24];
        const int k = ckpt_segments(e, h.depth, h.bot_stack != 0, h.bot_stack, h.has_lsm != 0, seg);
        size_t total = sizeof(CkptHeader);
        for (int i = 0; i < k; ++i) total += seg[i].bytes;
        if ((int64_t)total != bytes) ARG_FAIL("checkpoint size does not match its header (truncated blob?)");
    }
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    if (h.bot_stack && !e->bot_state) {          // (allocates only; on failure the handle is unchanged)
        int rc = bot_alloc(e, h.bot_stack);
        if (rc != BBAI_OK) return rc;
    }
    if (h.period != e->period) {
        // The handle chose its look-ahead period from the memory that was free when it was created (bbai_create); the blob's
        // ring has the saving handle's.  The ring is part of the state: take the blob's shape.  The new ring is allocated BEFORE the
        // old one is let go, so that a failure leaves the handle as it was.
        const size_t D = (size_t)h.depth, slot_bytes = (size_t)e->n * e->cfg.rec_bytes;
        uint8_t* nrec = nullptr; Hot* nhot = nullptr; uint8_t* nobs = nullptr;
        hipError_t r1 = hipMalloc((void**)&nrec, D * slot_bytes);
        hipError_t r2 = r1 == hipSuccess ? hipMalloc((void**)&nhot, D * (size_t)e->n * sizeof(Hot)) : r1;
        hipError_t r3 = r2;
        if (r3 == hipSuccess && e->inplace) r3 = hipMalloc((void**)&nobs, D * (size_t)e->n * OBS_SLOT);
        if (r3 != hipSuccess) {
            (void)hipGetLastError();
            if (nrec) (void)hipFree(nrec);
            if (nhot) (void)hipFree(nhot);
            if (nobs) (void)hipFree(nobs);
            snprintf(g_err, sizeof(g_err), "checkpoint_load: no memory for the checkpoint's look-ahead ring (period %d); the handle is unchanged", h.period);
            return BBAI_ERR_NOMEM;
        }
        (void)hipFree(e->next_rec); (void)hipFree(e->next_hot);
        if (e->next_obs) (void)hipFree(e->next_obs);
        e->next_rec = nrec; e->next_hot = nhot; e->next_obs = nobs;
        e->period = h.period;
        e->depth = h.depth;
    }
    const bool had_bot = e->bot_state != nullptr;
    if (!h.bot_stack && had_bot) {               // checkpoint without expert state: every env gets a fresh Bot
        HIP_TRY(hipMemset(e->bot_state, 0, (size_t)e->n * bot_state_bytes(e->bot_stack)));
        HIP_TRY(hipMemset(e->bot_stats, 0, 16));
    }
    Seg seg[24];
    const int k = ckpt_segments(e, e->depth, h.bot_stack != 0, h.bot_stack, h.has_lsm != 0, seg);
    const uint8_t* src = (const uint8_t*)host_buf + sizeof(CkptHeader);
    // (from here on a failing copy leaves a half-loaded handle: it must not be stepped)
    e->seeded = e->live = false;
    for (int i = 0; i < k; ++i) { HIP_TRY(hipMemcpy(seg[i].p, src, seg[i].bytes, hipMemcpyHostToDevice)); src += seg[i].bytes; }
    e->step_parity = h.step_parity; e->next_counter_clean = h.next_counter_clean != 0; e->seeded = h.seeded != 0; e->live = h.live != 0;
    e->tick = h.tick;
    // (the saved run was idle: every refill it had launched has landed, and flow[FLOW_REFILLED] in the blob says so)
    for (int i = 0; i < NWIN; ++i) HIP_TRY(hipEventRecord(e->ev_refill[i], e->side));
    HIP_TRY(hipDeviceSynchronize());
    { int rc = mt_sync(e); if (rc != BBAI_OK) return rc; }
    return sync_view(e, 0, e->n);
}

int bbai_get_programs(bbai_env* e, int64_t first, int64_t count, uint8_t* prog) {
    if (!e || !prog || first < 0 || count < 0 || first + count > e->n) ARG_FAIL("null pointer or env range out of bounds");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    { int rc = live_copy(e, first, count, 0); if (rc != BBAI_OK) return rc; }
    HIP_TRY(hipMemcpy2D(prog, sizeof(Prog), e->rec + first * e->cfg.rec_bytes + e->cfg.off_prog, (size_t)e->cfg.rec_bytes,
                        sizeof(Prog), (size_t)count, hipMemcpyDeviceToHost));
    return BBAI_OK;
}

static int bot_alloc(bbai_env* e, int cap) {             // the expert's state: all three buffers or none
    const int64_t threads = std::min<int64_t>((e->n + 63) / 64 * 64, 256 * 8 * 64);
    void *st = nullptr, *wk = nullptr, *ss = nullptr, *rw = nullptr;
    const size_t sbytes = bot_state_bytes(cap);
    hipError_t err = hipMalloc(&st, (size_t)e->n * sbytes);
    if (err == hipSuccess) err = hipMalloc(&wk, (size_t)threads * BOT_WORK_WORDS * sizeof(uint16_t));
    if (err == hipSuccess) err = hipMalloc(&rw, (size_t)threads * (R_ALL - R_FAST) * MAX_W * sizeof(uint32_t));
    if (err == hipSuccess) err = hipMalloc(&ss, 16);
    if (err == hipSuccess) err = hipMemset(st, 0, (size_t)e->n * sbytes);
    if (err == hipSuccess) err = hipMemset(ss, 0, 16);
    if (err != hipSuccess) {
        (void)hipFree(st); (void)hipFree(wk); (void)hipFree(ss); (void)hipFree(rw);
        snprintf(g_err, sizeof(g_err), "allocating the expert's state failed: %s", hipGetErrorString(err));
        return BBAI_ERR_NOMEM;
    }
    e->bot_stack = cap;
    e->bot_state = (uint8_t*)st; e->bot_work = (uint16_t*)wk; e->bot_stats = (uint64_t*)ss; e->bot_threads = threads;
    e->bot_rows = (uint32_t*)rw;
    { const char* ev = getenv("BBAI_BOT_EAGER"); e->bot_eager = ev ? atoi(ev) != 0 : 1; }
    return BBAI_OK;
}

// k_bot on stream s (allocates the expert's state on first use); the caller has entered the call
static int bot_launch(bbai_env* e, const uint8_t* prev_actions, uint8_t* actions, int dead_action, uint8_t* gave_up, hipStream_t s) {
    if (!e->bot_state) {                                   // first use
        // Subgoal stack depth per env.  The reference's list is unbounded; 48 covers every plan that makes progress (the
        // rare bot that loops without progress grows its stack until max_steps and fails the episode -- here it gives up
        // when the stack is full, counted in bbai_bot_stats().capacity).  Raise it to follow such a bot further.
        const char* ev = getenv("BBAI_BOT_STACK");
        int rc = bot_alloc(e, ev ? std::max(8, std::min(atoi(ev), 4096)) : BOT_STACK);
        if (rc != BBAI_OK) return rc;
    }
    // Occupancy target, measured (profiles/r01/bot_bench.jsonl): the fully inlined expert wants ~400
    // registers; capping it at 256 (2 waves/SIMD, spills to scratch) is +35 % on maze levels (BossLevel 1M envs 22.8 ->
    // 17.0 ms) and -10 % on single rooms, 128 registers (4 waves/SIMD) loses everywhere, real calls instead of inlining too.
    const bool maze = e->cfg.num_rows * e->cfg.num_cols > 1;
    unsigned long long* stats = (unsigned long long*)e->bot_stats;
#if !BBAI_BOT_GROUP_BUILD
    if (e->bot_group) { snprintf(g_err, sizeof(g_err), "this library was built without the lane-group expert (k_botg: an experiment build, -DBBAI_BOT_GROUP_BUILD=1)"); return BBAI_ERR_ARG; }
#else
    if (e->bot_group) {                                     // one lane group per env (k_botg)
        // Measured (profiles/r05/NOTES.md section 11): same decisions, 1.9-2.4 x SLOWER than lane = env -- a group executes 3.4 x the
        // instructions per env (the subgoal machine runs redundantly on the group's lanes, 4 envs share a wave's issue slots instead
        // of 64) and only gets 1.8 x the instructions per cycle back.  Kept as an option for experiments, off by default.
        constexpr int G = 16;
        if (e->bot_group != G) { snprintf(g_err, sizeof(g_err), "no k_botg build for %d lanes per env (only 16)", e->bot_group); return BBAI_ERR_ARG; }
        const int64_t threads = std::min<int64_t>((e->n * G + 63) / 64 * 64, (int64_t)256 * 8 * 64);
        const size_t glds = (size_t)(64 / G) * botg_group_words(e->cfg) * 4;     // BossLevel: 4 x 2.5 KB
        hipLaunchKernelGGL((k_botg<G, 2>), dim3((unsigned)(threads / 64)), dim3(64), glds, s, e->cfg, e->n, e->rec, e->inplace ? e->next_rec : nullptr, e->depth, e->hot,
                           e->stale, e->bot_state, e->bot_stack, e->bot_work, e->bot_eager, prev_actions, actions, stats, dead_action, gave_up);
        HIP_TRY(hipGetLastError());
        return BBAI_OK;
    }
#endif
    const dim3 grid((unsigned)(e->bot_threads / 64)), block(64);
    const size_t lds = (size_t)R_FAST * e->cfg.H * 64 * 4 + (size_t)BOT_RING * 64 * 2;     // BossLevel: 11.3 + 8 KB -> 8 waves per CU
    // ONE instantiation for every geometry: round 5's packed lists took the kernel from ~400 wanted registers to a shape that fits 256 with 24 spilled, and
    // the single rooms then prefer two waves per SIMD as well (profiles/r05/bot_round5_experiments.jsonl: PickupLoc 262 144 1.09 -> 0.95 ms per decision + step,
    // GoToLocal 65 536 0.357 -> 0.353); round 1's k_bot<1> for them is gone.
    (void)maze;
    hipLaunchKernelGGL(k_bot<2>, grid, block, lds, s, e->cfg, e->n, e->rec, e->inplace ? e->next_rec : nullptr, e->depth, e->hot, e->stale, e->bot_state, e->bot_stack, e->bot_work,
                       e->bot_rows, e->bot_eager, prev_actions, actions, stats, dead_action, gave_up);
    HIP_TRY(hipGetLastError());
    return BBAI_OK;
}

int bbai_bot_act(bbai_env* e, const uint8_t* prev_actions, uint8_t* actions, void* stream) {
    if (!e || !actions) ARG_FAIL("null handle or output buffer");
    if (!e->live) { snprintf(g_err, sizeof(g_err), "bot_act before reset"); return BBAI_ERR_STATE; }
    ON_DEVICE(e->device);
    hipStream_t s = (hipStream_t)stream;
    CallScope call(e, s);
    if (call.rc != BBAI_OK) return call.rc;
    { int rc = bot_launch(e, prev_actions, actions, BOT_DEAD, nullptr, s); if (rc != BBAI_OK) return rc; }
    return call.leave();
}

// T expert decisions + steps with no host round trip in between (the inner loop of generate_demos).
int bbai_bot_rollout(bbai_env* e, int T, uint8_t* image, uint8_t* dirs, uint8_t* images_out, uint8_t* dirs_out, uint8_t* tokens_out,
                     uint8_t* actions_out, float* rewards_out, uint8_t* dones_out, uint8_t* gave_up_out, void* stream) {
    if (!e || T < 1 || !image || !dirs || !images_out || !dirs_out || !actions_out || !rewards_out || !dones_out || !gave_up_out)
        ARG_FAIL("null handle or buffer, or T < 1");
    if (!e->live || !e->seeded) { snprintf(g_err, sizeof(g_err), "bot_rollout before seed + reset"); return BBAI_ERR_STATE; }
    if (tokens_out && !e->tokens) { snprintf(g_err, sizeof(g_err), "bot_rollout: tokens_out needs a registered token buffer (bbai_set_token_buffer)"); return BBAI_ERR_STATE; }
    ON_DEVICE(e->device);
    hipStream_t s = (hipStream_t)stream;
    CallScope call(e, s);
    if (call.rc != BBAI_OK) return call.rc;
    const size_t n = (size_t)e->n;
    for (int t = 0; t < T; ++t) {
        // what the expert decides on: the observation (and mission) BEFORE the step
        HIP_TRY(hipMemcpyAsync(images_out + (size_t)t * n * OBS_BYTES, image, n * OBS_BYTES, hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(dirs_out + (size_t)t * n, dirs, n, hipMemcpyDeviceToDevice, s));
        if (tokens_out) HIP_TRY(hipMemcpyAsync(tokens_out + (size_t)t * n * TOK_MAX, e->tokens, n * TOK_MAX, hipMemcpyDeviceToDevice, s));
        int rc = bot_launch(e, nullptr, actions_out + (size_t)t * n, A_RESET_ENV, gave_up_out + (size_t)t * n, s);
        if (rc != BBAI_OK) return rc;
        rc = step_launch(e, actions_out + (size_t)t * n, image, dirs, rewards_out + (size_t)t * n, nullptr, dones_out + (size_t)t * n, 1, s,
                         1 /* the expert's `done` is the enum member (bot.py:593) */);
        if (rc != BBAI_OK) return rc;
    }
    return call.leave();
}

#if defined(BBAI_BOT_PROF)
int bbai_bot_prof_read(unsigned long long* out32, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_bot_prof), 32 * 8));
    if (reset) { unsigned long long z[32] = {}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_bot_prof), z, sizeof(z))); }
    return BBAI_OK;
}
#endif

int bbai_bot_stats(bbai_env* e, uint64_t* gave_up, uint64_t* capacity) {
    if (!e || !gave_up || !capacity) ARG_FAIL("null pointer");
    ON_DEVICE(e->device);
    *gave_up = *capacity = 0;
    if (!e->bot_stats) return BBAI_OK;
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipMemcpy(v, e->bot_stats, 16, hipMemcpyDeviceToHost));
    *gave_up = v[0]; *capacity = v[1];
    return BBAI_OK;
}

static int tap_launch(int64_t count, int64_t pix_count, const int64_t* ids, const uint8_t* image, const uint8_t* dirs, const double* rew64,
                      const uint8_t* dones, const uint8_t* pixels, uint8_t* image_out, uint8_t* dirs_out, double* rew64_out, uint8_t* dones_out,
                      uint8_t* pixels_out, void* stream) {
    if (count <= 0 || pix_count < 0 || pix_count > count || !image || !dirs || !rew64 || !dones || !image_out || !dirs_out || !rew64_out || !dones_out ||
        (pix_count && (!pixels || !pixels_out || ((uintptr_t)pixels & 15) || ((uintptr_t)pixels_out & 15))))
        ARG_FAIL("null / misaligned pointer or empty tap");
    const int64_t work = std::max<int64_t>(count * OBS_BYTES, pix_count * (PIX_BYTES / 16));
    hipLaunchKernelGGL(k_tap, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, count, pix_count, ids,
                       image, dirs, rew64, dones, pixels, image_out, dirs_out, rew64_out, dones_out, pixels_out);
    HIP_TRY(hipGetLastError());
    return BBAI_OK;
}

int bbai_tap(int64_t count, int64_t pix_count, const uint8_t* image, const uint8_t* dirs, const double* rew64, const uint8_t* dones,
             const uint8_t* pixels, uint8_t* image_out, uint8_t* dirs_out, double* rew64_out, uint8_t* dones_out, uint8_t* pixels_out,
             void* stream) {
    return tap_launch(count, pix_count, nullptr, image, dirs, rew64, dones, pixels, image_out, dirs_out, rew64_out, dones_out, pixels_out, stream);
}

int bbai_tap_ids(int64_t count, int64_t pix_count, const int64_t* ids_dev, const uint8_t* image, const uint8_t* dirs, const double* rew64,
                 const uint8_t* dones, const uint8_t* pixels, uint8_t* image_out, uint8_t* dirs_out, double* rew64_out, uint8_t* dones_out,
                 uint8_t* pixels_out, void* stream) {
    if (!ids_dev) ARG_FAIL("null id list");
    return tap_launch(count, pix_count, ids_dev, image, dirs, rew64, dones, pixels, image_out, dirs_out, rew64_out, dones_out, pixels_out, stream);
}

// T steps of the hot path with no host round trip in between (include/bbai.h): what a caller's loop of bbai_step [+ bbai_render]
// [+ bbai_tap_ids] enqueues, enqueued from here.  (Built on the suspicion that an interpreter's per-step overhead does not stay
// ahead of a 40-us step -- a rocprofv3 trace shows 10-us gaps between the kernels of one step, profiles/r04/kernel_trace_gaps_*.txt --
// and measured equal to the Python loop on every config: the gaps are the tracer's.  Kept as the open-loop entry it is.)
int bbai_rollout(bbai_env* e, int T, const uint8_t* actions, uint8_t* image, uint8_t* dirs, float* rewards, double* rewards64, uint8_t* dones,
                 int auto_reset, uint8_t* pixels, const bbai_tap_log* tap, void* stream) {
    if (!e || T < 1 || !actions || !image || !dirs || !rewards || !dones) ARG_FAIL("null handle or buffer, or T < 1");
    if (!e->live) { snprintf(g_err, sizeof(g_err), "rollout before reset"); return BBAI_ERR_STATE; }
    if (auto_reset && !e->seeded) { snprintf(g_err, sizeof(g_err), "auto-reset rollout before seed"); return BBAI_ERR_STATE; }
    if (pixels && e->n_tiles <= 0) { snprintf(g_err, sizeof(g_err), "rollout with pixels before set_atlas"); return BBAI_ERR_STATE; }
    if (tap && (!rewards64 || tap->count <= 0 || tap->pix_count < 0 || tap->pix_count > tap->count || !tap->image_out || !tap->dir_out ||
                !tap->reward64_out || !tap->done_out || (tap->pix_count && (!pixels || !tap->pixels_out))))
        ARG_FAIL("tap log incomplete (it needs reward64_dev, and pixels_dev for its pixel rows)");
    ON_DEVICE(e->device);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)e->n;
    const bool own_tap = tap && !tap->ids_dev;          // the envs of bbai_step_tap_set, logged by the stepping lanes
    if (own_tap && (tap->count != e->tap_count || tap->pix_count != 0)) {
        snprintf(g_err, sizeof(g_err), "rollout: a tap log without ids_dev follows bbai_step_tap_set's list (%lld envs, no pixel rows)", (long long)e->tap_count);
        return BBAI_ERR_ARG;
    }
    // Encoded observations, the finished envs moved on inside k_step (or nobody resets): the launches carry as many ticks as the look-ahead
    // window has left -- the same bytes as tick-by-tick launches (tests/test_gpu_parity.py::test_rollout_*), without their boundaries.
    // An unfused auto-reset (k_consume between the steps), a token buffer (k_tokens reads every step's dones) or a tap that is a launch
    // keep one step per launch.
    const bool fused = auto_reset && (e->inplace || use_fused_consume(e));
    if (!pixels && (fused || !auto_reset) && !e->tokens && (!tap || own_tap) && (e->rollout_multi || own_tap)) {
        CallScope call(e, s);
        if (call.rc != BBAI_OK) return call.rc;
        for (int t = 0; t < T;) {
            StepPlan p;
            { int rc = step_prepare(e, auto_reset, s, p); if (rc != BBAI_OK) return rc; }
            int ticks = e->rollout_multi ? T - t : 1;
            if (fused) ticks = std::min(ticks, e->period - tick_pos(e).pos);
            if (own_tap) {
                const size_t c = (size_t)tap->count, orow = (size_t)(tap->obs_row0 + t), row = (size_t)(tap->row0 + t);
                p.tap = TapArgs{e->tap_mask, e->tap_rank0, e->tap_perm, tap->image_out + orow * c * OBS_BYTES, tap->dir_out + orow * c,
                                tap->reward64_out + row * c, tap->done_out + row * c, tap->count};
            }
            { int rc = step_kernel(e, p, actions + (size_t)t * n, image, dirs, rewards, rewards64, dones, auto_reset, s, e->done_action_enum, 0, step_blocks(e), ticks); if (rc != BBAI_OK) return rc; }
            if (fused) { int rc = window_end(e, s, 2, dones, ticks); if (rc != BBAI_OK) return rc; }
            t += ticks;
        }
        return call.leave();
    }
    if (own_tap && pixels) { snprintf(g_err, sizeof(g_err), "rollout: a tap log without ids_dev has no pixel rows (encoded observations only)"); return BBAI_ERR_ARG; }
    for (int t = 0; t < T; ++t) {
        if (own_tap) {          // (unfused auto-reset / a token buffer: one bbai_step_tapped per step)
            CallScope call(e, s);
            if (call.rc != BBAI_OK) return call.rc;
            const size_t c = (size_t)tap->count, orow = (size_t)(tap->obs_row0 + t), row = (size_t)(tap->row0 + t);
            const TapRows rows = {tap->image_out + orow * c * OBS_BYTES, tap->dir_out + orow * c, tap->reward64_out + row * c, tap->done_out + row * c};
            { int rc = step_launch(e, actions + (size_t)t * n, image, dirs, rewards, rewards64, dones, auto_reset, s, e->done_action_enum, &rows); if (rc != BBAI_OK) return rc; }
            { int rc = call.leave(); if (rc != BBAI_OK) return rc; }
            continue;
        }
        { int rc = step_render_launch(e, actions + (size_t)t * n, image, dirs, rewards, rewards64, dones, auto_reset, pixels, s, e->done_action_enum); if (rc != BBAI_OK) return rc; }
        if (tap) {
            const size_t c = (size_t)tap->count, orow = (size_t)(tap->obs_row0 + t), row = (size_t)(tap->row0 + t);
            int rc = tap_launch(tap->count, tap->pix_count, tap->ids_dev, image, dirs, rewards64, dones, pixels, tap->image_out + orow * c * OBS_BYTES,
                                tap->dir_out + orow * c, tap->reward64_out + row * c, tap->done_out + row * c,
                                tap->pix_count ? tap->pixels_out + orow * (size_t)tap->pix_count * PIX_BYTES : nullptr, stream);
            if (rc != BBAI_OK) return rc;
        }
    }
    return BBAI_OK;
}

int bbai_gae(int64_t num_envs, int num_frames, const float* rewards, const float* values, const float* masks, const float* last_mask,
             const float* last_value, double discount, double gae_lambda, float* advantage, float* returnn, void* stream) {
    if (num_envs <= 0 || num_frames <= 0 || !rewards || !values || !masks || !last_mask || !last_value || !advantage || !returnn)
        ARG_FAIL("null pointer or empty rollout");
    hipLaunchKernelGGL(k_gae, dim3((unsigned)((num_envs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, num_envs, num_frames, rewards, values,
                       masks, last_mask, last_value, (float)discount, (float)(discount * gae_lambda), advantage, returnn);
    HIP_TRY(hipGetLastError());
    return BBAI_OK;
}

int bbai_set_call_events(bbai_env* e, int enable) {
    if (!e) ARG_FAIL("null handle");
    if (enable && !e->call_events && e->have_stream) {      // switching on mid-run: the event must describe the work so far
        ON_DEVICE(e->device);
        HIP_TRY(hipEventRecord(e->ev_switch, e->last_stream));
    }
    e->call_events = enable != 0;
    return BBAI_OK;
}


// The reference's BABYAI_DONE_ACTIONS verifier mode for this handle (on by default iff that variable was non-empty at
// bbai_create, as the reference reads it at import).  Switching it resets every env's lastStepMatch bits.
int bbai_set_done_actions(bbai_env* e, int enable) {
    if (!e) ARG_FAIL("null handle");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    if (enable && !e->lsm) HIP_TRY(hipMalloc((void**)&e->lsm, (size_t)e->n));
    if (!enable && e->lsm) { HIP_TRY(hipFree(e->lsm)); e->lsm = nullptr; }
    if (e->lsm) HIP_TRY(hipMemset(e->lsm, 0, (size_t)e->n));
    return BBAI_OK;
}
int bbai_get_done_actions(bbai_env* e) { return e && e->lsm ? 1 : 0; }

// Performance knobs by name (never semantics: every setting produces the same bytes).  What the BBAI_* environment variables
// set at bbai_create, switchable on a live handle so that measurements can alternate settings inside ONE process on ONE box
// (tools/ab.py).  Synchronises the device first.
int bbai_set_option(bbai_env* e, const char* name, int64_t value) {
    if (!e || !name) ARG_FAIL("null handle or name");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    const int v = (int)value;
    if (!strcmp(name, "render_queue")) e->render_queue = v;
    else if (!strcmp(name, "render_queue_bpc")) e->render_queue_bpc = v;
    else if (!strcmp(name, "render_queue_blocks")) e->render_queue_blocks = v;
    else if (!strcmp(name, "render_pace")) e->render_pace = v < 0 ? 0 : v;
    else if (!strcmp(name, "render_group")) e->render_group = v;
    else if (!strcmp(name, "render_tpb")) e->render_tpb = v;
    else if (!strcmp(name, "step_prio")) e->step_prio = v;
    else if (!strcmp(name, "pregen_group")) e->pregen_group = v;
    else if (!strcmp(name, "pregen_lane")) {
        if (!e->mtt) { if (v) ARG_FAIL("pregen_lane: this level kind / layout is not covered by the lane generator"); }
        else if ((v != 0) != (e->pregen_lane != 0)) {
            HIP_TRY(hipDeviceSynchronize());
            // the lane-group kernel reads and writes the canonical form; the lane kernel needs the tempered halves behind it
            { int rc = mt_canon(e); if (rc != BBAI_OK) return rc; }
            { int rc = mt_sync(e); if (rc != BBAI_OK) return rc; }
            e->pregen_lane = v != 0;
        }
    }
    else if (!strcmp(name, "lane_blocks")) e->lane_blocks = std::max(1, v);
    else if (!strcmp(name, "pregen_blocks")) e->pregen_cap = std::max(64, v);
    else if (!strcmp(name, "pregen_min")) e->pregen_min = std::max(0, v);
    else if (!strcmp(name, "pregen_per_group")) e->pregen_per_group = std::max(1, v);
    else if (!strcmp(name, "consume_fused")) e->consume_fused = v;
    else if (!strcmp(name, "gate_strict")) e->gate_strict = v != 0;
    else if (!strcmp(name, "gate_probe")) { e->gate_probe = v; e->n_probed = 0; e->gate_forced = 0; }      // (forget the verdicts: the next window probes again)
    else if (!strcmp(name, "gate_fault_inject")) e->host_flags[0] = v != 0;       // tests: what a timed-out gate leaves behind
    else if (!strcmp(name, "bot_group")) e->bot_group = v;
    else if (!strcmp(name, "step_render_split")) e->step_render_split = v;
    else if (!strcmp(name, "rollout_multi")) e->rollout_multi = v;
    else if (!strcmp(name, "lookahead_streams")) { int rc = set_sides(e, v); if (rc != BBAI_OK) return rc; }
    else if (!strcmp(name, "done_action_enum")) e->done_action_enum = v != 0;       // (the one SEMANTIC switch in this list: include/bbai.h bbai_set_done_actions)
    else {
        snprintf(g_err, sizeof(g_err), "set_option: unknown option '%s'", name);
        return BBAI_ERR_ARG;
    }
    return BBAI_OK;
}

static int read_flow(bbai_env* e, int word, unsigned long long* out) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, e->flow + word, 8, hipMemcpyDeviceToHost));
    return BBAI_OK;
}

// Read back a knob: the names of bbai_set_option, plus "render_pace_effective" and "lookahead_period" (the refill period the handle chose).
int bbai_get_option(bbai_env* e, const char* name, int64_t* out) {
    if (!e || !name || !out) ARG_FAIL("null handle, name or output");
    if (!strcmp(name, "render_queue")) *out = e->render_queue;
    else if (!strcmp(name, "render_queue_bpc")) *out = e->render_queue_bpc;
    else if (!strcmp(name, "render_queue_blocks")) *out = e->render_queue_blocks;
    else if (!strcmp(name, "render_pace")) *out = e->render_pace;
    else if (!strcmp(name, "render_group")) *out = e->render_group;
    else if (!strcmp(name, "render_tpb")) *out = e->render_tpb;
    else if (!strcmp(name, "step_prio")) *out = e->step_prio;
    else if (!strcmp(name, "pregen_group")) *out = e->pregen_group;
    else if (!strcmp(name, "pregen_lane")) *out = e->pregen_lane;
    else if (!strcmp(name, "lane_blocks")) *out = e->lane_blocks;
    else if (!strcmp(name, "pregen_blocks")) *out = e->pregen_cap;
    else if (!strcmp(name, "pregen_min")) *out = e->pregen_min;
    else if (!strcmp(name, "pregen_per_group")) *out = e->pregen_per_group;
    else if (!strcmp(name, "consume_fused")) *out = e->consume_fused;
    else if (!strcmp(name, "gate_strict")) *out = e->gate_strict;
    else if (!strcmp(name, "gate_probe")) *out = e->gate_probe;
    else if (!strcmp(name, "gate_forced_strict")) *out = e->gate_forced;          // 1: the caller's stream failed the concurrency probe
    else if (!strcmp(name, "gate_fault")) *out = e->host_flags ? (int64_t)e->host_flags[0] : 0;     // sticky; no synchronisation
    else if (!strcmp(name, "bot_group")) *out = e->bot_group;
    else if (!strcmp(name, "step_render_split")) *out = e->step_render_split;
    else if (!strcmp(name, "rollout_multi")) *out = e->rollout_multi;
    else if (!strcmp(name, "lookahead_streams")) *out = e->n_sides;
    else if (!strcmp(name, "profile_step_ticks")) *out = e->prof_step_ticks;
    else if (!strcmp(name, "inplace")) *out = e->inplace;
    else if (!strcmp(name, "cplane")) *out = e->cplane ? 1 : 0;
    else if (!strcmp(name, "done_action_enum")) *out = e->done_action_enum;
    else if (!strcmp(name, "lookahead_period")) *out = e->period;
    else if (!strcmp(name, "gate_timeouts")) {      // (synchronises) window gates that gave up waiting for a refill: must be 0 (k_gate)
        ON_DEVICE(e->device);
        unsigned long long v = 0;
        int rc = read_flow(e, FLOW_GATE_TIMEOUTS, &v);
        if (rc != BBAI_OK) return rc;
        *out = (int64_t)v;
    }
    else if (!strcmp(name, "render_pace_effective")) *out = e->render_pace > 0 ? e->render_pace : 0;
    else {
        snprintf(g_err, sizeof(g_err), "get_option: unknown option '%s'", name);
        return BBAI_ERR_ARG;
    }
    return BBAI_OK;
}

int bbai_profile(bbai_env* e, int enable) {
    if (!e) ARG_FAIL("null handle");
    e->prof_on = enable != 0;
    // 1: start from zero; 2: resume (totals kept: callers that bracket every other block of a run); 0: pause, totals readable
    if (enable == 1) e->prof_step_ticks = 0;
    if (enable == 1) for (int k = 0; k < 3; ++k) { e->prof_ms[k] = 0; e->prof_n[k] = 0; for (int i = 0; i < PROF_RING; ++i) e->prof[k][i].used = false; }
    return BBAI_OK;
}

int bbai_profile_read(bbai_env* e, double* ms_total /* [3] */, int64_t* launches /* [3] */) {
    if (!e || !ms_total || !launches) ARG_FAIL("null pointer");
    ON_DEVICE(e->device);
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < PROF_RING; ++i) prof_fold(e, k, i);
        ms_total[k] = e->prof_ms[k]; launches[k] = e->prof_n[k];
    }
    return BBAI_OK;
}

int bbai_generator_failures(bbai_env* e, uint64_t* out) {
    if (!e || !out) ARG_FAIL("null pointer");
    ON_DEVICE(e->device);
    unsigned long long v = 0;
    { int rc = read_flow(e, FLOW_GEN_FAILURES, &v); if (rc != BBAI_OK) return rc; }
    *out = (uint64_t)v;
    return BBAI_OK;
}

int bbai_reset_count(bbai_env* e, uint64_t* out) {
    if (!e || !out) ARG_FAIL("null pointer");
    ON_DEVICE(e->device);
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long shards[SHARDS * SHARD_U64], v = 0;
    HIP_TRY(hipMemcpy(shards, e->totals, sizeof(shards), hipMemcpyDeviceToHost));
    for (int k = 0; k < SHARDS; ++k) v += shards[k * SHARD_U64];
    *out = (uint64_t)v;
    return BBAI_OK;
}

}  // extern "C"
