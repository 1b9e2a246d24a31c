// hostsim.cpp -- TEST HARNESS ONLY: compiles the engine's per-env core (generator, step,
// verifier, observation) for the host CPU with one "lane", so the logic can be checked
// against the Python oracle in a container without a GPU.  Never loaded by the product
// package (babyai_amd loads only the HIP library and fails without it).
#include <string.h>
#include "../../babyai_amd/csrc/bbai_types.hpp"
#include "../../babyai_amd/csrc/bbai_gen.hpp"
#include "../../babyai_amd/csrc/bbai_genl.hpp"
#include "../../babyai_amd/csrc/bbai_step.hpp"
#include "../../babyai_amd/csrc/bbai_view.hpp"
static long long g_bot_counts[8];
#define BBAI_BOT_COUNT(what) (++g_bot_counts[what])
static bool g_bot_aligned_ok = true;         // hs_bot_set_aligned(0): _find_obj_pos always through the packed lists (property test of the shortcut)
#define BBAI_BOT_ALIGNED_OK g_bot_aligned_ok
#include "../../babyai_amd/csrc/bbai_bot.hpp"
#include <ucontext.h>
#include <cstdlib>
#include "../../babyai_amd/csrc/bbai_seed.hpp"

using namespace bbai;

struct HostCtx {
    static constexpr int kLanes = 1;
    int lane() const { return 0; }
    int nlanes() const { return 1; }
    void sync() const {}
    uint32_t shfl_up1(uint32_t) const { return 0; }
    uint32_t shfl_down1(uint32_t) const { return 0; }
    uint32_t shfl(uint32_t v, int) const { return v; }
    bool any(bool p) const { return p; }
};

extern "C" {

int hs_fill_layout(LevelCfg* cfg) { return fill_layout(*cfg); }

void hs_seed(uint64_t seed, uint32_t* mt) { seed_env(seed, mt); }

// Generate the next level of the stream (mt, *mti) into rec / hot; stale := 0.
int hs_generate(const LevelCfg* cfg, uint32_t* mt, int32_t* mti, uint8_t* rec, Hot* hot) {
    int last_locked = hot->last_locked == NONE8 ? -1 : hot->last_locked;
    static thread_local GenWork w;
    memset((void*)&w, 0xA5, sizeof(w));          // (the device's working set starts as whatever LDS held: nothing may be read before it is written)
    uint32_t tw[MT_CH];                          // the tempered chunk (device: LDS, behind the state)
    memset(tw, 0xA5, sizeof(tw));
    Gen<HostCtx> g(HostCtx(), *cfg, w, mt, tw, *mti, last_locked);      // (the generator advances the caller's state in place)
    int max_steps = g.generate();
    *mti = g.mti;
    memset(rec, 0, cfg->rec_bytes);
    memcpy(rec, w.E, cfg->ES * cfg->EH);
    memcpy(rec + cfg->off_I, w.I, cfg->W * cfg->H);
    // object tables: unused entries are written as zeros / NONE8, exactly like the device write-out (k_pregen)
    for (int o = 0; o < cfg->maxo; ++o) {
        const bool used = o < g.nobj;
        rec[cfg->off_app + o] = used ? w.app[o] : 0;
        rec[cfg->off_pos + 2 * o] = used ? w.px[o] : 0;
        rec[cfg->off_pos + 2 * o + 1] = used ? w.py[o] : 0;
    }
    for (int o = 0; o < cfg->maxo; ++o) rec[cfg->off_cont + o] = o < g.nobj ? w.cont[o] : NONE8;
    memcpy(rec + cfg->off_prog, &w.prog, sizeof(Prog));
    Hot h;
    memset(&h, 0, sizeof(h));
    h.ax = g.ax; h.ay = g.ay; h.dir = g.adir; h.carry = NONE8;
    h.step = 0; h.max_steps = (uint16_t)max_steps;
    h.pre4 = 0xFFFFFFFFu;
    h.last_locked = g.last_locked < 0 ? NONE8 : (uint8_t)g.last_locked;
    *hot = h;
    return g.nobj;
}

// The lane = level generator (bbai_genl.hpp) on the host: a lane's word array and a plain MT19937 state.  Same outputs as hs_generate
// expected, byte for byte (tests/test_hostsim_genl.py).  Returns nobj, or -1 when the level kind is not covered by this generator.
struct HostLaneMem {
    uint32_t w[160];
    uint32_t* mt; int mti;
    uint32_t ld(int k) const { return w[k]; }
    void st(int k, uint32_t v) { w[k] = v; }
    uint32_t next_u32() {
        if (mti >= MT_N) { mt_twist(HostCtx(), mt); mti = 0; }
        return mt_temper(mt[mti++]);
    }
    uint32_t draw_masked(uint32_t mask, uint32_t rng) { uint32_t v; do { v = next_u32() & mask; } while (v > rng); return v; }
    void topup() {}
};
int hs_generate_lane(const LevelCfg* cfg, uint32_t* mt, int32_t* mti, uint8_t* rec, Hot* hot) {
    if (!lane_gen_ok(*cfg)) return -1;
    static thread_local HostLaneMem mem;
    memset((void*)mem.w, 0xA5, sizeof(mem.w));   // (the device's LDS starts as whatever it held)
    mem.mt = mt; mem.mti = *mti;
    static thread_local uint8_t tmpl[4096];
    lane_build_template(*cfg, tmpl);
    int last_locked = hot->last_locked == NONE8 ? -1 : hot->last_locked;
    GenL<HostLaneMem> g(mem, *cfg, last_locked);
    for (int attempts = 0;; ++attempts) {
        if (attempts >= Gen<HostCtx>::MAX_ATTEMPTS) return -2;
        bool ok = cfg->kind == K_LEVELGEN ? g.attempt<K_LEVELGEN>() : g.attempt<K_GOTO>();
        if (ok) break;
    }
    *mti = mem.mti;
    memset(rec, 0xEE, cfg->rec_bytes);           // (the record is written completely)
    alignas(16) static thread_local uint8_t arec[4096];
    g.write_record(arec, tmpl);
    memcpy(rec, arec, cfg->rec_bytes);
    Hot h;
    memset(&h, 0, sizeof(h));
    h.ax = g.ax; h.ay = g.ay; h.dir = g.adir; h.carry = NONE8;
    h.step = 0; h.max_steps = (uint16_t)g.max_steps();
    h.pre4 = 0xFFFFFFFFu;
    h.last_locked = g.last_locked < 0 ? NONE8 : (uint8_t)g.last_locked;
    *hot = h;
    return g.nobj;
}
// ... and with the DEVICE's RNG plumbing emulated: the env's stream as two tempered generations + a signed position (LaneRng), the
// cooperative twist at the top of every attempt (k_pregen_lane: every lane whose position is >= 0), the lone twist inside a draw.
// State in: canonical (mt, mti) ; out: canonical again (what k_mt_canon makes of it), so that it compares with hs_generate.
struct HostLaneMem2 : LaneRng<HostLaneMem2> {
    uint32_t* w;                                 // (a pointer, as on the device: lane_fill works on a COPY of the policy object)
    uint32_t ld(int k) const { return w[k]; }
    void st(int k, uint32_t v) { w[k] = v; }
    void topup() { if (low()) refill(); }
};
// `st` = the env's persistent device-side RNG state between calls: [2 * MT_N] tempered halves, then par, then a "valid" word (0: build it
// from the canonical (mt, *mti), as k_mt_sync does after a seed / import).  *mti is the SIGNED position on the way out unless `canon`.
int hs_generate_lane2(const LevelCfg* cfg, uint32_t* mt, int32_t* mti, uint8_t* rec, Hot* hot, int32_t* lone_twists, uint32_t* st, int canon) {
    if (!lane_gen_ok(*cfg)) return -1;
    static thread_local HostLaneMem2 mem;
    static thread_local uint32_t words[160];
    uint32_t* mtt = st;
    mem.w = words;
    memset((void*)words, 0xA5, sizeof(words));
    if (!st[2 * MT_N + 1]) {
        for (int k = 0; k < MT_N; ++k) { mtt[k] = mt_temper(mt[k]); mtt[MT_N + k] = 0xDEADBEEFu; }       // k_mt_sync
        st[2 * MT_N] = 0; st[2 * MT_N + 1] = 1;
    }
    mem.mts_env = mt; mem.mtt_env = mtt; mem.fifo0 = lane_layout(*cfg).fifo; mem.start(*mti, (int)st[2 * MT_N]);
    static thread_local uint8_t tmpl[4096];
    lane_build_template(*cfg, tmpl);
    int last_locked = hot->last_locked == NONE8 ? -1 : hot->last_locked;
    GenL<HostLaneMem2> g(mem, *cfg, last_locked);
    int coop = 0, total = 0;
    for (int attempts = 0;; ++attempts) {
        if (attempts >= Gen<HostCtx>::MAX_ATTEMPTS) return -2;
        if (mem.position() >= 0) {               // the wave's twist of this lane's env
            mt_twist(HostCtx(), mt);
            for (int k = 0; k < MT_N; ++k) mtt[(mem.par ^ 1) * MT_N + k] = mt_temper(mt[k]);
            mem.twisted(); ++coop;
        }
        const int p0 = mem.position(), q0 = mem.par;
        bool ok = cfg->kind == K_LEVELGEN ? g.attempt<K_LEVELGEN>() : g.attempt<K_GOTO>();
        (void)p0; if (mem.par != q0) ++total;
        if (ok) break;
    }
    if (lone_twists) *lone_twists = total;
    // canonical form back (k_mt_canon)
    int pos = mem.position();
    if (canon && pos < 0) {
        const uint32_t* prev = mtt + (mem.par ^ 1) * MT_N;
        for (int k = 0; k < MT_N; ++k) mt[k] = mt_untemper(prev[k]);
        pos += MT_N;
        mem.par ^= 1;
    }
    *mti = pos;
    st[2 * MT_N] = (uint32_t)mem.par;
    alignas(16) static thread_local uint8_t arec[4096];
    g.write_record(arec, tmpl);
    memcpy(rec, arec, cfg->rec_bytes);
    Hot h;
    memset(&h, 0, sizeof(h));
    h.ax = g.ax; h.ay = g.ay; h.dir = g.adir; h.carry = NONE8;
    h.step = 0; h.max_steps = (uint16_t)g.max_steps();
    h.pre4 = 0xFFFFFFFFu;
    h.last_locked = g.last_locked < 0 ? NONE8 : (uint8_t)g.last_locked;
    *hot = h;
    return g.nobj;
}
uint32_t hs_untemper(uint32_t y) { return mt_untemper(y); }
uint32_t hs_temper(uint32_t y) { return mt_temper(y); }

// PutNext*Carrying (bonus_levels.py:821-829): AFTER the first observation the object leaves the grid into the
// agent's hands.  Call once after hs_generate + hs_observe; returns 1 if something was picked up.
int hs_start_carry(const LevelCfg* cfg, uint8_t* rec, Hot* hot, uint64_t* stale) {
    const Prog* p = (const Prog*)(rec + cfg->off_prog);
    if (p->start_carry == NONE8) return 0;
    apply_start_carry(*cfg, rec, *hot, *stale, p->start_carry);
    return 1;
}

int hs_step64(const LevelCfg* cfg, uint8_t* rec, Hot* hot, uint64_t* stale, int action, double* reward) {
    const Prog* p = (const Prog*)(rec + cfg->off_prog);
    uint64_t sets[8];
    for (int k = 0; k < 8; ++k) sets[k] = p->set[k >> 1][k & 1];
    VProg vp; vp.bind(vhead_pack(*p), sets, 1);
    return step_env_cmd(*cfg, rec, vp, *hot, *stale, action, *reward) ? 1 : 0;
}
// the step in k_step's order of operations (bbai_step.hpp step_env_prefetch); lsm may be NULL (normal mode)
int hs_step64_prefetch(const LevelCfg* cfg, uint8_t* rec, Hot* hot, uint64_t* stale, int action, double* reward, uint32_t* lsm) {
    const Prog* p = (const Prog*)(rec + cfg->off_prog);
    uint64_t sets[8];
    for (int k = 0; k < 8; ++k) sets[k] = p->set[k >> 1][k & 1];
    VProg vp; vp.bind(vhead_pack(*p), sets, 1);
    if (action == A_RESET_ENV) { *reward = 0.0; return 1; }
    return step_env_prefetch(*cfg, rec, vp, *hot, *stale, action, *reward, lsm) ? 1 : 0;
}
// the same step in the reference's BABYAI_DONE_ACTIONS mode; *lsm = the env's lastStepMatch bits (0 at episode start)
int hs_step64_done(const LevelCfg* cfg, uint8_t* rec, Hot* hot, uint64_t* stale, int action, double* reward, uint32_t* lsm) {
    const Prog* p = (const Prog*)(rec + cfg->off_prog);
    uint64_t sets[8];
    for (int k = 0; k < 8; ++k) sets[k] = p->set[k >> 1][k & 1];
    VProg vp; vp.bind(vhead_pack(*p), sets, 1);
    return step_env_cmd(*cfg, rec, vp, *hot, *stale, action, *reward, lsm) ? 1 : 0;
}
// ... with this step's `done` counting as the ENUM member (verifier.py:543-545 `action is self.env.actions.done`); order: 0 = the
// reference's order of operations, 1 = k_step's
int hs_step64_done_enum(const LevelCfg* cfg, uint8_t* rec, Hot* hot, uint64_t* stale, int action, double* reward, uint32_t* lsm, int order) {
    const Prog* p = (const Prog*)(rec + cfg->off_prog);
    uint64_t sets[8];
    for (int k = 0; k < 8; ++k) sets[k] = p->set[k >> 1][k & 1];
    VProg vp; vp.bind(vhead_pack(*p), sets, 1);
    if (action == A_RESET_ENV) { *reward = 0.0; return 1; }
    return (order ? step_env_prefetch(*cfg, rec, vp, *hot, *stale, action, *reward, lsm, true)
                  : step_env(*cfg, rec, vp, *hot, *stale, action, *reward, lsm, true)) ? 1 : 0;
}
int hs_step(const LevelCfg* cfg, uint8_t* rec, Hot* hot, uint64_t* stale, int action, float* reward) {
    double r = 0.0;
    const int d = hs_step64(cfg, rec, hot, stale, action, &r);
    *reward = (float)r;
    return d;
}

void hs_observe(const LevelCfg* cfg, const uint8_t* rec, const Hot* hot, uint8_t* out) {
    observe_env(*cfg, rec, *hot, out);
}
// the same observation through k_step's register pipeline (bbai_view.hpp view_cells_perm + encode_cells); returns the front cell's appearance
int hs_observe_perm(const LevelCfg* cfg, const uint8_t* rec, const Hot* hot, int nfe, uint8_t* out) {
    alignas(16) uint8_t rows[ROWS_FRONT + OBS_BYTES + 16];
    memset(rows, 0xEE, sizeof(rows));
    const int fe2 = observe_env_perm(*cfg, rec, *hot, nfe, rows);
    memcpy(out, rows + ROWS_FRONT, OBS_BYTES);
    return fe2;
}

// ... and through the C plane path of the small single rooms (bbai_view.hpp window_rows_cpl + view_rows_perm): the env's row is built from the
// record (cpl_from_record: what the generator / k_sync_cpl write), the observation from the row alone.  row_out: cpl_bytes(cfg) bytes.
int hs_cpl_ok(const LevelCfg* cfg) { return cpl_ok(*cfg) ? cpl_bytes(*cfg) : 0; }
int hs_observe_cpl(const LevelCfg* cfg, const uint8_t* rec, const Hot* hot, int nfe, uint8_t* out, uint8_t* row_out) {
    alignas(16) uint8_t rows[ROWS_FRONT + OBS_BYTES + 16];
    alignas(16) uint8_t row[CPL_PLANE + CPL_MAX_IDS];
    memset(rows, 0xEE, sizeof(rows));
    cpl_from_record(*cfg, rec, row);
    const uint32_t ce = hot->carry != NONE8 ? rec[cfg->off_app + hot->carry] : (uint32_t)E_EMPTY;
    const int fe2 = observe_cpl_perm(row, cfg->H, *hot, ce, nfe, rows);
    memcpy(out, rows + ROWS_FRONT, OBS_BYTES);
    if (row_out) memcpy(row_out, row, cpl_bytes(*cfg));
    return fe2;
}
int hs_cid_lookup(const uint8_t* ids, int nbytes, int pos) {
    uint32_t d[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    memcpy(d, ids, nbytes);
    return cid_lookup(d, nbytes / 4, pos);
}

// ---- k_step's LDS row packing (bbai_step.hpp RowPacker), one lane at a time in the caller's lane order ---------------------
// rows_in: [n][37] dwords (147 bytes + one pad byte each); lds: ROWS_FRONT + n * 147 + 16 bytes, pre-filled by the caller;
// scratch_fill != 0 also scribbles over every lane's window scratch first (it must stay inside the lane's own row).
void hs_pack_rows(const uint32_t* rows_in, const int32_t* order, int n, uint8_t* lds, int scratch_fill) {
    uint8_t* rows = lds + ROWS_FRONT;
    if (scratch_fill)
        for (int i = 0; i < n; ++i) memset(rows + row_scratch(i), 0xEE, 56);
    for (int k = 0; k < n; ++k) {
        const int r = order[k];
        RowPacker o(rows, r);
        for (int j = 0; j < 37; ++j) o.put(j, rows_in[r * 37 + j]);
        o.finish();
    }
}
int hs_row_scratch(int r) { return row_scratch(r); }

// ---- the expert (bbai_bot.hpp) -------------------------------------------------------------------------------
int hs_bot_state_bytes(int stack_cap) { return (int)bot_state_bytes(stack_cap); }
int hs_bot_dead_reason(const uint8_t* state) { return ((const BotState*)state)->dead; }
int hs_bot_stack_depth(const uint8_t* state) { return ((const BotState*)state)->sp; }

// One Bot.replan decision; `first` != 0 starts a fresh Bot (new episode).  action_taken < 0 = None.
// Returns the suggested action, or 255 once the bot is dead (state->dead says why).
static int g_bot_eager = 0;
void hs_bot_set_eager(int on) { g_bot_eager = on; }
void hs_bot_set_aligned(int on) { g_bot_aligned_ok = on != 0; }

}  // extern "C"

// Lane groups on the host: the group's lanes are fibers (ucontext) that run ONE AT A TIME from collective to collective -- every
// ballot() / sync() of bbai_bot.hpp's Ctx contract hands control back to a scheduler, which resumes the lanes once all of them have
// arrived.  Stricter than a wavefront (a lane runs its whole segment before the next lane starts it, so code that only works in
// lockstep, or that misses a sync() between one lane's write and another's read, fails here), deterministic, single-threaded.
struct EmuGroup {
    static constexpr int MAXL = 64;
    static constexpr size_t STACK = 512 * 1024;
    int lanes = 0, cur = 0;
    ucontext_t main_ctx, fib[MAXL];
    char* stacks = nullptr;
    bool done[MAXL];
    int waiting_kind[MAXL];          // 0 = running / done, 1 = sync, 2 = ballot
    unsigned long long bits = 0, result = 0;
    int ret[MAXL];
    const char* error = nullptr;
    // the decision's arguments
    const LevelCfg* cfg; const uint8_t* rec; const Hot* hot; uint64_t stale; BotState* state; Subgoal* stk; int cap; BotWork work; bool first; int action;
};
static thread_local EmuGroup* g_emu = nullptr;
static unsigned long long emu_collective(int kind, bool p) {
    EmuGroup* g = g_emu;
    const int me = g->cur;
    if (p) g->bits |= 1ull << me;
    g->waiting_kind[me] = kind;
    swapcontext(&g->fib[me], &g->main_ctx);
    return g->result;
}
template <int G>
struct EmuCtx {
    static constexpr int kLanes = G;
    int lane() const { return g_emu->cur; }
    int nlanes() const { return G; }
    void sync() const { (void)emu_collective(1, false); }
    unsigned long long ballot(bool p) const { return emu_collective(2, p); }
};
template <int G>
static void emu_lane_entry() {
    EmuGroup* g = g_emu;
    const int me = g->cur;
    const int a = bot_decide(EmuCtx<G>(), *g->cfg, g->rec, *g->hot, g->stale, *g->state, g->stk, g->cap, g->work, g->first, g->action);
    g = g_emu;
    g->ret[me] = a;
    g->done[me] = true;
    g->waiting_kind[me] = 0;
}
template <int G>
static int emu_decide(EmuGroup& g) {
    if (!g.stacks) g.stacks = (char*)malloc(EmuGroup::MAXL * EmuGroup::STACK);
    g.lanes = G; g.bits = 0; g.result = 0; g.error = nullptr;
    for (int l = 0; l < G; ++l) {
        g.done[l] = false; g.waiting_kind[l] = 0; g.ret[l] = -2;
        getcontext(&g.fib[l]);
        g.fib[l].uc_stack.ss_sp = g.stacks + (size_t)l * EmuGroup::STACK;
        g.fib[l].uc_stack.ss_size = EmuGroup::STACK;
        g.fib[l].uc_link = &g.main_ctx;
        makecontext(&g.fib[l], (void (*)())emu_lane_entry<G>, 0);
    }
    g_emu = &g;
    for (;;) {
        for (int l = 0; l < G; ++l)
            if (!g.done[l]) { g.cur = l; swapcontext(&g.main_ctx, &g.fib[l]); }
        int n_done = 0, kind = 0;
        bool mixed = false;
        for (int l = 0; l < G; ++l) {
            if (g.done[l]) { ++n_done; continue; }
            if (kind && g.waiting_kind[l] != kind) mixed = true;
            kind = g.waiting_kind[l];
        }
        if (n_done == G) break;
        if (n_done || mixed) { g.error = "the lanes of a group diverged around a collective"; break; }
        g.result = g.bits; g.bits = 0;
    }
    g_emu = nullptr;
    if (g.error) return -3;
    for (int l = 1; l < G; ++l) if (g.ret[l] != g.ret[0]) return -4;      // the lanes disagree on the decision
    return g.ret[0];
}
extern "C" {
void hs_bot_counts(long long* out, int reset) { for (int k = 0; k < 8; ++k) { out[k] = g_bot_counts[k]; if (reset) g_bot_counts[k] = 0; } }
static int g_bot_lanes = 1;
void hs_bot_set_lanes(int lanes) { g_bot_lanes = lanes; }

int hs_bot_decide(const LevelCfg* cfg, const uint8_t* rec, const Hot* hot, const uint64_t* stale, uint8_t* state, int stack_cap,
                  int first, int action_taken) {
    static thread_local uint16_t buf[BOT_WORK_WORDS];
    static thread_local uint32_t rows[R_ALL * MAX_W];
    BotWork work; work.base = buf; work.stride = 1; work.cells = cfg->W * cfg->H;
    work.eager = g_bot_eager;
    if (g_bot_lanes > 1) {            // the lane-group kernel's layout: all four row masks together, search 1's arrays apart, no ring
        static thread_local uint16_t near_q[2 * BOT_MAX_CELLS];
        static thread_local EmuGroup group;
        work.near_q = near_q;
        work.ring = nullptr; work.ring_stride = 0; work.ring_size = 0;
        work.rows_fast = rows; work.rstride_fast = 1; work.rows_h = cfg->H; work.fast_n = R_ALL; work.rows_slow = nullptr; work.rstride_slow = 0;
        group.cfg = cfg; group.rec = rec; group.hot = hot; group.stale = *stale; group.state = (BotState*)state;
        group.stk = (Subgoal*)((BotState*)state + 1); group.cap = stack_cap; group.work = work; group.first = first != 0; group.action = action_taken;
        switch (g_bot_lanes) {
        case 4: return emu_decide<4>(group);
        case 8: return emu_decide<8>(group);
        case 16: return emu_decide<16>(group);
        case 32: return emu_decide<32>(group);
        default: return -5;
        }
    }
    static thread_local uint16_t ring[8]; work.ring = ring; work.ring_stride = 1; work.ring_size = 8;   /* tiny: the fall-back to the queue proper is exercised */ work.rows_fast = rows; work.rstride_fast = 1; work.rows_h = MAX_W; work.rows_slow = rows + R_FAST * MAX_W; work.rstride_slow = 1;
    return bot_decide(*cfg, rec, *hot, *stale, *(BotState*)state, stack_cap, work, first != 0, action_taken);
}

}  // extern "C"
