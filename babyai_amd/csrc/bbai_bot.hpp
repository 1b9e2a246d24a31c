// bbai_bot.hpp -- the GOFAI expert of the reference (babyai/bot.py), one decision per env-step, as scalar
// per-env code over the engine's env record (lane = env on the device; the same header builds for the host in
// tests/hostsim).  It reproduces `Bot.replan(action_taken)` DECISION FOR DECISION, including the points where the
// reference bot gives up (assertion / DisappearedBoxError / AttributeError -> `dead`), because demonstrations are
// only comparable when the two experts act identically (babyai/utils/agent.py:139-146, scripts/make_agent_demos.py).
//
// Follows (reference file:line, /root/reference/babyai/bot.py):
//   Subgoal._plan_undo_action                         :109-137
//   Close / Open / Drop / Pickup subgoals             :139-275
//   GoNextToSubgoal                                   :277-449
//   ExploreSubgoal                                    :452-500
//   Bot.__init__ / _process_instr                     :524-545, :900-939
//   Bot.replan                                        :547-597
//   Bot._find_obj_pos                                 :599-656   (obj_set / obj_poss index quirk kept, see find_obj_pos)
//   Bot._process_obs / _remember_current_state        :658-695
//   Bot._closest_wall_or_door_given_dir               :697-708
//   Bot._breadth_first_search / _shortest_path        :710-806
//   Bot._find_drop_pos                                :808-898
//   Bot._check_erroneous_box_opening                  :941-949
//   ObjDesc.find_matching_objs (key descriptors)      babyai/levels/verifier.py:96-161
//
// BFS notes: (1) the reference appends every neighbour to a FIFO and discards already-visited positions when they are
// popped; since the first entry queued for a position is also the first popped, marking positions when they are
// QUEUED visits the same positions in the same order with the same predecessors, and bounds the queue by W*H.
// (2) every search of one decision runs on the same grid from the same state, so two resumable search trees per
// decision (see `search`) replace the reference's one-or-two searches per query.
//
// Execution model (round 5): Bot<Ctx> -- ONE LANE GROUP = ONE ENV, Ctx::kLanes lanes wide, as in bbai_gen.hpp.  The subgoal machine is
// group-uniform code (every lane of the group computes the same values; shared state is only written with values all lanes agree on),
// the data-parallel parts split over the lanes: the 49 view cells of _process_obs, the grid rows of the search masks, the four
// neighbours of a popped position (queue order kept by a ballot prefix), the acceptance scan over the positions a search has popped,
// the grid scan for keys.  Ctx contract: kLanes, lane(), nlanes(), sync() (the group's memory accesses before / after are ordered),
// ballot(pred) (bit k = lane k of the group).  OneLane (the host build, and k_bot's lane = env form) makes all of it sequential.
#pragma once
#include "bbai_types.hpp"
#include "bbai_step.hpp"

// Optional phase timers of the expert (experiment builds only: -DBBAI_BOT_PROF, tools/bot_prof.sh): wave wall-clock ticks
// per phase, added by the first active lane of a scope.  Expands to nothing in the product build.
#if defined(BBAI_BOT_PROF) && defined(__HIP_DEVICE_COMPILE__)
extern __device__ unsigned long long g_bot_prof[32];
struct BotProfScope {
    int ph; unsigned long long t0;
    __device__ BotProfScope(int p) : ph(p), t0(wall_clock64()) {}
    __device__ ~BotProfScope() {
        const unsigned long long dt = wall_clock64() - t0;
        const unsigned long long act = __ballot(1);
        if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) { atomicAdd(&g_bot_prof[ph], dt); atomicAdd(&g_bot_prof[16 + ph], 1ull); }
    }
};
#define BOT_PROF(ph) BotProfScope bot_prof_scope_##ph(ph)
#else
#define BOT_PROF(ph)
#endif
enum { BP_DECIDE = 0, BP_OBS = 1, BP_AFTER = 2, BP_FIND_OBJ = 3, BP_PATH = 4, BP_SEARCH = 5, BP_ROWS = 6, BP_DROP_POS = 7, BP_BEFORE = 8,
       BP_INIT = 9, BP_KEYS = 10 };

#ifndef BBAI_BOT_COUNT
#define BBAI_BOT_COUNT(what)          // (tests/hostsim counts pops / queries per decision with this; nothing in the product)
#endif

#ifndef BBAI_BOT_ALIGNED_OK
#define BBAI_BOT_ALIGNED_OK true     // _find_obj_pos: take the list-free pairing when every object of the descriptor is still where the episode put it
#endif

namespace bbai {

struct OneLane {
    static constexpr int kLanes = 1;
    BB_HD int lane() const { return 0; }
    BB_HD int nlanes() const { return 1; }
    BB_HD void sync() const {}
    BB_HD unsigned long long ballot(bool p) const { return p ? 1ull : 0ull; }
};

constexpr int BOT_STACK = 48;       // default subgoal stack depth (overflow => dead, counted); BBAI_BOT_STACK raises it
constexpr int BOT_KEYS = 12;        // same-colour keys a key descriptor can list (overflow => dead, counted)
constexpr int BOT_MAX_CELLS = MAX_W * MAX_W;
constexpr int BOT_MAX_ITERS = 1000; // replanning rounds per decision (the reference would spin for ever)
constexpr int BOT_DEAD = 0xFF;      // action value reported for a dead bot
constexpr int BOT_RING = 64;        // LDS ring of the first search's newest queue entries (k_bot)

enum : uint8_t { SG_CLOSE = 0, SG_OPEN, SG_DROP, SG_PICKUP, SG_GONEXT, SG_EXPLORE };
enum : uint8_t { RS_NONE = 0, RS_UNLOCK, RS_KEEPKEY, RS_PUTNEXT, RS_EXPLORE, RS_OPEN };
enum : uint8_t { DT_NONE = 0, DT_DESC, DT_KEYS, DT_OBJ, DT_POS };
enum : uint8_t { DEAD_NO = 0, DEAD_REFERENCE = 1, DEAD_CAPACITY = 2 };   // 2: our fixed-size structures overflowed

struct Subgoal {            // 32 bytes
    uint8_t kind, reason, dtype;
    uint8_t a, b;           // DT_DESC: a = 2*leaf+slot; DT_OBJ: a = object; DT_POS: (a, b) = (x, y) as int8
    uint8_t nkeys;          // DT_KEYS: frozen obj_set / obj_poss of ObjDesc('key', colour)
    uint16_t keys[BOT_KEYS];// object << 10 | x << 5 | y
    uint8_t pad[2];
};
static_assert(sizeof(Subgoal) == 32, "Subgoal layout");

struct BotState {
    uint32_t vis[MAX_W];            // vis[y] bit x : Bot.vis_mask
    uint8_t ipos[MAX_OBJ][2];       // object positions when the episode started (order of ObjDesc.obj_set)
    uint16_t sp;                    // subgoal stack: `cap` entries of 32 bytes stored BEHIND this struct
    uint8_t dead, pad0;
    uint8_t prev_ax, prev_ay, prev_carry;
    uint8_t door_was_open;          // 0 / 1, 2 = attribute never set
    uint8_t prev_fwd_type;          // T_* of prev_fwd_cell, 0 = None
    uint8_t pad;
    uint16_t next_step;             // step_count at which the next decision of THIS plan is expected
};
static_assert(sizeof(BotState) % 4 == 0, "the subgoal stack follows the state");
BB_HD size_t bot_state_bytes(int stack_cap) { return sizeof(BotState) + (size_t)stack_cap * sizeof(Subgoal); }

// BFS scratch: four uint16 arrays of BOT_MAX_CELLS (predecessors + queue, two searches alive at once in
// _shortest_path) behind a strided view, so the kernel can choose the layout (k_bot: contiguous per thread).
constexpr int BOT_WORK_WORDS = 4 * BOT_MAX_CELLS;
enum { WK_PREV1 = 0, WK_Q1 = 1, WK_PREV2 = 2, WK_Q2 = 3 };
// Row bitmasks (bit x of row y) that turn the per-pop grid / visited look-ups of the searches into bit tests: what a
// position may be expanded through (search 1: free cells and open doors; search 2: blockers too) and which positions a
// search has queued.  The two of search 1 are the hot ones (k_bot keeps them in LDS, [row][lane]: conflict-free, H rows
// each), search 2's live in slower memory (it only runs when search 1 failed).
enum { R_EXP1 = 0, R_VIS1 = 1, R_FAST = 2, R_EXP2 = 2, R_VIS2 = 3, R_ALL = 4 };
struct BotWork {
    uint16_t* base;
    int stride;
    int cells;                      // W * H of the level: the four arrays are packed to the grid actually in use
    uint16_t* near_q = nullptr;     // lane-group kernel: search 1's predecessor + queue arrays [2][cells] in LDS (search 2's stay in `base`)
    uint32_t* rows_fast;            // [fast_n][rows_h] x rstride_fast
    uint32_t* rows_slow;            // [R_ALL - fast_n][MAX_W] x rstride_slow
    int fast_n = R_FAST;            // row-mask arrays held in rows_fast (lane-group kernel: all four)
    int rstride_fast, rstride_slow, rows_h;
    uint16_t* ring;                 // the newest `ring_size` (power of two) queue entries of search 1, entry i at
    int ring_stride, ring_size;     // ring[(i & (ring_size - 1)) * ring_stride]; 0 = no ring (host).  A FIFO's live part is
                                    // its frontier, which is short: the eager expansion pops from LDS instead of waiting
                                    // for a dependent global load per position
    int eager;                      // 1: run search 1 to exhaustion at the top of every decision, while the lanes of a wave are
                                    // still together (queries then only look things up); 0: expand lazily inside the queries
    BB_HD uint16_t& at(int arr, int i) const {
        if (near_q && arr < WK_PREV2) return near_q[arr * cells + i];
        return base[(int64_t)(arr * cells + i) * stride];
    }
    BB_HD uint32_t& row(int arr, int y) const {
        return arr < fast_n ? rows_fast[(arr * rows_h + y) * rstride_fast] : rows_slow[(int64_t)((arr - fast_n) * MAX_W + y) * rstride_slow];
    }
};

template <class Ctx>
struct Bot {
    Ctx ctx;
    const LevelCfg& c;
    const uint8_t* rec;
    const Hot& h;
    uint64_t stale;
    BotState& s;                    // shared by the group's lanes: written only with values every lane agrees on
    Subgoal* stk;                   // the subgoal stack (`cap` entries)
    int cap;
    BotWork w;
    const uint8_t *E, *I, *app, *pos;
    const Prog* prog;
    bool raised;
    // per-lane copies of what a decision reads of the previous one and rewrites at its end (a lane must not see another lane's update)
    int sp;
    uint8_t prev_ax, prev_ay, prev_carry, door_was_open, prev_fwd_type;

    BB_HD Bot(Ctx ctx_, const LevelCfg& c_, const uint8_t* rec_, const Hot& h_, uint64_t stale_, BotState& s_, Subgoal* stk_, int cap_, const BotWork& w_)
        : ctx(ctx_), c(c_), rec(rec_), h(h_), stale(stale_), s(s_), stk(stk_), cap(cap_), w(w_), raised(false) {
        E = rec; I = rec + c.off_I; app = rec + c.off_app; pos = rec + c.off_pos;
        prog = (const Prog*)(rec + c.off_prog);
        sp = s.sp; prev_ax = s.prev_ax; prev_ay = s.prev_ay; prev_carry = s.prev_carry; door_was_open = s.door_was_open; prev_fwd_type = s.prev_fwd_type;
    }

    // ---- small helpers -------------------------------------------------------------------------------------
    BB_HD void die(int why = DEAD_REFERENCE) { if (!raised) { raised = true; s.dead = (uint8_t)why; } }
    BB_HD int cell(int x, int y) const { return E[e_index(c, x, y)]; }               // margin makes +-5 safe
    BB_HD bool in_grid(int x, int y) const { return x >= 0 && y >= 0 && x < c.W && y < c.H; }
    BB_HD bool seen(int x, int y) const { return in_grid(x, y) && (s.vis[y] >> x & 1); }
    BB_HD static bool is_none(int e) { return e == E_EMPTY; }
    BB_HD static bool open_door(int e) { return e_type(e) == T_DOOR && e_state(e) == S_OPEN; }
    BB_HD int fx() const { return h.ax + dir_dx(h.dir); }
    BB_HD int fy() const { return h.ay + dir_dy(h.dir); }
    BB_HD int rx() const { return -dir_dy(h.dir); }                                   // right_vec = (-dy, dx)
    BB_HD int ry() const { return dir_dx(h.dir); }
    BB_HD bool carrying() const { return h.carry != NONE8; }
    BB_HD bool obj_in_grid(int o) const {
        int x = pos[2 * o], y = pos[2 * o + 1];
        return in_grid(x, y) && I[i_index(c, x, y)] == o + 2;
    }

    BB_HD void push(const Subgoal& g) {
        if (sp >= cap) { die(DEAD_CAPACITY); return; }
        stk[sp++] = g;
    }
    BB_HD void pop() { if (sp) --sp; }
    BB_HD static Subgoal mk(int kind, int reason = RS_NONE) {
        Subgoal g = {};
        g.kind = (uint8_t)kind; g.reason = (uint8_t)reason; g.dtype = DT_NONE;
        return g;
    }
    BB_HD static Subgoal go_pos(int x, int y, int reason = RS_NONE) {
        Subgoal g = mk(SG_GONEXT, reason);
        g.dtype = DT_POS; g.a = (uint8_t)(int8_t)x; g.b = (uint8_t)(int8_t)y;
        return g;
    }
    // GoNextToSubgoal(bot, drop_pos) where _find_drop_pos may have returned None: tuple(None) raises when used
    BB_HD static Subgoal go_maybe(bool ok, int x, int y) { return ok ? go_pos(x, y) : mk(SG_GONEXT); }
    BB_HD static Subgoal go_desc(int k, int reason = RS_NONE) {
        Subgoal g = mk(SG_GONEXT, reason);
        g.dtype = DT_DESC; g.a = (uint8_t)k;
        return g;
    }
    BB_HD static Subgoal go_obj(int o, int reason) {
        Subgoal g = mk(SG_GONEXT, reason);
        g.dtype = DT_OBJ; g.a = (uint8_t)o;
        return g;
    }

    // ---- BFS (bot.py:710-806) ---------------------------------------------------------------------------------
    enum { ACC_POS, ACC_UNSEEN, ACC_DOOR_UNLOCKED, ACC_DOOR, ACC_UNBLOCK, ACC_EMPTY };
    struct Accept { int kind, x, y; bool has_except; int ex, ey; };

    BB_HD bool match_empty(int x, int y, const Accept& a) const {                     // :873-885
        if (x == h.ax && y == h.ay) return false;
        if (a.has_except && x == a.ex && y == a.ey) return false;
        return seen(x, y) && is_none(cell(x, y));
    }
    BB_HD bool match_unblock(int x, int y, const Accept& a) const {                   // :815-871
        if (!match_empty(x, y, a)) return false;
        const int dk[8] = {-1, 0, 1, 1, 1, 0, -1, -1}, dl[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
        int cls[8];
        for (int q = 0; q < 8; ++q) {
            const int nx = x + dk[q], ny = y + dl[q];
            const int e = cell(nx, ny);
            const bool v = seen(nx, ny);
            if (v && e_type(e) == T_WALL) cls[q] = 1;
            else if (v && (is_none(e) || open_door(e) || (nx == h.ax && ny == h.ay)) && !(a.has_except && nx == a.ex && ny == a.ey)) cls[q] = 0;
            else cls[q] = 2;
        }
        int changes = 0;
        for (int q = 0; q < 8; ++q) if ((cls[(q + 1) % 8] != 0) != (cls[q] != 0)) ++changes;
        for (int q = 0; q < 8; ++q)
            if (cls[q] == 2 && cls[(q + 7) % 8] != 0 && cls[(q + 1) % 8] != 0) return false;
        return changes <= 2;
    }
    BB_HD bool accept(const Accept& a, int x, int y, int e) const {
        switch (a.kind) {
        case ACC_POS: return x == a.x && y == a.y;
        case ACC_UNSEEN: return !seen(x, y);
        case ACC_DOOR_UNLOCKED: return e_type(e) == T_DOOR && e_state(e) == S_CLOSED;
        case ACC_DOOR: return e_type(e) == T_DOOR && e_state(e) != S_OPEN;
        case ACC_UNBLOCK: return match_unblock(x, y, a);
        default: return match_empty(x, y, a);
        }
    }

    // The searches of one decision share their work.  Which positions a search pops, in which order and with which
    // predecessor does not depend on the acceptance test (the reference merely stops at the first accepted position
    // it pops), and the env does not change inside Bot.replan.  So each of the two searches of _shortest_path (from the
    // agent / from everything the first one reached, through blockers) is kept as ONE resumable tree per decision:
    // a query first looks through the positions already popped, in pop order, then keeps popping until the test
    // accepts.  Queue entries: packed position | dir << 10; prev: predecessor (packed), 0xFFFE = None (an initial state).
    mutable int head1 = 0, qn1 = -1, head2 = 0, qn2 = -1;          // qn < 0: search not started in this decision

    // (prev array: predecessor of every queued position, WRITTEN when it is queued and only read back along the path of
    // an answer; "already queued" is the R_VIS row bit, "may be expanded" the R_EXP row bit)
    // Positions travel through the searches PACKED as y << 5 | x (no division by the grid width per pop); the predecessor
    // arrays are indexed y * W + x.
    BB_HD static int pk(int x, int y) { return y << 5 | x; }
    BB_HD int pidx(int p) const { return (p >> 5) * c.W + (p & 31); }
    // (a group's lanes leave expand() with the queue, the predecessors and the row masks ordered: sync() on every path that wrote)
    BB_HD void expand(int prev, int q, int& qn, int st, bool ignore_blockers) const {
        const int p = st & 1023, d = st >> 10;
        const int x = p & 31, y = p >> 5;
        BBAI_BOT_COUNT(ignore_blockers ? 1 : 0);
        // seen, and (empty | open door | with ignore_blockers: any object that is not a wall or a closed door)
        if (!(w.row(ignore_blockers ? R_EXP2 : R_EXP1, y) >> x & 1)) return;
        const int rv = ignore_blockers ? R_VIS2 : R_VIS1;
        if constexpr (Ctx::kLanes >= 4) {
            // lanes 0..3 take one neighbour each, in the reference's order (di,dj), (dj,di), (-dj,-di), (-di,-dj); the ballot's prefix count
            // gives each accepted neighbour its place in the queue.  Neighbours 0 / 3 and 1 / 2 are opposite: the horizontal pair shares
            // row y of the queued-mask, so each of the two writes BOTH bits (same value from both lanes); the vertical pair owns its rows.
            const int k = ctx.lane();
            const int ndk = k == 0 ? d : k == 1 ? (d ^ 1) : k == 2 ? (3 - d) : (d ^ 2);
            const int nx = x + dir_dx(ndk), ny = y + dir_dy(ndk);
            const bool ok = k < 4 && in_grid(nx, ny) && !(w.row(rv, ny) >> nx & 1);
            const unsigned m = (unsigned)ctx.ballot(ok) & 15u;
            if (!m) return;                                        // (group-uniform)
            if (ok) {
                uint32_t bits = 1u << nx;
                if (ny == y && (m >> (3 - k) & 1)) bits |= 1u << (2 * x - nx);
                w.row(rv, ny) |= bits;
                w.at(prev, ny * c.W + nx) = (uint16_t)p;
                w.at(q, qn + __builtin_popcount(m & ((1u << k) - 1u))) = (uint16_t)(pk(nx, ny) | ndk << 10);
            }
            qn += __builtin_popcount(m);
            ctx.sync();
            return;
        }
        const int nd[4] = {d, d ^ 1, 3 - d, d ^ 2};              // (di,dj), (dj,di), (-dj,-di), (-di,-dj)
        for (int k = 0; k < 4; ++k) {
            const int nx = x + dir_dx(nd[k]), ny = y + dir_dy(nd[k]);
            if (!in_grid(nx, ny)) continue;                       // (never happens: the border is wall)
            uint32_t& vr = w.row(rv, ny);
            if (vr >> nx & 1) continue;
            vr |= 1u << nx;
            w.at(prev, ny * c.W + nx) = (uint16_t)p;
            const uint16_t ent = (uint16_t)(pk(nx, ny) | nd[k] << 10);
            if (w.ring_size && q == WK_Q1) w.ring[(qn & (w.ring_size - 1)) * w.ring_stride] = ent;
            w.at(q, qn++) = ent;
        }
    }
    BB_HD int q1_get(int i, int qn) const {                       // entry i of search 1's queue (the ring holds the newest)
        if (w.ring_size && qn - i <= w.ring_size) return w.ring[(i & (w.ring_size - 1)) * w.ring_stride];
        return w.at(WK_Q1, i);
    }
    BB_HD bool queued(int rv, int p) const { return w.row(rv, p >> 5) >> (p & 31) & 1; }
    // first position (packed) in pop order that the test accepts, -1 if the search ends without one
    BB_HD int search(int prev, int q, int& head, int& qn, const Accept& a, bool ignore_blockers) const {
        BOT_PROF(BP_SEARCH);
        const int rv = ignore_blockers ? R_VIS2 : R_VIS1;
        if (a.kind == ACC_POS) {                                  // queued already => it will be popped and accepted
            if (!in_grid(a.x, a.y)) { while (head < qn) expand(prev, q, qn, w.at(q, head++), ignore_blockers); return -1; }
            const int target = pk(a.x, a.y);
            while (!queued(rv, target) && head < qn) expand(prev, q, qn, w.at(q, head++), ignore_blockers);
            return queued(rv, target) ? target : -1;
        }
        for (int base = 0; base < head; base += ctx.nlanes()) {   // the popped positions, nlanes() at a time, first accepted wins
            const int i = base + ctx.lane();
            bool ok = false;
            if (i < head) {
                const int p = w.at(q, i) & 1023;
                ok = accept(a, p & 31, p >> 5, cell(p & 31, p >> 5));
            }
            const unsigned long long m = ctx.ballot(ok);
            if (m) return w.at(q, base + __builtin_ctzll(m)) & 1023;
        }
        while (head < qn) {
            const int st = w.at(q, head);
            const int p = st & 1023;
            if (accept(a, p & 31, p >> 5, cell(p & 31, p >> 5))) return p;       // stays at the head for later queries
            expand(prev, q, qn, st, ignore_blockers);
            ++head;
        }
        return -1;
    }
    // the row masks of this decision's searches: one pass over the grid instead of a cell + visibility look-up per pop
    BB_HD void build_rows(bool second) const {
        BOT_PROF(BP_ROWS);
        // a grid row = W appearance bytes at a dword-aligned pitch: fetched as 7 independent dwords (25 cells + the margin's
        // odd byte) and tested from registers -- no load waits on another
        constexpr int LEAD = MARGIN & 3;
        for (int y = ctx.lane(); y < c.H; y += ctx.nlanes()) {
            const uint32_t sv = s.vis[y];
            const uint32_t* rp = (const uint32_t*)(E + (y + MARGIN) * c.ES + (MARGIN & ~3));
            uint32_t d[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) d[k] = 4 * k < c.W + LEAD ? rp[k] : 0u;
            uint32_t ex = 0;
#pragma unroll
            for (int x = 0; x < MAX_W; ++x) {
                const int b = x + LEAD;
                const int e = (d[b >> 2] >> (8 * (b & 3))) & 0xFF;
                bool ok = is_none(e) || open_door(e);
                if (second) { const int t = e_type(e); ok = ok || (t != T_WALL && t != T_DOOR); }
                ex |= (uint32_t)ok << x;
            }
            ex &= sv;                                              // seen cells only (vis has no bits beyond the grid)
            if (second) { w.row(R_EXP2, y) = ex; w.row(R_VIS2, y) = w.row(R_VIS1, y); }     // search 2 starts from all search 1 reached
            else { w.row(R_EXP1, y) = ex; w.row(R_VIS1, y) = 0; }
        }
        ctx.sync();
    }
    BB_HD void start1() const {
        if (qn1 >= 0) return;
        build_rows(false);
        const int start = pk(h.ax, h.ay);
        head1 = qn1 = 0;
        w.row(R_VIS1, h.ay) |= 1u << h.ax;
        w.at(WK_PREV1, pidx(start)) = 0xFFFE;
        if (w.ring_size) w.ring[0] = (uint16_t)(start | h.dir << 10);
        w.at(WK_Q1, qn1++) = (uint16_t)(start | h.dir << 10);
        ctx.sync();
    }
    BB_HD void start2() const {                                    // needs search 1 complete (it is: its query just failed)
        if (qn2 >= 0) return;
        build_rows(true);
        head2 = 0;
        for (int i = ctx.lane(); i < qn1; i += ctx.nlanes()) {     // every position search 1 reached, direction (1,0)
            const int p = w.at(WK_Q1, i) & 1023;
            w.at(WK_PREV2, pidx(p)) = 0xFFFE;
            w.at(WK_Q2, i) = (uint16_t)p;
        }
        qn2 = qn1;
        ctx.sync();
    }

    struct Path { bool found; bool nonempty; int len; int nx, ny; int fxp, fyp; bool with_blockers; };

    BB_HD Path shortest_path(const Accept& a, bool try_with_blockers) const {          // :772-806
        BOT_PROF(BP_PATH);
        BBAI_BOT_COUNT(2);
        Path p = {};
        start1();
        int len = 0, next = -1;
        int finish = search(WK_PREV1, WK_Q1, head1, qn1, a, false);
        if (finish >= 0) {
            for (int v = finish, u; (u = w.at(WK_PREV1, pidx(v))) != 0xFFFE; v = u) { ++len; next = v; }
        } else if (try_with_blockers) {
            p.with_blockers = true;
            start2();
            finish = search(WK_PREV2, WK_Q2, head2, qn2, a, true);
            if (finish >= 0) {
                int v = finish, u;
                for (; (u = w.at(WK_PREV2, pidx(v))) != 0xFFFE; v = u) { ++len; next = v; }
                int len1 = 0, next1 = -1;
                for (; (u = w.at(WK_PREV1, pidx(v))) != 0xFFFE; v = u) { ++len1; next1 = v; }
                len += len1;
                if (len1) next = next1;
            }
        }
        p.found = finish >= 0;
        if (p.found) {
            p.len = len; p.nonempty = len > 0;
            p.fxp = finish & 31; p.fyp = finish >> 5;
            if (len) { p.nx = next & 31; p.ny = next >> 5; }
        }
        return p;
    }
    BB_HD static Accept acc_pos(int x, int y) { Accept a = {}; a.kind = ACC_POS; a.x = x; a.y = y; return a; }

    BB_HD bool find_drop_pos(bool has_except, int ex, int ey, int& ox, int& oy) const { // :808-898
        BOT_PROF(BP_DROP_POS);
        Accept a = {};
        a.has_except = has_except; a.ex = ex; a.ey = ey;
        const int kinds[4] = {ACC_UNBLOCK, ACC_EMPTY, ACC_UNBLOCK, ACC_EMPTY};
        for (int k = 0; k < 4; ++k) {
            a.kind = kinds[k];
            Path p = shortest_path(a, k >= 2);
            if (p.found) { ox = p.fxp; oy = p.fyp; return true; }
        }
        return false;
    }

    BB_HD int closest_wall_or_door(int dx, int dy) const {                             // :697-708
        // positions h.a + d*(+-right_vec) stay inside the 7x7 view for d <= 3 (agent at view column 3)
        for (int d = 1; d <= 3; ++d) {
            const int t = e_type(cell(h.ax + d * dx, h.ay + d * dy));
            if (t == T_DOOR || t == T_WALL) return d;
        }
        return 3;
    }

    // ---- descriptors ------------------------------------------------------------------------------------------
    // ObjDesc('key', colour).find_matching_objs(env): x-major scan of the whole grid (verifier.py:96-161)
    BB_HD Subgoal go_keys(int color) {
        BOT_PROF(BP_KEYS);
        Subgoal g = mk(SG_GONEXT);
        g.dtype = DT_KEYS;
        const int cells = c.W * c.H;
        for (int base = 0; base < cells; base += ctx.nlanes()) {       // x-major, nlanes() cells at a time
            const int idx = base + ctx.lane();
            bool hit = false;
            if (idx < cells) {
                const int e = cell(idx / c.H, idx % c.H);
                hit = e_type(e) == T_KEY && e_color(e) == color;
            }
            for (unsigned long long m = ctx.ballot(hit); m; m &= m - 1) {
                const int j = base + __builtin_ctzll(m), x = j / c.H, y = j % c.H;
                if (g.nkeys >= BOT_KEYS) { die(DEAD_CAPACITY); return g; }
                g.keys[g.nkeys++] = (uint16_t)((I[i_index(c, x, y)] - 2) << 10 | x << 5 | y);
            }
        }
        return g;
    }

    // Bot._find_obj_pos: closest visible object of a descriptor.  obj_set keeps its episode-start order while
    // obj_poss is rebuilt (shorter) on every drop action, and the reference indexes both with the same i
    // (IndexError swallowed, :649-653) -- reproduced as is.
    BB_HD bool find_obj_pos(const Subgoal& g, bool adjacent, int& obj, int& ox, int& oy) {
        BOT_PROF(BP_FIND_OBJ);
        // ObjDesc(type=None, colour='grey') also matches every WALL cell (bonus_levels.py PickupDist; walls are grey):
        // the walls sit in obj_set / obj_poss in scan order with the objects, and the first visible wall the search
        // cannot reach (a corner) trips the reference's assertion.  Rare, so the merged lists are streamed.
        const bool walls = g.dtype == DT_DESC && prog->desc[g.a >> 1][g.a & 1].type == 0 &&
                           prog->desc[g.a >> 1][g.a & 1].color == C_GREY;
        if (walls) return find_obj_pos_with_walls(prog->set[g.a >> 1][g.a & 1], adjacent, obj, ox, oy);
        // One candidate of the reference's loop over zip(obj_set, obj_poss); `order` = its index in that loop.  The loop keeps the FIRST
        // minimum of d, i.e. the minimum of (d, order): candidates may be offered in any order.
        int best = 999, best_order = 0x7FFFFFFF;
        bool have = false;
        auto consider = [&](int so, int px, int py, int order) {
            if (so == h.carry) return;
            if (!seen(px, py)) return;
            Path p = shortest_path(acc_pos(px, py), true);
            if (!p.found) { die(); return; }                       // assert shortest_path_to_obj is not None
            int d = p.len;
            if (p.with_blockers) d = p.len + (carrying() ? 7 : 4);
            if (d == 0) d = adjacent ? 3 : 2;
            if (adjacent && d == 1) d = 3;
            if (d < best || (d == best && order < best_order)) { best = d; best_order = order; have = true; obj = so; ox = px; oy = py; }
        };
        if (g.dtype == DT_KEYS) {                                  // frozen lists, already in the reference's order
            if (g.nkeys == 0) { die(); return false; }             // assert len(obj_desc.obj_set) > 0
            for (int i = 0; i < g.nkeys && !raised; ++i) consider(g.keys[i] >> 10, (g.keys[i] >> 5) & 31, g.keys[i] & 31, i);
            return have && !raised;
        }
        const uint64_t set = prog->set[g.a >> 1][g.a & 1];
        if (!set) { die(); return false; }
        // Common case: every object of the descriptor is still recorded, at the position the episode started it on.  Then obj_set
        // (sorted by the start positions) and obj_poss (sorted by the recorded ones) pair every object with ITS position, and the
        // loop index of a pair is the rank of its position key: no list has to be built (the lists live in per-lane scratch, and their
        // insertion sorts were a third of a BossLevel decision: profiles/r05/NOTES.md section 11).
        // (BBAI_BOT_ALIGNED_OK: the host harness can switch the shortcut off -- tests/test_hostsim_bot.py replays every reference-bot fixture
        // through the packed lists alone and must see the same decisions)
        bool aligned = BBAI_BOT_ALIGNED_OK;
        for (uint64_t m = set; m && aligned; m &= m - 1) {
            const int o = __builtin_ctzll(m);
            aligned = (obj_in_grid(o) || (stale >> o & 1)) && pos[2 * o] == s.ipos[o][0] && pos[2 * o + 1] == s.ipos[o][1];
        }
        if (aligned) {
            for (uint64_t m = set; m && !raised; m &= m - 1) {
                const int o = __builtin_ctzll(m);
                consider(o, pos[2 * o], pos[2 * o + 1], pos[2 * o] << 8 | pos[2 * o + 1]);
            }
            return have && !raised;
        }
        // Otherwise the two lists, each ONE packed array (a shift of the insertion sort moves one element per list):
        // ks[i] = start-position key << 8 | object (obj_set order), kp[i] = recorded position x << 8 | y (obj_poss order)
        uint32_t ks[MAX_OBJ];
        uint16_t kp[MAX_OBJ];
        int n_set = 0, n_poss = 0;
        for (uint64_t m = set; m; m &= m - 1) {
            const int o = __builtin_ctzll(m);
            {   // obj_set: order of the scan at reset (the keys are distinct positions; the object id in the low byte never decides)
                const uint32_t v = (uint32_t)(s.ipos[o][0] << 8 | s.ipos[o][1]) << 8 | (uint32_t)o;
                int j = n_set++;
                for (; j > 0 && ks[j - 1] > v; --j) ks[j] = ks[j - 1];
                ks[j] = v;
            }
            if (obj_in_grid(o) || (stale >> o & 1)) {   // obj_poss: still recorded since the last refresh
                const uint16_t k = (uint16_t)(pos[2 * o] << 8 | pos[2 * o + 1]);
                int j = n_poss++;
                for (; j > 0 && kp[j - 1] > k; --j) kp[j] = kp[j - 1];
                kp[j] = k;
            }
        }
        for (int i = 0; i < n_set && i < n_poss && !raised; ++i)   // (i >= len(obj_poss): IndexError -> pass)
            consider((int)(ks[i] & 0xFFu), kp[i] >> 8, kp[i] & 0xFF, i);
        return have && !raised;
    }

    // next entry of obj_set (which = 0, episode-start positions) / obj_poss (which = 1, recorded positions) at or after
    // x-major scan index `cur`, walls included.  Returns false at the end; `o` = object or NONE8 for a wall.
    BB_HD bool next_with_walls(uint64_t set, int which, int& cur, int& o, int& x, int& y) const {
        for (; cur < c.W * c.H; ++cur) {
            x = cur / c.H; y = cur % c.H;
            if (e_type(cell(x, y)) == T_WALL) { o = NONE8; ++cur; return true; }
            for (uint64_t m = set; m; m &= m - 1) {
                const int k = __builtin_ctzll(m);
                const bool hit = which == 0 ? (s.ipos[k][0] == x && s.ipos[k][1] == y)
                                            : ((obj_in_grid(k) || (stale >> k & 1)) && pos[2 * k] == x && pos[2 * k + 1] == y);
                if (hit) { o = k; ++cur; return true; }
            }
        }
        return false;
    }
    BB_HD bool find_obj_pos_with_walls(uint64_t set, bool adjacent, int& obj, int& ox, int& oy) {
        int cs = 0, cp = 0, best = 999;
        bool have = false;
        for (;;) {
            int so, sx, sy, po, px, py;
            if (!next_with_walls(set, 0, cs, so, sx, sy)) break;
            const bool has_pos = next_with_walls(set, 1, cp, po, px, py);      // both cursors advance with i
            if (so != NONE8 && so == h.carry) continue;
            if (!has_pos) continue;                                            // IndexError -> pass
            if (!seen(px, py)) continue;
            Path p = shortest_path(acc_pos(px, py), true);
            if (!p.found) { die(); return false; }
            int d = p.len;
            if (p.with_blockers) d = p.len + (carrying() ? 7 : 4);
            if (d == 0) d = adjacent ? 3 : 2;
            if (adjacent && d == 1) d = 3;
            if (d < best) { best = d; have = true; obj = so; ox = px; oy = py; }
        }
        return have;
    }

    // ---- subgoals -----------------------------------------------------------------------------------------------
    BB_HD void plan_undo(int action) {                                                  // :109-137
        if (action == A_FORWARD) {
            if (prev_ax != h.ax || prev_ay != h.ay) push(go_pos(h.ax, h.ay));
        } else if (action == A_LEFT) {
            push(go_pos(h.ax + rx(), h.ay + ry()));
        } else if (action == A_RIGHT) {
            push(go_pos(h.ax - rx(), h.ay - ry()));
        } else if (action == A_DROP && prev_carry != h.carry) {
            const int t = e_type(cell(fx(), fy()));
            if (!(t == T_KEY || t == T_BOX || t == T_BALL)) { die(); return; }
            push(mk(SG_PICKUP));
        } else if (action == A_PICKUP && prev_carry != h.carry) {
            push(mk(SG_DROP));
        } else if (action == A_TOGGLE) {
            const int e = cell(fx(), fy());
            if (e_type(e) == T_DOOR) {
                if (door_was_open == 2) { die(); return; }          // AttributeError: fwd_door_was_open
                const int open = e_state(e) == S_OPEN;
                if (door_was_open != open) push(mk(open ? SG_CLOSE : SG_OPEN));
            }
        }
    }
    BB_HD static bool is_move(int a) { return a == A_FORWARD || a == A_LEFT || a == A_RIGHT; }

    BB_HD void after_action(const Subgoal g, int action) {      // replan_after_action; action < 0 = None
        BOT_PROF(BP_AFTER);
        const bool none = action < 0;
        switch (g.kind) {
        case SG_CLOSE:
            if (none || action == A_TOGGLE) pop();
            else if (is_move(action)) plan_undo(action);
            break;
        case SG_OPEN:
            if (none || action == A_TOGGLE) {
                pop();
                if (g.reason == RS_UNLOCK) {
                    int dx = 0, dy = 0;
                    const bool ok = find_drop_pos(false, 0, 0, dx, dy);
                    push(mk(SG_DROP));
                    push(go_maybe(ok, dx, dy));
                }
            } else plan_undo(action);
            break;
        case SG_DROP:
            if (none || action == A_DROP) pop();
            else if (is_move(action)) plan_undo(action);
            break;
        case SG_PICKUP:
            if (none || action == A_PICKUP) pop();
            else if (action == A_LEFT || action == A_RIGHT) plan_undo(action);
            break;
        case SG_GONEXT:
            if (action == A_PICKUP || action == A_DROP || action == A_TOGGLE) plan_undo(action);
            break;
        default: break;
        }
    }
    BB_HD static bool exploratory(const Subgoal& g) {
        return g.kind == SG_EXPLORE || (g.kind == SG_GONEXT && g.reason == RS_EXPLORE);
    }

    BB_HD bool got_key_for(int door_e) const {
        if (!carrying()) return false;
        const int ce = app[h.carry];
        return e_type(ce) == T_KEY && e_color(ce) == e_color(door_e);
    }

    // replan_before_action: returns an action, or -1 when the stack changed and replanning continues
    BB_HD int before_open(const Subgoal g) {                                           // :170-233
        const int fe = cell(fx(), fy());
        if (e_type(fe) != T_DOOR) { die(); return -1; }
        if (e_state(fe) == S_LOCKED && !got_key_for(fe)) {
            Subgoal keys = go_keys(e_color(fe));
            if (raised) return -1;
            if (carrying()) {
                pop();
                int dx = 0, dy = 0;
                const bool ok = find_drop_pos(false, 0, 0, dx, dy);
                push(mk(SG_PICKUP));
                push(go_maybe(ok, dx, dy));
                push(mk(SG_OPEN));
                push(go_pos(fx(), fy()));
                push(mk(SG_PICKUP));
                push(keys);
                push(mk(SG_DROP));
                push(go_maybe(ok, dx, dy));
            } else {
                pop();
                push(mk(SG_OPEN));
                push(go_pos(fx(), fy()));
                push(mk(SG_PICKUP));
                push(keys);
            }
            return -1;
        }
        if (e_state(fe) == S_OPEN) { push(mk(SG_CLOSE)); return -1; }
        if (e_state(fe) == S_LOCKED && g.reason == RS_NONE) { pop(); push(mk(SG_OPEN, RS_UNLOCK)); return -1; }
        return A_TOGGLE;
    }

    BB_HD int before_gonext(const Subgoal g) {                                         // :295-442
        int tobj = -1, tx = 0, ty = 0;
        if (g.dtype == DT_DESC || g.dtype == DT_KEYS) {
            if (!find_obj_pos(g, g.reason == RS_PUTNEXT, tobj, tx, ty)) {
                if (raised) return -1;
                push(mk(SG_EXPLORE));
                return -1;
            }
            if (tobj == NONE8) tobj = -1;                           // a wall "object" (grey type-less descriptor)
        } else if (g.dtype == DT_OBJ) {
            tobj = g.a; tx = pos[2 * tobj]; ty = pos[2 * tobj + 1];
        } else if (g.dtype == DT_POS) {
            tx = (int8_t)g.a; ty = (int8_t)g.b;
        } else { die(); return -1; }                                // tuple(None)

        if (g.reason == RS_OPEN && tobj >= 0) {
            const int te = app_now(tobj);
            if (e_type(te) == T_DOOR && e_state(te) == S_LOCKED) {
                if (!carrying()) {
                    Subgoal keys = go_keys(e_color(te));
                    if (raised) return -1;
                    pop();
                    push(go_obj(tobj, RS_OPEN));
                    push(mk(SG_PICKUP));
                    push(keys);
                    return -1;
                }
            }
        }
        const int px = h.ax, py = h.ay, fxx = fx(), fyy = fy();
        const int fe = cell(fxx, fyy);
        const int dist = (tx > px ? tx - px : px - tx) + (ty > py ? ty - py : py - ty);
        if (dist == (g.reason == RS_PUTNEXT ? 1 : 0)) {
            if (is_none(fe) || open_door(fe)) return A_FORWARD;
            int e = cell(px + rx(), py + ry());
            if (is_none(e) || open_door(e)) return A_RIGHT;
            e = cell(px - rx(), py - ry());
            if (is_none(e) || open_door(e)) return A_LEFT;
            return A_LEFT;
        }
        if (g.reason == RS_PUTNEXT) {
            const int dfw = (tx > fxx ? tx - fxx : fxx - tx) + (ty > fyy ? ty - fyy : fyy - ty);
            if (dfw == 1) {
                if (is_none(fe)) { pop(); return -1; }
                if (open_door(fe)) { push(go_pos(fxx + 2 * dir_dx(h.dir), fyy + 2 * dir_dy(h.dir))); return -1; }
            }
        } else if (tx == fxx && ty == fyy) { pop(); return -1; }

        Path p = shortest_path(acc_pos(tx, ty), false);
        if (!p.nonempty) p = shortest_path(acc_pos(tx, ty), true);
        if (!p.nonempty) { push(mk(SG_EXPLORE)); return -1; }

        if (p.nx == fxx && p.ny == fyy) {
            if (!is_none(fe)) {
                if (e_type(fe) == T_DOOR) {
                    if (e_state(fe) == S_LOCKED) { die(); return -1; }       // assert not is_locked
                    if (e_state(fe) != S_OPEN) { push(mk(SG_OPEN)); return -1; }
                    return A_FORWARD;
                }
                if (carrying()) {
                    int cx = 0, cy = 0, bx = 0, by = 0;
                    const bool okc = find_drop_pos(false, 0, 0, cx, cy);
                    // _find_drop_pos(drop_pos_cur): `except_pos and ...` -- None means no exclusion
                    const bool okb = find_drop_pos(okc, cx, cy, bx, by);
                    push(mk(SG_PICKUP));
                    push(go_maybe(okc, cx, cy));
                    push(mk(SG_DROP));
                    push(go_maybe(okb, bx, by));
                    push(mk(SG_PICKUP));
                    push(go_pos(fxx, fyy));
                    push(mk(SG_DROP));
                    push(go_maybe(okc, cx, cy));
                    return -1;
                }
                int dx = 0, dy = 0;
                const bool ok = find_drop_pos(false, 0, 0, dx, dy);
                push(mk(SG_DROP));
                push(go_maybe(ok, dx, dy));
                push(mk(SG_PICKUP));
                return -1;
            }
            return A_FORWARD;
        }
        if (p.nx - px == rx() && p.ny - py == ry()) return A_RIGHT;
        if (p.nx - px == -rx() && p.ny - py == -ry()) return A_LEFT;
        const int dr = closest_wall_or_door(rx(), ry()), dl = closest_wall_or_door(-rx(), -ry());
        return dl > dr ? A_LEFT : A_RIGHT;
    }
    // appearance of object o as it is NOW (a door's state lives in the E plane)
    BB_HD int app_now(int o) const {
        if (obj_in_grid(o)) return cell(pos[2 * o], pos[2 * o + 1]);
        return app[o];
    }

    BB_HD int before_explore() {                                                       // :453-497
        Accept a = {};
        a.kind = ACC_UNSEEN;
        Path p = shortest_path(a, true);
        if (p.found) { push(go_pos(p.fxp, p.fyp, RS_EXPLORE)); return -1; }
        a.kind = ACC_DOOR_UNLOCKED;
        p = shortest_path(a, true);
        if (!p.found) { a.kind = ACC_DOOR; p = shortest_path(a, true); }
        if (p.found) {
            const int de = cell(p.fxp, p.fyp);
            const int o = I[i_index(c, p.fxp, p.fyp)] - 2;
            const bool keep = e_state(de) == S_LOCKED && got_key_for(de);
            pop();
            push(mk(SG_OPEN, keep ? RS_KEEPKEY : RS_NONE));
            push(go_obj(o, RS_OPEN));
            return -1;
        }
        die();                                                       // assert False, "nothing left to explore"
        return -1;
    }

    BB_HD int before_action(const Subgoal g) {
        const int fe = cell(fx(), fy());
        switch (g.kind) {
        case SG_CLOSE:                                              // :141-145
            if (e_type(fe) != T_DOOR || e_state(fe) != S_OPEN) { die(); return -1; }
            return A_TOGGLE;
        case SG_OPEN: return before_open(g);
        case SG_DROP:                                               // :252-255
            if (!carrying() || !is_none(fe)) { die(); return -1; }
            return A_DROP;
        case SG_PICKUP:                                             // :266-268
            if (carrying()) { die(); return -1; }
            return A_PICKUP;
        case SG_GONEXT: return before_gonext(g);
        default: return before_explore();
        }
    }

    // ---- Bot ------------------------------------------------------------------------------------------------------
    BB_HD void process_instr_leaf(int leaf) {                                          // :900-924
        const int k = prog->kind[leaf];
        if (k == L_GOTO) {
            push(go_desc(2 * leaf));
        } else if (k == L_OPEN) {
            push(mk(SG_OPEN));
            push(go_desc(2 * leaf, RS_OPEN));
        } else if (k == L_PICKUP) {
            push(mk(SG_DROP));
            push(mk(SG_PICKUP));
            push(go_desc(2 * leaf));
        } else if (k == L_PUTNEXT) {
            push(mk(SG_DROP));
            push(go_desc(2 * leaf + 1, RS_PUTNEXT));
            push(mk(SG_PICKUP));
            push(go_desc(2 * leaf));
        } else die();
    }
    // a side is one ActionInstr or And(first, second): And pushes b then a (:926-929)
    BB_HD void process_side(int base, int n) {
        if (n == 2) process_instr_leaf(base + 1);
        process_instr_leaf(base);
    }
    BB_HD void init() {                                                                // Bot.__init__
        for (int y = 0; y < MAX_W; ++y) s.vis[y] = 0;
        for (int o = 0; o < c.maxo; ++o) { s.ipos[o][0] = pos[2 * o]; s.ipos[o][1] = pos[2 * o + 1]; }
        sp = 0; s.dead = DEAD_NO;
        prev_ax = prev_ay = 0; prev_carry = NONE8; door_was_open = 2; prev_fwd_type = 0;
        const int root = prog->root;
        if (root == R_ACTION || root == R_AND) process_side(0, prog->n_a);
        else if (root == R_BEFORE) { process_side(2, prog->n_b); process_side(0, prog->n_a); }      // b then a
        else { process_side(0, prog->n_a); process_side(2, prog->n_b); }                            // After: a then b
    }

    BB_HD void process_obs() {                                                         // :658-687
        BOT_PROF(BP_OBS);
        uint32_t opq[VIEW], vis[VIEW];
        unsigned long long opaque = 0;                             // bit vj * 7 + vi, the 49 view cells nlanes() at a time
        for (int base = 0; base < VIEW * VIEW; base += ctx.nlanes()) {
            const int idx = base + ctx.lane();
            bool o = false;
            if (idx < VIEW * VIEW) {
                int x, y; view_to_world(h.ax, h.ay, h.dir, idx % VIEW, idx / VIEW, x, y);
                o = e_opaque(cell(x, y));
            }
            opaque |= ctx.ballot(o) << base;
        }
        for (int vj = 0; vj < VIEW; ++vj) opq[vj] = (uint32_t)(opaque >> (VIEW * vj)) & 127u;
        process_vis_rows(opq, vis);
        // one world row per lane: the view axis that runs along world y is vi when the agent faces +-x, vj otherwise
        const bool vi_is_y = dir_dx(h.dir) != 0;
        for (int l = ctx.lane(); l < VIEW; l += ctx.nlanes()) {
            uint32_t mask = 0;
            int row = -1;
            for (int m = 0; m < VIEW; ++m) {
                const int vi = vi_is_y ? l : m, vj = vi_is_y ? m : l;
                if (!(vis[vj] >> vi & 1)) continue;
                int x, y; view_to_world(h.ax, h.ay, h.dir, vi, vj, x, y);
                if (in_grid(x, y)) { mask |= 1u << x; row = y; }
            }
            if (row >= 0) s.vis[row] |= mask;
        }
        ctx.sync();
    }

    // Bot.replan(action_taken).  action_taken < 0 = None.  Returns the suggested action or BOT_DEAD.
    BB_HD int replan(int action_taken) {
        if (s.dead) return BOT_DEAD;
        process_obs();
        if (w.eager && sp) {                // same tree, same pop order: only WHEN it is expanded changes
            BOT_PROF(BP_INIT);
            start1();
            while (head1 < qn1) { const int st = q1_get(head1, qn1); ++head1; expand(WK_PREV1, WK_Q1, qn1, st, false); }
        }
        if (action_taken == A_TOGGLE && prev_fwd_type == T_BOX) { die(); return BOT_DEAD; }     // DisappearedBoxError
        // (top(): every lane has read the subgoal before any lane's pushes overwrite its slot)
        if (sp) after_action(top(), action_taken);
        if (raised) return BOT_DEAD;
        while (sp && exploratory(stk[sp - 1])) pop();
        int suggested = -1;
        int iters = 0;
        while (sp) {
            { BOT_PROF(BP_BEFORE); suggested = before_action(top()); }
            if (raised) return BOT_DEAD;
            if (suggested >= 0) break;
            if (++iters > BOT_MAX_ITERS) { die(); return BOT_DEAD; }
        }
        if (!sp) suggested = A_DONE;
        // _remember_current_state (:689-695)
        prev_ax = h.ax; prev_ay = h.ay; prev_carry = h.carry;
        const int fe = cell(fx(), fy());
        if (e_type(fe) == T_DOOR) door_was_open = e_state(fe) == S_OPEN;
        prev_fwd_type = is_none(fe) ? 0 : (uint8_t)e_type(fe);
        return suggested;
    }
    BB_HD Subgoal top() const { const Subgoal g = stk[sp - 1]; ctx.sync(); return g; }
    // what the next decision reads back (every lane stores the same values)
    BB_HD void remember() const {
        s.sp = (uint16_t)sp;
        s.prev_ax = prev_ax; s.prev_ay = prev_ay; s.prev_carry = prev_carry; s.door_was_open = door_was_open; s.prev_fwd_type = prev_fwd_type;
    }
};

// One decision for one env.  A fresh Bot (`Bot(env)`, action_taken = None) is started when the caller says so, when
// the episode has just begun (step_count == 0), and when the expert was not consulted on the previous step of this
// episode (so it can be switched on in the middle of an episode, like constructing `Bot(env)` there).  A Bot started
// mid-episode orders each descriptor's obj_set by the objects' positions at that moment; the reference uses their
// positions at reset, which differ only if a described object was carried somewhere else before.
// `s` and `stk` (the subgoal stack, `stack_cap` entries) may live in different memories (k_botg stages `s` in LDS); a lane group
// passes its Ctx and shares s / stk / w.
template <class Ctx>
BB_HD int bot_decide(Ctx ctx, const LevelCfg& c, const uint8_t* rec, const Hot& h, uint64_t stale, BotState& s, Subgoal* stk, int stack_cap,
                     const BotWork& w, bool first, int action_taken) {
    BOT_PROF(BP_DECIDE);
    first = first || h.step == 0 || s.next_step != h.step;
    ctx.sync();                                       // (every lane has read next_step)
    Bot<Ctx> b(ctx, c, rec, h, stale, s, stk, stack_cap, w);
    s.next_step = (uint16_t)(h.step + 1);
    if (first) { b.init(); action_taken = -1; }
    const int a = b.raised ? BOT_DEAD : b.replan(action_taken);
    ctx.sync();                                       // (no lane still reads what remember() rewrites)
    b.remember();
    return a;
}
BB_HD int bot_decide(const LevelCfg& c, const uint8_t* rec, const Hot& h, uint64_t stale, BotState& s, int stack_cap, const BotWork& w,
                     bool first, int action_taken) {
    return bot_decide(OneLane(), c, rec, h, stale, s, (Subgoal*)(&s + 1), stack_cap, w, first, action_taken);
}

}  // namespace bbai
