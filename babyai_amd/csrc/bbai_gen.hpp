// bbai_gen.hpp -- level generator: RoomGrid layout + LevelGen / GoTo-family missions,
// consuming the env's persistent numpy-RandomState (MT19937) stream draw-for-draw.
//
// Follows (reference file:line, /root/reference):
//   RoomGridLevel._gen_grid rejection loop        babyai/levels/levelgen.py:77-102
//   validate_instrs                               babyai/levels/levelgen.py:104-155
//   check_objs_reachable                          babyai/levels/levelgen.py:201-253
//   LevelGen.gen_mission / add_locked_room        babyai/levels/levelgen.py:293-352
//   LevelGen.rand_obj / rand_instr                babyai/levels/levelgen.py:354-460
//   ObjDesc.find_matching_objs (use_location)     babyai/levels/verifier.py:96-161
//   GoToRedBall / GoToObj / GoToLocal / GoTo      babyai/levels/iclr19_levels.py:40-124,224-257
//   RoomGrid / MiniGridEnv placement helpers      gym_minigrid (absent dependency), restated
//                                                 per SURVEY.md Appendix B3-B7
//
// Execution model: ONE LANE GROUP = ONE ENV, Ctx::kLanes lanes wide (device: 32 -> two envs per wavefront by default, 16 / 64 selectable, k_pregen in
// bbai_engine.hip; host: 1).  Inside a group control flow is uniform (every lane runs the same scalar program on the
// same RNG draws); the working set (MT state, both grid planes, room / object tables) sits in LDS, one GenWork per
// group; the group's lanes split the data-parallel parts (MT twist, grid fill, reachability rows, record write-out).
// Groups of one wave diverge from each other like any SIMT lanes do (different draws, different rejection counts): what
// was a wave-uniform SCALAR program when a whole wave served one env (round 2: 16 k of 22 k instructions per level on the
// CU's one scalar unit) is VECTOR work shared by the wave's groups now.  Ctx::sync() therefore orders LDS accesses
// within the wave (it is reached under divergent control flow: never a workgroup barrier); shuffles and ballots are
// group-relative.  The same code compiles for the host with kLanes == 1 (unit tests only, never a product path).
#pragma once
#include "bbai_types.hpp"

namespace bbai {

constexpr int GEN_ES = 36;                       // >= round_up(MAX_W + 2*MARGIN, 4)
constexpr int GEN_EH = MAX_W + 2 * MARGIN;       // 35

struct GenWork {                                 // lives in LDS on the device; the env's MT19937 state sits next to it (k_pregen: s_mt) and reaches
                                                 // Gen as a POINTER ARGUMENT, never through a field: a pointer loaded from memory is a generic
                                                 // pointer, and every draw became a flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) instead of a ds_read
    uint8_t E[GEN_ES * GEN_EH];
    uint8_t I[MAX_W * MAX_W + 3];
    uint8_t app[MAX_OBJ], px[MAX_OBJ], py[MAX_OBJ];
    uint8_t cont[MAX_OBJ];                       // box contents (object id revealed on toggle) or NONE8; px == NONE8: hidden
    uint8_t door_x[MAX_ROOMS][4], door_y[MAX_ROOMS][4];
    uint32_t rowbuf[2][MAX_W + 2];               // reachability rows
    uint64_t masks[14];                          // object-id sets: [0..3] by type (door,key,ball,box),
                                                 // [4..9] by colour, [10..13] by location (left,right,front,behind)
    Prog prog;
};

// Ctx contract: static constexpr int kLanes; lane() in [0, kLanes), nlanes() == kLanes, sync() (LDS accesses of the group
// before / after it are ordered), shfl_up1 / shfl_down1(v) (neighbour lane's value inside the group, 0 at the ends),
// shfl(v, src) (lane src of the group), any(pred) over the group.  Host: one lane, no-ops.
// Optional phase profiling (tools/genprof.hip): a Ctx that defines `static constexpr bool kProfile = true` and
// `now()` gets per-phase cycle totals in prof[]; the product contexts leave it off and the hooks vanish.
template <class C, class = void> struct ctx_profiles { static constexpr bool value = false; };
template <class C> struct ctx_profiles<C, decltype((void)C::kProfile)> { static constexpr bool value = C::kProfile; };
enum : int { PH_BUILD = 0, PH_LOCK = 1, PH_CONNECT = 2, PH_DISTS = 3, PH_AGENT = 4, PH_REACH = 5, PH_INSTR = 6, PH_VALIDATE = 7,
             PH_ATTEMPTS = 8, PH_DRAWS = 9, PH_TWISTS = 10, PH_N = 11 };

// MT19937 state transition (numpy legacy RandomState bit stream), the lanes of the context splitting each chunk.  Needed
// once per 624 draws but reachable from every draw: NOT inlined -- as part of next_u32() it was copied to ~50 call sites of
// the level generators (~2 KB each; k_pregen<LevelGen> 208 KB of code against a 64 KB instruction cache).
BB_HD uint32_t mt_mix(uint32_t u, uint32_t v) {
    uint32_t y = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (y >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
template <class Ctx>
BB_HD void mt_twist_chunk(Ctx ctx, uint32_t* mt, int lo, int hi) {
    // new[k] = src[k+397 mod] ^ mix(old[k], old[k+1]) for k in [lo,hi); reads complete
    // before any write of the chunk (sync), so lanes never see half-updated inputs.
    // (chunks are <= 227 long: Q strided elements per lane cover them)
    constexpr int Q = (227 + Ctx::kLanes - 1) / Ctx::kLanes;
    uint32_t tmp[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        int k = lo + ctx.lane() + q * ctx.nlanes();
        if (k < hi) {
            int m = k + 397; if (m >= MT_N) m -= MT_N;
            tmp[q] = mt[m] ^ mt_mix(mt[k], mt[k + 1 < MT_N ? k + 1 : 0]);
        }
    }
    ctx.sync();
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        int k = lo + ctx.lane() + q * ctx.nlanes();
        if (k < hi) mt[k] = tmp[q];
    }
    ctx.sync();
}
template <class Ctx>
BB_COLD void mt_twist(Ctx ctx, uint32_t* mt) {
    if constexpr (Ctx::kLanes == 1) {  // host, or lane = level on the device: plain sequential generation
        for (int k = 0; k < MT_N; ++k) {
            int m = k + 397; if (m >= MT_N) m -= MT_N;
            mt[k] = mt[m] ^ mt_mix(mt[k], mt[k + 1 < MT_N ? k + 1 : 0]);
        }
    } else {
        ctx.sync();
        mt_twist_chunk(ctx, mt, 0, 227);      // uses old[k+397]
        mt_twist_chunk(ctx, mt, 227, 454);    // uses new[k-227] from the first chunk
        mt_twist_chunk(ctx, mt, 454, 623);    // uses new[k-227] from the second chunk
        mt_twist_chunk(ctx, mt, 623, 624);    // uses new[396] and new[0]
    }
}

// MT19937 output tempering.  The generator's inner loops are chains of draw -> mask -> compare -> redraw, and a draw used to be ten
// vector instructions of tempering executed by every lane of the group for one value.  Now the group's lanes temper MT_CH consecutive
// state words at once into a small buffer (`tw`, next to the state in LDS) whenever the output index crosses a multiple of MT_CH
// (624 = 39 x 16, so the twist boundary is one of them): a draw is a read of that buffer.
constexpr int MT_CH = 16;
BB_HD uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

template <class Ctx>
struct Gen {
    Ctx ctx;
    unsigned long long prof[PH_N] = {};
    unsigned long long t_last = 0;
    BB_HD void tick(int phase) {
        if constexpr (ctx_profiles<Ctx>::value) {
            unsigned long long t = ctx.now();
            prof[phase] += t - t_last;
            t_last = t;
        }
    }
    BB_HD void count(int what) {
        if constexpr (ctx_profiles<Ctx>::value) prof[what]++;
    }
    const LevelCfg& cfg;
    GenWork& w;
    uint32_t* const mt;      // the env's MT19937 state (device: LDS; host: the caller's array, advanced in place)
    uint32_t* const tw;      // MT_CH tempered outputs: words [mti & ~(MT_CH - 1), + MT_CH) of the state (device: LDS, behind the state)
    int mti;                 // MT19937 output index (wave-uniform)
    uint32_t nxt;            // the output for index mti, fetched one draw AHEAD: the generator is a chain of draw -> test -> branch -> draw,
                             // and every link used to start with an LDS round trip
    bool twisted;            // the state words were regenerated at least once (k_pregen: only then does the state go back to memory)
    uint64_t occ;            // grids of <= 64 cells (every single room up to 8 x 8): bit y * W + x = the cell holds a wall or an object --
                             // the placement loops test a register bit instead of waiting for an id-plane byte from LDS
    uint32_t seen;           // bit (type - T_KEY) * 6 + colour: a key / ball / box of that look exists (add_distractors(all_unique))
    bool small;              // cfg.W * cfg.H <= 64
    int nobj;
    int ax, ay, adir;
    bool agent_set;
    int locked_room;         // room index or -1: the locked room created in THIS attempt
    int last_locked;         // room index or -1: `self.locked_room` as the reference keeps it -- it is
                             // only ever assigned in add_locked_room, so it survives attempts and
                             // episodes; a stale value never `is` a current room (levelgen.py:305-307)
                             // but still feeds rand_obj's implicit_unlock filter (levelgen.py:384-392)
    int S, rows, cols;
    bool gave_up;            // generate() hit the last-resort attempt cap
    uint64_t doors;          // bit (16*k + r): room r has a door on side k (0 right, 1 down, 2 left, 3 up)
    uint32_t locked_mask;    // bit r: room r is behind a locked door (Room.locked)
    uint32_t inv_cols, inv_s1, inv_es;   // 2^16/d + 1: exact small-range division without the divider

    BB_HD Gen(Ctx c, const LevelCfg& cf, GenWork& wk, uint32_t* mt_, uint32_t* tw_, int mti_, int last_locked_)
        : ctx(c), cfg(cf), w(wk), mt(mt_), tw(tw_), mti(mti_), nxt(0), twisted(false), occ(0), seen(0), small(cf.W * cf.H <= 64),
          nobj(0), ax(0), ay(0), adir(0), agent_set(false),
          locked_room(-1), last_locked(last_locked_), S(cf.room_size), rows(cf.num_rows), cols(cf.num_cols),
          gave_up(false), doors(0), locked_mask(0), inv_cols(65536u / (uint32_t)cf.num_cols + 1u),
          inv_s1(65536u / (uint32_t)(cf.room_size - 1) + 1u), inv_es(65536u / (uint32_t)cf.ES + 1u) {
        // in the middle of a chunk (the previous level stopped there): temper it again; on a boundary the first draw does
        if (mti & (MT_CH - 1)) { fill_chunk(mti & ~(MT_CH - 1)); nxt = tw[mti & (MT_CH - 1)]; }
    }

    BB_HD int div_cols(int v) const { return (int)(((uint32_t)v * inv_cols) >> 16); }     // v < 64
    BB_HD int div_s1(int v) const { return (int)(((uint32_t)v * inv_s1) >> 16); }         // v < 64
    BB_HD int div_es(int v) const { return (int)(((uint32_t)v * inv_es) >> 16); }         // v < 2048

    // ---------------- MT19937 (numpy legacy RandomState bit stream) ----------------
    BB_HD void twist() { mt_twist(ctx, mt); }
    // tw[0 .. MT_CH) = tempered mt[base .. base + MT_CH), base a multiple of MT_CH below MT_N: one word per lane
    BB_HD void fill_chunk(int base) {
        ctx.sync();
        if constexpr (Ctx::kLanes == 1) {
            for (int k = 0; k < MT_CH; ++k) tw[k] = mt_temper(mt[base + k]);
        } else {
            static_assert(Ctx::kLanes == 1 || Ctx::kLanes >= MT_CH, "one tempered word per lane");
            const int l = ctx.lane();
            if (l < MT_CH) tw[l] = mt_temper(mt[base + l]);
        }
        ctx.sync();
    }
    BB_HD uint32_t next_u32() {
        if ((mti & (MT_CH - 1)) == 0) {          // every MT_CH draws.  (Inline on purpose: a non-inlined MEMBER would pin the whole Gen object to
                                                 // scratch memory -- 464 bytes per lane when it was tried; only the twist itself is a call)
            if (mti >= MT_N) { twist(); mti = 0; twisted = true; count(PH_TWISTS); }
            fill_chunk(mti);
            nxt = tw[0];
        }
        count(PH_DRAWS);
        const uint32_t y = nxt;
        ++mti;
        nxt = tw[mti & (MT_CH - 1)];            // (for the next draw; on a chunk boundary refill() replaces it)
        return y;
    }
    // RandomState.randint(lo, hi): masked rejection on 32-bit outputs; range 1 => no draw.
    BB_HD int rand_int(int lo, int hi) {
        uint32_t rng = (uint32_t)(hi - lo - 1);
        if (rng == 0) return lo;
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(rng);       // smallest 2^k - 1 >= rng (numpy's mask), rng >= 1
        uint32_t v;
        do { v = next_u32() & mask; } while (v > rng);
        return lo + (int)v;
    }
    BB_HD bool rand_bool() { return rand_int(0, 2) == 0; }
    BB_HD double rand_float01() {   // RandomState.uniform(0,1): 53-bit double from two draws
        uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    BB_HD int rand_color() { return color_name_to_idx(rand_int(0, 6)); }

    // ---------------- grid helpers ----------------
    BB_HD int eidx(int x, int y) const { return (y + MARGIN) * cfg.ES + (x + MARGIN); }
    BB_HD int iidx(int x, int y) const { return y * cfg.W + x; }
    BB_HD void set_cell(int x, int y, int e, int id) {
        w.E[eidx(x, y)] = (uint8_t)e; w.I[iidx(x, y)] = (uint8_t)id;
        if (small) { const uint64_t b = 1ull << (y * cfg.W + x); occ = id ? (occ | b) : (occ & ~b); }
    }
    // the cells build_rooms makes walls, as a bitboard (grids of <= 64 cells): whole rows on the room boundaries, the boundary columns elsewhere
    BB_HD uint64_t wall_board() const {
        uint64_t in_row = 0, b = 0;
        for (int x = 0; x < cfg.W; x += S - 1) in_row |= 1ull << x;
        const uint64_t full_row = (1ull << cfg.W) - 1ull;
        for (int y = 0; y < cfg.H; ++y) b |= ((y - div_s1(y) * (S - 1)) == 0 ? full_row : in_row) << (y * cfg.W);
        return b;
    }
    BB_HD bool occupied(int x, int y) const { return small ? (occ >> (y * cfg.W + x) & 1ull) != 0 : w.I[iidx(x, y)] != 0; }
    BB_HD void note_obj(int e) { if (e_type(e) >= T_KEY) seen |= 1u << ((e_type(e) - T_KEY) * 6 + e_color(e)); }
    BB_HD int room_of(int x, int y) const { return div_s1(y) * cols + div_s1(x); }
    BB_HD void room_ij(int r, int& i, int& j) const { j = div_cols(r); i = r - j * cols; }
    BB_HD bool has_neighbor(int r, int k) const {
        int i, j; room_ij(r, i, j);
        return k == 0 ? i < cols - 1 : k == 1 ? j < rows - 1 : k == 2 ? i > 0 : j > 0;
    }
    BB_HD bool has_door(int r, int k) const { return doors >> (16 * k + r) & 1; }
    BB_HD int neighbor(int r, int k) const { return k == 0 ? r + 1 : k == 1 ? r + cols : k == 2 ? r - 1 : r - cols; }
    BB_HD void front_of(int& fx, int& fy) const {
        fx = ax + (adir == 0) - (adir == 2);
        fy = ay + (adir == 1) - (adir == 3);
    }

    // RoomGrid._gen_grid: walls on every multiple of (S-1), door slots drawn per room.
    BB_HD void build_rooms() {
        ctx.sync();
        const int ncell = cfg.ES * cfg.EH;
        for (int idx = ctx.lane(); idx < ncell; idx += ctx.nlanes()) {
            int row = div_es(idx);
            int x = idx - row * cfg.ES - MARGIN, y = row - MARGIN;
            bool inside = x >= 0 && x < cfg.W && y >= 0 && y < cfg.H;
            bool wall = true;
            if (inside) wall = x - div_s1(x) * (S - 1) == 0 || y - div_s1(y) * (S - 1) == 0;
            w.E[idx] = wall ? E_WALL : E_EMPTY;
            if (inside) w.I[iidx(x, y)] = wall ? 1 : 0;
        }
        doors = 0;
        locked_mask = 0;
        seen = 0;
        occ = small ? wall_board() : 0ull;
        ctx.sync();
        for (int j = 0; j < rows; ++j)
            for (int i = 0; i < cols; ++i) {
                int r = j * cols + i;
                int tx = i * (S - 1), ty = j * (S - 1);
                int x_l = tx + 1, y_l = ty + 1, x_m = tx + S - 1, y_m = ty + S - 1;
                if (i < cols - 1) {
                    int y = rand_int(y_l, y_m);
                    w.door_x[r][0] = x_m; w.door_y[r][0] = y;
                    w.door_x[r + 1][2] = x_m; w.door_y[r + 1][2] = y;
                }
                if (j < rows - 1) {
                    int x = rand_int(x_l, x_m);
                    w.door_x[r][1] = x; w.door_y[r][1] = y_m;
                    w.door_x[r + cols][3] = x; w.door_y[r + cols][3] = y_m;
                }
            }
        ax = (cols / 2) * (S - 1) + S / 2;
        ay = (rows / 2) * (S - 1) + S / 2;
        adir = 0;
        agent_set = true;
        nobj = 0;
        locked_room = -1;
        ctx.sync();
    }

    // MiniGridEnv.place_obj restricted to a room rectangle, max_tries = 1000.
    BB_HD bool place_pos(int r, bool reject_next, int& ox, int& oy) {
        int ri, rj; room_ij(r, ri, rj);
        int tx = ri * (S - 1), ty = rj * (S - 1);
        int xh = tx + S < cfg.W ? tx + S : cfg.W, yh = ty + S < cfg.H ? ty + S : cfg.H;
        int tries = 0;
        for (;;) {
            if (tries > 1000) return false;          // RecursionError('rejection sampling failed')
            ++tries;
            int x = rand_int(tx, xh);
            int y = rand_int(ty, yh);
            if (occupied(x, y)) continue;
            if (agent_set && x == ax && y == ay) continue;
            if (reject_next) {
                int d = (x > ax ? x - ax : ax - x) + (y > ay ? y - ay : ay - y);
                if (d < 2) continue;
            }
            ox = x; oy = y;
            return true;
        }
    }
    // RoomGrid.add_object -> place_in_room (kind and colour always given on this path).
    BB_HD int add_object(int r, int type, int color) {
        int x, y;
        if (!place_pos(r, true, x, y)) return -1;
        if (nobj >= cfg.maxo) return -1;
        int id = nobj++;
        int e = e_make(type, color, 0);
        w.app[id] = e; w.px[id] = x; w.py[id] = y; w.cont[id] = NONE8;
        set_cell(x, y, e, id + 2);
        note_obj(e);
        return id;
    }
    // RoomGrid.add_door with explicit index / colour / locked flag.
    BB_HD int add_door(int r, int k, int color, bool is_locked) {
        if (nobj >= cfg.maxo) return -1;
        int id = nobj++;
        int x = w.door_x[r][k], y = w.door_y[r][k];
        int e = e_make(T_DOOR, color, is_locked ? S_LOCKED : S_CLOSED);
        locked_mask = (locked_mask & ~(1u << r)) | ((is_locked ? 1u : 0u) << r);   // room.locked = locked
        w.app[id] = e; w.px[id] = x; w.py[id] = y; w.cont[id] = NONE8;
        set_cell(x, y, e, id + 2);
        doors |= 1ull << (16 * k + r);
        doors |= 1ull << (16 * ((k + 2) & 3) + neighbor(r, k));
        return id;
    }
    // RoomGrid.place_agent(i=None, j=None): room drawn, then pose re-drawn until the
    // front cell is empty or a wall.
    BB_HD bool place_agent(int room = -1) {
        int r = room;
        if (r < 0) {
            int i = rand_int(0, cols);
            int j = rand_int(0, rows);
            r = j * cols + i;
        }
        // TERMINATION GUARD (deliberate deviation, shared with the oracle shim): the reference re-draws the pose for
        // ever; in a crowded room whose free cells all face objects or doors no pose is ever accepted and the
        // reference never returns (e.g. MiniBossLevel, seed 100758, 2nd level).  After 1000 rejected poses the
        // attempt is abandoned like the other RecursionError cases and the level is re-generated.
        for (int pose_tries = 0; pose_tries <= 1000; ++pose_tries) {
            agent_set = false;
            int x, y;
            if (!place_pos(r, false, x, y)) return false;
            ax = x; ay = y; agent_set = true;
            adir = rand_int(0, 4);
            int fx, fy; front_of(fx, fy);
            int id = w.I[iidx(fx, fy)];
            if (id == 0 || id == 1) return true;
        }
        return false;
    }
    // RoomGrid.connect_all: random doors until every room is reachable from the agent's.
    // RoomGrid.add_door(i, j, door_idx=None, color=None, locked=...): side drawn until it has a neighbour and
    // no door yet, then the colour.
    BB_HD int add_door_random_side(int r, bool is_locked, int& color_out) {
        int k;
        for (;;) {
            k = rand_int(0, 4);
            if (has_neighbor(r, k) && !has_door(r, k)) break;
        }
        color_out = rand_color();
        return add_door(r, k, color_out, is_locked);
    }
    // position of a colour in COLOR_NAMES (sorted): blue green grey purple red yellow
    BB_HD static int color_name_pos(int c) { return (0x253014 >> (4 * c)) & 0xF; }   // red4 green1 blue0 purple3 yellow5 grey2

    BB_HD bool connect_all(int excl_color = -1) {
        int start = room_of(ax, ay);
        int nrooms = rows * cols;
        int itrs = 0;
        uint32_t reach = 1u << start;
        bool grew = true;                             // the closure only changes when a door was added: most trips of the loop draw a
        for (;;) {                                    // triple that is rejected, and keep the closure they came with
            if (itrs > 5000) return false;            // RecursionError('connect_all failed')
            ++itrs;
            if (grew) {
                // rooms reachable from the start room through existing doors: bit-parallel closure (it only ever grows)
                const uint32_t d0 = (uint32_t)doors & 0x1FFu, d1 = (uint32_t)(doors >> 16) & 0x1FFu;
                const uint32_t d2 = (uint32_t)(doors >> 32) & 0x1FFu, d3 = (uint32_t)(doors >> 48) & 0x1FFu;
                for (int pass = 0; pass < nrooms; ++pass) {
                    uint32_t nr = reach | ((reach & d0) << 1) | ((reach & d1) << cols) | ((reach & d2) >> 1) | ((reach & d3) >> cols);
                    if (nr == reach) break;
                    reach = nr;
                }
                grew = false;
            }
            if (__builtin_popcount(reach) == nrooms) return true;
            int i = rand_int(0, cols);
            int j = rand_int(0, rows);
            int k = rand_int(0, 4);
            int r = j * cols + i;
            if (!has_neighbor(r, k) || has_door(r, k)) continue;
            if ((locked_mask >> r & 1) || (locked_mask >> neighbor(r, k) & 1)) continue;
            int color;
            if (excl_color < 0) {
                color = rand_color();
            } else {            // door_colors = COLOR_NAMES without the locked door's colour (iclr19_levels.py:443-445)
                int q = rand_int(0, 5);
                if (q >= color_name_pos(excl_color)) ++q;
                color = color_name_to_idx(q);
            }
            if (add_door(r, k, color, false) < 0) return false;
            grew = true;
        }
    }
    // RoomGrid.add_distractors(i=None, j=None).  first_id receives the first new id.
    BB_HD bool add_distractors(int num, bool all_unique, int room = -1) {
        int count = 0;
        while (count < num) {
            int color = rand_color();
            int type = T_KEY + rand_int(0, 3);        // ['key','ball','box']
            if (all_unique && (seen >> ((type - T_KEY) * 6 + color) & 1u)) continue;     // (type, colour) of an existing key / ball / box
            int r = room;
            if (r < 0) {
                int ri = rand_int(0, cols);
                int rj = rand_int(0, rows);
                r = rj * cols + ri;
            }
            if (add_object(r, type, color) < 0) return false;
            ++count;
        }
        return true;
    }

    // check_objs_reachable (levelgen.py:201-253) as a row-bitmask flood fill: spread through
    // passable cells (empty or door); an object is reached if it lies in, or 4-adjacent to,
    // the flooded region.
    // Grids of <= 64 cells: the same flood fill on ONE 64-bit board in registers (no LDS rows, no shuffles, no ballots): passable =
    // free cells + doors, the flood spreads by four masked shifts, an object is reached when its cell lies in the flood's 4-neighbourhood.
    BB_HD bool objs_reachable_small() {
        const int W = cfg.W, H = cfg.H;
        uint64_t col0 = 0;
        for (int y = 0; y < H; ++y) col0 |= 1ull << (y * W);
        const uint64_t board = W * H == 64 ? ~0ull : (1ull << (W * H)) - 1ull;
        const uint64_t not0 = ~col0, notL = ~(col0 << (W - 1));
        uint64_t objs = occ & ~wall_board();             // (recomputed, not kept: two registers less in every placement loop; an object hidden inside a box is not in occ)
        uint64_t pass = ~occ & board;
        if (doors)                                       // doors stand where walls were: passable, and objects that must be reached (single rooms have none)
            for (int o = 0; o < nobj; ++o)
                if (e_type(w.app[o]) == T_DOOR) { const uint64_t b = 1ull << (w.py[o] * W + w.px[o]); pass |= b; objs |= b; }
        uint64_t f = 1ull << (ay * W + ax);
        for (;;) {
            const uint64_t g = (f | ((f << 1) & not0) | ((f >> 1) & notL) | (f << W) | (f >> W)) & pass;
            if ((g | f) == f) break;
            f |= g;
        }
        const uint64_t near = f | ((f << 1) & not0) | ((f >> 1) & notL) | (f << W) | (f >> W);
        return (objs & ~near) == 0;
    }
    BB_HD bool objs_reachable() {
        if (small) return objs_reachable_small();
        const int W = cfg.W, H = cfg.H;
        uint32_t* pass = w.rowbuf[0];
        uint32_t* fl = w.rowbuf[1];
        ctx.sync();
        if constexpr (Ctx::kLanes > 1) {
            // device: lane l owns grid rows l, l + kLanes, ... (K rows per lane); rows exchange their flood masks with
            // group shuffles, the row above the first row of pass k being the last lane's row of pass k - 1
            constexpr int NL = Ctx::kLanes;
            constexpr int K = (MAX_W + NL - 1) / NL;
            const int l = ctx.lane();
            uint32_t p[K], f[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int y = k * NL + l;
                p[k] = 0;
                if (y < H)
                    for (int x = 0; x < W; ++x) {
                        int e = w.E[eidx(x, y)];
                        if (e == E_EMPTY || e_type(e) == T_DOOR) p[k] |= 1u << x;
                    }
                f[k] = (y == ay) ? (1u << ax) : 0u;
            }
            for (;;) {
                bool changed = false;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    uint32_t up = ctx.shfl_up1(f[k]), down = ctx.shfl_down1(f[k]);
                    if (K > 1) {
                        const uint32_t wrap_up = k > 0 ? ctx.shfl(f[k > 0 ? k - 1 : 0], NL - 1) : 0u;      // row k*NL - 1
                        const uint32_t wrap_dn = k + 1 < K ? ctx.shfl(f[k + 1 < K ? k + 1 : k], 0) : 0u;   // row (k+1)*NL
                        if (l == 0) up = wrap_up;
                        if (l == NL - 1) down = wrap_dn;
                    }
                    uint32_t g = (f[k] | up | down) & p[k];
                    for (;;) {                      // horizontal closure inside the row
                        uint32_t g2 = (g | (g << 1) | (g >> 1)) & p[k];
                        if (g2 == g) break;
                        g = g2;
                    }
                    changed |= g != f[k];
                    f[k] = g;
                }
                if (!ctx.any(changed)) break;
            }
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k * NL + l < H) fl[k * NL + l] = f[k];
            ctx.sync();
        } else {
            for (int y = 0; y < H; ++y) {
                uint32_t p = 0;
                for (int x = 0; x < W; ++x) {
                    int e = w.E[eidx(x, y)];
                    if (e == E_EMPTY || e_type(e) == T_DOOR) p |= 1u << x;
                }
                pass[y] = p;
                fl[y] = (y == ay) ? (1u << ax) : 0u;
            }
            for (;;) {
                bool changed = false;
                for (int y = 0; y < H; ++y) {
                    uint32_t f = fl[y];
                    uint32_t g = f | (y > 0 ? fl[y - 1] : 0u) | (y + 1 < H ? fl[y + 1] : 0u);
                    g &= pass[y];
                    for (;;) {
                        uint32_t g2 = (g | (g << 1) | (g >> 1)) & pass[y];
                        if (g2 == g) break;
                        g = g2;
                    }
                    if (g != f) { fl[y] = g; changed = true; }
                }
                if (!changed) break;
            }
        }
        for (int o = 0; o < nobj; ++o) {
            int x = w.px[o], y = w.py[o];
            uint32_t near = fl[y] | (fl[y] << 1) | (fl[y] >> 1) | (y > 0 ? fl[y - 1] : 0u) | (y + 1 < H ? fl[y + 1] : 0u);
            if (!(near >> x & 1)) return false;
        }
        return true;
    }

    // ObjDesc.find_matching_objs(env, use_location=True) over the object table (every object is in
    // the grid during generation; type is never None on this path).  The agent pose and the object
    // set are final before any descriptor is drawn, so the per-type / per-colour / per-location id
    // sets are built once (prep_masks) and a match is three ANDs.
    BB_HD void prep_masks() {
        uint64_t mt_[4] = {0, 0, 0, 0}, mc_[6] = {0, 0, 0, 0, 0, 0}, ml_[4] = {0, 0, 0, 0};
        int r = room_of(ax, ay);
        int ri, rj; room_ij(r, ri, rj);
        int tx = ri * (S - 1), ty = rj * (S - 1);
        int d1x = (adir == 0) - (adir == 2), d1y = (adir == 1) - (adir == 3);
        int d2x = -d1y, d2y = d1x;
        for (int o = 0; o < nobj; ++o) {
            if (w.px[o] == NONE8) continue;              // hidden inside a box: not in the grid, matches nothing
            const uint64_t bit = 1ull << o;
            int e = w.app[o];
            int t = e_type(e) - T_DOOR, col = e_color(e);
#pragma unroll
            for (int q = 0; q < 4; ++q) mt_[q] |= (t == q) ? bit : 0;
#pragma unroll
            for (int q = 0; q < 6; ++q) mc_[q] |= (col == q) ? bit : 0;
            int x = w.px[o], y = w.py[o];
            if (x < tx || y < ty || x >= tx + S || y >= ty + S) continue;   // locations: agent's room only
            int vx = x - ax, vy = y - ay;
            int p2 = vx * d2x + vy * d2y, p1 = vx * d1x + vy * d1y;
            ml_[0] |= p2 < 0 ? bit : 0; ml_[1] |= p2 > 0 ? bit : 0;
            ml_[2] |= p1 > 0 ? bit : 0; ml_[3] |= p1 < 0 ? bit : 0;
        }
        ctx.sync();
        if (ctx.lane() == 0) {
            for (int q = 0; q < 4; ++q) w.masks[q] = mt_[q];
            for (int q = 0; q < 6; ++q) w.masks[4 + q] = mc_[q];
            for (int q = 0; q < 4; ++q) w.masks[10 + q] = ml_[q];
        }
        ctx.sync();
    }
    BB_HD uint64_t find_matching(int type, int color, int loc) const {
        uint64_t m = type == 0 ? (w.masks[0] | w.masks[1] | w.masks[2] | w.masks[3]) : w.masks[type - T_DOOR];
        if (color != 7) m &= w.masks[4 + color];
        if (loc != LOC_NONE) m &= w.masks[10 + loc - 1];
        return m;
    }

    // LevelGen.rand_obj.  types_mode: 0 = OBJ_TYPES, 1 = OBJ_TYPES_NOT_DOOR, 2 = ['door'].
    BB_HD bool rand_obj(int types_mode, int leaf, int slot) {
        int tries = 0;
        for (;;) {
            if (tries > 100) return false;            // RecursionError('failed to find suitable object')
            ++tries;
            int cv = rand_int(0, 7);
            int color = cv == 0 ? 7 : color_name_to_idx(cv - 1);
            int type = types_mode == 0 ? T_BOX - rand_int(0, 4) : types_mode == 1 ? T_BOX - rand_int(0, 3) : T_DOOR;
            int loc = LOC_NONE;
            if (cfg.locations && rand_bool()) loc = 1 + rand_int(0, 4);
            uint64_t m = find_matching(type, color, loc);
            if (m == 0) continue;
            if (!cfg.implicit_unlock && last_locked >= 0) {
                int li, lj; room_ij(last_locked, li, lj);
                int tx = li * (S - 1), ty = lj * (S - 1);
                bool any_out = false;
                for (int o = 0; o < nobj; ++o)
                    if (m >> o & 1) {
                        int x = w.px[o], y = w.py[o];
                        if (x < tx || y < ty || x >= tx + S || y >= ty + S) any_out = true;
                    }
                if (!any_out) continue;
            }
            w.prog.set[leaf][slot] = m;
            DescInfo d; d.type = type; d.color = color; d.loc = loc; d.count = (uint8_t)__builtin_popcountll(m);
            w.prog.desc[leaf][slot] = d;
            return true;
        }
    }
    BB_HD bool rand_action(int leaf) {
        int a = cfg.action_kinds[rand_int(0, cfg.n_action_kinds)];
        if (a == AK_GOTO) { w.prog.kind[leaf] = L_GOTO; return rand_obj(0, leaf, 0); }
        if (a == AK_PICKUP) { w.prog.kind[leaf] = L_PICKUP; return rand_obj(1, leaf, 0); }
        if (a == AK_OPEN) { w.prog.kind[leaf] = L_OPEN; return rand_obj(2, leaf, 0); }
        w.prog.kind[leaf] = L_PUTNEXT;
        return rand_obj(1, leaf, 0) && rand_obj(0, leaf, 1);
    }
    BB_HD void clear_prog() {
        ctx.sync();
        uint32_t* p = (uint32_t*)&w.prog;
        for (int k = ctx.lane(); k < (int)(sizeof(Prog) / 4); k += ctx.nlanes()) p[k] = 0;
        ctx.sync();
        w.prog.start_carry = NONE8;
    }
    // LevelGen.rand_instr (depth <= 2: seq -> and -> action).
    BB_HD bool rand_instr() {
        clear_prog();
        int kind = cfg.instr_kinds[rand_int(0, cfg.n_instr_kinds)];
        if (kind == IK_ACTION) {
            w.prog.root = R_ACTION; w.prog.n_a = 1;
            return rand_action(0);
        }
        if (kind == IK_AND) {
            w.prog.root = R_AND; w.prog.n_a = 2;
            return rand_action(0) && rand_action(1);
        }
        for (int side = 0; side < 2; ++side) {
            int base = side * 2;
            int k2 = rand_int(0, 2);                  // ['action', 'and']
            int n = k2 == 0 ? 1 : 2;
            if (side == 0) w.prog.n_a = n; else w.prog.n_b = n;
            for (int q = 0; q < n; ++q)
                if (!rand_action(base + q)) return false;
        }
        w.prog.root = rand_int(0, 2) == 0 ? R_BEFORE : R_AFTER;
        return true;
    }

    // validate_instrs (levelgen.py:104-155); false => RejectSampling.
    BB_HD bool validate() {
        uint32_t locked_colors = 0;
        bool unb = cfg.kind == K_LEVELGEN && cfg.unblocking;
        if (unb)
            for (int o = 0; o < nobj; ++o)
                if (e_type(w.app[o]) == T_DOOR && e_state(w.E[eidx(w.px[o], w.py[o])]) == S_LOCKED)
                    locked_colors |= 1u << e_color(w.app[o]);
        for (int leaf = 0; leaf < 4; ++leaf) {
            int k = w.prog.kind[leaf];
            if (k == L_NONE) continue;
            if (k == L_PUTNEXT) {
                uint64_t mv = w.prog.set[leaf][0], fx = w.prog.set[leaf][1];
                if (mv & fx) return false;
                for (int o = 0; o < nobj; ++o)
                    if (mv >> o & 1) {
                        int x = w.px[o], y = w.py[o];
                        const int nx[4] = {x + 1, x - 1, x, x}, ny[4] = {y, y, y + 1, y - 1};
                        for (int q = 0; q < 4; ++q) {
                            int id = w.I[iidx(nx[q], ny[q])];
                            if (id >= 2 && (fx >> (id - 2) & 1)) return false;
                        }
                    }
            }
            if (unb)
                for (int s = 0; s < 2; ++s) {
                    DescInfo d = w.prog.desc[leaf][s];
                    if ((s == 0 || k == L_PUTNEXT) && d.type == T_KEY && d.color != 7 && (locked_colors >> d.color & 1)) return false;
                }
        }
        return true;
    }

    // LevelGen.gen_mission
    BB_HD bool mission_levelgen() {
        tick(PH_BUILD);
        if (rand_float01() < cfg.locked_room_prob) {
            // add_locked_room
            int door_color;
            for (;;) {
                int i = rand_int(0, cols);
                int j = rand_int(0, rows);
                int k = rand_int(0, 4);
                locked_room = last_locked = j * cols + i;
                if (!has_neighbor(locked_room, k)) continue;
                door_color = rand_color();
                if (add_door(locked_room, k, door_color, true) < 0) return false;
                break;
            }
            for (;;) {
                int i = rand_int(0, cols);
                int j = rand_int(0, rows);
                if (j * cols + i == locked_room) continue;
                if (add_object(j * cols + i, T_KEY, door_color) < 0) return false;
                break;
            }
        }
        tick(PH_LOCK);
        if (!connect_all()) return false;
        tick(PH_CONNECT);
        if (!add_distractors(cfg.num_dists, false)) return false;
        tick(PH_DISTS);
        for (;;) {
            if (!place_agent()) return false;
            if (room_of(ax, ay) == locked_room) continue;
            break;
        }
        tick(PH_AGENT);
        if (!cfg.unblocking && !objs_reachable()) return false;
        tick(PH_REACH);
        prep_masks();
        bool ok = rand_instr();
        tick(PH_INSTR);
        return ok;
    }

    // ObjDesc(type, color, loc) resolved against the final scene.  type 0 = any type, color 7 = any colour.
    // With type None the reference also matches wall cells of that colour (verifier.py:112-127): they cannot be
    // picked up or opened, but they count for the "a"/"the" article.
    BB_HD void set_desc_tcl(int leaf, int slot, int type, int color, int loc) {
        uint64_t m = find_matching(type, color, loc);
        w.prog.set[leaf][slot] = m;
        int cnt = __builtin_popcountll(m);
        if (type == 0 && color == C_GREY && loc == LOC_NONE) cnt += 2;      // the room walls are grey: plural for sure
        DescInfo d; d.type = type; d.color = color; d.loc = loc; d.count = (uint8_t)(cnt > 255 ? 255 : cnt);
        w.prog.desc[leaf][slot] = d;
    }
    BB_HD void set_desc(int leaf, int slot, int obj) {
        set_desc_tcl(leaf, slot, e_type(w.app[obj]), e_color(w.app[obj]), LOC_NONE);
    }

    // ---------------- helpers of the bonus scripts ----------------
    // _rand_subset(COLOR_NAMES, k): repeated _rand_elem on the shrinking (sorted-name) list
    BB_HD void rand_subset_colors(int k, int* out) {
        uint32_t avail = 0x3F;
        for (int q = 0; q < k; ++q) {
            int pick = rand_int(0, 6 - q);
            int pos = 0;
            for (int b = 0; b < 6; ++b)
                if (avail >> b & 1) { if (pick-- == 0) { pos = b; break; } }
            avail &= ~(1u << pos);
            out[q] = color_name_to_idx(pos);
        }
    }
    // _rand_subset(list of n ids starting at first, 2)
    BB_HD void rand_subset2(int first, int n, int& a, int& b) {
        int i = rand_int(0, n);
        int j = rand_int(0, n - 1);
        if (j >= i) ++j;
        a = first + i; b = first + j;
    }
    // RoomGrid.add_object(i, j, kind=None|k, color=None|c)
    BB_HD int add_object_rand(int r, int type /* 0 = random */, int color /* -1 = random */) {
        if (type == 0) type = T_KEY + rand_int(0, 3);
        if (color < 0) color = rand_color();
        return add_object(r, type, color);
    }
    // RoomGrid.add_door(i, j, door_idx=k|None, color=c|None, locked=b|None): draws in that order
    BB_HD int add_door_opt(int r, int k, int color, int locked /* -1 = random */) {
        if (k < 0)
            for (;;) {
                k = rand_int(0, 4);
                if (has_neighbor(r, k) && !has_door(r, k)) break;
            }
        if (color < 0) color = rand_color();
        if (locked < 0) locked = rand_bool() ? 1 : 0;
        return add_door(r, k, color, locked != 0);
    }
    // RoomGrid.remove_wall(i, j, wall_idx): the wall segment between two rooms becomes floor; counts as a connection
    BB_HD void remove_wall(int r, int k) {
        int ri, rj; room_ij(r, ri, rj);
        int tx = ri * (S - 1), ty = rj * (S - 1);
        for (int q = 1; q < S - 1; ++q) {
            int x = k == 0 ? tx + S - 1 : k == 2 ? tx : tx + q;
            int y = k == 1 ? ty + S - 1 : k == 3 ? ty : ty + q;
            set_cell(x, y, E_EMPTY, 0);
        }
        doors |= 1ull << (16 * k + r);
        doors |= 1ull << (16 * ((k + 2) & 3) + neighbor(r, k));
    }
    // grid.set(x, y, obj) / place_obj(obj, (x, y), (1, 1)): an object at an exact cell, no sampling, no draws
    BB_HD int put_fixed(int type, int color, int x, int y) {
        if (nobj >= cfg.maxo) return 0;
        int id = nobj++;
        w.app[id] = e_make(type, color, 0); w.px[id] = x; w.py[id] = y; w.cont[id] = NONE8;
        set_cell(x, y, w.app[id], id + 2);
        note_obj(e_make(type, color, 0));
        return id;
    }
    BB_HD void one_leaf(int kind, bool strict = false) {
        clear_prog();
        prep_masks();
        w.prog.root = R_ACTION; w.prog.n_a = 1; w.prog.kind[0] = (uint8_t)kind;
        w.prog.strict = strict ? 1 : 0;
        w.prog.start_carry = NONE8;
    }

    // gen_mission of the bonus levels (babyai/levels/bonus_levels.py), one case per level class.
    BB_HD bool mission_bonus() {
        const int mid = (rows > 1 ? cols : 0) + (cols > 1 ? 1 : 0);       // room (1,1) of a 3x3 maze, (1,0) of a 1x3
        const int R11 = rows >= 2 && cols >= 2 ? 1 * cols + 1 : mid;
        int colors[6];
        switch (cfg.script) {
        case BS_GOTO_REDBLUE_BALL: {                    // :7-40
            if (!place_agent()) return false;
            int first = nobj;
            if (!add_distractors(cfg.num_dists, false)) return false;
            for (int o = first; o < nobj; ++o)
                if (e_type(w.app[o]) == T_BALL && (e_color(w.app[o]) == C_BLUE || e_color(w.app[o]) == C_RED)) return false;
            int color = rand_int(0, 2) == 0 ? C_RED : C_BLUE;
            int obj = add_object(0, T_BALL, color);
            if (obj < 0) return false;
            if (!objs_reachable()) return false;
            one_leaf(L_GOTO); set_desc(0, 0, obj);
            return true;
        }
        case BS_OPEN_RED_DOOR: {                        // :43-62
            if (add_door(0, 0, C_RED, false) < 0) return false;
            if (!place_agent(0)) return false;
            one_leaf(L_OPEN); set_desc_tcl(0, 0, T_DOOR, C_RED, LOC_NONE);
            return true;
        }
        case BS_OPEN_DOOR: {                            // :65-99  sp[0]: 0 random, 1 color, 2 loc; sp[1]: debug
            rand_subset_colors(4, colors);
            int first = nobj;
            for (int k = 0; k < 4; ++k)
                if (add_door(R11, k, colors[k], false) < 0) return false;
            int sel = cfg.sp[0];
            if (sel == 0) sel = 1 + rand_int(0, 2);
            int loc = LOC_NONE;
            if (sel == 2) loc = 1 + rand_int(0, 4);
            if (!place_agent(R11)) return false;
            one_leaf(L_OPEN, cfg.sp[1] != 0);
            if (sel == 1) set_desc_tcl(0, 0, T_DOOR, e_color(w.app[first]), LOC_NONE);
            else set_desc_tcl(0, 0, T_DOOR, 7, loc);
            return true;
        }
        case BS_GOTO_DOOR: {                            // :150-171
            int first = nobj;
            for (int k = 0; k < 4; ++k)
                if (add_door_opt(R11, -1, -1, -1) < 0) return false;
            if (!place_agent(R11)) return false;
            int obj = first + rand_int(0, 4);
            one_leaf(L_GOTO); set_desc_tcl(0, 0, T_DOOR, e_color(w.app[obj]), LOC_NONE);
            return true;
        }
        case BS_GOTO_OBJ_DOOR: {                        // :174-197
            if (!place_agent(R11)) return false;
            int first = nobj;
            if (!add_distractors(8, false, R11)) return false;
            for (int k = 0; k < 4; ++k)
                if (add_door_opt(R11, -1, -1, -1) < 0) return false;
            if (!objs_reachable()) return false;
            int obj = first + rand_int(0, 12);
            one_leaf(L_GOTO); set_desc(0, 0, obj);
            return true;
        }
        case BS_ACTION_OBJ_DOOR: {                      // :200-234
            int first = nobj;
            if (!add_distractors(5, true, R11)) return false;
            for (int k = 0; k < 4; ++k)
                if (add_door_opt(R11, -1, -1, 0) < 0) return false;
            if (!place_agent(R11)) return false;
            int obj = first + rand_int(0, 9);
            bool go = rand_bool();
            int kind = go ? L_GOTO : (e_type(w.app[obj]) == T_DOOR ? L_OPEN : L_PICKUP);
            one_leaf(kind); set_desc(0, 0, obj);
            return true;
        }
        case BS_UNLOCK_LOCAL: {                         // :237-264  sp[0]: distractors
            int door = add_door_opt(R11, -1, -1, 1);
            if (door < 0) return false;
            if (add_object(R11, T_KEY, e_color(w.app[door])) < 0) return false;
            if (cfg.sp[0] && !add_distractors(3, true, R11)) return false;
            if (!place_agent(R11)) return false;
            one_leaf(L_OPEN); set_desc_tcl(0, 0, T_DOOR, 7, LOC_NONE);
            return true;
        }
        case BS_KEY_IN_BOX: {                           // :267-287
            int door = add_door_opt(R11, -1, -1, 1);
            if (door < 0) return false;
            int box_color = rand_color();
            int box = add_object(R11, T_BOX, box_color);
            if (box < 0 || nobj >= cfg.maxo) return false;
            int key = nobj++;                            // lives inside the box until it is toggled
            w.app[key] = e_make(T_KEY, e_color(w.app[door]), 0); w.px[key] = NONE8; w.py[key] = NONE8; w.cont[key] = NONE8;
            note_obj(e_make(T_KEY, e_color(w.app[door]), 0));
            w.cont[box] = (uint8_t)key;
            if (!place_agent(R11)) return false;
            one_leaf(L_OPEN); set_desc_tcl(0, 0, T_DOOR, 7, LOC_NONE);
            return true;
        }
        case BS_UNLOCK_PICKUP: {                        // :290-329  1x2 rooms; sp[0]: distractors
            int obj = add_object_rand(1, T_BOX, -1);
            if (obj < 0) return false;
            int door = add_door_opt(0, 0, -1, 1);
            if (door < 0) return false;
            if (add_object(0, T_KEY, e_color(w.app[door])) < 0) return false;
            if (cfg.sp[0] && !add_distractors(4, true)) return false;
            if (!place_agent(0)) return false;
            one_leaf(L_PICKUP); set_desc(0, 0, obj);
            return true;
        }
        case BS_BLOCKED_UNLOCK_PICKUP: {                // :332-361
            int obj = add_object_rand(1, T_BOX, -1);
            if (obj < 0) return false;
            int door = add_door_opt(0, 0, -1, 1);
            if (door < 0) return false;
            int color = rand_color();
            if (nobj >= cfg.maxo) return false;
            int ball = nobj++;                           // grid.set(): put right in front of the door, no sampling
            int bx = w.px[door] - 1, by = w.py[door];
            w.app[ball] = e_make(T_BALL, color, 0); w.px[ball] = bx; w.py[ball] = by; w.cont[ball] = NONE8;
            set_cell(bx, by, w.app[ball], ball + 2);
            note_obj(e_make(T_BALL, color, 0));
            if (add_object(0, T_KEY, e_color(w.app[door])) < 0) return false;
            if (!place_agent(0)) return false;
            one_leaf(L_PICKUP); set_desc_tcl(0, 0, T_BOX, 7, LOC_NONE);
            return true;
        }
        case BS_UNLOCK_TO_UNLOCK: {                     // :364-398  1x3 rooms
            rand_subset_colors(2, colors);
            if (add_door(0, 0, colors[0], true) < 0) return false;
            if (add_object(2, T_KEY, colors[0]) < 0) return false;
            if (add_door(1, 0, colors[1], true) < 0) return false;
            if (add_object(1, T_KEY, colors[1]) < 0) return false;
            if (add_object_rand(0, T_BALL, -1) < 0) return false;
            if (!place_agent(1)) return false;
            one_leaf(L_PICKUP); set_desc_tcl(0, 0, T_BALL, 7, LOC_NONE);
            return true;
        }
        case BS_PICKUP_DIST: {                          // :401-444  sp[0]: debug (strict)
            int first = nobj;
            if (!add_distractors(5, true)) return false;
            if (!place_agent(0)) return false;
            int obj = first + rand_int(0, 5);
            int sel = rand_int(0, 3);                    // "type", "color", "both"
            int type = e_type(w.app[obj]), color = e_color(w.app[obj]);
            if (sel == 1) type = 0; else if (sel == 0) color = 7;
            one_leaf(L_PICKUP, cfg.sp[0] != 0); set_desc_tcl(0, 0, type, color, LOC_NONE);
            return true;
        }
        case BS_PICKUP_ABOVE: {                         // :447-469
            int obj = add_object_rand(0 * cols + 1, 0, -1);
            if (obj < 0) return false;
            if (add_door_opt(R11, 3, -1, 0) < 0) return false;
            if (!place_agent(R11)) return false;
            if (!connect_all()) return false;
            one_leaf(L_PICKUP); set_desc(0, 0, obj);
            return true;
        }
        case BS_OPEN_TWO_DOORS: {                       // :472-562  sp[0], sp[1]: fixed colours + 1 (0 = random); sp[2]: strict
            rand_subset_colors(2, colors);
            int c1 = cfg.sp[0] ? cfg.sp[0] - 1 : colors[0];
            int c2 = cfg.sp[1] ? cfg.sp[1] - 1 : colors[1];
            if (add_door(R11, 2, c1, false) < 0) return false;
            if (add_door(R11, 0, c2, false) < 0) return false;
            if (!place_agent(R11)) return false;
            clear_prog(); prep_masks();
            w.prog.root = R_BEFORE; w.prog.n_a = 1; w.prog.n_b = 1; w.prog.kind[0] = L_OPEN; w.prog.kind[2] = L_OPEN;
            w.prog.strict = cfg.sp[2] ? 1 : 0; w.prog.start_carry = NONE8;
            set_desc_tcl(0, 0, T_DOOR, c1, LOC_NONE);
            set_desc_tcl(2, 0, T_DOOR, c2, LOC_NONE);
            return true;
        }
        case BS_FIND_OBJ: {                             // :565-611
            int i = rand_int(0, rows);
            int j = rand_int(0, cols);
            int obj = add_object_rand(j * cols + i, 0, -1);
            if (obj < 0) return false;
            if (!place_agent(R11)) return false;
            if (!connect_all()) return false;
            one_leaf(L_PICKUP); set_desc_tcl(0, 0, e_type(w.app[obj]), 7, LOC_NONE);
            return true;
        }
        case BS_KEY_CORRIDOR: {                         // :614-704  3 columns, `rows` rows
            for (int j = 1; j < rows; ++j) remove_wall(j * cols + 1, 3);
            int room_idx = rand_int(0, rows);
            int door = add_door_opt(room_idx * cols + 2, 2, -1, 1);
            if (door < 0) return false;
            int obj = add_object_rand(room_idx * cols + 2, T_BALL, -1);
            if (obj < 0) return false;
            int kj = rand_int(0, rows);
            if (add_object(kj * cols + 0, T_KEY, e_color(w.app[door])) < 0) return false;
            if (!place_agent((rows / 2) * cols + 1)) return false;
            if (!connect_all()) return false;
            one_leaf(L_PICKUP); set_desc_tcl(0, 0, T_BALL, 7, LOC_NONE);
            return true;
        }
        case BS_ONE_ROOM: {                             // :707-763
            if (add_object_rand(0, T_BALL, -1) < 0) return false;
            if (!place_agent()) return false;
            one_leaf(L_PICKUP); set_desc_tcl(0, 0, T_BALL, 7, LOC_NONE);
            return true;
        }
        case BS_PUT_NEXT: {                             // :766-904  1x2 rooms; num_dists per room; sp[0]: start carrying
            if (!place_agent(0)) return false;
            int fl = nobj;
            if (!add_distractors(cfg.num_dists, true, 0)) return false;
            int fr = nobj;
            if (!add_distractors(cfg.num_dists, true, 1)) return false;
            remove_wall(0, 0);
            int a = fl + rand_int(0, cfg.num_dists);
            int b = fr + rand_int(0, cfg.num_dists);
            if (rand_bool()) { int t = a; a = b; b = t; }
            one_leaf(L_PUTNEXT);
            set_desc(0, 0, a); set_desc(0, 1, b);
            if (cfg.sp[0]) w.prog.start_carry = (uint8_t)a;
            return true;
        }
        case BS_MOVE_TWO_ACROSS: {                      // :907-971
            if (!place_agent(0)) return false;
            int fl = nobj;
            if (!add_distractors(cfg.num_dists, true, 0)) return false;
            int fr = nobj;
            if (!add_distractors(cfg.num_dists, true, 1)) return false;
            remove_wall(0, 0);
            int a, d, b, c2;
            rand_subset2(fl, cfg.num_dists, a, d);
            rand_subset2(fr, cfg.num_dists, b, c2);
            clear_prog(); prep_masks();
            w.prog.root = R_BEFORE; w.prog.n_a = 1; w.prog.n_b = 1; w.prog.kind[0] = L_PUTNEXT; w.prog.kind[2] = L_PUTNEXT;
            w.prog.strict = 0; w.prog.start_carry = NONE8;
            set_desc(0, 0, a); set_desc(0, 1, b);
            set_desc(2, 0, c2); set_desc(2, 1, d);
            return true;
        }
        case BS_OPEN_DOORS_ORDER: {                     // :974-1049  sp[0]: num_doors, sp[1]: debug
            const int nd = cfg.sp[0];
            rand_subset_colors(nd, colors);
            int first = nobj;
            for (int k = 0; k < nd; ++k)
                if (add_door_opt(R11, -1, colors[k], 0) < 0) return false;
            if (!place_agent(R11)) return false;
            int d1, d2;
            rand_subset2(first, nd, d1, d2);
            int mode = rand_int(0, 3);
            clear_prog(); prep_masks();
            const bool dbg = cfg.sp[1] != 0;
            w.prog.start_carry = NONE8;
            w.prog.kind[0] = L_OPEN; w.prog.n_a = 1;
            set_desc_tcl(0, 0, T_DOOR, e_color(w.app[d1]), LOC_NONE);
            if (mode == 0) {
                w.prog.root = R_ACTION; w.prog.strict = dbg ? 1 : 0;
            } else {
                w.prog.root = mode == 1 ? R_BEFORE : R_AFTER;
                w.prog.kind[2] = L_OPEN; w.prog.n_b = 1; w.prog.strict = dbg ? 5 : 0;
                set_desc_tcl(2, 0, T_DOOR, e_color(w.app[d2]), LOC_NONE);
            }
            return true;
        }
        // ---- test_levels.py: fixed layouts (objects put at exact cells, agent pose assigned) ----
        case BS_TEST_GOTO_BLOCKED: {                    // test_levels.py:13-38
            if (!place_agent()) return false;           // draws are consumed, the pose is then overwritten
            ax = 3; ay = 3; adir = 0;
            int obj = put_fixed(T_BALL, C_YELLOW, 1, 1);
            for (int i = 1; i <= 3; ++i)
                for (int j = 1; j <= 3; ++j)
                    if (!((i == 1 && j == 1) || (i == 3 && j == 3))) put_fixed(T_BALL, C_RED, i, j);
            one_leaf(L_GOTO); set_desc(0, 0, obj);
            return true;
        }
        case BS_TEST_PUTNEXT_BLOCKED: {                 // :41-66
            if (!place_agent()) return false;
            ax = 3; ay = 3; adir = 0;
            int o1 = put_fixed(T_BALL, C_YELLOW, 4, 4);
            int o2 = put_fixed(T_BALL, C_BLUE, 1, 1);
            put_fixed(T_BALL, C_RED, 1, 2);
            put_fixed(T_BALL, C_RED, 2, 1);
            one_leaf(L_PUTNEXT); set_desc(0, 0, o1); set_desc(0, 1, o2);
            return true;
        }
        case BS_TEST_PUTNEXT_DOOR1:
        case BS_TEST_PUTNEXT_DOOR2: {                   // :69-107  2x1 rooms
            ax = 3; ay = 3; adir = 0;
            int door = add_door_opt(0, -1, C_RED, 0);
            if (door < 0) return false;
            int o1 = put_fixed(T_BALL, C_YELLOW, 4, 4);
            int o2 = put_fixed(T_BALL, C_BLUE, w.px[door], w.py[door] + 1);
            clear_prog(); prep_masks();
            w.prog.strict = 0;
            if (cfg.script == BS_TEST_PUTNEXT_DOOR1) {
                w.prog.root = R_BEFORE; w.prog.n_a = 1; w.prog.n_b = 1; w.prog.kind[0] = L_OPEN; w.prog.kind[2] = L_PUTNEXT;
                set_desc_tcl(0, 0, T_DOOR, C_RED, LOC_NONE);
                set_desc(2, 0, o1); set_desc(2, 1, o2);
            } else {
                w.prog.root = R_ACTION; w.prog.n_a = 1; w.prog.kind[0] = L_PUTNEXT;
                set_desc(0, 0, o1); set_desc(0, 1, o2);
            }
            return true;
        }
        case BS_TEST_PUTNEXT_IDENTICAL: {               // :110-138
            ax = 3; ay = 3; adir = 0;
            put_fixed(T_BOX, C_YELLOW, 1, 1);
            put_fixed(T_BALL, C_BLUE, 4, 4);
            put_fixed(T_BALL, C_RED, 2, 2);
            clear_prog(); prep_masks();
            w.prog.root = R_BEFORE; w.prog.n_a = 1; w.prog.n_b = 1; w.prog.kind[0] = L_PUTNEXT; w.prog.kind[2] = L_PUTNEXT;
            set_desc_tcl(0, 0, T_BALL, C_BLUE, LOC_NONE); set_desc_tcl(0, 1, T_BOX, C_YELLOW, LOC_NONE);
            set_desc_tcl(2, 0, T_BOX, C_YELLOW, LOC_NONE); set_desc_tcl(2, 1, T_BALL, 7, LOC_NONE);
            return true;
        }
        case BS_TEST_UNBLOCKING_LOOP: {                 // :141-168  2x2 rooms
            ax = 15; ay = 4; adir = 2;
            if (add_door(0, 1, C_RED, false) < 0) return false;
            if (add_door(1 * cols + 0, 0, C_RED, false) < 0) return false;
            if (add_door(1 * cols + 1, 3, C_BLUE, false) < 0) return false;
            put_fixed(T_BOX, C_YELLOW, 9, 1);
            put_fixed(T_BALL, C_BLUE, 5, 3);
            put_fixed(T_BALL, C_YELLOW, 6, 2);
            put_fixed(T_KEY, C_BLUE, 15, 15);
            clear_prog(); prep_masks();
            w.prog.root = R_BEFORE; w.prog.n_a = 1; w.prog.n_b = 2;
            w.prog.kind[0] = L_PUTNEXT; w.prog.kind[2] = L_GOTO; w.prog.kind[3] = L_GOTO;
            set_desc_tcl(0, 0, T_KEY, C_BLUE, LOC_NONE); set_desc_tcl(0, 1, T_DOOR, C_BLUE, LOC_NONE);
            set_desc_tcl(2, 0, T_BALL, C_YELLOW, LOC_NONE); set_desc_tcl(3, 0, T_BOX, C_YELLOW, LOC_NONE);
            return true;
        }
        case BS_TEST_PUTNEXT_CLOSE_DOOR: {              // :171-201  2x2 rooms
            ax = 5; ay = 10; adir = 2;
            int d1 = add_door(0, 1, C_RED, false);
            if (d1 < 0) return false;
            if (add_door(1 * cols + 0, 0, C_RED, false) < 0) return false;
            if (add_door(1 * cols + 1, 3, C_BLUE, false) < 0) return false;
            const int px1 = w.px[d1], py1 = w.py[d1];
            put_fixed(T_BALL, C_BLUE, px1, py1 - 1);
            put_fixed(T_BALL, C_BLUE, px1, py1 - 2);
            if (px1 - 1 >= 1) put_fixed(T_BOX, C_GREEN, px1 - 1, py1 - 1);
            if (px1 + 1 < 8) put_fixed(T_BOX, C_GREEN, px1 + 1, py1 - 1);
            put_fixed(T_BOX, C_YELLOW, 3, 15);
            one_leaf(L_PUTNEXT);
            set_desc_tcl(0, 0, T_BOX, C_YELLOW, LOC_NONE); set_desc_tcl(0, 1, T_BALL, C_BLUE, LOC_NONE);
            return true;
        }
        case BS_TEST_LOTS_OF_BLOCKERS: {                // :204-232
            ax = 5; ay = 5; adir = 0;
            const int bx[6] = {2, 2, 2, 3, 2, 1}, by[6] = {1, 2, 3, 4, 6, 3};
            for (int q = 0; q < 6; ++q) put_fixed(T_BOX, C_YELLOW, bx[q], by[q]);
            put_fixed(T_BALL, C_BLUE, 1, 2);
            put_fixed(T_BALL, C_RED, 3, 6);
            one_leaf(L_PUTNEXT);
            set_desc_tcl(0, 0, T_BALL, C_RED, LOC_NONE); set_desc_tcl(0, 1, T_BALL, C_BLUE, LOC_NONE);
            return true;
        }
        default: return false;
        }
    }

    // The hand-written single-instruction levels (gen_mission of GoToRedBall[Grey] / GoToObj / GoToLocal / GoTo /
    // Pickup / UnblockPickup / Open / PutNext[Local] and the lock-first pair Unlock / GoToImpUnlock), driven by
    // the cfg fields; RNG draw order follows iclr19_levels.py line by line.
    BB_HD bool mission_goto() {
        int target = -1, target2 = -1, locked_door = -1, lock_color = 0;
        int first = 0, ndist = 0;
        if (cfg.lock) {
            // Unlock :424-437 / GoToImpUnlock :311-325 : locked door on a random side of a random room, key elsewhere
            int id = rand_int(0, cols);
            int jd = rand_int(0, rows);
            locked_room = jd * cols + id;
            locked_door = add_door_random_side(locked_room, true, lock_color);
            if (locked_door < 0) return false;
            for (;;) {
                int ik = rand_int(0, cols);
                int jk = rand_int(0, rows);
                if (jk * cols + ik == locked_room) continue;
                if (add_object(jk * cols + ik, T_KEY, lock_color) < 0) return false;
                break;
            }
            int excl = -1;
            if (cfg.lock_color_excl && rand_bool()) excl = lock_color;
            if (!connect_all(excl)) return false;
            first = nobj;
            for (int i = 0; i < cols; ++i)
                for (int j = 0; j < rows; ++j)
                    if (j * cols + i != locked_room)
                        if (!add_distractors(cfg.num_dists, false, j * cols + i)) return false;
            for (;;) {
                if (!place_agent()) return false;
                if (room_of(ax, ay) != locked_room) break;
            }
            if (cfg.check_reach && !objs_reachable()) return false;
            if (cfg.target == TG_LOCKED_ROOM_OBJ) {
                target = nobj;
                if (!add_distractors(1, false, locked_room)) return false;
            } else {
                target = locked_door;
            }
        } else {
            tick(PH_BUILD);
            if (!place_agent()) return false;
            tick(PH_AGENT);
            if (cfg.redball) {
                target = add_object(0, T_BALL, C_RED);
                if (target < 0) return false;
            }
            if (cfg.connect && !connect_all()) return false;
            tick(PH_CONNECT);
            first = nobj;
            ndist = cfg.num_dists;
            if (!add_distractors(ndist, cfg.all_unique != 0)) return false;
            tick(PH_DISTS);
            if (cfg.grey_dists)
                for (int o = first; o < nobj; ++o) {
                    int e = e_make(e_type(w.app[o]), C_GREY, 0);
                    w.app[o] = e;
                    w.E[eidx(w.px[o], w.py[o])] = e;
                }
            if (cfg.check_reach == 1 && !objs_reachable()) { tick(PH_REACH); return false; }
            if (cfg.check_reach == 2 && objs_reachable()) { tick(PH_REACH); return false; }     // UnblockPickup :383-386
            tick(PH_REACH);
            if (cfg.target == TG_DIST) {
                target = first + rand_int(0, ndist);                        // _rand_elem(objs)
            } else if (cfg.target == TG_TWO_DISTS) {                        // _rand_subset(objs, 2)
                int a = rand_int(0, ndist);
                int b = rand_int(0, ndist - 1);
                if (b >= a) ++b;
                target = first + a; target2 = first + b;
            } else if (cfg.target == TG_DOOR) {
                // Open :401-411 : doors listed room by room (i outer, j inner), side by side => every door twice
                int n = 0;
                for (int r = 0; r < rows * cols; ++r)
                    for (int k = 0; k < 4; ++k) n += has_door(r, k) ? 1 : 0;
                if (n == 0) return false;
                int pick = rand_int(0, n);
                for (int i = 0; i < cols && target < 0; ++i)
                    for (int j = 0; j < rows && target < 0; ++j)
                        for (int k = 0; k < 4; ++k)
                            if (has_door(j * cols + i, k)) {
                                if (pick-- == 0) { target = w.I[iidx(w.door_x[j * cols + i][k], w.door_y[j * cols + i][k])] - 2; break; }
                            }
            }
        }
        clear_prog();
        prep_masks();
        w.prog.root = R_ACTION; w.prog.n_a = 1; w.prog.kind[0] = (uint8_t)cfg.instr;
        set_desc(0, 0, target);
        if (cfg.instr == L_PUTNEXT) set_desc(0, 1, target2);
        if (cfg.doors_open)
            for (int o = 0; o < nobj; ++o)
                if (e_type(w.app[o]) == T_DOOR) {
                    int e = e_make(T_DOOR, e_color(w.app[o]), S_OPEN);
                    w.E[eidx(w.px[o], w.py[o])] = e;
                }
        return true;
    }

    // RoomGridLevel._gen_grid: retry until a mission is generated and validated.
    // Returns max_steps (levelgen.py:42-45).
    BB_HD int generate() {
        return cfg.kind == K_LEVELGEN ? generate_kind<K_LEVELGEN>() : cfg.kind == K_BONUS ? generate_kind<K_BONUS>() : generate_kind<K_GOTO>();
    }
    // KIND is the level family (cfg.kind): a kernel instantiated per family carries only that family's mission code
    template <int KIND>
    BB_HD int generate_kind() {
        if constexpr (ctx_profiles<Ctx>::value) t_last = ctx.now();
        gave_up = false;
        for (int attempts = 0;; ++attempts) {
            // Last-resort guard against a generation that can never succeed (every known unbounded loop of the
            // reference is bounded above; this only keeps an unknown one from hanging the GPU): the caller freezes
            // the env and counts the failure (bbai_generator_failures) instead of spinning for ever.
            if (attempts >= MAX_ATTEMPTS) { gave_up = true; break; }
            if (attempt<KIND>()) break;
        }
        return finish();
    }
    static constexpr int MAX_ATTEMPTS = 200000;
    // ONE pass of the rejection loop of RoomGridLevel._gen_grid (levelgen.py:80-102): layout, mission, validation.
    // k_pregen drives this directly, so that the groups of a wave stay together at attempt granularity: a group whose
    // attempt succeeded moves on to its next level while its neighbours retry.
    template <int KIND>
    BB_HD bool attempt() {
        count(PH_ATTEMPTS);
        build_rooms();
        bool ok;
        if constexpr (KIND == K_LEVELGEN) ok = mission_levelgen();
        else if constexpr (KIND == K_BONUS) ok = mission_bonus();
        else ok = mission_goto();
        tick(PH_INSTR);
        bool v = ok && validate();
        tick(PH_VALIDATE);
        return v;
    }
    // max_steps of the accepted level (levelgen.py:42-45)
    BB_HD int finish() {
        ctx.sync();
        int navs = 0;
        for (int leaf = 0; leaf < 4; ++leaf)
            navs += w.prog.kind[leaf] == L_PUTNEXT ? 2 : w.prog.kind[leaf] != L_NONE ? 1 : 0;
        return navs * S * S * rows * cols;
    }
};

}  // namespace bbai
