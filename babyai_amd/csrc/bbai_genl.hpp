// bbai_genl.hpp -- the level generator with ONE LANE = ONE LEVEL (k_pregen_lane in bbai_engine.hip).
//
// Same reference semantics, draw for draw, as bbai_gen.hpp (which stays the generator of the lock-first and bonus families and the
// yardstick this file is tested against: tests/test_hostsim_genl.py generates every eligible level with both on the host and compares
// record, pose, program and RNG position byte for byte):
//   RoomGridLevel._gen_grid rejection loop        babyai/levels/levelgen.py:77-102
//   validate_instrs                               babyai/levels/levelgen.py:104-155
//   check_objs_reachable                          babyai/levels/levelgen.py:201-253
//   LevelGen.gen_mission / add_locked_room        babyai/levels/levelgen.py:293-352
//   LevelGen.rand_obj / rand_instr                babyai/levels/levelgen.py:354-460
//   ObjDesc.find_matching_objs (use_location)     babyai/levels/verifier.py:96-161
//   GoToRedBall / GoToObj / GoToLocal / GoTo ...  babyai/levels/iclr19_levels.py:40-124,187-257,330-411
//   RoomGrid / MiniGridEnv placement helpers      gym_minigrid (absent dependency), restated per SURVEY.md Appendix B3-B7
//
// Why a second form.  The generator is one scalar program per env.  With a lane GROUP per env (bbai_gen.hpp: 32 lanes, two envs per wave)
// every instruction of that program is issued for one or two envs: 4 100 wave instructions per GoToLocal level, 19 000 per GoTo level
// (profiles/r06/generator_instruction_counters.txt), the vector ALU 64 % busy -- the chip generates 0.05 levels per ns because it issues
// that many instructions, not because it waits.  Here every lane of a wave runs its own level: an instruction serves up to 64 levels, and
// what is lost to divergence (loops run for their slowest lane) is a factor of 2-3, not 30.  What made round 5's first attempt at this
// shape 2 x SLOWER was its working set: the lane-group generator's 1.3-KB appearance plane, id plane and tables, per lane, in global
// memory -- every cell test a memory round trip.  This form has NO planes: the grid is what the reference's placement code actually asks
// of it --
//   * occupancy: one bit per cell (a 64-bit board in registers for grids of <= 64 cells, one 32-bit row per grid row in LDS for the
//     mazes); walls are arithmetic (every (S - 1)-th row / column), a door is "an object bit on a wall line";
//   * objects: one LDS word each (appearance | x << 8 | y << 16 | contents << 24);
//   * the per-room door slots drawn by RoomGrid._gen_grid: one LDS word per room;
//   * the instruction (Prog, 28 words) in its final layout;
// -- 76 words per lane for a single room, about 110 for a 3 x 3 maze, interleaved over the wave's lanes (word k of lane l at dword
// 64 k + l: conflict-free).  Reachability is a flood fill over the bit rows (Kogge-Stone fills along a row, sweeps down and up), descriptor
// matching a loop over the object words.  The record the rest of the engine reads (planes with their wall margin, tables, program) is
// materialised ONCE, at write-out: a per-level-kind template (walls and empty cells, built on the host at bbai_create) copied with 16-byte
// stores, the objects scattered over it.
//
// Random numbers.  A lane cannot afford MT19937's 624-word state next to it, and a twist inside a divergent draw would be paid by the
// whole wave for one lane.  So the env's stream is read from memory: next to the raw state (`mts`, the latest generation of 624 words)
// the engine keeps the TEMPERED outputs of the latest AND the previous generation (`mtt[env][2][624]`); a lane's position is relative to
// the latest generation's start, in [-624, 624].  Twists happen where the wave is converged (k_pregen_lane, top of an attempt): every lane
// whose position is >= 0 has its env twisted by the whole wave (the 64 lanes share the 624 words), after which it has at least 624 draws
// in front of it.  An attempt that needs more (a crowded room's placement loop) ends up at position 624 inside a draw and twists by
// itself, one lane, sequentially -- correct and slow, and rare.  M (the memory / RNG policy) hides all of it: the host build draws from
// a plain MT19937 state.
#pragma once
#include "bbai_types.hpp"
#include "bbai_gen.hpp"

#ifndef BBAI_GENL_TRACE
#define BBAI_GENL_TRACE(m, tag, val)      // (tools/genl_check.hip defines it: where device and host part ways)
#endif

namespace bbai {

// The families this generator covers: every LevelGen parameterisation and the hand-written single-instruction levels without the
// lock-first prologue.  (Bonus scripts and lock-first levels stay on the lane-group generator.)
BB_HD bool lane_gen_ok(const LevelCfg& c) { return c.kind == K_LEVELGEN || (c.kind == K_GOTO && !c.lock); }

struct LaneLayout { int obj, row, door, prog, fifo, words; };     // word offsets of a lane's regions
constexpr int LANE_FIFO_WORDS = 16;                        // the draw FIFO (LaneRng below)
constexpr int LANE_PROG_WORDS = (int)(sizeof(Prog) / 4);    // 28; the flood rows of the reachability test alias them (H <= 25)
BB_HD LaneLayout lane_layout(const LevelCfg& c) {
    LaneLayout L;
    const bool small = c.W * c.H <= 64;
    L.obj = 0;
    L.row = c.maxo;
    L.door = L.row + (small ? 0 : c.H);
    L.prog = L.door + c.num_rows * c.num_cols;
    L.fifo = L.prog + LANE_PROG_WORDS;
    L.words = L.fifo + LANE_FIFO_WORDS;
    return L;
}
static_assert(MAX_W <= LANE_PROG_WORDS, "flood rows alias the program words");

// inverse of mt_temper (canonical state of an env whose position lies in the previous generation: k_mt_canon)
BB_HD uint32_t mt_untemper(uint32_t y) {
    y ^= y >> 18;
    y ^= (y << 15) & 0xefc60000u;
    uint32_t t = y;                              // y ^= (y << 7) & 0x9d2c5680, inverted 7 bits at a time
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    t = y ^ ((t << 7) & 0x9d2c5680u);
    y = t;
    t = y ^ (y >> 11);                           // y ^= y >> 11, inverted 11 bits at a time
    y = y ^ (t >> 11);
    return y;
}

// A lane's view of its env's MT19937 stream (device: LaneMemDev in bbai_genlane.hip; host: the emulation in tests/hostsim): the raw state
// of the latest generation, the tempered outputs of the latest (half `par`) and the previous generation, and a FIFO of LANE_FIFO outputs
// in the lane's own words.  A draw out of memory is a dependent round trip of a microsecond on a chain that is nothing but draw -> test ->
// branch (the first device build: 3 000 cycles per draw); a draw out of the FIFO is an LDS read.  The FIFO is topped up where the wave
// is converged -- GenL calls topup() at the head of every loop that draws -- with all of a lane's loads in flight together; a draw that
// finds it empty all the same (a long placement loop) refills it by itself.  `p` = position, relative to the latest generation's start, of
// the next word to FETCH; the consumer is avail() words behind.
BB_COLD_FN void lane_twist_alone(uint32_t* mt, uint32_t* tt) {
    // A fetch past the end of the latest generation INSIDE an attempt (it consumed more than the 624 draws the wave guarantees at its top):
    // this one lane regenerates its env's state by itself -- the textbook in-place recurrence, then the tempered copy into the other half.
    for (int k = 0; k < MT_N; ++k) {
        int m = k + 397; if (m >= MT_N) m -= MT_N;
        mt[k] = mt[m] ^ mt_mix(mt[k], mt[k + 1 < MT_N ? k + 1 : 0]);
    }
    for (int k = 0; k < MT_N; ++k) tt[k] = mt_temper(mt[k]);
}
constexpr int LANE_FIFO = LANE_FIFO_WORDS;   // words (a power of two)
constexpr int LANE_FIFO_LOW = 8;     // topup() fills when fewer are left
struct LaneFill { int p, par; uint32_t wr; };
// Fill the FIFO as far as the generation in hand reaches (never across a twist, never across the seam between the two halves: the next
// call continues).  Only when it is EMPTY and the latest generation is used up does the lane twist, alone.  Out of line: one copy per
// kernel, called from the draw's slow path and from topup().
template <class Mem>
BB_COLD LaneFill lane_fill(Mem m, uint32_t* mts_env, uint32_t* mtt_env, int p, int par, uint32_t rd, uint32_t wr, int fifo0) {
    if (p >= MT_N && rd == wr) {
        lane_twist_alone(mts_env, mtt_env + (par ^ 1) * MT_N);
        par ^= 1;
        p = 0;
    }
    const int room = LANE_FIFO - (int)(wr - rd);
    const int left = p < 0 ? -p : MT_N - p;
    const int n = room < left ? room : left;
    const uint32_t* src = mtt_env + (p < 0 ? (par ^ 1) * MT_N + MT_N + p : par * MT_N + p);
    uint32_t v[LANE_FIFO];
#pragma unroll
    for (int j = 0; j < LANE_FIFO; ++j) v[j] = j < n ? src[j] : 0u;          // (all loads in flight before the first store)
#pragma unroll
    for (int j = 0; j < LANE_FIFO; ++j) if (j < n) m.st(fifo0 + (int)((wr + (uint32_t)j) & (LANE_FIFO - 1)), v[j]);
    LaneFill r; r.p = p + n; r.par = par; r.wr = wr + (uint32_t)n;
    return r;
}
template <class Mem>
struct LaneRng {
    uint32_t* mts_env;
    uint32_t* mtt_env;               // [2][MT_N]
    int p, par;
    uint32_t rd, wr;
    int fifo0;                       // the FIFO's first word in the lane's words (lane_layout)
    BB_HD Mem& self() { return *static_cast<Mem*>(this); }
    BB_HD int avail() const { return (int)(wr - rd); }
    BB_HD int position() const { return p - avail(); }          // the consumer's: what goes back to mtis
    BB_HD void start(int pos, int par_) { p = pos; par = par_; rd = wr = 0; }
    BB_HD void twisted() { p -= MT_N; par ^= 1; }               // the wave twisted this lane's env (k_pregen_lane): same words, one generation back
    BB_HD void refill() {
        const LaneFill r = lane_fill(self(), mts_env, mtt_env, p, par, rd, wr, fifo0);
        p = r.p; par = r.par; wr = r.wr;
    }
    BB_HD uint32_t next_u32() {
        if (rd == wr) refill();
        const uint32_t y = self().ld(fifo0 + (int)(rd & (LANE_FIFO - 1)));
        ++rd;
        return y;
    }
    BB_HD bool low() const { return avail() < LANE_FIFO_LOW; }
    // RandomState.randint's masked rejection: the first of the next outputs with (output & mask) <= rng.  As a loop over next_u32() the wave
    // runs it for its unluckiest lane (64 lanes, a quarter of the colour draws rejected: four trips where a lane needs 1.3), each trip an LDS
    // round trip.  Here the next FOUR outputs are read at once and the first acceptable one is picked -- straight-line code; the loop is only
    // left for a lane whose FIFO holds fewer than four words or whose four candidates were all rejected (a range of 3, 5 or 6 values: 0.4 %).
    BB_HD uint32_t draw_masked(uint32_t mask, uint32_t rng) {
        if (mask != rng && avail() >= 4) {       // (a power-of-two range accepts its first output: one plain draw)
            const uint32_t v0 = self().ld(fifo0 + (int)(rd & (LANE_FIFO - 1))) & mask;
            const uint32_t v1 = self().ld(fifo0 + (int)((rd + 1) & (LANE_FIFO - 1))) & mask;
            const uint32_t v2 = self().ld(fifo0 + (int)((rd + 2) & (LANE_FIFO - 1))) & mask;
            const uint32_t v3 = self().ld(fifo0 + (int)((rd + 3) & (LANE_FIFO - 1))) & mask;
            if (v0 <= rng) { rd += 1; return v0; }
            if (v1 <= rng) { rd += 2; return v1; }
            if (v2 <= rng) { rd += 3; return v2; }
            if (v3 <= rng) { rd += 4; return v3; }
            rd += 4;
        }
        uint32_t v;
        do { v = next_u32() & mask; } while (v > rng);
        return v;
    }
};

// The record template of a level kind: what every level of the kind has in common -- wall margin, wall lines, empty cells, id plane
// (1 on walls), empty tables (contents NONE8), zeroed program -- rec_bytes, followed by the 64-byte C plane of a small single room
// (bbai_types.hpp; walls beyond W x H).  Built on the host (bbai_create / the host tests), read by every lane at write-out.
inline int lane_template_bytes(const LevelCfg& c) { return c.rec_bytes + CPL_PLANE; }
inline void lane_build_template(const LevelCfg& c, uint8_t* t) {
    for (int k = 0; k < lane_template_bytes(c); ++k) t[k] = 0;
    const int S = c.room_size;
    for (int k = 0; k < c.ES * c.EH; ++k) t[k] = (uint8_t)E_WALL;
    for (int y = 0; y < c.H; ++y)
        for (int x = 0; x < c.W; ++x) {
            const bool wall = x % (S - 1) == 0 || y % (S - 1) == 0;
            t[e_index(c, x, y)] = wall ? (uint8_t)E_WALL : (uint8_t)E_EMPTY;
            t[c.off_I + i_index(c, x, y)] = wall ? 1 : 0;
        }
    for (int o = 0; o < c.maxo; ++o) t[c.off_cont + o] = (uint8_t)NONE8;
    uint8_t* p = t + c.rec_bytes;
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) p[8 * y + x] = (x < c.W && y < c.H) ? t[e_index(c, x, y)] : (uint8_t)E_WALL;
}

// M contract: ld(word) / st(word, value) on the lane's word array (lane_layout), next_u32() = the env's next MT19937 output, topup() =
// "the wave is converged here and about to draw" (a hint: the device tops the lanes' draw FIFOs up, the host does nothing).
template <class M>
struct GenL {
    M& m;
    const LevelCfg& cfg;
    const LaneLayout L;
    uint64_t occ;            // grids of <= 64 cells: bit y * W + x = an object (or a door) stands on the cell
    uint64_t wallb;          //   ... the cells RoomGrid._gen_grid makes walls
    uint32_t wrow_full, wrow_in;     // larger grids: a wall-line row / any other row as bit rows
    uint32_t seen;           // bit (type - T_KEY) * 6 + colour: a key / ball / box of that look exists (add_distractors(all_unique))
    bool small;
    int nobj;
    int ax, ay, adir;
    bool agent_set;
    int locked_room, last_locked;
    int S, rows, cols;
    uint64_t doors;          // bit (16 k + r): room r has a door on side k (0 right, 1 down, 2 left, 3 up)
    uint32_t locked_mask;
    uint32_t inv_cols, inv_s1;

    BB_HD GenL(M& mem, const LevelCfg& cf, int last_locked_)
        : m(mem), cfg(cf), L(lane_layout(cf)), occ(0), wallb(0), wrow_full(0), wrow_in(0), seen(0), small(cf.W * cf.H <= 64), nobj(0),
          ax(0), ay(0), adir(0), agent_set(false), locked_room(-1), last_locked(last_locked_), S(cf.room_size), rows(cf.num_rows),
          cols(cf.num_cols), doors(0), locked_mask(0), inv_cols(65536u / (uint32_t)cf.num_cols + 1u),
          inv_s1(65536u / (uint32_t)(cf.room_size - 1) + 1u) {
        uint32_t in_row = 0;
        for (int x = 0; x < cfg.W; x += S - 1) in_row |= 1u << x;
        wrow_in = in_row;
        wrow_full = (1u << cfg.W) - 1u;
        if (small) {
            uint64_t b = 0;
            for (int y = 0; y < cfg.H; ++y) b |= (uint64_t)(on_line(y) ? wrow_full : wrow_in) << (y * cfg.W);
            wallb = b;
        }
    }

    BB_HD int div_cols(int v) const { return (int)(((uint32_t)v * inv_cols) >> 16); }     // v < 64
    BB_HD int div_s1(int v) const { return (int)(((uint32_t)v * inv_s1) >> 16); }         // v < 64
    BB_HD bool on_line(int v) const { return v - div_s1(v) * (S - 1) == 0; }

    // ---------------- draws (numpy legacy RandomState) ----------------
    BB_HD uint32_t next_u32() { return m.next_u32(); }
    BB_HD int rand_int(int lo, int hi) {
        const uint32_t rng = (uint32_t)(hi - lo - 1);
        if (rng == 0) return lo;
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz(rng);
        return lo + (int)m.draw_masked(mask, rng);
    }
    BB_HD bool rand_bool() { return rand_int(0, 2) == 0; }
    BB_HD double rand_float01() {
        const uint32_t a = next_u32() >> 5, b = next_u32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
    BB_HD int rand_color() { return color_name_to_idx(rand_int(0, 6)); }

    // ---------------- the grid as bits ----------------
    BB_HD uint32_t wall_row(int y) const { return on_line(y) ? wrow_full : wrow_in; }
    BB_HD bool obj_at(int x, int y) const { return small ? (occ >> (y * cfg.W + x) & 1ull) != 0 : (m.ld(L.row + y) >> x & 1u) != 0; }
    BB_HD bool occupied(int x, int y) const {
        if (small) return ((occ | wallb) >> (y * cfg.W + x) & 1ull) != 0;
        return ((m.ld(L.row + y) | wall_row(y)) >> x & 1u) != 0;
    }
    BB_HD void mark(int x, int y) {
        if (small) occ |= 1ull << (y * cfg.W + x);
        else m.st(L.row + y, m.ld(L.row + y) | (1u << x));
    }
    BB_HD void note_obj(int e) { if (e_type(e) >= T_KEY) seen |= 1u << ((e_type(e) - T_KEY) * 6 + e_color(e)); }
    BB_HD int room_of(int x, int y) const { return div_s1(y) * cols + div_s1(x); }
    BB_HD void room_ij(int r, int& i, int& j) const { j = div_cols(r); i = r - j * cols; }
    BB_HD bool has_neighbor(int r, int k) const {
        int i, j; room_ij(r, i, j);
        return k == 0 ? i < cols - 1 : k == 1 ? j < rows - 1 : k == 2 ? i > 0 : j > 0;
    }
    BB_HD bool has_door(int r, int k) const { return doors >> (16 * k + r) & 1; }
    BB_HD int neighbor(int r, int k) const { return k == 0 ? r + 1 : k == 1 ? r + cols : k == 2 ? r - 1 : r - cols; }
    // object word: appearance | x << 8 | y << 16 | contents << 24
    BB_HD uint32_t obj(int o) const { return m.ld(L.obj + o); }
    BB_HD static int o_app(uint32_t w) { return (int)(w & 0xFFu); }
    BB_HD static int o_x(uint32_t w) { return (int)(w >> 8 & 0xFFu); }
    BB_HD static int o_y(uint32_t w) { return (int)(w >> 16 & 0xFFu); }
    // the door slot of room r on side k (RoomGrid._gen_grid: the right / lower neighbour shares it)
    BB_HD void door_xy(int r, int k, int& x, int& y) const {
        const int src = k < 2 ? r : k == 2 ? r - 1 : r - cols;
        const uint32_t d = m.ld(L.door + src);
        int i, j; room_ij(src, i, j);
        if ((k & 1) == 0) { x = i * (S - 1) + S - 1; y = (int)(d & 0xFFu); }
        else { x = (int)(d >> 8 & 0xFFu); y = j * (S - 1) + S - 1; }
    }

    BB_HD void build_rooms() {
        doors = 0; locked_mask = 0; seen = 0; occ = 0;
        if (!small) for (int y = 0; y < cfg.H; ++y) m.st(L.row + y, 0u);
        for (int j = 0; j < rows; ++j)
            for (int i = 0; i < cols; ++i) {
                const int tx = i * (S - 1), ty = j * (S - 1);
                uint32_t d = 0;
                if (i < cols - 1) d |= (uint32_t)rand_int(ty + 1, ty + S - 1);
                if (j < rows - 1) d |= (uint32_t)rand_int(tx + 1, tx + S - 1) << 8;
                m.st(L.door + j * cols + i, d);
            }
        ax = (cols / 2) * (S - 1) + S / 2;
        ay = (rows / 2) * (S - 1) + S / 2;
        adir = 0;
        agent_set = true;
        nobj = 0;
        locked_room = -1;
    }

    // MiniGridEnv.place_obj restricted to a room rectangle, max_tries = 1000
    BB_HD bool place_pos(int r, bool reject_next, int& ox, int& oy) {
        int ri, rj; room_ij(r, ri, rj);
        const int tx = ri * (S - 1), ty = rj * (S - 1);
        const int xh = tx + S < cfg.W ? tx + S : cfg.W, yh = ty + S < cfg.H ? ty + S : cfg.H;
        int tries = 0;
        for (;;) {
            if (tries > 1000) return false;
            ++tries;
            if ((tries & 3) == 0) m.topup();
            const int x = rand_int(tx, xh);
            const int y = rand_int(ty, yh);
            if (occupied(x, y)) continue;
            if (agent_set && x == ax && y == ay) continue;
            if (reject_next) {
                const int d = (x > ax ? x - ax : ax - x) + (y > ay ? y - ay : ay - y);
                if (d < 2) continue;
            }
            ox = x; oy = y;
            return true;
        }
    }
    BB_HD int add_object(int r, int type, int color) {
        int x, y;
        if (!place_pos(r, true, x, y)) return -1;
        if (nobj >= cfg.maxo) return -1;
        const int id = nobj++;
        const int e = e_make(type, color, 0);
        m.st(L.obj + id, (uint32_t)e | (uint32_t)x << 8 | (uint32_t)y << 16 | (uint32_t)NONE8 << 24);
        mark(x, y);
        note_obj(e);
        return id;
    }
    BB_HD int add_door(int r, int k, int color, bool is_locked) {
        if (nobj >= cfg.maxo) return -1;
        const int id = nobj++;
        int x, y; door_xy(r, k, x, y);
        const int e = e_make(T_DOOR, color, is_locked ? S_LOCKED : S_CLOSED);
        locked_mask = (locked_mask & ~(1u << r)) | ((is_locked ? 1u : 0u) << r);
        m.st(L.obj + id, (uint32_t)e | (uint32_t)x << 8 | (uint32_t)y << 16 | (uint32_t)NONE8 << 24);
        mark(x, y);
        doors |= 1ull << (16 * k + r);
        doors |= 1ull << (16 * ((k + 2) & 3) + neighbor(r, k));
        return id;
    }
    // RoomGrid.place_agent: a room, then poses (place_obj + a direction) until the cell in front is empty or a wall.  The reference nests the
    // placement's rejection loop inside the pose loop; here ONE loop makes one placement try per trip -- the lanes of a wave are at different
    // poses and tries, and a nested loop would run every inner loop for its unluckiest lane.  Same draws in the same order for every lane.
    BB_HD bool place_agent(int room = -1) {
        int r = room;
        if (r < 0) {
            const int i = rand_int(0, cols);
            const int j = rand_int(0, rows);
            r = j * cols + i;
        }
        int ri, rj; room_ij(r, ri, rj);
        const int tx = ri * (S - 1), ty = rj * (S - 1);
        const int xh = tx + S < cfg.W ? tx + S : cfg.W, yh = ty + S < cfg.H ? ty + S : cfg.H;
        int pose_tries = 0, tries = 0;
        agent_set = false;
        for (;;) {
            m.topup();
            if (tries > 1000) return false;      // place_obj's RecursionError
            ++tries;
            const int x = rand_int(tx, xh);
            const int y = rand_int(ty, yh);
            if (occupied(x, y)) continue;
            ax = x; ay = y; agent_set = true;
            adir = rand_int(0, 4);
            const int fx = ax + (adir == 0) - (adir == 2), fy = ay + (adir == 1) - (adir == 3);
            if (!obj_at(fx, fy)) return true;            // the front cell is empty or a wall
            if (++pose_tries > 1000) return false;       // (termination guard: bbai_gen.hpp place_agent)
            agent_set = false;
            tries = 0;
        }
    }
    BB_HD bool connect_all() {
        const int start = room_of(ax, ay);
        const int nrooms = rows * cols;
        int itrs = 0;
        uint32_t reach = 1u << start;
        bool grew = true;
        for (;;) {
            if (itrs > 5000) return false;
            ++itrs;
            m.topup();
            if (grew) {
                const uint32_t d0 = (uint32_t)doors & 0x1FFu, d1 = (uint32_t)(doors >> 16) & 0x1FFu;
                const uint32_t d2 = (uint32_t)(doors >> 32) & 0x1FFu, d3 = (uint32_t)(doors >> 48) & 0x1FFu;
                for (int pass = 0; pass < nrooms; ++pass) {
                    const uint32_t nr = reach | ((reach & d0) << 1) | ((reach & d1) << cols) | ((reach & d2) >> 1) | ((reach & d3) >> cols);
                    if (nr == reach) break;
                    reach = nr;
                }
                grew = false;
            }
            if (__builtin_popcount(reach) == nrooms) return true;
            const int i = rand_int(0, cols);
            const int j = rand_int(0, rows);
            const int k = rand_int(0, 4);
            const int r = j * cols + i;
            if (!has_neighbor(r, k) || has_door(r, k)) continue;
            if ((locked_mask >> r & 1) || (locked_mask >> neighbor(r, k) & 1)) continue;
            const int color = rand_color();
            if (add_door(r, k, color, false) < 0) return false;
            grew = true;
        }
    }
    // RoomGrid.add_distractors(i=None, j=None): look, room, then place_obj's rejection loop -- as ONE loop, one placement try per trip
    // (see place_agent).
    BB_HD bool add_distractors(int num, bool all_unique) {
        int count = 0, tries = 0, color = 0, type = 0, tx = 0, ty = 0, xh = 0, yh = 0;
        bool fresh = true;
        while (count < num) {
            m.topup();
            if (fresh) {
                color = rand_color();
                type = T_KEY + rand_int(0, 3);
                if (all_unique && (seen >> ((type - T_KEY) * 6 + color) & 1u)) continue;
                const int ri = rand_int(0, cols);
                const int rj = rand_int(0, rows);
                tx = ri * (S - 1); ty = rj * (S - 1);
                xh = tx + S < cfg.W ? tx + S : cfg.W; yh = ty + S < cfg.H ? ty + S : cfg.H;
                tries = 0;
                fresh = false;
            }
            if (tries > 1000) return false;
            ++tries;
            const int x = rand_int(tx, xh);
            const int y = rand_int(ty, yh);
            if (occupied(x, y)) continue;
            if (agent_set && x == ax && y == ay) continue;
            if ((x > ax ? x - ax : ax - x) + (y > ay ? y - ay : ay - y) < 2) continue;
            if (nobj >= cfg.maxo) return false;
            const int id = nobj++;
            const int e = e_make(type, color, 0);
            m.st(L.obj + id, (uint32_t)e | (uint32_t)x << 8 | (uint32_t)y << 16 | (uint32_t)NONE8 << 24);
            mark(x, y);
            note_obj(e);
            ++count;
            fresh = true;
        }
        return true;
    }

    // check_objs_reachable.  passable = empty cells and doors = NOT (wall XOR object bit): a door is an object bit on a wall line, every
    // other object stands off the lines.  The flood is the 4-connected component of the agent's cell; an object is reached when its cell
    // lies in the flood's 4-neighbourhood.
    BB_HD bool objs_reachable() {
        const int W = cfg.W, H = cfg.H;
        if (small) {
            uint64_t col0 = 0;
            for (int y = 0; y < H; ++y) col0 |= 1ull << (y * W);
            const uint64_t board = W * H == 64 ? ~0ull : (1ull << (W * H)) - 1ull;
            const uint64_t not0 = ~col0, notL = ~(col0 << (W - 1));
            const uint64_t pass = ~(wallb ^ occ) & board;
            uint64_t f = 1ull << (ay * W + ax);
            for (;;) {
                const uint64_t g = (f | ((f << 1) & not0) | ((f >> 1) & notL) | (f << W) | (f >> W)) & pass;
                if ((g | f) == f) break;
                f |= g;
            }
            const uint64_t near = f | ((f << 1) & not0) | ((f >> 1) & notL) | (f << W) | (f >> W);
            return (occ & ~near) == 0;
        }
        const int FL = L.prog;                   // flood rows (the program is generated after this test)
        const uint32_t rowmask = (1u << W) - 1u;
        for (int y = 0; y < H; ++y) m.st(FL + y, y == ay ? 1u << ax : 0u);
        for (;;) {
            bool changed = false;
            uint32_t carry = 0;
            for (int y = 0; y < H; ++y) {        // sweep down: a row takes what the (already updated) row above holds
                const uint32_t p = ~(wall_row(y) ^ m.ld(L.row + y)) & rowmask;
                const uint32_t f = m.ld(FL + y);
                const uint32_t g = fill_row((f | carry) & p, p);
                if (g != f) { m.st(FL + y, g); changed = true; }
                carry = g;
            }
            carry = 0;
            for (int y = H - 1; y >= 0; --y) {   // sweep up
                const uint32_t p = ~(wall_row(y) ^ m.ld(L.row + y)) & rowmask;
                const uint32_t f = m.ld(FL + y);
                const uint32_t g = fill_row((f | carry) & p, p);
                if (g != f) { m.st(FL + y, g); changed = true; }
                carry = g;
            }
            if (!changed) break;
        }
        uint32_t prev = 0, cur = m.ld(FL);
        bool ok = true;
        for (int y = 0; y < H; ++y) {
            const uint32_t nxt = y + 1 < H ? m.ld(FL + y + 1) : 0u;
            const uint32_t near = cur | (cur << 1) | (cur >> 1) | prev | nxt;
            if (m.ld(L.row + y) & ~near) ok = false;
            prev = cur; cur = nxt;
        }
        return ok;
    }
    // g, a subset of p, spread along the row through p in both directions (Kogge-Stone occluded fill, rows of <= 32 cells)
    BB_HD static uint32_t fill_row(uint32_t g, uint32_t p) {
        uint32_t a = g, q = p;
        a |= q & (a << 1); q &= q << 1;
        a |= q & (a << 2); q &= q << 2;
        a |= q & (a << 4); q &= q << 4;
        a |= q & (a << 8); q &= q << 8;
        a |= q & (a << 16);
        uint32_t b = g; q = p;
        b |= q & (b >> 1); q &= q >> 1;
        b |= q & (b >> 2); q &= q >> 2;
        b |= q & (b >> 4); q &= q >> 4;
        b |= q & (b >> 8); q &= q >> 8;
        b |= q & (b >> 16);
        return a | b;
    }

    // ObjDesc.find_matching_objs(env, use_location=True) over the object words.  type 0 = any type, color 7 = any colour.
    BB_HD uint64_t find_matching(int type, int color, int loc) const {
        const int r = room_of(ax, ay);
        int ri, rj; room_ij(r, ri, rj);
        const int tx = ri * (S - 1), ty = rj * (S - 1);
        const int d1x = (adir == 0) - (adir == 2), d1y = (adir == 1) - (adir == 3);
        const int d2x = -d1y, d2y = d1x;
        uint64_t mm = 0;
        for (int o = 0; o < nobj; ++o) {
            const uint32_t w = obj(o);
            const int e = o_app(w);
            bool ok = (type == 0 || e_type(e) == type) && (color == 7 || e_color(e) == color);
            if (loc != LOC_NONE) {
                const int x = o_x(w), y = o_y(w);
                const bool inroom = !(x < tx || y < ty || x >= tx + S || y >= ty + S);
                const int vx = x - ax, vy = y - ay;
                const int p2 = vx * d2x + vy * d2y, p1 = vx * d1x + vy * d1y;
                const bool side = loc == LOC_LEFT ? p2 < 0 : loc == LOC_RIGHT ? p2 > 0 : loc == LOC_FRONT ? p1 > 0 : p1 < 0;
                ok = ok && inroom && side;
            }
            mm |= ok ? 1ull << o : 0ull;
        }
        return mm;
    }

    // ---------------- the program (Prog layout, 28 words) ----------------
    BB_HD void p_set(int leaf, int slot, uint64_t v) {
        m.st(L.prog + 2 * (2 * leaf + slot), (uint32_t)v);
        m.st(L.prog + 2 * (2 * leaf + slot) + 1, (uint32_t)(v >> 32));
    }
    BB_HD uint64_t p_get(int leaf, int slot) const {
        return (uint64_t)m.ld(L.prog + 2 * (2 * leaf + slot)) | (uint64_t)m.ld(L.prog + 2 * (2 * leaf + slot) + 1) << 32;
    }
    BB_HD void p_desc(int leaf, int slot, int type, int color, int loc, int count) {
        m.st(L.prog + 16 + 2 * leaf + slot, (uint32_t)type | (uint32_t)color << 8 | (uint32_t)loc << 16 | (uint32_t)count << 24);
    }
    BB_HD uint32_t p_desc_get(int leaf, int slot) const { return m.ld(L.prog + 16 + 2 * leaf + slot); }
    BB_HD void p_kind(int leaf, int kind) {
        const uint32_t w = m.ld(L.prog + 24);
        m.st(L.prog + 24, (w & ~(0xFFu << (8 * leaf))) | (uint32_t)kind << (8 * leaf));
    }
    BB_HD int p_kind_get(int leaf) const { return (int)(m.ld(L.prog + 24) >> (8 * leaf) & 0xFFu); }
    BB_HD void p_shape(int root, int n_a, int n_b) { m.st(L.prog + 25, (uint32_t)root | (uint32_t)n_a << 8 | (uint32_t)n_b << 16); }   // strict = 0
    BB_HD void clear_prog() {
        for (int k = 0; k < LANE_PROG_WORDS; ++k) m.st(L.prog + k, 0u);
        m.st(L.prog + 26, (uint32_t)NONE8);      // start_carry
    }

    // LevelGen.rand_obj.  types_mode: 0 = OBJ_TYPES, 1 = OBJ_TYPES_NOT_DOOR, 2 = ['door'].
    BB_HD bool rand_obj(int types_mode, int leaf, int slot) {
        int tries = 0;
        for (;;) {
            if (tries > 100) return false;
            ++tries;
            m.topup();
            const int cv = rand_int(0, 7);
            const int color = cv == 0 ? 7 : color_name_to_idx(cv - 1);
            const int type = types_mode == 0 ? T_BOX - rand_int(0, 4) : types_mode == 1 ? T_BOX - rand_int(0, 3) : T_DOOR;
            int loc = LOC_NONE;
            if (cfg.locations && rand_bool()) loc = 1 + rand_int(0, 4);
            const uint64_t mm = find_matching(type, color, loc);
            BBAI_GENL_TRACE(m, 40, cv | type << 8 | loc << 16);
            BBAI_GENL_TRACE(m, 41, (int)(uint32_t)mm);
            if (mm == 0) continue;
            if (!cfg.implicit_unlock && last_locked >= 0) {
                int li, lj; room_ij(last_locked, li, lj);
                const int tx = li * (S - 1), ty = lj * (S - 1);
                bool any_out = false;
                for (int o = 0; o < nobj; ++o)
                    if (mm >> o & 1) {
                        const uint32_t w = obj(o);
                        const int x = o_x(w), y = o_y(w);
                        if (x < tx || y < ty || x >= tx + S || y >= ty + S) any_out = true;
                    }
                if (!any_out) continue;
            }
            p_set(leaf, slot, mm);
            p_desc(leaf, slot, type, color, loc, __builtin_popcountll(mm));
            BBAI_GENL_TRACE(m, 10 + leaf * 2 + slot, (int)(uint32_t)mm);
            return true;
        }
    }
    BB_HD bool rand_action(int leaf) {
        const int a = cfg.action_kinds[rand_int(0, cfg.n_action_kinds)];
        BBAI_GENL_TRACE(m, 20 + leaf, a);
        if (a == AK_GOTO) { p_kind(leaf, L_GOTO); return rand_obj(0, leaf, 0); }
        if (a == AK_PICKUP) { p_kind(leaf, L_PICKUP); return rand_obj(1, leaf, 0); }
        if (a == AK_OPEN) { p_kind(leaf, L_OPEN); return rand_obj(2, leaf, 0); }
        p_kind(leaf, L_PUTNEXT);
        return rand_obj(1, leaf, 0) && rand_obj(0, leaf, 1);
    }
    BB_HD bool rand_instr() {
        clear_prog();
        const int kind = cfg.instr_kinds[rand_int(0, cfg.n_instr_kinds)];
        if (kind == IK_ACTION) {
            p_shape(R_ACTION, 1, 0);
            return rand_action(0);
        }
        if (kind == IK_AND) {
            p_shape(R_AND, 2, 0);
            return rand_action(0) && rand_action(1);
        }
        int n_a = 0, n_b = 0;
        for (int side = 0; side < 2; ++side) {
            const int k2 = rand_int(0, 2);
            const int n = k2 == 0 ? 1 : 2;
            if (side == 0) n_a = n; else n_b = n;
            for (int q = 0; q < n; ++q)
                if (!rand_action(side * 2 + q)) return false;
        }
        p_shape(rand_int(0, 2) == 0 ? R_BEFORE : R_AFTER, n_a, n_b);
        return true;
    }

    // validate_instrs; false => RejectSampling
    BB_HD bool validate() {
        uint32_t locked_colors = 0;
        const bool unb = cfg.kind == K_LEVELGEN && cfg.unblocking;
        if (unb)
            for (int o = 0; o < nobj; ++o) {
                const int e = o_app(obj(o));
                if (e_type(e) == T_DOOR && e_state(e) == S_LOCKED) locked_colors |= 1u << e_color(e);
            }
        for (int leaf = 0; leaf < 4; ++leaf) {
            const int k = p_kind_get(leaf);
            if (k == L_NONE) continue;
            if (k == L_PUTNEXT) {
                const uint64_t mv = p_get(leaf, 0), fx = p_get(leaf, 1);
                if (mv & fx) return false;
                // an object of the move set next to one of the fixed set (levelgen.py:113-124: the four neighbours of its cell)
                for (int a = 0; a < nobj; ++a)
                    if (mv >> a & 1) {
                        const uint32_t wa = obj(a);
                        for (int b = 0; b < nobj; ++b)
                            if (fx >> b & 1) {
                                const uint32_t wb = obj(b);
                                const int dx = o_x(wa) - o_x(wb), dy = o_y(wa) - o_y(wb);
                                if ((dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy) == 1) return false;
                            }
                    }
            }
            if (unb)
                for (int s = 0; s < 2; ++s) {
                    const uint32_t d = p_desc_get(leaf, s);
                    const int dtype = (int)(d & 0xFFu), dcolor = (int)(d >> 8 & 0xFFu);
                    if ((s == 0 || k == L_PUTNEXT) && dtype == T_KEY && dcolor != 7 && (locked_colors >> dcolor & 1)) return false;
                }
        }
        return true;
    }

    // LevelGen.gen_mission
    BB_HD bool mission_levelgen() {
        if (rand_float01() < cfg.locked_room_prob) {
            int door_color;
            for (;;) {
                const int i = rand_int(0, cols);
                const int j = rand_int(0, rows);
                const int k = rand_int(0, 4);
                locked_room = last_locked = j * cols + i;
                if (!has_neighbor(locked_room, k)) continue;
                door_color = rand_color();
                if (add_door(locked_room, k, door_color, true) < 0) return false;
                break;
            }
            for (;;) {
                const int i = rand_int(0, cols);
                const int j = rand_int(0, rows);
                if (j * cols + i == locked_room) continue;
                if (add_object(j * cols + i, T_KEY, door_color) < 0) return false;
                break;
            }
        }
        BBAI_GENL_TRACE(m, 1, nobj);
        if (!connect_all()) return false;
        BBAI_GENL_TRACE(m, 2, nobj);
        if (!add_distractors(cfg.num_dists, false)) return false;
        BBAI_GENL_TRACE(m, 3, nobj);
        for (;;) {
            if (!place_agent()) return false;
            if (room_of(ax, ay) == locked_room) continue;
            break;
        }
        BBAI_GENL_TRACE(m, 4, ax | ay << 8 | adir << 16);
        if (!cfg.unblocking && !objs_reachable()) return false;
        const bool ri = rand_instr();
        BBAI_GENL_TRACE(m, 5, ri);
        return ri;
    }

    BB_HD void set_desc(int leaf, int slot, int o) {
        const int e = o_app(obj(o));
        const int type = e_type(e), color = e_color(e);
        const uint64_t mm = find_matching(type, color, LOC_NONE);
        p_set(leaf, slot, mm);
        const int cnt = __builtin_popcountll(mm);
        p_desc(leaf, slot, type, color, LOC_NONE, cnt > 255 ? 255 : cnt);
    }

    // the hand-written single-instruction levels without the lock-first prologue (bbai_gen.hpp mission_goto, second arm)
    BB_HD bool mission_goto() {
        int target = -1, target2 = -1;
        if (!place_agent()) return false;
        if (cfg.redball) {
            target = add_object(0, T_BALL, C_RED);
            if (target < 0) return false;
        }
        if (cfg.connect && !connect_all()) return false;
        const int first = nobj;
        const int ndist = cfg.num_dists;
        if (!add_distractors(ndist, cfg.all_unique != 0)) return false;
        if (cfg.grey_dists)
            for (int o = first; o < nobj; ++o) {
                const uint32_t w = obj(o);
                m.st(L.obj + o, (w & ~0xFFu) | (uint32_t)e_make(e_type(o_app(w)), C_GREY, 0));
            }
        if (cfg.check_reach == 1 && !objs_reachable()) return false;
        if (cfg.check_reach == 2 && objs_reachable()) return false;
        if (cfg.target == TG_DIST) {
            target = first + rand_int(0, ndist);
        } else if (cfg.target == TG_TWO_DISTS) {
            const int a = rand_int(0, ndist);
            int b = rand_int(0, ndist - 1);
            if (b >= a) ++b;
            target = first + a; target2 = first + b;
        } else if (cfg.target == TG_DOOR) {
            int n = 0;
            for (int r = 0; r < rows * cols; ++r)
                for (int k = 0; k < 4; ++k) n += has_door(r, k) ? 1 : 0;
            if (n == 0) return false;
            int pick = rand_int(0, n);
            for (int i = 0; i < cols && target < 0; ++i)
                for (int j = 0; j < rows && target < 0; ++j)
                    for (int k = 0; k < 4; ++k)
                        if (has_door(j * cols + i, k)) {
                            if (pick-- == 0) {
                                int dx, dy; door_xy(j * cols + i, k, dx, dy);
                                for (int o = 0; o < nobj; ++o) {
                                    const uint32_t w = obj(o);
                                    if (o_x(w) == dx && o_y(w) == dy) target = o;
                                }
                                break;
                            }
                        }
        }
        clear_prog();
        p_shape(R_ACTION, 1, 0);
        p_kind(0, cfg.instr);
        set_desc(0, 0, target);
        if (cfg.instr == L_PUTNEXT) set_desc(0, 1, target2);
        return true;
    }

    // ONE pass of the rejection loop of RoomGridLevel._gen_grid
    template <int KIND>
    BB_HD bool attempt() {
        m.topup();
        build_rooms();
        bool ok;
        if constexpr (KIND == K_LEVELGEN) ok = mission_levelgen();
        else ok = mission_goto();
        const bool v = ok && validate();
        BBAI_GENL_TRACE(m, 30, v);
        return v;
    }
    BB_HD int max_steps() const {
        int navs = 0;
        for (int leaf = 0; leaf < 4; ++leaf) {
            const int k = p_kind_get(leaf);
            navs += k == L_PUTNEXT ? 2 : k != L_NONE ? 1 : 0;
        }
        return navs * S * S * rows * cols;
    }

    // The accepted level as the engine's record: the kind's template, the objects scattered over its planes, the tables, the program.
    // `rec` is 16-byte aligned, `tmpl` = lane_build_template's bytes.
    BB_HD void write_record(uint8_t* rec, const uint8_t* tmpl) const {
        struct alignas(16) V4 { uint32_t a, b, c, d; };
        const V4* src = (const V4*)tmpl;
        V4* dst = (V4*)rec;
        const int nv = cfg.off_prog >> 4;        // (off_prog is a 16-byte multiple; the program follows from the lane's own words)
        for (int k = 0; k < nv; ++k) dst[k] = src[k];
        for (int o = 0; o < nobj; ++o) {
            const uint32_t w = obj(o);
            const int e = o_app(w), x = o_x(w), y = o_y(w);
            rec[e_index(cfg, x, y)] = (uint8_t)((cfg.doors_open && e_type(e) == T_DOOR) ? e_make(T_DOOR, e_color(e), S_OPEN) : e);
            rec[cfg.off_I + i_index(cfg, x, y)] = (uint8_t)(o + 2);
        }
        // tables, four objects per dword (off_app, off_pos are dword multiples: off_app by construction, maxo a multiple of 8)
        for (int q = 0; q < cfg.maxo; q += 4) {
            uint32_t a = 0, p0 = 0, p1 = 0;
            for (int b = 0; b < 4; ++b) {
                const int o = q + b;
                const uint32_t w = o < nobj ? obj(o) : 0u;
                a |= (w & 0xFFu) << (8 * b);
                const uint32_t xy = (w >> 8) & 0xFFFFu;
                if (b < 2) p0 |= xy << (16 * b); else p1 |= xy << (16 * (b - 2));
            }
            *(uint32_t*)(rec + cfg.off_app + q) = a;
            *(uint32_t*)(rec + cfg.off_pos + 2 * q) = p0;
            *(uint32_t*)(rec + cfg.off_pos + 2 * q + 4) = p1;
        }
        V4* pd = (V4*)(rec + cfg.off_prog);
        for (int k = 0; k < LANE_PROG_WORDS / 4; ++k) {
            V4 v;
            v.a = m.ld(L.prog + 4 * k); v.b = m.ld(L.prog + 4 * k + 1); v.c = m.ld(L.prog + 4 * k + 2); v.d = m.ld(L.prog + 4 * k + 3);
            pd[k] = v;
        }
        const int tail0 = cfg.off_prog + (int)sizeof(Prog);
        for (int k = tail0 >> 4; k < (cfg.rec_bytes >> 4); ++k) dst[k] = src[k];
    }
};

}  // namespace bbai
