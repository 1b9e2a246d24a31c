// bbai_types.hpp -- data layout of the batched BabyAI engine (MI355X / gfx950).
//
// One environment ("env") = one RoomGridLevel instance of the reference
// (babyai/levels/levelgen.py:17-66).  State lives in HBM as
//   * a fixed-size per-env RECORD (cold-ish, gathered per lane / streamed per wave):
//       E plane : appearance byte per cell, 5-cell wall margin on every side so the
//                 7x7 egocentric window never needs a bounds check
//       I plane : object id per cell (0 empty, 1 wall, 2+k = object k)
//       app[k]  : appearance byte of object k (type | colour<<3 | door-state<<6)
//       pos[k]  : last grid position of object k (x,y)
//       Prog    : compiled instruction tree (verifier program + mission descriptor)
//   * struct-of-arrays HOT state, one element per env, lane == env, fully coalesced:
//       Hot (16 B): agent x,y,dir, carried object, step_count, max_steps,
//                   per-leaf preCarrying, Seq/And progress bits
//       stale (8 B): objects that left the grid since the verifier's positions were
//                   last refreshed (reference: ObjDesc.obj_poss staleness,
//                   babyai/levels/verifier.py:96-161, levelgen.py:53-54)
//   * MT19937 state per env (624 words + index) -- numpy RandomState bit stream.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BB_HD __host__ __device__ __forceinline__
#define BB_COLD __host__ __device__ __attribute__((noinline))      // rare and large: ONE copy per kernel instead of one per call site
#define BB_COLD_FN inline __host__ __device__ __attribute__((noinline))     // ... for a plain (non-template) function in a header
#else
#define BB_HD inline
#define BB_COLD inline
#define BB_COLD_FN inline
#endif

namespace bbai {

// ---- appearance byte ------------------------------------------------------
// bits 2:0 = OBJECT_TO_IDX (1 empty, 2 wall, 4 door, 5 key, 6 ball, 7 box)
// bits 5:3 = COLOR_TO_IDX (red0 green1 blue2 purple3 yellow4 grey5)
// bits 7:6 = door state (0 open, 1 closed, 2 locked)
// => the 3 observation channels of a visible cell are plain bit fields of E.
enum : int {
    T_UNSEEN = 0, T_EMPTY = 1, T_WALL = 2, T_DOOR = 4, T_KEY = 5, T_BALL = 6, T_BOX = 7,
};
enum : int { C_RED = 0, C_GREEN = 1, C_BLUE = 2, C_PURPLE = 3, C_YELLOW = 4, C_GREY = 5 };
enum : int { S_OPEN = 0, S_CLOSED = 1, S_LOCKED = 2 };
enum : int { E_EMPTY = T_EMPTY, E_WALL = T_WALL | (C_GREY << 3) };
enum : int { A_LEFT = 0, A_RIGHT = 1, A_FORWARD = 2, A_PICKUP = 3, A_DROP = 4, A_TOGGLE = 5, A_DONE = 6 };

BB_HD int e_type(int e) { return e & 7; }
BB_HD int e_color(int e) { return (e >> 3) & 7; }
BB_HD int e_state(int e) { return (e >> 6) & 3; }
BB_HD int e_make(int t, int c, int s) { return t | (c << 3) | (s << 6); }
// Wall, or a door that is not open, blocks sight (WorldObj.see_behind).
BB_HD bool e_opaque(int e) { return e_type(e) == T_WALL || (e_type(e) == T_DOOR && e_state(e) != S_OPEN); }

constexpr int MARGIN = 5;        // agent x>=1 and the view reaches 6 cells ahead
constexpr int VIEW = 7;
constexpr int OBS_BYTES = VIEW * VIEW * 3;   // 147
constexpr int NONE8 = 0xFF;      // "no object" (carrying / preCarrying)
constexpr int MAX_ROOMS = 9;
constexpr int MAX_W = 25;
constexpr int MAX_OBJ = 48;      // ids fit a 64-bit set
constexpr int MT_N = 624;
constexpr int TILE = 8;          // RGBImgPartialObsWrapper tile size
constexpr int PIX = VIEW * TILE; // 56
constexpr int PIX_BYTES = PIX * PIX * 3;     // 9408
constexpr int TILE_BYTES = TILE * TILE * 3;  // 192
constexpr int MAX_TILES = 96;

// COLOR_NAMES = sorted(COLORS) = blue, green, grey, purple, red, yellow
// (gym_minigrid.minigrid; used through _rand_elem so ORDER is part of the RNG contract)
BB_HD int color_name_to_idx(int k) {
    // index into sorted names -> COLOR_TO_IDX
    return (0x403512 >> (4 * k)) & 0xF;   // nibble k: blue2 green1 grey5 purple3 red0 yellow4
}

// ---- instruction program --------------------------------------------------
enum : int { L_NONE = 0, L_GOTO = 1, L_PICKUP = 2, L_OPEN = 3, L_PUTNEXT = 4 };
enum : int { R_ACTION = 0, R_AND = 1, R_BEFORE = 2, R_AFTER = 3 };
enum : int { LOC_NONE = 0, LOC_LEFT = 1, LOC_RIGHT = 2, LOC_FRONT = 3, LOC_BEHIND = 4 };
// verifier.py:7  OBJ_TYPES = ['box', 'ball', 'key', 'door']  (index -> OBJECT_TO_IDX = 7 - index)

struct DescInfo {           // mission-surface descriptor (verifier.py:64-94)
    uint8_t type;           // OBJECT_TO_IDX of the described type
    uint8_t color;          // COLOR_TO_IDX, or 7 = no colour given
    uint8_t loc;            // LOC_*
    uint8_t count;          // number of matching objects at generation (article: >1 => "a")
};

struct Prog {               // 112 bytes
    uint64_t set[4][2];     // [leaf][0]=desc / desc_move, [leaf][1]=desc_fixed : obj_set as id bitmask
    DescInfo desc[4][2];
    uint8_t kind[4];        // L_* ; leaves 0,1 = side A (two => And), leaves 2,3 = side B
    uint8_t root;           // R_*
    uint8_t n_a, n_b;       // leaves on each side
    uint8_t strict;         // bit k < 4: leaf k verifies in strict mode (verifier.py:270-272,343-346,398-401);
                            // bit 4: the Before/After root is strict (verifier.py:466-469,507-510)
    uint8_t start_carry;    // NONE8, or the object the agent holds right after reset (bonus_levels.py:821-829)
    uint8_t pad[7];
};
static_assert(sizeof(Prog) == 112, "Prog layout");

// The verifier's per-step view of the program, kept OUT of the record as struct-of-arrays so that k_step reads
// it coalesced: a packed head word (tree shape + leaf kinds) and the obj_set bitmasks, set[k][env] with
// k = 2*leaf + slot -- a single-leaf mission touches 4 + 8 bytes per env-step instead of a 112-byte gather.
BB_HD uint32_t vhead_pack(const Prog& p) {
    return (uint32_t)p.root | ((uint32_t)p.n_a << 2) | ((uint32_t)p.n_b << 4) | ((uint32_t)p.kind[0] << 8) |
           ((uint32_t)p.kind[1] << 11) | ((uint32_t)p.kind[2] << 14) | ((uint32_t)p.kind[3] << 17) |
           ((uint32_t)(p.strict & 31) << 20);
}
struct VProg {
    uint32_t head;
    const uint64_t* sets;       // set(leaf, slot) = sets[(2*leaf + slot) * stride]
    int64_t stride;
    uint64_t set00;             // set(0, 0), fetched with the rest of the env's hot state (every mission's first leaf needs it:
                                // one memory round trip less on the verifier's path)
    BB_HD void bind(uint32_t head_, const uint64_t* sets_, int64_t stride_) { head = head_; sets = sets_; stride = stride_; set00 = sets_[0]; }
    BB_HD int root() const { return head & 3; }
    BB_HD int n_a() const { return (head >> 2) & 3; }
    BB_HD int n_b() const { return (head >> 4) & 3; }
    BB_HD int kind(int leaf) const { return (head >> (8 + 3 * leaf)) & 7; }
    BB_HD bool strict(int leaf) const { return (head >> (20 + leaf)) & 1; }
    BB_HD bool strict_seq() const { return (head >> 24) & 1; }
    BB_HD uint64_t set(int leaf, int slot) const { return (leaf | slot) == 0 ? set00 : sets[(int64_t)(2 * leaf + slot) * stride]; }
};

struct Hot {                // 16 bytes, one per env, SoA array => one dwordx4 per lane
    uint8_t ax, ay, dir, carry;
    uint16_t step, max_steps;
    uint32_t pre4;          // preCarrying per leaf, byte k = leaf k (verifier.py:321-334,373-395); packed so
                            // the dynamic leaf index is a shift, not a scratch array
    uint8_t vstate;         // bit0 Seq first part done; bits1,2 side-A And a/b; bits3,4 side-B And a/b
    uint8_t frozen;         // ManyEnvs semantics: finished, waiting for an explicit reset
    uint8_t last_locked;    // LevelGen.locked_room survives episodes (levelgen.py:284,325,384): room idx or NONE8
    uint8_t slot;           // look-ahead ring: which slot holds this env's next level
};
static_assert(sizeof(Hot) == 16, "Hot layout");

// ---- level configuration ---------------------------------------------------
enum : int { K_GOTO = 0, K_LEVELGEN = 1, K_BONUS = 2 };
enum : int { BS_GOTO_REDBLUE_BALL = 1, BS_OPEN_RED_DOOR, BS_OPEN_DOOR, BS_GOTO_DOOR, BS_GOTO_OBJ_DOOR, BS_ACTION_OBJ_DOOR,
             BS_UNLOCK_LOCAL, BS_KEY_IN_BOX, BS_UNLOCK_PICKUP, BS_BLOCKED_UNLOCK_PICKUP, BS_UNLOCK_TO_UNLOCK, BS_PICKUP_DIST,
             BS_PICKUP_ABOVE, BS_OPEN_TWO_DOORS, BS_FIND_OBJ, BS_KEY_CORRIDOR, BS_ONE_ROOM, BS_PUT_NEXT, BS_MOVE_TWO_ACROSS,
             BS_OPEN_DOORS_ORDER,
             // test_levels.py: hand-built bot-regression layouts
             BS_TEST_GOTO_BLOCKED, BS_TEST_PUTNEXT_BLOCKED, BS_TEST_PUTNEXT_DOOR1, BS_TEST_PUTNEXT_DOOR2,
             BS_TEST_PUTNEXT_IDENTICAL, BS_TEST_UNBLOCKING_LOOP, BS_TEST_PUTNEXT_CLOSE_DOOR, BS_TEST_LOTS_OF_BLOCKERS, BS_COUNT };
enum : int { TG_REDBALL = 0, TG_DIST = 1, TG_DOOR = 2, TG_TWO_DISTS = 3, TG_LOCKED_DOOR = 4, TG_LOCKED_ROOM_OBJ = 5 };
enum : int { AK_GOTO = 0, AK_PICKUP = 1, AK_OPEN = 2, AK_PUTNEXT = 3 };
enum : int { IK_ACTION = 0, IK_AND = 1, IK_SEQ = 2 };

struct LevelCfg {
    int32_t kind;
    int32_t room_size, num_rows, num_cols, num_dists;
    // K_GOTO family (iclr19_levels.py:40-63,66-124,224-257)
    int32_t redball;        // place a red ball first and make it the target
    int32_t connect;        // connect_all()
    int32_t check_reach;    // 0 none, 1 check_objs_reachable(), 2 reject when everything IS reachable (UnblockPickup)
    int32_t doors_open;     // open_all_doors() at the end
    int32_t all_unique;     // add_distractors(all_unique=...)
    // single-instruction hand-written levels (iclr19_levels.py:10-37,187-221,304-491)
    int32_t instr;          // leaf kind of the mission: L_GOTO / L_PICKUP / L_OPEN / L_PUTNEXT
    int32_t target;         // TG_*: how the described object(s) are chosen
    int32_t lock;           // 1: a locked door + its key are placed first; agent placed last, outside the locked room
    int32_t lock_color_excl;// Unlock: with probability 1/2 connect_all() avoids the locked door's colour
    int32_t dists_per_room; // 1: num_dists distractors in every room except the locked one
    int32_t grey_dists;     // GoToRedBallGrey: distractors are recoloured grey
    // K_BONUS: hand-written gen_mission scripts of bonus_levels.py (BS_*), with up to four integer parameters
    int32_t script, sp[4];
    // K_LEVELGEN (levelgen.py:256-460)
    int32_t locations, unblocking, implicit_unlock;
    int32_t n_action_kinds, action_kinds[4];
    int32_t n_instr_kinds, instr_kinds[3];
    double locked_room_prob;
    // derived layout (fill_layout)
    int32_t W, H, ES, EH, maxo;
    int32_t off_I, off_app, off_pos, off_cont, off_prog, rec_bytes;
};

BB_HD int round_up(int v, int m) { return (v + m - 1) / m * m; }

inline int fill_layout(LevelCfg& c) {
    if (c.room_size < 3 || c.num_rows < 1 || c.num_cols < 1) return -1;
    if (c.num_rows * c.num_cols > MAX_ROOMS) return -1;
    c.W = (c.room_size - 1) * c.num_cols + 1;
    c.H = (c.room_size - 1) * c.num_rows + 1;
    if (c.W > MAX_W || c.H > MAX_W) return -1;
    c.ES = round_up(c.W + 2 * MARGIN, 4);
    c.EH = c.H + 2 * MARGIN;
    int ndoors = c.num_rows * (c.num_cols - 1) + c.num_cols * (c.num_rows - 1);
    int nd = c.dists_per_room ? c.num_dists * c.num_rows * c.num_cols : c.num_dists;
    if (c.kind == K_BONUS) nd = 24;               // bonus scripts place at most 2*9 + a few objects
    c.maxo = round_up(nd + 2 + ndoors, 8);
    if (c.maxo > MAX_OBJ) return -1;
    c.off_I = c.ES * c.EH;
    c.off_app = round_up(c.off_I + c.W * c.H, 4);
    c.off_pos = c.off_app + c.maxo;
    c.off_cont = c.off_pos + 2 * c.maxo;          // contains[k]: object revealed when box k is toggled (NONE8 = empty)
    c.off_prog = round_up(c.off_cont + c.maxo, 16);
    c.rec_bytes = round_up(c.off_prog + (int)sizeof(Prog), 64);
    return 0;
}

// ---- window plane ("V plane"): a redundant, window-shaped copy of the appearance plane for k_step ----------------------
// HBM moves 128-byte lines.  A 7x7 window cut out of the row-major appearance plane (7 rows x 12 B at a 32-byte pitch)
// touches 2-3 lines for 49 useful bytes (profiles/r03/fetchcal*: the gather costs whole lines).  The V plane stores, for
// every window-origin class (xo, yo) = (tx >> 3, ty >> 1) -- tx, ty = top-left cell of the window in plane coordinates --
// ONE line of 8 rows x 16 cells starting at plane cell (8 xo, 2 yo): the window of any agent pose lies inside exactly one
// line (columns tx - 8 xo + 0..6 <= 13, rows ty - 2 yo + 0..6 <= 7).  Each cell is stored in up to 8 lines; only k_step's
// own rare grid writes (pickup / drop / toggle) and the per-episode rebuild pay for that.  BossLevel: 4 x 13 lines =
// 6.5 KB per env next to the 1 KB plane -- what 288 GB of HBM is for.
constexpr int VLINE = 128;
BB_HD int v_nxo(const LevelCfg& c) { return ((c.ES - VIEW) >> 3) + 1; }
BB_HD int v_nyo(const LevelCfg& c) { return ((c.EH - VIEW) >> 1) + 1; }
BB_HD int v_bytes(const LevelCfg& c) { return v_nxo(c) * v_nyo(c) * VLINE; }

// ---- compact plane ("C plane"): the whole grid of a small single room, per env, next to the SoA state ----------------------------
// A single room of at most 8 x 8 cells IS 64 appearance bytes (everything outside is wall), and none of them depends on the agent's
// pose.  k_step's in-place path used to fetch the front cell (record), then -- the pose known -- the 7 x 7 window out of the record's
// margin plane (7 rows at pitch ES = 20: 2-3 lines for 49 bytes; profiles/r05: 2.3-2.7 x the algorithmic bytes), then the id-plane
// entry of the front cell: three DEPENDENT round trips on a kernel that is nothing but a chain of them.  The C plane row of an env --
//   bytes 0 .. 63           E[y][x] at pitch 8, y < 8 (cells outside W x H hold E_WALL)
//   bytes 64 .. 64 + ids    cid[k] = (y << 3 | x) of object k while it stands on the grid, 0xFF otherwise (carried, consumed box, hidden)
// -- is contiguous over the envs ([n][cpl_bytes]: a wave reads 64 rows as one dense span, every fetched byte used) and is loaded WITH the
// SoA state in the step's first round trip; the front cell, the window (wall fill + byte alignment in registers, rows through LDS) and
// the front cell's object id (a byte search over cid) then cost no memory access at all.  Derived state like the window plane of the
// mazes: the generator writes a row next to every look-ahead level (next_obs slot, CPL_OFF), a finished env copies its next one in,
// k_step's rare grid writes patch it, imports and checkpoint loads rebuild it from the records (k_sync_cpl).
constexpr int CPL_PLANE = 64;
constexpr int CPL_MAX_IDS = 32;
BB_HD bool cpl_ok(const LevelCfg& c) { return c.num_rows * c.num_cols == 1 && c.W <= 8 && c.H <= 8 && c.maxo <= CPL_MAX_IDS; }
BB_HD int cpl_ids(const LevelCfg& c) { return c.maxo <= 16 ? 16 : 32; }
BB_HD int cpl_bytes(const LevelCfg& c) { return CPL_PLANE + cpl_ids(c); }          // 80 or 96: multiples of 16
constexpr int OBS_SLOT = 256;           // bytes per look-ahead level in next_obs (in-place layout): the first observation (147 of 160) ...
constexpr int CPL_OFF = 160;            // ... and its C plane row (<= 96 bytes)

BB_HD int e_index(const LevelCfg& c, int x, int y) { return (y + MARGIN) * c.ES + (x + MARGIN); }
BB_HD int i_index(const LevelCfg& c, int x, int y) { return y * c.W + x; }

}  // namespace bbai
