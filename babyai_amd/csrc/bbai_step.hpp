// bbai_step.hpp -- per-env transition + instruction verifier + egocentric observation.
// One lane = one env (scalar code per lane; the kernels in bbai_engine.hip wrap it with
// coalesced SoA loads / LDS staging).
//
// Follows (reference file:line, /root/reference):
//   RoomGridLevel.step                  babyai/levels/levelgen.py:49-66
//   update_objs_poss on every drop      babyai/levels/levelgen.py:53-54,68-75
//   GoToInstr / OpenInstr / PickupInstr / PutNextInstr .verify_action
//                                       babyai/levels/verifier.py:257-274,296-303,330-350,393-417
//   BeforeInstr / AfterInstr / AndInstr .verify
//                                       babyai/levels/verifier.py:449-471,490-512,536-550
//   MiniGridEnv.step / _reward / gen_obs_grid / Grid.process_vis / Grid.encode
//                                       gym_minigrid (absent dependency) restated per
//                                       SURVEY.md section 8a rows 1,2,4; geometry pinned by
//                                       babyai/bot.py:658-687
#pragma once
#include "bbai_types.hpp"

namespace bbai {

struct EnvRef {             // views into one env's record + its verifier program (SoA)
    uint8_t* E; uint8_t* I; uint8_t* app; uint8_t* pos; uint8_t* cont; VProg prog;
};
enum : int { V_CONTINUE = 0, V_SUCCESS = 1, V_FAILURE = 2, V_NONE = 3 /* done-action mode: ActionInstr.verify returned None */ };
BB_HD EnvRef env_ref(const LevelCfg& c, uint8_t* rec, const VProg& vp) {
    EnvRef r;
    r.E = rec; r.I = rec + c.off_I; r.app = rec + c.off_app; r.pos = rec + c.off_pos; r.cont = rec + c.off_cont;
    r.prog = vp;
    return r;
}

BB_HD int dir_dx(int d) { return (d == 0) - (d == 2); }
BB_HD int dir_dy(int d) { return (d == 1) - (d == 3); }

// One ActionInstr.verify_action.  `visited` semantics: preCarrying is only updated when the
// leaf is actually evaluated (verifier.py:331-334,394-396).
// `fe2` = appearance byte of the cell in front of the agent AFTER the action (callers that have the 7x7 window at hand pass
// its cell (3,5); step_env reads it from the appearance plane).
// `idf` = the id-plane entry of that front cell when the caller already has it (k_step fetches it together with the window,
// so that the common verifications need no further memory round trip), or -1 = read it here when needed.
BB_HD int verify_leaf_action(const LevelCfg& c, const EnvRef& r, Hot& h, uint64_t stale, int leaf, int action, int fe2, int idf) {
    const int kind = r.prog.kind(leaf);
    const uint64_t set0 = r.prog.set(leaf, 0);
    if (kind == L_GOTO) {
        // success iff front_pos is one of the recorded positions (obj_poss): live tracked
        // object in the front cell, or the remembered cell of one that left the grid since
        // the last refresh.  The id plane is only consulted when the appearance plane (whose
        // lines the observation needs anyway) shows an object there.
        int fx = h.ax + dir_dx(h.dir), fy = h.ay + dir_dy(h.dir);
        if (e_type(fe2) >= T_DOOR) {
            int id = idf >= 0 ? idf : (int)r.I[i_index(c, fx, fy)];
            if (id >= 2 && (set0 >> (id - 2) & 1)) return V_SUCCESS;
        }
        uint64_t m = set0 & stale;
        while (m) {
            int o = __builtin_ctzll(m);
            m &= m - 1;
            if (r.pos[2 * o] == fx && r.pos[2 * o + 1] == fy) return V_SUCCESS;
        }
        return V_CONTINUE;
    }
    if (kind == L_OPEN) {
        if (action != A_TOGGLE) return V_CONTINUE;
        int fx = h.ax + dir_dx(h.dir), fy = h.ay + dir_dy(h.dir);
        int fe = fe2;
        if (e_type(fe) != T_DOOR) return V_CONTINUE;
        if (e_state(fe) == S_OPEN) {
            int id = idf >= 0 ? idf : (int)r.I[i_index(c, fx, fy)];
            if (id >= 2 && (set0 >> (id - 2) & 1)) return V_SUCCESS;
        }
        // strict: toggling any door without completing the instruction fails (verifier.py:270-272)
        return r.prog.strict(leaf) ? V_FAILURE : V_CONTINUE;
    }
    // Pickup / PutNext share the preCarrying protocol.
    const int sh = 8 * leaf;
    int pre = (h.pre4 >> sh) & 0xFF;
    h.pre4 = (h.pre4 & ~(0xFFu << sh)) | ((uint32_t)h.carry << sh);
    if (kind == L_PICKUP) {
        if (action != A_PICKUP) return V_CONTINUE;
        if (pre == NONE8 && h.carry != NONE8 && (set0 >> h.carry & 1)) return V_SUCCESS;
        // strict: holding anything after a pickup action that did not complete the instruction (verifier.py:343-346)
        return (r.prog.strict(leaf) && h.carry != NONE8) ? V_FAILURE : V_CONTINUE;
    }
    // L_PUTNEXT
    // strict: a pickup action while holding something fails the instruction (verifier.py:398-401)
    if (r.prog.strict(leaf) && action == A_PICKUP && h.carry != NONE8) return V_FAILURE;
    if (action != A_DROP) return V_CONTINUE;
    if (pre == NONE8 || !(set0 >> pre & 1)) return V_CONTINUE;
    if (h.carry == pre) return V_CONTINUE;          // drop failed: cur_pos == (-1,-1)
    const uint64_t set1 = r.prog.set(leaf, 1);
    int x = r.pos[2 * pre], y = r.pos[2 * pre + 1];
    const int nx[4] = {x + 1, x - 1, x, x}, ny[4] = {y, y, y + 1, y - 1};
    for (int q = 0; q < 4; ++q) {
        int id = r.I[i_index(c, nx[q], ny[q])];
        if (id >= 2 && (set1 >> (id - 2) & 1)) return V_SUCCESS;
    }
    return V_CONTINUE;
}

// ActionInstr.verify (verifier.py:216-230).  `lsm` = NULL: the normal mode, verify_action's result.  Otherwise the
// BABYAI_DONE_ACTIONS mode (verifier.py:17): bit `leaf` of *lsm is the instruction's lastStepMatch; a `done` action (and,
// by include/bbai.h's definition, every byte above 7) succeeds iff the previous evaluated action completed the instruction
// and FAILS otherwise; any other action only records whether it did and returns None (V_NONE: neither success nor
// failure for the callers, exactly as the reference's missing `return` behaves).
// (The bits travel as a value + a mode flag, by reference: a NULLABLE POINTER to them kept the kernel's copy in scratch memory --
// a select between an address and NULL cannot be promoted to a register -- 8 bytes of scratch per lane in every k_step.)
struct Lsm { uint32_t bits; bool on; };
BB_HD int verify_leaf(const LevelCfg& c, const EnvRef& r, Hot& h, uint64_t stale, int leaf, int action, int fe2, Lsm& lsm, int idf) {
    if (!lsm.on) return verify_leaf_action(c, r, h, stale, leaf, action, fe2, idf);
    if (action >= A_DONE) return (lsm.bits >> leaf & 1u) ? V_SUCCESS : V_FAILURE;
    const int res = verify_leaf_action(c, r, h, stale, leaf, action, fe2, idf);
    lsm.bits = (lsm.bits & ~(1u << leaf)) | ((res == V_SUCCESS ? 1u : 0u) << leaf);
    return V_NONE;
}

// One side of a Seq (an ActionInstr, or an AndInstr of two).  bit_a/bit_b: And progress bits.
// An AndInstr never reports failure (verifier.py:536-550) -- except in done-action mode for a `done` that IS the enum member
// (verifier.py:543-545 `action is self.env.actions.done`): if both of its instructions just failed on it, so does the And.
// An int 6 never passes that identity test (every vectorised caller of the reference steps with ints, penv.py:8); the
// reference's own bot returns the member (bot.py:593, fed to env.step by scripts/make_agent_demos.py:93-107).  `enum_done`
// says which kind this step's `done` actions are: bbai_bot_rollout (the expert's own actions) and the "done_action_enum"
// option set it.  A lone ActionInstr passes its failure through.
//
// The tree is walked as ONE loop over at most four leaf evaluations -- the first side's one or two leaves, then the second side's -- around a
// single copy of verify_leaf (round 6).  Written as the reference writes it (verify_root -> verify_side -> verify_leaf, everything inlined) the
// kernel held up to twelve copies of verify_leaf_action, and a wave whose envs carry different program shapes (BossLevel) walked through most of
// them on every step: 630 vector + 400 scalar instructions per wave-step more than GoTo's single leaf (profiles/r06/k_step_ticks_instruction_mix.txt).
// Which leaves are evaluated, in which order, and with which side effects (preCarrying, And progress bits, lastStepMatch) is unchanged:
//   * a side's leaf is evaluated unless the side is an And and the leaf's progress bit is already set (verifier.py:536-550);
//   * Before: a then b; After: b then a.  The second part is verified with the SAME action in the step the first part completes
//     (verifier.py:463-464,504-505); a failure of either part fails; while the first part is pending, a strict Seq PROBES the second part -- a
//     verify() with its side effects -- and fails if it succeeds (verifier.py:466-469,507-510).
BB_HD int side_status(const Hot& h, int n, int bit_a, int sa, int sb, bool and_fails) {
    if (n == 1) return sa;
    if (and_fails && sa == V_FAILURE && sb == V_FAILURE) return V_FAILURE;
    return (h.vstate >> bit_a & 3) == 3 ? V_SUCCESS : V_CONTINUE;
}
BB_HD int verify_root(const LevelCfg& c, const EnvRef& r, Hot& h, uint64_t stale, int action, int fe2, Lsm& lsm, int idf, bool enum_done = false) {
    const VProg* p = &r.prog;
    const int root = p->root();
    // a lone ActionInstr (every single-instruction level; a fifth of BossLevel's missions): leaf 0, with everything that depends on the leaf
    // number folded -- the loop's copy of verify_leaf indexes the program by a variable (150 vector instructions per wave-step more on GoToLocal
    // when it was the only copy)
    if (root == R_ACTION) return verify_leaf(c, r, h, stale, 0, action, fe2, lsm, idf);
    const bool seq = root != R_AND;
    const bool before = root == R_BEFORE;
    const int na = p->n_a(), nb = p->n_b();
    // first side / second side: base leaf, leaf count, first And progress bit
    const int b1 = (seq && !before) ? 2 : 0, n1 = (seq && !before) ? nb : na, s1 = (seq && !before) ? 3 : 1;
    const int b2 = before ? 2 : 0, n2 = before ? nb : na, s2 = before ? 3 : 1;
    const bool first_done = seq && (h.vstate & 1);
    const bool and_fails = lsm.on && enum_done && action == A_DONE;
    int ra = V_SUCCESS, rb = V_SUCCESS;     // the current side's leaf results (a leaf that succeeded earlier is not verified again: it stays 'success')
    bool second = first_done;               // is the second side evaluated on this step?  (decided after the first side's leaves)
    int st1 = V_SUCCESS;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int trip = 0; trip < 4; ++trip) {
        const int k = trip & 1;
        if (trip == 2) {
            if (!seq) break;
            if (!first_done) {
                st1 = side_status(h, n1, s1, ra, rb, and_fails);
                second = st1 == V_SUCCESS || (st1 != V_FAILURE && p->strict_seq());
            }
            ra = V_SUCCESS; rb = V_SUCCESS;
        }
        const bool side2 = trip >= 2;
        const int base = side2 ? b2 : b1, n = side2 ? n2 : n1, bit = (side2 ? s2 : s1) + k;
        const bool live = (side2 ? second : !first_done) && k < n && (n == 1 || !(h.vstate >> bit & 1));
        if (live) {
            const int v = verify_leaf(c, r, h, stale, base + k, action, fe2, lsm, idf);
            if (n == 2 && v == V_SUCCESS) h.vstate |= 1 << bit;
            if (k == 0) ra = v; else rb = v;
        }
    }
    if (!seq) return side_status(h, n1, s1, ra, rb, and_fails);
    const int st2 = side_status(h, n2, s2, ra, rb, and_fails);       // (meaningful where the second side was evaluated)
    if (first_done) return st2;
    if (st1 == V_FAILURE) return st1;
    if (st1 != V_SUCCESS) {             // the first part is still pending: a strict Seq has probed the second
        if (p->strict_seq() && st2 == V_SUCCESS) return V_FAILURE;
        return st1;
    }
    h.vstate |= 1;
    return st2;
}

// reward = 1 - 0.9 * (step_count / max_steps) in float64 (MiniGridEnv._reward, returned as a Python float at
// levelgen.py:59-61), no FMA contraction.  The f64 value crosses the ABI as is; the f32 output is its rounding.
BB_HD double success_reward(int step, int max_steps) {
#if defined(__HIP_DEVICE_COMPILE__)
    double q = __ddiv_rn((double)step, (double)max_steps);
    return __dsub_rn(1.0, __dmul_rn(0.9, q));
#else
    volatile double q = (double)step / (double)max_steps;
    volatile double m = 0.9 * q;
    return 1.0 - m;
#endif
}

// MiniGridEnv.step's effect on the world, first half of a step, in two pieces so that k_step can put the window fetch of the
// NEW pose between them (the pose only needs the action and the front cell; the object actions never move the agent):
//   apply_pose     step counter + turn / move;
//   apply_objects  pickup / drop / toggle: the record updates (appearance + id plane, positions, stale set).
// `fe` = appearance byte of the cell in front of the agent BEFORE the action; `ce` = appearance byte of what the agent carries
// (E_EMPTY: nothing) -- both are passed in so that callers which cache them (k_step: 2 bytes per env) never touch the record
// on a plain move / turn; `ce` is kept current.  apply_objects returns the NEW appearance byte of the front cell when the
// action changed it, else -1 (callers that keep derived copies of the appearance plane patch them with it; the changed cell
// is view cell (3, 5) of the observation that follows).  `idf` = the id-plane entry of the front cell if the caller fetched it
// already (-1: read here); `nid` receives the entry this call wrote there (-1: none).
BB_HD void apply_pose(Hot& h, int action, int fe) {
    h.step = (uint16_t)(h.step + 1);
    if (action == A_LEFT) h.dir = (h.dir + 3) & 3;
    else if (action == A_RIGHT) h.dir = (h.dir + 1) & 3;
    else if (action == A_FORWARD && (fe == E_EMPTY || (e_type(fe) == T_DOOR && e_state(fe) == S_OPEN))) {
        h.ax = (uint8_t)(h.ax + dir_dx(h.dir)); h.ay = (uint8_t)(h.ay + dir_dy(h.dir));
    }
}
BB_HD int apply_objects(const LevelCfg& c, const EnvRef& r, Hot& h, uint64_t& stale, int action, int fe, int& ce, int idf = -1, int* nid = nullptr) {
    int nfe = -1, wid = -1;          // new appearance / new id-plane entry of the front cell (-1: unchanged)
    if (action == A_PICKUP || action == A_DROP || action == A_TOGGLE) {
        const int fx = h.ax + dir_dx(h.dir), fy = h.ay + dir_dy(h.dir);
        const int ei = e_index(c, fx, fy), ii = i_index(c, fx, fy);
        if (action == A_PICKUP) {
            if (e_type(fe) >= T_KEY && h.carry == NONE8) {
                int o = (idf >= 0 ? idf : (int)r.I[ii]) - 2;
                h.carry = (uint8_t)o;
                ce = fe;                                 // a key / ball / box looks the same on the floor and in the hand
                nfe = E_EMPTY; r.I[ii] = 0; wid = 0;
                stale |= 1ull << o;                      // its recorded position is now stale
            }
        } else if (action == A_DROP) {
            if (fe == E_EMPTY && h.carry != NONE8) {
                int o = h.carry;
                nfe = ce; r.I[ii] = (uint8_t)(o + 2); wid = o + 2;
                r.pos[2 * o] = (uint8_t)fx; r.pos[2 * o + 1] = (uint8_t)fy;
                h.carry = NONE8;
                ce = E_EMPTY;
            }
        } else if (e_type(fe) == T_DOOR) {
            if (e_state(fe) == S_LOCKED) {
                if (h.carry != NONE8 && e_type(ce) == T_KEY && e_color(ce) == e_color(fe)) nfe = e_make(T_DOOR, e_color(fe), S_OPEN);
            } else {
                nfe = e_make(T_DOOR, e_color(fe), e_state(fe) == S_OPEN ? S_CLOSED : S_OPEN);
            }
        } else if (e_type(fe) == T_BOX) {                // box is replaced by its contents (nothing, or a hidden object)
            int o = (idf >= 0 ? idf : (int)r.I[ii]) - 2;
            int inner = r.cont[o];
            if (inner == NONE8) {
                nfe = E_EMPTY; r.I[ii] = 0; wid = 0;
            } else {
                nfe = r.app[inner]; r.I[ii] = (uint8_t)(inner + 2); wid = inner + 2;
                r.pos[2 * inner] = (uint8_t)fx; r.pos[2 * inner + 1] = (uint8_t)fy;
            }
            stale |= 1ull << o;
        }
        if (nfe >= 0) r.E[ei] = (uint8_t)nfe;
        // every drop ACTION refreshes the tracked positions (levelgen.py:53-54)
        if (action == A_DROP) stale = 0;
    }
    if (nid) *nid = wid;                                  // (callers that fetched the front cell's id before this ran)
    return nfe;                                           // (done, and by definition every byte above 7: nothing -- include/bbai.h)
}
BB_HD int apply_action(const LevelCfg& c, const EnvRef& r, Hot& h, uint64_t& stale, int action, int fe, int& ce) {
    apply_pose(h, action, fe);
    return apply_objects(c, r, h, stale, action, fe, ce);
}

// Second half: the instruction verifier and the episode end (RoomGridLevel.step, levelgen.py:56-66).  `fe2` = appearance
// byte of the front cell of the pose AFTER the action.  Returns done; reward by reference.
BB_HD bool finish_step(const LevelCfg& c, const EnvRef& r, Hot& h, uint64_t stale, int action, int fe2, double& reward, Lsm& lsm, int idf = -1,
                       bool enum_done = false) {
    const int status = verify_root(c, r, h, stale, action, fe2, lsm, idf, enum_done);
    bool done = h.step >= h.max_steps;
    reward = 0.0;
    if (status == V_SUCCESS) { done = true; reward = success_reward(h.step, h.max_steps); }
    else if (status == V_FAILURE) done = true;      // levelgen.py:62-64
    return done;
}

// MiniGridEnv.step + RoomGridLevel.step for one env, everything read from the record (host build, reference form of the
// two halves above).  Returns done; reward by reference.
BB_HD bool step_env(const LevelCfg& c, uint8_t* rec, const VProg& vp, Hot& h, uint64_t& stale, int action, double& reward, uint32_t* lsm = nullptr,
                    bool enum_done = false) {
    EnvRef r = env_ref(c, rec, vp);
    const int fe = r.E[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
    int ce = h.carry != NONE8 ? r.app[h.carry] : (int)E_EMPTY;
    apply_action(c, r, h, stale, action, fe, ce);
    const int fe2 = r.E[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
    Lsm l = {lsm ? *lsm : 0u, lsm != nullptr};
    const bool done = finish_step(c, r, h, stale, action, fe2, reward, l, -1, enum_done);
    if (lsm) *lsm = l.bits;
    return done;
}

// The same step in k_step's order of operations: pose, then the front cell's id fetched BEFORE the object actions (with the
// window, in the kernel) and corrected by what they wrote, then the verifier on that id.  The host build runs the golden
// traces through this form too (tests/test_hostsim_golden.py), so the bookkeeping is checked without a GPU.
BB_HD bool step_env_prefetch(const LevelCfg& c, uint8_t* rec, const VProg& vp, Hot& h, uint64_t& stale, int action, double& reward, uint32_t* lsm = nullptr,
                             bool enum_done = false) {
    EnvRef r = env_ref(c, rec, vp);
    const int fe = r.E[e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
    int ce = h.carry != NONE8 ? r.app[h.carry] : (int)E_EMPTY;
    apply_pose(h, action, fe);
    const int fi = e_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir));
    int fe2 = r.E[fi];                                                          // (k_step: view cell (3, 5) of the fetched window)
    int idf = r.I[i_index(c, h.ax + dir_dx(h.dir), h.ay + dir_dy(h.dir))];
    int nid = -1;
    const int nfe = apply_objects(c, r, h, stale, action, fe, ce, idf, &nid);
    if (nfe >= 0) fe2 = nfe;
    if (nid >= 0) idf = nid;
    Lsm l = {lsm ? *lsm : 0u, lsm != nullptr};
    const bool done = finish_step(c, r, h, stale, action, fe2, reward, l, idf, enum_done);
    if (lsm) *lsm = l.bits;
    return done;
}

// Not a MiniGrid action: "env.reset() for THIS env, now" -- what a ParallelEnv worker does on a `reset` command
// (babyai/rl/utils/penv.py:12-14) and what make_agent_demos.py:84-88 does after a bot crash.  The episode ends with
// done = 1, reward = 0 and (auto-reset) the next observation is the first one of the env's next level.
constexpr int A_RESET_ENV = 7;
BB_HD bool step_env_cmd(const LevelCfg& c, uint8_t* rec, const VProg& vp, Hot& h, uint64_t& stale, int action, double& reward, uint32_t* lsm = nullptr,
                        bool enum_done = false) {
    if (action == A_RESET_ENV) { reward = 0.0; return true; }
    return step_env(c, rec, vp, h, stale, action, reward, lsm, enum_done);
}

// bonus_levels.py:821-829: right after reset (and after the first observation was produced) the object is taken
// off the grid and put in the agent's hands; its recorded position stays behind (stale), exactly as if picked up.
BB_HD void apply_start_carry(const LevelCfg& c, uint8_t* rec, Hot& h, uint64_t& stale, int obj) {
    uint8_t* pos = rec + c.off_pos;
    const int x = pos[2 * obj], y = pos[2 * obj + 1];
    rec[e_index(c, x, y)] = E_EMPTY;
    rec[c.off_I + i_index(c, x, y)] = 0;
    h.carry = (uint8_t)obj;
    stale |= 1ull << obj;
}

// Grid.process_vis on 7-bit row masks.  opq[vj] bit vi = cell (vi,vj) blocks sight.
// Returns vis[vj] rows.  Agent at (3,6).
BB_HD void process_vis_rows(const uint32_t opq[VIEW], uint32_t vis[VIEW]) {
    uint32_t m = 1u << 3;
    for (int vj = VIEW - 1; vj >= 0; --vj) {
        const uint32_t t = ~opq[vj] & 0x7Fu;
        // left-to-right sweep (i = 0..5): closure of  m[i+1] |= m[i] & t[i]
        for (int k = 0; k < 6; ++k) m |= ((m & t & 0x3Fu) << 1);
        uint32_t a1 = m & t & 0x3Fu;
        uint32_t up = a1 | (a1 << 1);
        // right-to-left sweep (i = 6..1)
        for (int k = 0; k < 6; ++k) m |= ((m & t & 0x7Eu) >> 1);
        uint32_t a2 = m & t & 0x7Eu;
        up |= a2 | (a2 >> 1);
        vis[vj] = m & 0x7Fu;
        m = up & 0x7Fu;
    }
}

// ---- k_step's observation rows in LDS, packed at the OUTPUT pitch ----------------------------------------------------------
// A block's 256 observations leave as one contiguous span of 256 x 147 bytes.  With the rows parked in LDS at that same
// 147-byte pitch the copy-out is a plain 16-byte-per-lane stream (ds_read_b128 -> global_store_dwordx4); parked at a
// dword-aligned 148-byte pitch instead (round 2 / early round 3) every output dword had to be re-assembled from three LDS
// dwords -- 26 VALU instructions per dword, a third of k_step's vector work.  The price is paid once per lane here: the
// lane's 37 encoding dwords are shifted by its row's byte phase (one v_alignbyte_b32 each) and written as ALIGNED dwords;
// the first and last bytes of a row share a dword with the neighbouring rows and go out as byte writes, so lanes never
// touch each other's bytes and no barrier is needed before the block-wide one.
BB_HD uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t s) {     // ({hi, lo} >> 8 s) & 0xFFFFFFFF, s = 0..3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (s & 3)));
#endif
}
constexpr int ROWS_FRONT = 16;       // bytes in front of row 0 (row 0's "previous dword" exists; keeps the span 16-byte aligned)
struct RowPacker {
    uint8_t* p;          // dword j of the lane's shifted stream lives at p + 4 j
    uint32_t prev, s;
    int shp;             // 1..4: bytes of the row's first LDS dword that belong to the previous row
    // `rows` = LDS byte ROWS_FRONT of the block's row area (16-byte aligned), row r = bytes [147 r, 147 r + 147)
    BB_HD RowPacker(uint8_t* rows, int r) {
        const int base = r * OBS_BYTES;
        shp = (base & 3) ? (base & 3) : 4;
        p = rows + (base - shp);
        s = (uint32_t)(4 - shp);
        prev = 0;
    }
    // dword j of the row (j = 0..36, in order; dword 36 = the last three bytes + one zero byte)
    BB_HD void put(int j, uint32_t d) {
        const uint32_t w = align_bytes(d, prev, s);
        if (j == 0) {        // stream bytes 0 .. 3 - shp, at byte positions shp .. 3
            if (shp <= 1) p[1] = (uint8_t)(w >> 8);
            if (shp <= 2) p[2] = (uint8_t)(w >> 16);
            if (shp <= 3) p[3] = (uint8_t)(w >> 24);
        } else {
            *(uint32_t*)(p + 4 * j) = w;
        }
        prev = d;
    }
    BB_HD void finish() {    // stream bytes 148 - shp .. 146: the shp - 1 low bytes of dword 37
        const uint32_t w = align_bytes(0u, prev, s);
        uint8_t* q = p + 4 * 37;
        if (shp >= 2) q[0] = (uint8_t)w;
        if (shp >= 3) q[1] = (uint8_t)(w >> 8);
        if (shp >= 4) q[2] = (uint8_t)(w >> 16);
    }
};
// the lane's 56 bytes of dword-aligned scratch (the 7x7 window in world orientation) inside its own row
BB_HD int row_scratch(int r) { return (r * OBS_BYTES + 88 + 3) & ~3; }

// World cell shown at view cell (vi, vj): pos + f*(6-vj) + r*(vi-3), r = (-f.y, f.x).
BB_HD void view_to_world(int ax, int ay, int dir, int vi, int vj, int& x, int& y) {
    int fx = dir_dx(dir), fy = dir_dy(dir);
    int rx = -fy, ry = fx;
    x = ax + fx * (6 - vj) + rx * (vi - 3);
    y = ay + fy * (6 - vj) + ry * (vi - 3);
}

// gen_obs_grid + encode for one env: writes 147 bytes, image[vi][vj][ch].
// (straightforward per-lane form; the step kernel uses an LDS-staged equivalent)
BB_HD void observe_env(const LevelCfg& c, const uint8_t* rec, const Hot& h, uint8_t* out) {
    const uint8_t* E = rec;
    const uint8_t* app = rec + c.off_app;
    uint8_t cell[VIEW][VIEW];
    uint32_t opq[VIEW], vis[VIEW];
    for (int vj = 0; vj < VIEW; ++vj) {
        uint32_t o = 0;
        for (int vi = 0; vi < VIEW; ++vi) {
            int x, y; view_to_world(h.ax, h.ay, h.dir, vi, vj, x, y);
            int e = E[e_index(c, x, y)];
            cell[vj][vi] = (uint8_t)e;
            if (e_opaque(e)) o |= 1u << vi;
        }
        opq[vj] = o;
    }
    process_vis_rows(opq, vis);
    // the agent's own cell shows what it carries
    cell[6][3] = h.carry != NONE8 ? app[h.carry] : (uint8_t)E_EMPTY;
    for (int vi = 0; vi < VIEW; ++vi)
        for (int vj = 0; vj < VIEW; ++vj) {
            int e = cell[vj][vi];
            bool v = vis[vj] >> vi & 1;
            uint8_t* o = out + (vi * VIEW + vj) * 3;
            o[0] = v ? e_type(e) : 0; o[1] = v ? e_color(e) : 0; o[2] = v ? e_state(e) : 0;
        }
}

}  // namespace bbai
