// bbai_view.hpp -- the 7x7 egocentric view of k_step in REGISTERS: rotation, occlusion and masking as byte permutes on packed
// dwords (v_perm_b32 / v_alignbyte_b32 / v_dot4_u32_u8), no LDS round trip and no per-cell loop.
//
// Reference semantics: gym_minigrid MiniGridEnv.gen_obs_grid = Grid.slice + rotate_left x (dir + 1) + process_vis + the carried object on
// the agent's cell, then Grid.encode (absent dependency, restated in oracle/shim; geometry pinned by /root/reference/babyai/bot.py:658-687);
// the straightforward per-cell form is bbai_step.hpp observe_env, which the host build checks this file against on every golden
// trace and on random grids (tests/test_hostsim_view.py).
//
// Rounds 2-4 parked the window's 7 rows in LDS and read the 49 cells back one ds_read_u8 at a time in view order, testing opacity
// and building the visibility byte masks cell by cell: ~900 of k_step's ~2 600 vector instructions and 63 LDS operations per
// env-step (profiles/r04: the kernel is half VALU-bound).  Here the window stays in 14 registers (7 rows x 8 bytes):
//   * W = the rows as fetched (world orientation), T = their 8x8 byte transpose (30 permutes);
//   * the view's y-rows Y[vj] (bytes over vi) are rows of W or of T, taken forwards or backwards, with the bytes in order or
//     reversed -- per-lane selects and permute selectors, since the direction differs from lane to lane;
//   * opacity = "wall, or door state != open" is 8 SWAR operations per four cells (a non-door never carries a state), and the
//     seven 7-bit row masks process_vis wants fall out of two v_dot4_u32_u8 per row;
//   * the visibility rows come back as byte masks over Y (bit -> byte spreads), the masked rows are transposed once more into
//     the encoding's x-major order (30 permutes) and packed into 13 dwords of 49 cells.
#pragma once
#include "bbai_types.hpp"
#include "bbai_step.hpp"

namespace bbai {

// D = v_perm_b32(S0, S1, sel): result byte k is picked by selector byte k -- 0..3: byte of S1, 4..7: byte (sel - 4) of S0, 0x0C: 0x00,
// >= 0x0D: 0xFF (8..11, the sign fills, are not used here)
BB_HD uint32_t bb_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(s0, s1, sel);
#else
    uint32_t d = 0;
    for (int k = 0; k < 4; ++k) {
        const uint32_t s = (sel >> (8 * k)) & 0xFFu;
        uint32_t b;
        if (s <= 3) b = (s1 >> (8 * s)) & 0xFFu;
        else if (s <= 7) b = (s0 >> (8 * (s - 4))) & 0xFFu;
        else if (s == 0x0C) b = 0;
        else if (s >= 0x0D) b = 0xFFu;
        else b = 0xAAu;                  // (sign fills: unused; a value no test can mistake for a cell)
        d |= b << (8 * k);
    }
    return d;
#endif
}
BB_HD uint32_t bb_alignbyte(uint32_t hi, uint32_t lo, uint32_t s) {     // ({hi, lo} >> 8 s) & 0xFFFFFFFF, s = 0..3
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, s);
#else
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (s & 3)));
#endif
}
BB_HD uint32_t bb_udot4(uint32_t a, uint32_t b, uint32_t c) {           // sum of the four byte products + c
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    uint32_t r = c;
    for (int k = 0; k < 4; ++k) r += ((a >> (8 * k)) & 0xFFu) * ((b >> (8 * k)) & 0xFFu);
    return r;
#endif
}

// 4x4 byte transpose: a_r byte c = M[r][c]  ->  t_c byte r = M[r][c]
BB_HD void transpose4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t& t0, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
    const uint32_t x01l = bb_perm(a1, a0, 0x05010400u), x01h = bb_perm(a1, a0, 0x07030602u);     // [a0.b0 a1.b0 a0.b1 a1.b1], [a0.b2 a1.b2 a0.b3 a1.b3]
    const uint32_t x23l = bb_perm(a3, a2, 0x05010400u), x23h = bb_perm(a3, a2, 0x07030602u);
    t0 = bb_perm(x23l, x01l, 0x05040100u); t1 = bb_perm(x23l, x01l, 0x07060302u);
    t2 = bb_perm(x23h, x01h, 0x05040100u); t3 = bb_perm(x23h, x01h, 0x07060302u);
}
// ... of a 7x7 matrix held as 7 rows of 8 bytes (lo = columns 0..3, hi = columns 4..6 and a pad byte): out row c = column c.  Row 7 is zero;
// column 7 (the pad) is dropped, so the pad byte of every output row is zero.
BB_HD void transpose7(const uint32_t* lo, const uint32_t* hi, uint32_t* tlo, uint32_t* thi) {
    uint32_t drop;
    transpose4(lo[0], lo[1], lo[2], lo[3], tlo[0], tlo[1], tlo[2], tlo[3]);       // rows 0..3 x columns 0..3
    transpose4(lo[4], lo[5], lo[6], 0u, thi[0], thi[1], thi[2], thi[3]);          // rows 4..6 x columns 0..3
    transpose4(hi[0], hi[1], hi[2], hi[3], tlo[4], tlo[5], tlo[6], drop);         // rows 0..3 x columns 4..6
    transpose4(hi[4], hi[5], hi[6], 0u, thi[4], thi[5], thi[6], drop);            // rows 4..6 x columns 4..6
    (void)drop;
}

// four cells -> four 0/1 bytes: the cell blocks sight (WorldObj.see_behind is false): a wall, or a door that is not open.  Only doors
// carry a state (bbai_types.hpp: every other appearance byte is made with state 0), so "state != 0" alone says "closed or locked door".
BB_HD uint32_t opaque_flags(uint32_t x) {
    const uint32_t not_wall = (((x & 0x07070707u) ^ 0x02020202u) + 0x07070707u) >> 3;      // bit 0 of every byte: type != wall
    const uint32_t has_state = (((x >> 6) & 0x03030303u) + 0x03030303u) >> 2;              // bit 0 of every byte: state != 0
    return (~not_wall | has_state) & 0x01010101u;
}
// bits 0..3 of v -> 0x00 / 0xFF in bytes 0..3
BB_HD uint32_t spread_mask(uint32_t v) {
    const uint32_t s = (v | (v << 7) | (v << 14) | (v << 21)) & 0x01010101u;
    return (s << 8) - s;
}

// The view of k_step.  wd[3 r .. 3 r + 2] = the three aligned dwords that hold window row r (world orientation, r = 0..6), `off` = byte
// offset of the window's first column inside wd[3 r]; dir = the agent's direction; ce = appearance of what the agent carries (E_EMPTY:
// nothing), shown on its own cell (3, 6); nfe >= 0: the new appearance of the cell in front of the agent, view cell (3, 5), which an
// object action of this very step changed after the window was fetched.
// Out: cp[13] = the 49 cells in the encoding's order (cell (vi, vj) = byte 7 vi + vj), ALREADY MASKED by visibility (an unseen cell is 0,
// whose three channels encode as 0, 0, 0); fe2 = the (unmasked) appearance of view cell (3, 5) for the verifier and the next step.
// view_rows_perm: the same from the window's rows themselves -- wl[r] = columns 0..3 of world row r of the window, wh[r] = columns 4..6 (+ a pad byte).
BB_HD void view_rows_perm(const uint32_t* wl, const uint32_t* wh, int dir, uint32_t ce, int nfe, uint32_t* cp, int& fe2);
BB_HD void view_cells_perm(const uint32_t* wd, int off, int dir, uint32_t ce, int nfe, uint32_t* cp, int& fe2) {
    uint32_t wl[VIEW], wh[VIEW];
#pragma unroll
    for (int r = 0; r < VIEW; ++r) {
        wl[r] = bb_alignbyte(wd[3 * r + 1], wd[3 * r], (uint32_t)off);
        wh[r] = bb_alignbyte(wd[3 * r + 2], wd[3 * r + 1], (uint32_t)off);
    }
    view_rows_perm(wl, wh, dir, ce, nfe, cp, fe2);
}
BB_HD void view_rows_perm(const uint32_t* wl, const uint32_t* wh, int dir, uint32_t ce, int nfe, uint32_t* cp, int& fe2) {
    uint32_t tl[VIEW], th[VIEW];
    transpose7(wl, wh, tl, th);
    // view (vi, vj) -> window (row, column): dir 3 (vj, vi), dir 0 (vi, 6 - vj), dir 1 (6 - vj, 6 - vi), dir 2 (6 - vi, vj).  So the view's
    // y-row Y[vj] (bytes over vi) is   dir 3: W[vj]   dir 1: W[6 - vj] reversed   dir 0: T[6 - vj]   dir 2: T[vj] reversed.
    const bool odd = (dir & 1) != 0, flip = dir == 0 || dir == 1, rev = dir == 1 || dir == 2;
    const uint32_t sel_lo = rev ? 0x03040506u : 0x03020100u;      // reversed: [hi.b2 hi.b1 hi.b0 lo.b3]
    const uint32_t sel_hi = rev ? 0x0C000102u : 0x0C060504u;      // reversed: [lo.b2 lo.b1 lo.b0 0]; in order: [hi.b0 hi.b1 hi.b2 0] (the pad byte goes)
    uint32_t bl[VIEW], bh[VIEW], yl[VIEW], yh[VIEW];
#pragma unroll
    for (int k = 0; k < VIEW; ++k) { bl[k] = odd ? wl[k] : tl[k]; bh[k] = odd ? wh[k] : th[k]; }
#pragma unroll
    for (int vj = 0; vj < VIEW; ++vj) {
        const uint32_t sl = flip ? bl[VIEW - 1 - vj] : bl[vj], sh = flip ? bh[VIEW - 1 - vj] : bh[vj];
        yl[vj] = bb_perm(sh, sl, sel_lo);
        yh[vj] = bb_perm(sh, sl, sel_hi);
    }
    if (nfe >= 0) yl[5] = (yl[5] & 0x00FFFFFFu) | ((uint32_t)nfe << 24);       // view cell (3, 5)
    fe2 = (int)(yl[5] >> 24);
    // process_vis on 7-bit rows: opq[vj] bit vi
    uint32_t opq[VIEW], vis[VIEW];
#pragma unroll
    for (int vj = 0; vj < VIEW; ++vj)
        opq[vj] = bb_udot4(opaque_flags(yl[vj]), 0x08040201u, bb_udot4(opaque_flags(yh[vj]), 0x00402010u, 0u));
    process_vis_rows(opq, vis);
#pragma unroll
    for (int vj = 0; vj < VIEW; ++vj) {
        yl[vj] &= spread_mask(vis[vj] & 0xFu);
        yh[vj] &= spread_mask((vis[vj] >> 4) & 0x7u);
    }
    yl[6] = (yl[6] & 0x00FFFFFFu) | (ce << 24);                 // the agent's own cell (3, 6) shows what it carries (always visible)
    // x-major for the encoding: X[vi] = bytes over vj
    uint32_t xl[VIEW], xh[VIEW];
    transpose7(yl, yh, xl, xh);
    // 7 rows of 7 bytes -> 49 contiguous bytes
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                // rows 4 q .. 4 q + 3 fill dwords 7 q .. 7 q + 6 (the phase repeats every 28 bytes)
        const int r = 4 * q, d = 7 * q;
        cp[d] = xl[r];
        cp[d + 1] = bb_perm(xl[r + 1], xh[r], 0x04020100u);                         // [h0 h1 h2 | L0]
        cp[d + 2] = bb_perm(xh[r + 1], xl[r + 1], 0x04030201u);                     // [L1 L2 L3 | H0]
        cp[d + 3] = bb_perm(xl[r + 2], xh[r + 1], 0x05040201u);                     // [H1 H2 | l0 l1]
        cp[d + 4] = bb_perm(xh[r + 2], xl[r + 2], 0x05040302u);                     // [l2 l3 | h0 h1]
        if (q == 0) {
            cp[d + 5] = bb_perm(xl[r + 3], xh[r + 2], 0x06050402u);                                     // [h2 | L0 L1 L2]
            cp[d + 6] = bb_perm(xh[r + 3], xl[r + 3], 0x06050403u);                                     // [L3 | H0 H1 H2]
        }
    }
    cp[12] = xh[6] >> 16 & 0xFFu;                                // cell 48 = row 6, byte 6
}

// The window's rows out of an env's C plane (bbai_types.hpp): `pl` = 8 rows of 8 appearance bytes (row y at pl + 8 y, 8-byte aligned; cells
// beyond W hold E_WALL), H = rows that exist.  World row ty + r of the window is a plane row or all wall; its columns tx .. tx + 6 are
// bytes tx + 8 .. tx + 14 of the 24-byte virtual row [8 x wall | the plane row | 8 x wall]: three of its six dwords picked by tx, aligned
// by tx & 3.  (k_step: `pl` is the lane's copy of the row in LDS, loaded with the SoA state -- no memory access depends on the pose.)
BB_HD void window_rows_cpl(const uint8_t* pl, int H, int ax, int ay, int dir, uint32_t* wl, uint32_t* wh) {
    const int tx = ax + (dir == 0 ? 0 : dir == 2 ? -6 : -3);
    const int ty = ay + (dir == 1 ? 0 : dir == 3 ? -6 : -3);
    const int s = tx + 8;                        // 2 .. 15 (the agent stands inside the walls: 1 <= ax <= 6)
    const int q = s >> 2;
    const uint32_t o = (uint32_t)(s & 3);
    constexpr uint32_t WW = (uint32_t)E_WALL * 0x01010101u;
#pragma unroll
    for (int r = 0; r < VIEW; ++r) {
        const int y = ty + r;
        const bool inr = (unsigned)y < (unsigned)H;
        const uint32_t* row = (const uint32_t*)(pl + 8 * (inr ? y : 0));
        const uint32_t l0 = row[0], h0 = row[1];
        const uint32_t lo = inr ? l0 : WW, hi = inr ? h0 : WW;
        const uint32_t a = q == 2 ? lo : q == 3 ? hi : WW;
        const uint32_t b = q == 1 ? lo : q == 2 ? hi : WW;
        const uint32_t c = q == 0 ? lo : q == 1 ? hi : WW;
        wl[r] = bb_alignbyte(b, a, o);
        wh[r] = bb_alignbyte(c, b, o);
    }
}
// cid[k] == pos for the smallest such k, as id-plane entry (k + 2); 0 when no object stands there.  `ids` = the C plane row's id bytes as
// dwords (4 or 8 of them), pos = (y << 3 | x).  SWAR byte equality: a byte of x ^ (pos * 0x01010101) is zero iff it matches.
BB_HD int cid_lookup(const uint32_t* ids, int ndw, int pos) {
    const uint32_t pp = (uint32_t)pos * 0x01010101u;
    int found = 0;
#pragma unroll
    for (int d = 7; d >= 0; --d) {
        if (d >= ndw) continue;
        const uint32_t x = ids[d] ^ pp;
        const uint32_t z = (x - 0x01010101u) & ~x & 0x80808080u;       // bit 7 of byte b set if byte b of x is zero (no false positives below the first hit)
        if (z) found = 4 * d + (__builtin_ctz(z) >> 3) + 2;
    }
    return found;
}
BB_HD void encode_cells(const uint32_t* cp, RowPacker o);
// observe_env through the C plane path (host build / tests): plane -> rows -> view -> encoding.  `pl` as above; ce = the carried object's appearance.
BB_HD int observe_cpl_perm(const uint8_t* pl, int H, const Hot& h, uint32_t ce, int nfe, uint8_t* rows) {
    uint32_t wl[VIEW], wh[VIEW], cp[13];
    int fe2;
    window_rows_cpl(pl, H, h.ax, h.ay, h.dir, wl, wh);
    view_rows_perm(wl, wh, h.dir, ce, nfe, cp, fe2);
    encode_cells(cp, RowPacker(rows + ROWS_FRONT, 0));
    return fe2;
}
// An env's C plane row from its record (generator write-out on the host, k_sync_cpl): plane + id bytes.
BB_HD void cpl_from_record(const LevelCfg& c, const uint8_t* rec, uint8_t* row) {
    for (int y = 0; y < 8; ++y)
        for (int x = 0; x < 8; ++x) row[8 * y + x] = (x < c.W && y < c.H) ? rec[e_index(c, x, y)] : (uint8_t)E_WALL;
    const int nid = cpl_ids(c);
    for (int k = 0; k < nid; ++k) {
        uint8_t v = 0xFF;
        if (k < c.maxo) {
            const int x = rec[c.off_pos + 2 * k], y = rec[c.off_pos + 2 * k + 1];
            if (x < c.W && y < c.H && rec[c.off_I + i_index(c, x, y)] == k + 2) v = (uint8_t)(y << 3 | x);      // (on the grid iff the id plane says so)
        }
        row[CPL_PLANE + k] = v;
    }
}

// Grid.encode of the masked cells: a dword of four appearance bytes e0..e3 becomes the 12 encoding bytes
// t0 c0 s0 t1 | c1 s1 t2 c2 | s2 t3 c3 s3 (type = e & 7, colour = (e >> 3) & 7, state = e >> 6) with three field extractions on
// the whole dword and six byte permutes -- 9 instructions per four cells.
template <class Sink> BB_HD void encode_cells_to(const uint32_t* cp, Sink o);
BB_HD void encode_cells(const uint32_t* cp, RowPacker o) { encode_cells_to(cp, o); }
// (Sink: put(j, dword j of the 37) in order, then finish() -- RowPacker for k_step's LDS rows, a plain store for the lane generator's first observation)
template <class Sink> BB_HD void encode_cells_to(const uint32_t* cp, Sink o) {
#pragma unroll
    for (int k = 0; k < 13; ++k) {
        const uint32_t x = cp[k];
        const uint32_t t = x & 0x07070707u, c = (x >> 3) & 0x07070707u, st = (x >> 6) & 0x03030303u;
        if (k < 12) {
            o.put(3 * k, bb_perm(bb_perm(t, c, 0x050C0004u), st, 0x07000504u));
            o.put(3 * k + 1, bb_perm(bb_perm(c, st, 0x060C0105u), t, 0x07020504u));
            o.put(3 * k + 2, bb_perm(bb_perm(st, t, 0x070C0306u), c, 0x07030504u));
        } else {
            o.put(36, (t & 0xFFu) | ((c & 0xFFu) << 8) | ((st & 0xFFu) << 16));   // cell 48: three bytes, the row's last dword
        }
    }
    o.finish();
}

// observe_env through the kernel's pipeline (host build / tests): the window fetched as k_step fetches it from the record's appearance
// plane (7 rows x 3 aligned dwords), view_cells_perm, encode_cells into a one-row RowPacker area.  `rows` = ROWS_FRONT + 147 + 16 bytes.
BB_HD int observe_env_perm(const LevelCfg& c, const uint8_t* rec, const Hot& h, int nfe, uint8_t* rows) {
    const int dir = h.dir;
    const int txm = h.ax + MARGIN + (dir == 0 ? 0 : dir == 2 ? -6 : -3);
    const int tym = h.ay + MARGIN + (dir == 1 ? 0 : dir == 3 ? -6 : -3);
    const uint32_t* q = (const uint32_t*)(rec + ((tym * c.ES + txm) & ~3));
    uint32_t wd[3 * VIEW];
    for (int r = 0; r < VIEW; ++r) { wd[3 * r] = q[r * (c.ES >> 2)]; wd[3 * r + 1] = q[r * (c.ES >> 2) + 1]; wd[3 * r + 2] = q[r * (c.ES >> 2) + 2]; }
    const uint32_t ce = h.carry != NONE8 ? rec[c.off_app + h.carry] : (uint32_t)E_EMPTY;
    uint32_t cp[13];
    int fe2;
    view_cells_perm(wd, txm & 3, dir, ce, nfe, cp, fe2);
    encode_cells(cp, RowPacker(rows + ROWS_FRONT, 0));
    return fe2;
}

}  // namespace bbai
