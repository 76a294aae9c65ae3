// board.cuh — Tak rules on a CDNA4 wavefront: one 64-lane wave owns one game, lane = square.
//
// Replaces (reference paths): tak/src/game.rs (Game::play / result), tak/src/move_gen.rs
// (possible_moves), tak/src/board.rs (find_paths / full / flat_diff), tak/src/tile.rs.
// Layout: the packed state of include/takgpu.h is a struct of per-square arrays, so the wave loads
// a game with one coalesced 8-byte load per lane (stack bits) + one byte load (meta).  Per-colour /
// per-type bitboards are never stored: `__ballot` builds them from the lanes in one instruction,
// roads are flood-filled on the scalar unit with shifts, and spreads move stones between lanes with
// ds_bpermute shuffles.  Everything here is bit-exact integer work (no floating point except the
// fcd plane of the encoder, which reproduces the reference's f64 division).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/takgpu.h"

namespace tg {

enum : uint32_t { FLAT = 0, WALL = 1, CAP = 2 };
enum : uint32_t { UP = 0, DOWN = 1, LEFT = 2, RIGHT = 3 };

struct Geom {
    int n, nsq, slots, bytes;
    uint64_t all, row0, rowL, col0, colL;
};

__host__ __device__ inline Geom make_geom(int n) {
    Geom g;
    g.n = n;
    g.nsq = n * n;
    g.slots = n <= 5 ? 25 : 36;
    g.bytes = n <= 5 ? TG_STATE5_BYTES : TG_STATE6_BYTES;
    g.all = (g.nsq == 64) ? ~0ull : ((1ull << g.nsq) - 1);
    g.row0 = (1ull << n) - 1;
    g.rowL = g.row0 << (n * (n - 1));
    g.col0 = 0;
    for (int r = 0; r < n; r++) g.col0 |= 1ull << (r * n);
    g.colL = g.col0 << (n - 1);
    return g;
}

// reference tak/src/game.rs:10-20
__host__ __device__ inline void starting_stones(int n, int& stones, int& caps) {
    stones = n == 3 ? 10 : n == 4 ? 15 : n == 5 ? 21 : 30;
    caps = n <= 4 ? 0 : 1;
}
__host__ __device__ inline int board_channels(int n) { return (n + 2 + 6) * 2; }
__host__ __device__ inline int input_channels(int n) {
    int s, c;
    starting_stones(n, s, c);
    return board_channels(n) + 2 + 2 * s + 2 * c;
}

// Wave-distributed game: per-lane square data + wave-uniform header.
struct WState {
    uint64_t stack;   // lane = square: colour bits bottom→top
    uint32_t height;  // lane = square
    uint32_t top;     // lane = square: piece type of the top stone
    uint32_t to_move, ply, ws, wc, bs, bc, rev;
    int32_t half_komi;
};

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src);
    uint32_t hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

__device__ inline void ws_load(WState& s, const uint8_t* st, const Geom& g) {
    int lane = lane_id();
    bool on = lane < g.nsq;
    const uint64_t* stk = (const uint64_t*)st;
    const uint8_t* meta = st + 8 * g.slots;
    s.stack = on ? stk[lane] : 0ull;
    uint32_t m = on ? (uint32_t)meta[lane] : 0u;
    s.height = m & 63u;
    s.top = m >> 6;
    const uint32_t* h = (const uint32_t*)(st + g.bytes - 16);
    uint32_t h0 = uni(h[0]), h1 = uni(h[1]), h2 = uni(h[2]);
    s.to_move = (h0 >> 8) & 0xff;
    s.ply = h0 >> 16;
    s.ws = h1 & 0xff;
    s.wc = (h1 >> 8) & 0xff;
    s.bs = (h1 >> 16) & 0xff;
    s.bc = h1 >> 24;
    s.half_komi = (int32_t)(int8_t)(h2 & 0xff);
    s.rev = (h2 >> 8) & 0xff;
}

// ws_load in two halves, for a wave that stages several positions: the raw words of ALL of them are requested first
// (ws_load_raw), each becomes a WState when its turn comes (ws_unpack) — one memory round trip instead of one per position
struct WRaw { uint64_t stack; uint32_t meta, h0, h1, h2; };
__device__ inline WRaw ws_load_raw(const uint8_t* st, const Geom& g) {
    const int lane = lane_id();
    const int l = lane < g.nsq ? lane : 0;  // unconditional loads (hipcc waits right behind an exec-masked one)
    WRaw r;
    r.stack = ((const uint64_t*)st)[l];
    r.meta = (uint32_t)(st + 8 * g.slots)[l];
    const uint32_t* h = (const uint32_t*)(st + g.bytes - 16);
    r.h0 = h[0]; r.h1 = h[1]; r.h2 = h[2];
    return r;
}
__device__ inline void ws_unpack(WState& s, const WRaw& r, const Geom& g) {
    const bool on = lane_id() < g.nsq;
    s.stack = on ? r.stack : 0ull;
    const uint32_t m = on ? r.meta : 0u;
    s.height = m & 63u;
    s.top = m >> 6;
    const uint32_t h0 = uni(r.h0), h1 = uni(r.h1), h2 = uni(r.h2);
    s.to_move = (h0 >> 8) & 0xff;
    s.ply = h0 >> 16;
    s.ws = h1 & 0xff;
    s.wc = (h1 >> 8) & 0xff;
    s.bs = (h1 >> 16) & 0xff;
    s.bc = h1 >> 24;
    s.half_komi = (int32_t)(int8_t)(h2 & 0xff);
    s.rev = (h2 >> 8) & 0xff;
}

__device__ inline void ws_store(const WState& s, uint8_t* st, const Geom& g) {
    int lane = lane_id();
    uint64_t* stk = (uint64_t*)st;
    uint8_t* meta = st + 8 * g.slots;
    if (lane < g.slots) {
        bool on = lane < g.nsq;
        stk[lane] = on ? s.stack : 0ull;
        meta[lane] = on ? (uint8_t)(s.height | ((s.height ? s.top : 0u) << 6)) : (uint8_t)0;
    }
    // zero the pad so equal games are equal bytes
    int pad0 = 9 * g.slots, pad1 = g.bytes - 16;
    for (int i = pad0 + lane; i < pad1; i += 64) st[i] = 0;
    if (lane < 4) {
        uint32_t v = 0;
        if (lane == 0) v = (uint32_t)g.n | (s.to_move << 8) | (s.ply << 16);
        if (lane == 1) v = s.ws | (s.wc << 8) | (s.bs << 16) | (s.bc << 24);
        if (lane == 2) v = ((uint32_t)s.half_komi & 0xff) | ((s.rev & 0xff) << 8);
        ((uint32_t*)(st + g.bytes - 16))[lane] = v;
    }
}

__device__ inline void ws_start(WState& s, const Geom& g, int half_komi) {
    int st, cp;
    starting_stones(g.n, st, cp);
    s.stack = 0; s.height = 0; s.top = 0;
    s.to_move = 0; s.ply = 0; s.ws = s.bs = (uint32_t)st; s.wc = s.bc = (uint32_t)cp;
    s.rev = 0; s.half_komi = half_komi;
}

__device__ inline uint32_t top_color(const WState& s) { return s.height ? (uint32_t)((s.stack >> (s.height - 1)) & 1ull) : 0u; }

// Board::find_paths (board.rs:77-113) as a bitboard flood fill: does `b` connect `from` to `to`?
__device__ inline bool bb_connects(uint64_t b, uint64_t from, uint64_t to, const Geom& g) {
    uint64_t r = b & from;
    if (!r || !(b & to)) return false;
    for (;;) {
        uint64_t nx = r | (((r << g.n) | (r >> g.n) | ((r << 1) & ~g.col0) | ((r >> 1) & ~g.colL)) & b);
        if (nx == r) break;
        r = nx;
    }
    return (r & to) != 0;
}
__device__ inline bool bb_road(uint64_t b, const Geom& g) {
    return bb_connects(b, g.row0, g.rowL, g) || bb_connects(b, g.col0, g.colL, g);
}

// Game::result, game.rs:220-267.  Wave-uniform result (TgResult).
__device__ inline uint32_t ws_result(const WState& s, const Geom& g) {
    bool occ = s.height > 0;
    uint32_t tc = top_color(s);
    bool roadp = occ && s.top != WALL;
    uint64_t wroad = __ballot(roadp && tc == 0);
    uint64_t broad = __ballot(roadp && tc == 1);
    uint64_t occ_bb = __ballot(occ);
    uint64_t wflat = __ballot(occ && s.top == FLAT && tc == 0);
    uint64_t bflat = __ballot(occ && s.top == FLAT && tc == 1);
    uint32_t other = s.to_move ^ 1u;
    // dragon clause: the player who just moved is checked first
    if (bb_road(other ? broad : wroad, g)) return other ? TG_BLACK_ROAD : TG_WHITE_ROAD;
    if (bb_road(s.to_move ? broad : wroad, g)) return s.to_move ? TG_BLACK_ROAD : TG_WHITE_ROAD;
    if ((s.wc == 0 && s.ws == 0) || (s.bc == 0 && s.bs == 0) || occ_bb == g.all) {
        int fd = __popcll(wflat) - __popcll(bflat);
        int k = s.half_komi / 2;  // truncates toward zero like Rust's i8 division
        if (fd > k) return TG_WHITE_FLAT;
        if (fd < k) return TG_BLACK_FLAT;
        return (s.half_komi % 2 == 0) ? TG_DRAW : TG_BLACK_FLAT;
    }
    if (s.rev >= 50) return TG_DRAW_REVERSIBLE;
    return TG_ONGOING;
}

__device__ inline int flat_diff(const WState& s) {
    bool occ = s.height > 0;
    uint32_t tc = top_color(s);
    uint64_t wflat = __ballot(occ && s.top == FLAT && tc == 0);
    uint64_t bflat = __ballot(occ && s.top == FLAT && tc == 1);
    return __popcll(wflat) - __popcll(bflat);
}

// Game::play, game.rs:121-218.  Wave-uniform TgPlayError; on error the state is unchanged
// (Game::safe_play semantics).  All 64 lanes must call.
__device__ inline uint32_t ws_play(WState& s, uint32_t mv, const Geom& g) {
    const int lane = lane_id();
    const int n = g.n;
    uint32_t sq = mv & 63u, f = (mv >> 6) & 3u, pat = (mv >> 8) & 0xffu;
    bool swapped = s.ply < 2;
    uint32_t color = swapped ? (s.to_move ^ 1u) : s.to_move;
    if ((int)sq >= g.nsq) return TG_PLAY_OUT_OF_BOUNDS;
    // the move is wave-uniform: the source square's record comes by v_readlane, not through the LDS crossbar (four ds_bpermute
    // round trips at the head of every move of every descent)
    const int sql = (int)uni(sq);
    uint32_t src_h = (uint32_t)__builtin_amdgcn_readlane((int)s.height, sql);
    uint32_t src_top = (uint32_t)__builtin_amdgcn_readlane((int)s.top, sql);
    uint64_t src_stack = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(s.stack >> 32), sql) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)s.stack, sql);
    if (pat == 0) {  // execute_place, game.rs:147-169
        uint32_t piece = f;
        if (piece > CAP) return TG_PLAY_OUT_OF_BOUNDS;
        uint32_t stones = s.to_move == 0 ? s.ws : s.bs;
        uint32_t caps = s.to_move == 0 ? s.wc : s.bc;
        if (src_h) return TG_PLAY_ALREADY_OCCUPIED;
        if (piece == CAP && caps == 0) return TG_PLAY_NO_CAPSTONE;
        if (piece != CAP && stones == 0) return TG_PLAY_NO_STONES;
        if (swapped && piece != FLAT) return TG_PLAY_OPENING_NON_FLAT;
        if (lane == (int)sq) { s.stack = color; s.height = 1; s.top = piece; }
        if (piece != CAP) {
            if ((s.to_move == 0) != swapped) s.ws -= 1; else s.bs -= 1;  // dec_stones :103-109
        } else {
            if (s.to_move == 0) s.wc -= 1; else s.bc -= 1;               // dec_caps :111-116
        }
        s.rev = 0;
    } else {  // execute_spread, game.rs:171-209
        if (src_h == 0) return TG_PLAY_EMPTY_SQUARE;
        uint32_t src_tc = (uint32_t)((src_stack >> (src_h - 1)) & 1ull);
        if (src_tc != color) return TG_PLAY_STACK_NOT_OWNED;
        int k = 8 - __builtin_ctz(pat);  // Pattern::count_pieces
        int m = __popc(pat);             // squares covered
        if (k > n) return TG_PLAY_TAKE_CARRY_LIMIT;
        if (k > (int)src_h) return TG_PLAY_TAKE_STACK_SIZE;
        int sr = (int)sq / n, sc = (int)sq % n;
        int room = f == UP ? n - 1 - sr : f == DOWN ? sr : f == LEFT ? sc : n - 1 - sc;
        int lr = lane / n, lc = lane % n;
        int i = 0;  // this lane is the i-th square of the spread (0 = not on it)
        if (lane < g.nsq) {
            if (f == UP && lc == sc && lr > sr) i = lr - sr;
            else if (f == DOWN && lc == sc && lr < sr) i = sr - lr;
            else if (f == LEFT && lr == sr && lc < sc) i = sc - lc;
            else if (f == RIGHT && lr == sr && lc > sc) i = lc - sc;
        }
        if (i > m) i = 0;
        // stones for step i: carry bits [start, start+len)
        int start = 0, len = 0;
        {
            int cnt = 0, prev = 0;
            for (int b = 0; b < 8; b++) {
                if (pat & (0x80u >> b)) {
                    cnt++;
                    if (cnt == i) { start = prev; len = b + 1 - prev; }
                    prev = b + 1;
                }
            }
        }
        bool last = (i == m);
        uint32_t err = 0;
        if (i > 0 && s.height > 0) {  // Tile::stack, tile.rs:28-45
            if (s.top == CAP) err = TG_PLAY_STACK_CAP;
            else if (s.top == WALL && !(last && len == 1 && src_top == CAP)) err = TG_PLAY_STACK_WALL;
        }
        uint64_t err_bb = __ballot(err != 0);
        if (err_bb || m > room) {
            // first failing step in walking order decides
            int limit = m > room ? room : m;
            for (int t = 1; t <= limit; t++) {
                uint64_t bb = __ballot(i == t && err != 0);
                if (bb) return (uint32_t)__shfl((int)err, __builtin_ctzll(bb));
            }
            return TG_PLAY_SPREAD_OUT_OF_BOUNDS;
        }
        uint64_t carry = (src_stack >> (src_h - (uint32_t)k)) & ((1ull << k) - 1ull);
        if (i > 0) {
            uint64_t sub = (carry >> start) & ((1ull << len) - 1ull);
            s.stack |= sub << s.height;
            s.height += (uint32_t)len;
            s.top = last ? src_top : FLAT;  // only the last stone dropped keeps the carried top type
        }
        if (lane == (int)sq) {  // Tile::take, tile.rs:49-63
            s.height = src_h - (uint32_t)k;
            s.stack = src_stack & ((1ull << s.height) - 1ull);
            s.top = FLAT;
        }
        s.rev += 1;
    }
    s.ply += 1;
    s.to_move ^= 1u;
    return TG_PLAY_OK;
}

// Enumerate the spreads of one (square, direction) in the reference's order (move_gen.rs:54-102):
// pickup ascending; for each pickup the LIFO stack pops larger first drops first, which is ascending
// numeric order of the MSB-first drop pattern.  `free_run` = squares in the direction that accept
// any drop before the edge / a cap / a wall; `smash` = the blocker is a wall and the mover's top is
// a cap (it may take exactly one stone: the cap).
template <class F>
__device__ inline int enum_spreads(int max_carry, int free_run, bool smash, F&& emit) {
    int count = 0;
    for (int k = 1; k <= max_carry; k++) {
        int nv = 1 << (k - 1);
        for (int v = 0; v < nv; v++) {
            uint32_t bits = ((uint32_t)v << 1) | 1u;  // k bits, MSB = first carried stone
            int parts = __popc(bits);
            bool ok = parts <= free_run || (smash && parts == free_run + 1 && (k == 1 || (v & 1)));
            if (ok) { emit(count, bits << (8 - k)); count++; }
        }
    }
    return count;
}

// Number of patterns enum_spreads emits, in closed form: pickup k in p parts ↔ choosing p-1 of the k-1 gaps, all p ≤ free_run
// count; a smash needs p = free_run+1 with the last part = 1 (C(k-2, free_run-1) ways, or the single stone when k = 1).
struct SpreadCountTable {
    uint8_t c[9][8][2];
    constexpr SpreadCountTable() : c() {
        int binom[9][9] = {};
        for (int a = 0; a < 9; a++) {
            binom[a][0] = 1;
            for (int b = 1; b <= a; b++) binom[a][b] = binom[a - 1][b - 1] + (b <= a - 1 ? binom[a - 1][b] : 0);
        }
        for (int mc = 0; mc < 9; mc++)
            for (int fr = 0; fr < 8; fr++)
                for (int sm = 0; sm < 2; sm++) {
                    int total = 0;
                    for (int k = 1; k <= mc; k++) {
                        for (int p = 1; p <= k && p <= fr; p++) total += binom[k - 1][p - 1];
                        if (sm && fr + 1 <= k) total += k == 1 ? 1 : (fr >= 1 ? binom[k - 2][fr - 1] : 0);
                    }
                    c[mc][fr][sm] = (uint8_t)total;
                }
    }
};
__device__ inline int spread_count(int max_carry, int free_run, bool smash) {
    static constexpr SpreadCountTable T{};
    return T.c[max_carry][free_run][smash ? 1 : 0];
}

// The same enumeration as a bit mask, for boards up to 6×6 (pickups up to 6: 63 candidates).  Candidate b = 2^(k-1) - 1 + v is
// pickup k with drop pattern v in enum_spreads' order, so bit order = emission order; LE[f] holds the candidates in at most f
// parts, SM[f] those in exactly f + 1 parts whose last part is one stone (a cap flattening the wall behind f free squares).
// The patterns a (square, direction) emits are the set bits of (LE[free_run] | smash·SM[free_run]) below candidate 2^max_carry - 1:
// their number is a popcount (no table in memory) and the enumeration visits only the patterns it emits — enum_spreads walks
// all 2^max_carry - 1 candidates in every lane, the slowest lane setting the pace of the wave (a third of the tree kernel's
// move generation).
struct SpreadMasks {
    uint64_t le[8], sm[8];
    constexpr SpreadMasks() : le(), sm() {
        for (int f = 0; f < 8; f++) {
            uint64_t a = 0, c = 0;
            for (int b = 0; b < 63; b++) {
                int b1 = b + 1, k = 0;
                while ((1 << k) <= b1) k++;  // b1 has k bits
                const int v = b1 - (1 << (k - 1));
                int parts = 1;
                for (int t = v; t; t >>= 1) parts += t & 1;
                if (parts <= f) a |= 1ull << b;
                if (parts == f + 1 && (k == 1 || (v & 1))) c |= 1ull << b;
            }
            le[f] = a;
            sm[f] = c;
        }
    }
};
__device__ inline uint64_t spread_mask(int max_carry, int free_run, bool smash) {
    constexpr SpreadMasks M{};
    // (selected by compares on constants: free_run ≤ 5 on a 6×6 board)
    const int f = free_run;
    const uint64_t le = f <= 0 ? M.le[0] : f == 1 ? M.le[1] : f == 2 ? M.le[2] : f == 3 ? M.le[3] : f == 4 ? M.le[4] : f == 5 ? M.le[5] : M.le[6];
    const uint64_t sm = f <= 0 ? M.sm[0] : f == 1 ? M.sm[1] : f == 2 ? M.sm[2] : f == 3 ? M.sm[3] : f == 4 ? M.sm[4] : f == 5 ? M.sm[5] : M.sm[6];
    const uint64_t below = (1ull << ((1u << max_carry) - 1u)) - 1ull;  // candidates of pickups 1 … max_carry (≤ 6: ≤ 63 bits)
    return (le | (smash ? sm : 0ull)) & below;
}
template <class F>
__device__ inline void enum_spread_mask(uint64_t m, F&& emit) {
    int idx = 0;
    while (m) {
        const int b1 = __builtin_ctzll(m) + 1;
        m &= m - 1ull;
        const int k = 32 - __builtin_clz((unsigned)b1);
        const uint32_t v = (uint32_t)b1 - (1u << (k - 1));
        emit(idx++, ((v << 1) | 1u) << (8 - k));
    }
}

// Inclusive prefix sum over the 64 lanes with DPP adds (six v_add with a data-parallel-primitive operand: row_shr 1, 2, 4, 8
// inside the rows of 16 lanes, then row_bcast15 / row_bcast31 carry the row totals on) instead of six ds_bpermute round trips
// through the LDS crossbar with a select each.  Lanes that receive nothing add the identity 0 (`old` of update_dpp).
__device__ inline int wave_inclusive_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast15 → rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast31 → rows 2 and 3
    return v;
}

// Game::possible_moves, move_gen.rs:7-30, same order.  Calls emit(index, code) for every move and
// returns the (wave-uniform) count.  All 64 lanes must call.  emit is called with index < cap only.
template <class F>
__device__ inline int ws_movegen(const WState& s, const Geom& g, int cap, F&& emit) {
    const int lane = lane_id();
    const int n = g.n;
    bool occ = s.height > 0;
    uint32_t tc = top_color(s);
    bool swapped = s.ply < 2;
    uint32_t color = swapped ? (s.to_move ^ 1u) : s.to_move;
    uint64_t occ_bb = __ballot(occ);
    uint64_t own_bb = __ballot(occ && tc == color);
    uint64_t wall_bb = __ballot(occ && s.top == WALL);
    uint64_t cap_bb = __ballot(occ && s.top == CAP);
    uint32_t stones = s.to_move == 0 ? s.ws : s.bs;
    uint32_t caps = s.to_move == 0 ? s.wc : s.bc;
    int base = 0;
    const int items = 4 * g.nsq;  // (square in col-major order) × (Up, Down, Left, Right)
    for (int r0 = 0; r0 < items; r0 += 64) {
        int j = r0 + lane;
        bool valid = j < items;
        int jj = valid ? j : 0;
        int x = jj / (4 * n), y = (jj >> 2) % n, d = jj & 3;
        int sq = y * n + x;
        uint32_t h = (uint32_t)__shfl((int)s.height, sq);
        uint32_t tp = (uint32_t)__shfl((int)s.top, sq);
        int kind = 0;  // 0 nothing, 1 opening flat, 2 places, 3 spreads
        int max_carry = 0, free_run = 0;
        bool smash = false;
        if (valid) {
            bool sq_occ = (occ_bb >> sq) & 1ull;
            if (swapped) {
                if (d == 0 && !sq_occ) kind = 1;
            } else if (!sq_occ) {
                if (d == 0) kind = 2;
            } else if ((own_bb >> sq) & 1ull) {
                kind = 3;
                max_carry = (int)h < n ? (int)h : n;
                // squares in direction d that accept any drop before the edge / a cap / a wall (the walk of move_gen.rs:66-87),
                // without a per-lane loop: the first blocker on the ray is a find-first-bit of (caps | walls) under the ray's mask
                const int room = d == UP ? n - 1 - y : d == DOWN ? y : d == LEFT ? x : n - 1 - x;
                const uint64_t below = (1ull << sq) - 1ull, above = ~((2ull << sq) - 1ull);
                const uint64_t line = (d == UP || d == DOWN) ? (g.col0 << x) : (g.row0 << (y * n));
                const uint64_t hits = (cap_bb | wall_bb) & line & ((d == UP || d == RIGHT) ? above : below);
                free_run = room;
                if (hits) {
                    const int pos = (d == UP || d == RIGHT) ? __builtin_ctzll(hits) : 63 - __builtin_clzll(hits);
                    const int dist = (d == UP || d == RIGHT) ? pos - sq : sq - pos;
                    free_run = ((d == UP || d == DOWN) ? dist / n : dist) - 1;
                    smash = ((wall_bb >> pos) & 1ull) && tp == CAP;
                }
            }
        }
        int cnt = 0;
        if (kind == 1) cnt = 1;
        else if (kind == 2) cnt = (stones > 0 ? 2 : 0) + (caps > 0 ? 1 : 0);
        uint64_t smask = 0;
        if (kind == 3) {
            if (n <= 6) { smask = spread_mask(max_carry, free_run, smash); cnt = __popcll(smask); }
            else cnt = spread_count(max_carry, free_run, smash);
        }
        int incl = wave_inclusive_scan(cnt);
        int off = base + incl - cnt;
        if (kind == 1) {
            if (off < cap) emit(off, (uint32_t)sq | (FLAT << 6));
        } else if (kind == 2) {
            int o = off;
            if (stones > 0) {
                if (o < cap) emit(o, (uint32_t)sq | (FLAT << 6));
                if (o + 1 < cap) emit(o + 1, (uint32_t)sq | (WALL << 6));
                o += 2;
            }
            if (caps > 0 && o < cap) emit(o, (uint32_t)sq | (CAP << 6));
        } else if (kind == 3) {
            auto put = [&](int idx, uint32_t pat) {
                if (off + idx < cap) emit(off + idx, (uint32_t)sq | ((uint32_t)d << 6) | (pat << 8));
            };
            if (n <= 6) enum_spread_mask(smask, put);
            else enum_spreads(max_carry, free_run, smash, put);
        }
        base += __builtin_amdgcn_readlane(incl, 63);  // (v_readlane: no LDS round trip)
    }
    return base;
}

// game_repr (alpha-tak/src/repr/game.rs:19-51, board.rs:12-54, reserves.rs:4-28): value of channel
// c at square data (stack, height, top).  `fcd` is the wave-uniform fcd plane value.
__device__ inline float repr_value(int c, uint64_t stack, uint32_t height, uint32_t top, const WState& s, int n,
                                   int stones0, int caps0, float fcd) {
    const int bc = board_channels(n);
    if (c < 6) {
        if (!height) return 0.0f;
        uint32_t tcol = (uint32_t)((stack >> (height - 1)) & 1ull);
        int ch = 2 * (int)top + (tcol == s.to_move ? 0 : 1);
        return c == ch ? 1.0f : 0.0f;
    }
    if (c < bc) {
        int i = (c - 6) >> 1, who = (c - 6) & 1;  // i-th stone below the top
        if ((int)height < i + 2) return 0.0f;
        uint32_t col = (uint32_t)((stack >> (height - 2 - (uint32_t)i)) & 1ull);
        return ((col == s.to_move ? 0 : 1) == who) ? 1.0f : 0.0f;
    }
    c -= bc;
    bool w = s.to_move == 0;
    int my_st = w ? s.ws : s.bs, en_st = w ? s.bs : s.ws, my_cp = w ? s.wc : s.bc, en_cp = w ? s.bc : s.wc;
    if (c < stones0) return (my_st - 1 == c) ? 1.0f : 0.0f;
    c -= stones0;
    if (c < stones0) return (en_st - 1 == c) ? 1.0f : 0.0f;
    c -= stones0;
    if (c < caps0) return (my_cp - 1 == c) ? 1.0f : 0.0f;
    c -= caps0;
    if (c < caps0) return (en_cp - 1 == c) ? 1.0f : 0.0f;
    c -= caps0;
    if (c == 0) return w ? 1.0f : 0.0f;
    return fcd;
}

__device__ inline float fcd_value(const WState& s, const Geom& g) {
    int fcd = flat_diff(s) - s.half_komi / 2;
    return (float)((double)fcd / (double)g.nsq);  // f64 division then narrowing, game.rs:35-37
}

// Every input channel is 0/1 except the last (fcd), so one square's row of game_repr is a ≤128-bit mask:
// top one-hot, buried stones, reserve one-hots, colour (alpha-tak/src/repr/{board,reserves,game}.rs).
// The lane of a square builds its own mask — no cross-lane traffic.
struct RowMask { uint32_t w[4]; };

__device__ inline RowMask ws_row_mask(const WState& s, const Geom& g) {
    int st0, cp0;
    starting_stones(g.n, st0, cp0);
    const int bc = board_channels(g.n);
    RowMask m;
    m.w[0] = m.w[1] = m.w[2] = m.w[3] = 0u;
    auto setbit = [&](int c) {
        if (c < 32) m.w[0] |= 1u << c; else if (c < 64) m.w[1] |= 1u << (c - 32); else if (c < 96) m.w[2] |= 1u << (c - 64); else m.w[3] |= 1u << (c - 96);
    };
    if (s.height) {
        uint32_t tcol = (uint32_t)((s.stack >> (s.height - 1)) & 1ull);
        setbit(2 * (int)s.top + (tcol == s.to_move ? 0 : 1));
        const int depth = (int)s.height - 1 < g.n + 5 ? (int)s.height - 1 : g.n + 5;  // take(N+6).skip(1)
        for (int i = 0; i < depth; i++) {
            uint32_t col = (uint32_t)((s.stack >> (s.height - 2 - (uint32_t)i)) & 1ull);
            setbit(6 + 2 * i + (col == s.to_move ? 0 : 1));
        }
    }
    const bool w = s.to_move == 0;
    const int my_st = w ? s.ws : s.bs, en_st = w ? s.bs : s.ws, my_cp = w ? s.wc : s.bc, en_cp = w ? s.bc : s.wc;
    if (my_st > 0 && my_st <= st0) setbit(bc + my_st - 1);
    if (en_st > 0 && en_st <= st0) setbit(bc + st0 + en_st - 1);
    if (my_cp > 0 && my_cp <= cp0) setbit(bc + 2 * st0 + my_cp - 1);
    if (en_cp > 0 && en_cp <= cp0) setbit(bc + 2 * st0 + cp0 + en_cp - 1);
    if (w) setbit(bc + 2 * st0 + 2 * cp0);
    return m;
}

// channels 4k .. 4k+3 of the row as floats; channel C-1 is the fcd plane
__device__ inline float4 row_mask_value(const RowMask& m, int k, int C, float fcd) {
    const int c0 = k << 2;
    const uint32_t word = c0 < 32 ? m.w[0] : c0 < 64 ? m.w[1] : c0 < 96 ? m.w[2] : m.w[3];
    const uint32_t nib = (word >> (c0 & 31)) & 15u;
    float4 v;
    v.x = (nib & 1u) ? 1.0f : 0.0f;
    v.y = (nib & 2u) ? 1.0f : 0.0f;
    v.z = (nib & 4u) ? 1.0f : 0.0f;
    v.w = (nib & 8u) ? 1.0f : 0.0f;
    const int r = C - 1 - c0;
    if (r == 0) v.x = fcd; else if (r == 1) v.y = fcd; else if (r == 2) v.z = fcd; else if (r == 3) v.w = fcd;
    return v;
}

// Write the encoded planes of one game, fully coalesced.  NCHW: out[c*nsq + sq] (the reference
// tensor); NHWC: out[sq*cstride + c] with channels C..cstride-1 zero (the layout the conv kernels
// consume, rows padded to a multiple of 8 channels).
// CS: the NHWC row stride as a compile-time constant (0 = `cstride`): the division by cstride/4 of every store round then
// is a multiply-shift instead of a ≈ 20-instruction software division
template <bool NHWC, int CS = 0>
__device__ inline void ws_encode(const WState& s, const Geom& g, float* out, int cstride) {
    if (CS) cstride = CS;
    const int lane = lane_id();
    const int C = input_channels(g.n);
    if (!NHWC) cstride = C;
    int st0, cp0;
    starting_stones(g.n, st0, cp0);
    float fcd = fcd_value(s, g);
    if (NHWC && (cstride & 3) == 0 && cstride <= 128) {
        // Every lane builds the ≤128-bit channel mask of its own square; the planes of the position are then written as
        // one linear run of float4: lane i of a round owns float4 number idx = round·64 + i = (square, channel quad) and
        // fetches the mask word it needs from the square's lane — 1 KB contiguous per store instruction instead of 25 lanes
        // writing 16 B each at a pitch of one row (the encode was 64 % of the fused board pass: 3.69 → see DESIGN.md).
        RowMask m = ws_row_mask(s, g);
        const int per_sq = cstride >> 2;
        const int total = g.nsq * per_sq;
        float4* dst = (float4*)out;
        // channels c0..c0+3 as floats from the 32-bit mask word that holds them; the fcd plane is the last channel
        auto quad = [&](uint32_t word, int c0) {
            const uint32_t nib = word >> (c0 & 31);
            // bit → 0.0f / 1.0f without a compare: sign-extend the bit over the word and mask the bits of 1.0f
            auto one = [](uint32_t bits, int b) { return __uint_as_float((uint32_t)((int32_t)(bits << (31 - b)) >> 31) & 0x3f800000u); };
            float4 v;
            v.x = one(nib, 0);
            v.y = one(nib, 1);
            v.z = one(nib, 2);
            v.w = one(nib, 3);
            const int r = C - 1 - c0;
            if (r == 0) v.x = fcd; else if (r == 1) v.y = fcd; else if (r == 2) v.z = fcd; else if (r == 3) v.w = fcd;
            return v;
        };
        // words 1-3 of the mask are the same in every lane: scalars, selected by compares (indexing m.w[] by a lane-dependent
        // channel made hipcc park the mask in LDS and branch around the read in every round)
        const uint32_t m1 = uni(m.w[1]), m2 = uni(m.w[2]), m3 = uni(m.w[3]);
        auto const_word = [&](int c0) { return c0 < 64 ? m1 : c0 < 96 ? m2 : m3; };
        // Only word 0 of a mask belongs to the square (the board channels, ≤ 32 of them up to 8×8, come first): it is fetched
        // from the square's lane (every lane takes part in the shuffle); words 1-3 hold reserves and colour alone — the same
        // in every lane's own mask
        auto round = [&](int idx) {
            const int sq = idx / per_sq, k = idx - sq * per_sq;
            const int c0 = k << 2;
            const uint32_t w0 = (uint32_t)__shfl((int)m.w[0], sq);
            return quad(c0 < 32 ? w0 : const_word(c0), c0);
        };
        const int full = total & ~63, rem = total - full;
        for (int i0 = 0; i0 < full; i0 += 64) dst[i0 + lane] = round(i0 + lane);
        if (rem && rem <= per_sq - 8) {
            // the last few quads (2 of the 450 on 5×5 at 72 channels) are reserve / colour / fcd channels of the last square:
            // no square to look up
            const int c0 = (per_sq - rem + lane) << 2;
            const int cc = c0 < 4 * per_sq ? c0 : 4 * per_sq - 4;
            const float4 v = quad(const_word(cc), cc);
            if (lane < rem) dst[full + lane] = v;
        } else if (rem) {
            const int idx = full + lane;
            const float4 v = round(idx < total ? idx : total - 1);
            if (idx < total) dst[idx] = v;
        }
        return;
    }
    const int total = cstride * g.nsq;
    for (int e0 = 0; e0 < total; e0 += 64) {
        int e = e0 + lane;
        int ee = e < total ? e : 0;
        int sq = NHWC ? ee / cstride : ee % g.nsq;
        int c = NHWC ? ee % cstride : ee / g.nsq;
        uint64_t stk = shfl64(s.stack, sq);
        uint32_t h = (uint32_t)__shfl((int)s.height, sq);
        uint32_t tp = (uint32_t)__shfl((int)s.top, sq);
        float v = c < C ? repr_value(c, stk, h, tp, s, g.n, st0, cp0, fcd) : 0.0f;
        if (e < total) out[e] = v;
    }
}

// move_index, alpha-tak/src/search/move_map.rs:19-48.  lut5 = 25*4*32 entries (sq, dir, 5-bit
// pattern) → index into the legacy 1575 table (−1 = the reference panics), built on the host by rule.
__device__ inline int move_index_dev(uint32_t mv, int n, bool legacy5, const int16_t* __restrict__ lut5) {
    uint32_t sq = mv & 63u, f = (mv >> 6) & 3u, pat = (mv >> 8) & 0xffu;
    if ((int)sq >= n * n) return -1;  // not a square of this board (garbage from the caller must not index past the LUT)
    if (legacy5) {
        int row = (int)sq / 5, col = (int)sq % 5;
        if (pat == 0) return f > 2 ? -1 : (col * 5 + row) * 3 + (int)f;
        if (pat & 7u) return -1;
        return (int)lut5[((int)sq * 4 + (int)f) * 32 + (int)(pat >> 3)];
    }
    int row = (int)sq / n, col = (int)sq % n;
    int channel;
    if (pat == 0) channel = (int)f;
    else {
        int pattern_offset = (int)(pat >> (8 - n)) - 1;
        int d = f == UP ? 0 : f == RIGHT ? 1 : f == DOWN ? 2 : 3;
        channel = 3 + pattern_offset + ((1 << n) - 2) * d;
    }
    return channel * n * n + row * n + col;
}

}  // namespace tg
