// ORACLE — TEST INFRASTRUCTURE ONLY (see tak_rules.hpp).
// CPU restatement of the reference's state encoder (alpha-tak/src/repr/*.rs) and of the
// move → policy-index map (alpha-tak/src/search/move_map.rs).
#pragma once
#include <string>
#include <vector>

#include "tak_rules.hpp"

namespace orc {

constexpr int STACK_DEPTH_BEYOND_CARRY = 6;                                       // repr/board.rs:4
inline int board_channels(int n) { return (n + 2 + STACK_DEPTH_BEYOND_CARRY) * 2; }  // board.rs:6-8
inline int input_channels(int n) {                                                // repr/game.rs:12-15
    int s = 0, c = 0;
    default_starting_stones(n, s, c);
    return board_channels(n) + 1 + 1 + 2 * s + 2 * c;
}
inline int possible_patterns(int n) { return (1 << n) - 2; }                      // move_map.rs:15-17
inline int move_channels(int n) { return 3 + 4 * possible_patterns(n); }          // repr/moves.rs:20-25
inline int output_size(int n) { return n * n * move_channels(n); }                // repr/moves.rs:29-31
inline int possible_moves_count(int n) {                                          // repr/moves.rs:6-16
    switch (n) {
        case 3: return 126;
        case 4: return 480;
        case 5: return 1575;
        case 6: return 4572;
        case 7: return 12495;
        case 8: return 32704;
    }
    return -1;
}

// game_repr, repr/game.rs:19-51; out = C_in × N × N f32, index c*N² + row*N + col.
inline void game_repr(const Game& g, float* out) {
    const int n = g.n, nn = n * n;
    const int cin = input_channels(n);
    for (int i = 0; i < cin * nn; i++) out[i] = 0.0f;
    // board_repr, repr/board.rs:12-54
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        int off = n * y + x;
        const Tile& t = g.board[y][x];
        if (!t.empty()) {
            int ch = (t.piece == FLAT ? 0 : t.piece == WALL ? 2 : 4) + (t.stack[t.len - 1] == g.to_move ? 0 : 1);
            out[off + nn * ch] = 1.0f;
        }
        // stack.iter().rev().take(N + 6).skip(1).enumerate()
        int taken = 0;
        for (int k = t.len - 1; k >= 0 && taken < n + STACK_DEPTH_BEYOND_CARRY; k--, taken++) {
            if (taken == 0) continue;
            int i = taken - 1;
            int ch = 6 + 2 * i + (t.stack[k] == g.to_move ? 0 : 1);
            out[off + nn * ch] = 1.0f;
        }
    }
    // reserves_repr, repr/reserves.rs:4-28; order game.rs:39-48: my stones, en stones, my caps, en caps
    int stones = 0, caps = 0;
    default_starting_stones(n, stones, caps);
    int c = board_channels(n);
    auto one_hot = [&](int value, int max) {
        if (value > 0 && value <= max) for (int i = 0; i < nn; i++) out[(c + value - 1) * nn + i] = 1.0f;
        c += max;
    };
    bool w = g.to_move == WHITE;
    one_hot(w ? g.white_stones : g.black_stones, stones);
    one_hot(w ? g.black_stones : g.white_stones, stones);
    one_hot(w ? g.white_caps : g.black_caps, caps);
    one_hot(w ? g.black_caps : g.white_caps, caps);
    // colour layer game.rs:28-32
    if (w) for (int i = 0; i < nn; i++) out[c * nn + i] = 1.0f;
    c += 1;
    // fcd layer game.rs:35-37: i8 arithmetic, f64 division, stored as f32
    int fcd = g.flat_diff() - g.half_komi / 2;
    double rel = (double)fcd / (double)nn;
    for (int i = 0; i < nn; i++) out[c * nn + i] = (float)rel;
}

// ---------------------------------------------------------------------------------------
// move_index, move_map.rs:19-48.
// 5×5: position in the legacy 1575-string table POSSIBLE_MOVES_IN_5S (move_map.rs:51-201).  The
// table is regenerated here by rule (never copied): 75 placements (col a→e outer, row 1→5 inner,
// flat,S,C), then per square (col outer, row inner), directions in the order < - > + skipping
// those with no room, pickup 1→5, and for each pickup every composition into ≤ room parts with
// the first part descending, recursively.  tests/ pins the regenerated list by sha256 and, when
// /root/reference is present, against the source table itself.
// ---------------------------------------------------------------------------------------
inline void legacy5_gen_compositions(int hand, int room, std::vector<uint8_t>& cur, std::vector<std::vector<uint8_t>>& out) {
    if (hand == 0) { out.push_back(cur); return; }
    if (room == 0) return;
    for (int first = hand; first >= 1; first--) {
        cur.push_back((uint8_t)first);
        legacy5_gen_compositions(hand - first, room - 1, cur, out);
        cur.pop_back();
    }
}

inline std::vector<Move> legacy5_build() {
    std::vector<Move> table;
    const int n = 5;
    for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) for (int p = 0; p < 3; p++) {
        Move m; m.col = (uint8_t)x; m.row = (uint8_t)y; m.piece = (uint8_t)p;
        table.push_back(m);
    }
    static const uint8_t dir_order[4] = {LEFT, DOWN, RIGHT, UP};
    for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) for (int di = 0; di < 4; di++) {
        uint8_t d = dir_order[di];
        int room = d == LEFT ? x : d == DOWN ? y : d == RIGHT ? n - 1 - x : n - 1 - y;
        if (room == 0) continue;
        for (int pickup = 1; pickup <= n; pickup++) {
            std::vector<std::vector<uint8_t>> comps;
            std::vector<uint8_t> cur;
            legacy5_gen_compositions(pickup, room, cur, comps);
            for (auto& c : comps) {
                Move m; m.col = (uint8_t)x; m.row = (uint8_t)y; m.spread = true; m.dir = d;
                m.ndrops = (uint8_t)c.size();
                for (size_t i = 0; i < c.size(); i++) m.drops[i] = c[i];
                table.push_back(m);
            }
        }
    }
    return table;
}
inline const std::vector<Move>& legacy5_table() {
    static const std::vector<Move> table = legacy5_build();  // initialised once, thread-safe (C++11 static initialisation)
    return table;
}

// returns -1 where the reference panics ("could not map turn to index", move_map.rs:24).
// legacy5: the reference takes the legacy table for EVERY 5×5 move (move_map.rs:21-24) — its only 5×5 network, Net5, has the table's
// 1575 outputs.  A 5×5 network with Net6's conv head (output_size(5) = 3075 outputs) is not a reference configuration; the ABI allows
// it (TG_HEAD_CONV) and defines its index as the conv formula below (include/takgpu.h), so callers that know the head's size pass
// legacy5 = false there (move_index(m, n, policy_size)).
inline int move_index(const Move& m, int n, bool legacy5 = true) {
    if (n == 5 && legacy5) {
        static const std::vector<int> lut = [] {  // keyed by move code; thread-safe one-time initialisation
            std::vector<int> l(1 << 16, -1);
            const auto& t = legacy5_table();
            for (size_t i = 0; i < t.size(); i++) l[encode_move(t[i], 5)] = (int)i;
            return l;
        }();
        if (m.col >= 5 || m.row >= 5) return -1;
        if (!m.spread && m.piece > 2) return -1;
        return lut[encode_move(m, 5)];
    }
    int channel;
    if (!m.spread) channel = m.piece == FLAT ? 0 : m.piece == WALL ? 1 : 2;
    else {
        int pattern_offset = (m.mask() >> (8 - n)) - 1;
        int d = m.dir == UP ? 0 : m.dir == RIGHT ? 1 : m.dir == DOWN ? 2 : 3;
        channel = 3 + pattern_offset + possible_patterns(n) * d;
    }
    return channel * n * n + m.row * n + m.col;
}
// the same for a head of `policy_size` outputs: 5×5 with 1575 outputs is the legacy table, everything else the conv formula
inline int move_index(const Move& m, int n, int policy_size) { return move_index(m, n, n == 5 && policy_size == possible_moves_count(5)); }


// ---------------------------------------------------------------------------------------
// Symmetry, tak/src/symm.rs:7-97 — the 8 dihedral images of squares, directions, moves, boards, games.
// takparse's Square::rotate/mirror and Direction::rotate/mirror are not vendored; the orientation used here
// (rotate: (col,row) → (row, n-1-col), Up→Right→Down→Left; mirror: col → n-1-col, Left↔Right) is one of the
// mutually consistent choices that tak/tests/symm.rs pins.  Any consistent choice yields the same SET of 8
// images (used together for training), possibly in another order: "parity unpinned" for the order only.
// ---------------------------------------------------------------------------------------
inline void sq_rotate(int n, int& col, int& row) { int c = row, r = n - 1 - col; col = c; row = r; }
inline void sq_mirror(int n, int& col, int& row) { col = n - 1 - col; (void)row; }
inline uint8_t dir_rotate(uint8_t d) { return d == UP ? RIGHT : d == RIGHT ? DOWN : d == DOWN ? LEFT : UP; }
inline uint8_t dir_mirror(uint8_t d) { return d == LEFT ? RIGHT : d == RIGHT ? LEFT : d; }

// i-th symmetry (order of symm.rs:11-20): i<4 → rotate^i ; i≥4 → mirror then rotate^(i-4)
inline void sq_symmetry(int n, int i, int& col, int& row) {
    if (i >= 4) sq_mirror(n, col, row);
    for (int k = 0; k < (i & 3); k++) sq_rotate(n, col, row);
}
inline uint8_t dir_symmetry(int i, uint8_t d) {
    if (i >= 4) d = dir_mirror(d);
    for (int k = 0; k < (i & 3); k++) d = dir_rotate(d);
    return d;
}
inline Move move_symmetry(int n, int i, const Move& m) {  // symm.rs:40-52
    Move o = m;
    int c = m.col, r = m.row;
    sq_symmetry(n, i, c, r);
    o.col = (uint8_t)c; o.row = (uint8_t)r;
    if (m.spread) o.dir = dir_symmetry(i, m.dir);
    return o;
}
inline Game game_symmetry(int i, const Game& g) {  // symm.rs:55-97
    Game o = g;
    for (int y = 0; y < g.n; y++) for (int x = 0; x < g.n; x++) {
        int c = x, r = y;
        sq_symmetry(g.n, i, c, r);
        o.board[r][c] = g.board[y][x];
    }
    return o;
}

// Example::to_tensors, alpha-tak/src/example.rs:62-78: per symmetry the transformed game and the policy target
inline void example_symmetries(const Game& g, const std::vector<Move>& moves, const std::vector<uint32_t>& visits, int policy_size,
                               Game out_games[8], std::vector<float> out_pi[8]) {
    uint32_t total_u = 0;
    for (uint32_t v : visits) total_u += v;
    float total = (float)total_u;
    for (int i = 0; i < 8; i++) {
        out_games[i] = game_symmetry(i, g);
        out_pi[i].assign(policy_size, 0.0f);
        for (size_t k = 0; k < moves.size(); k++) {
            int idx = move_index(move_symmetry(g.n, i, moves[k]), g.n, policy_size);
            if (idx >= 0 && idx < policy_size) out_pi[i][idx] = (float)visits[k] / total;
        }
    }
}

}  // namespace orc
