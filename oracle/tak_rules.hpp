// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; nothing under tak_amd/ may
// include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg use it (as the checker / the reported CPU baseline).
//
// CPU restatement of the reference's Tak rules (ViliamVadocz/tak, `tak` crate), written from
// the source text; each function cites the reference file:line it follows.  Array-of-structs
// state with one little stack per square, exactly like the reference's Game<N>/Board<N>/Tile.
//
// Third-party dependency restated here: takparse 0.5.5 (Cargo.lock:1059-1062) — Move, Square,
// Direction, Pattern, PTN text, TPS text.  Its source is NOT under /root/reference; semantics
// are anchored on the reference's call sites and pinned by tak/tests/{perft,wins,tps}.rs.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../include/takgpu.h"  // packed-state layout + enum values only (no code)

namespace orc {

constexpr int MAXN = 6;  // packed states hold N ≤ 6 (max stack height 2·31 = 62 < 64)
enum : uint8_t { FLAT = 0, WALL = 1, CAP = 2 };
enum : uint8_t { WHITE = 0, BLACK = 1 };
// Direction numbering = iteration order of move_gen.rs:64 [Up, Down, Left, Right]
enum : uint8_t { UP = 0, DOWN = 1, LEFT = 2, RIGHT = 3 };

// tak/src/tile.rs:6-10.  `piece` is the type of the top stone; colours bottom→top.
struct Tile {
    uint8_t piece = FLAT;
    uint8_t len = 0;
    uint8_t stack[64];
    bool empty() const { return len == 0; }
    int size() const { return len; }
    bool operator==(const Tile& o) const {
        return piece == o.piece && len == o.len && std::memcmp(stack, o.stack, len) == 0;
    }
};

// takparse::Move restated: square + Place(piece) | Spread(direction, drop counts)
struct Move {
    uint8_t col = 0, row = 0;
    bool spread = false;
    uint8_t piece = FLAT;
    uint8_t dir = UP;
    uint8_t ndrops = 0;
    uint8_t drops[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int count_pieces() const {  // Pattern::count_pieces
        int c = 0;
        for (int i = 0; i < ndrops; i++) c += drops[i];
        return c;
    }
    // Pattern::mask(): MSB first, one bit per carried stone, 1 closes a drop group.
    uint8_t mask() const {
        uint32_t m = 0;
        int pos = 0;
        for (int i = 0; i < ndrops; i++) {
            pos += drops[i];
            m |= 1u << (8 - pos);
        }
        return (uint8_t)m;
    }
    bool operator==(const Move& o) const {
        if (col != o.col || row != o.row || spread != o.spread) return false;
        if (!spread) return piece == o.piece;
        return dir == o.dir && mask() == o.mask();
    }
};

inline TgMove encode_move(const Move& m, int n) {
    uint16_t sq = (uint16_t)(m.row * n + m.col);
    if (!m.spread) return (TgMove)(sq | (m.piece << 6));
    return (TgMove)(sq | (m.dir << 6) | (m.mask() << 8));
}

inline Move decode_move(TgMove code, int n) {
    Move m;
    int sq = code & 63;
    m.row = (uint8_t)(sq / n);
    m.col = (uint8_t)(sq % n);
    uint8_t pat = (uint8_t)(code >> 8);
    if (pat == 0) {
        m.spread = false;
        m.piece = (code >> 6) & 3;
    } else {
        m.spread = true;
        m.dir = (code >> 6) & 3;
        int run = 0;
        int total = 8 - __builtin_ctz(pat);
        for (int i = 0; i < total; i++) {
            run++;
            if (pat & (0x80 >> i)) {
                m.drops[m.ndrops++] = (uint8_t)run;
                run = 0;
            }
        }
    }
    return m;
}

// tak/src/game.rs:10-20
inline bool default_starting_stones(int width, int& stones, int& caps) {
    switch (width) {
        case 3: stones = 10; caps = 0; return true;
        case 4: stones = 15; caps = 0; return true;
        case 5: stones = 21; caps = 1; return true;
        case 6: stones = 30; caps = 1; return true;
        case 7: stones = 40; caps = 2; return true;
        case 8: stones = 50; caps = 2; return true;
    }
    return false;
}

constexpr uint8_t REVERSIBLE_PLIES = 50;  // game.rs:22

// tak/src/game.rs:24-35 + board.rs:7-10
struct Game {
    int n = 5;
    Tile board[MAXN][MAXN];  // [row y][col x], board.rs:24-27
    uint8_t to_move = WHITE;
    uint16_t ply = 0;
    uint8_t white_stones = 0, white_caps = 0, black_stones = 0, black_caps = 0;
    int8_t half_komi = 0;
    uint8_t reversible_plies = 0;

    // game.rs:37-54 (Default) / :58-63 (with_komi) / :67-72 (with_half_komi)
    static Game start(int n, int half_komi = 0) {
        Game g;
        g.n = n;
        int s = 0, c = 0;
        default_starting_stones(n, s, c);
        g.white_stones = g.black_stones = (uint8_t)s;
        g.white_caps = g.black_caps = (uint8_t)c;
        g.half_komi = (int8_t)half_komi;
        return g;
    }

    bool is_swapped() const { return ply < 2; }                                  // game.rs:84-86
    uint8_t color() const { return is_swapped() ? (to_move ^ 1) : to_move; }     // game.rs:88-94
    void get_counts(int& stones, int& caps) const {                              // game.rs:96-101
        if (to_move == WHITE) { stones = white_stones; caps = white_caps; }
        else { stones = black_stones; caps = black_caps; }
    }
    void dec_stones() {                                                          // game.rs:103-109
        if ((to_move == WHITE) ^ is_swapped()) white_stones -= 1; else black_stones -= 1;
    }
    void dec_caps() {                                                            // game.rs:111-116
        if (to_move == WHITE) white_caps -= 1; else black_caps -= 1;
    }
    bool has(int col, int row) const { return col >= 0 && row >= 0 && col < n && row < n; }  // board.rs:40-43

    // Tile::stack, tile.rs:28-45
    static int tile_stack(Tile& t, uint8_t piece, uint8_t color) {
        if (t.piece == WALL) { if (piece != CAP) return TG_PLAY_STACK_WALL; }
        else if (t.piece == CAP) return TG_PLAY_STACK_CAP;
        t.piece = piece;
        t.stack[t.len++] = color;
        return TG_PLAY_OK;
    }

    // execute_place, game.rs:147-169
    int execute_place(int col, int row, uint8_t piece) {
        int stones, caps;
        get_counts(stones, caps);
        if (!has(col, row)) return TG_PLAY_OUT_OF_BOUNDS;
        Tile& t = board[row][col];
        if (!t.empty()) return TG_PLAY_ALREADY_OCCUPIED;
        if (piece == CAP && caps == 0) return TG_PLAY_NO_CAPSTONE;
        if ((piece == FLAT || piece == WALL) && stones == 0) return TG_PLAY_NO_STONES;
        if (is_swapped() && (piece == WALL || piece == CAP)) return TG_PLAY_OPENING_NON_FLAT;
        t.piece = piece;
        t.len = 1;
        t.stack[0] = color();
        if (piece == FLAT || piece == WALL) dec_stones(); else dec_caps();
        return TG_PLAY_OK;
    }

    // execute_spread, game.rs:171-209 (+ Tile::take, tile.rs:49-63)
    int execute_spread(const Move& m) {
        if (!has(m.col, m.row)) return TG_PLAY_OUT_OF_BOUNDS;
        Tile& src = board[m.row][m.col];
        if (src.empty()) return TG_PLAY_EMPTY_SQUARE;
        if (src.stack[src.len - 1] != color()) return TG_PLAY_STACK_NOT_OWNED;
        int amount = m.count_pieces();
        if (amount == 0) return TG_PLAY_TAKE_ZERO;
        if (amount > n) return TG_PLAY_TAKE_CARRY_LIMIT;
        if (amount > src.size()) return TG_PLAY_TAKE_STACK_SIZE;
        // take: carry ordered top→bottom, source keeps the rest, its piece resets to Flat
        uint8_t carry[MAXN];
        for (int i = 0; i < amount; i++) carry[i] = src.stack[src.len - 1 - i];
        uint8_t piece = src.piece;
        src.len = (uint8_t)(src.len - amount);
        src.piece = FLAT;
        // pieces = [piece, Flat, Flat …]; both popped from the back: bottom of carry first,
        // and only the last stone dropped carries the original top type.
        uint8_t pieces[MAXN];
        pieces[0] = piece;
        for (int i = 1; i < amount; i++) pieces[i] = FLAT;
        int left = amount;
        int col = m.col, row = m.row;
        for (int d = 0; d < m.ndrops; d++) {
            // Square::checked_step: Up row+1, Down row-1, Left col-1, Right col+1
            switch (m.dir) {
                case UP: row += 1; break;
                case DOWN: row -= 1; break;
                case LEFT: col -= 1; break;
                default: col += 1; break;
            }
            if (!has(col, row)) return TG_PLAY_SPREAD_OUT_OF_BOUNDS;
            for (int k = 0; k < m.drops[d]; k++) {
                left--;
                int err = tile_stack(board[row][col], pieces[left], carry[left]);
                if (err) return err;
            }
        }
        return TG_PLAY_OK;
    }

    // Game::play, game.rs:121-130.  On error the state may be corrupt (as in the reference).
    int play(const Move& m) {
        int err = m.spread ? execute_spread(m) : execute_place(m.col, m.row, m.piece);
        if (err) return err;
        if (!m.spread) reversible_plies = 0; else reversible_plies += 1;  // update_reversible :211-218
        ply += 1;
        to_move ^= 1;
        return TG_PLAY_OK;
    }

    // Board::full, board.rs:61-63
    bool full() const {
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) if (board[y][x].empty()) return false;
        return true;
    }
    // Board::flat_diff, board.rs:65-75
    int flat_diff() const {
        int d = 0;
        for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
            const Tile& t = board[y][x];
            if (!t.empty() && t.piece == FLAT) d += (t.stack[t.len - 1] == WHITE) ? 1 : -1;
        }
        return d;
    }
    // Board::find_paths_recursive, board.rs:94-113
    void fill(int x, int y, uint8_t color, bool seen[MAXN][MAXN]) const {
        if (y >= n || x >= n || y < 0 || x < 0 || seen[y][x]) return;
        const Tile& t = board[y][x];
        if (t.empty()) return;
        if (t.stack[t.len - 1] == color && (t.piece == FLAT || t.piece == CAP)) {
            seen[y][x] = true;
            fill(x + 1, y, color, seen);
            fill(x, y + 1, color, seen);
            fill(x - 1, y, color, seen);
            fill(x, y - 1, color, seen);
        }
    }
    // Board::find_paths, board.rs:77-92
    bool find_paths(uint8_t color) const {
        bool seen[MAXN][MAXN];
        std::memset(seen, 0, sizeof seen);
        for (int x = 0; x < n; x++) fill(x, 0, color, seen);
        for (int x = 0; x < n; x++) if (seen[n - 1][x]) return true;
        std::memset(seen, 0, sizeof seen);
        for (int y = 0; y < n; y++) fill(0, y, color, seen);
        for (int y = 0; y < n; y++) if (seen[y][n - 1]) return true;
        return false;
    }

    // Game::result, game.rs:220-267
    uint8_t result() const {
        uint8_t other = to_move ^ 1;
        if (find_paths(other)) return other == WHITE ? TG_WHITE_ROAD : TG_BLACK_ROAD;
        if (find_paths(to_move)) return to_move == WHITE ? TG_WHITE_ROAD : TG_BLACK_ROAD;
        if ((white_caps == 0 && white_stones == 0) || (black_caps == 0 && black_stones == 0) || full()) {
            int fd = flat_diff();
            int k = half_komi / 2;  // i8 division truncates toward zero, like C
            if (fd > k) return TG_WHITE_FLAT;
            if (fd < k) return TG_BLACK_FLAT;
            if (half_komi % 2 == 0) return TG_DRAW;
            return TG_BLACK_FLAT;
        }
        if (reversible_plies >= REVERSIBLE_PLIES) return TG_DRAW_REVERSIBLE;
        return TG_ONGOING;
    }

    // Game::possible_moves, move_gen.rs:7-30
    void possible_moves(std::vector<Move>& moves) const {
        moves.clear();
        if (is_swapped()) {  // add_opening_moves, move_gen.rs:32-41
            for (int x = 0; x < n; x++) for (int y = 0; y < n; y++)
                if (board[y][x].empty()) { Move m; m.col = (uint8_t)x; m.row = (uint8_t)y; m.piece = FLAT; moves.push_back(m); }
            return;
        }
        for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) {
            const Tile& t = board[y][x];
            if (!t.empty()) {
                if (t.stack[t.len - 1] == color()) add_spreads(x, y, moves);
            } else {
                add_places(x, y, moves);
            }
        }
    }
    // add_places, move_gen.rs:43-52
    void add_places(int x, int y, std::vector<Move>& moves) const {
        int stones, caps;
        get_counts(stones, caps);
        Move m; m.col = (uint8_t)x; m.row = (uint8_t)y;
        if (stones > 0) { m.piece = FLAT; moves.push_back(m); m.piece = WALL; moves.push_back(m); }
        if (caps > 0) { m.piece = CAP; moves.push_back(m); }
    }
    // add_spreads, move_gen.rs:54-102: explicit LIFO stack of partial spreads, as the reference
    void add_spreads(int x, int y, std::vector<Move>& moves) const {
        struct Spread { int col, row, hand; uint8_t nd; uint8_t drops[MAXN]; };
        const Tile& tile = board[y][x];
        int max_carry = tile.size() < n ? tile.size() : n;
        static const uint8_t dirs[4] = {UP, DOWN, LEFT, RIGHT};
        std::vector<Spread> spreads;
        for (int di = 0; di < 4; di++) {
            uint8_t dir = dirs[di];
            for (int pickup = 1; pickup <= max_carry; pickup++) {
                spreads.clear();
                Spread s0; s0.col = x; s0.row = y; s0.hand = pickup; s0.nd = 0;
                spreads.push_back(s0);
                while (!spreads.empty()) {
                    Spread sp = spreads.back();
                    spreads.pop_back();
                    if (sp.hand == 0) {
                        Move m; m.col = (uint8_t)x; m.row = (uint8_t)y; m.spread = true; m.dir = dir;
                        m.ndrops = sp.nd;
                        for (int i = 0; i < sp.nd; i++) m.drops[i] = sp.drops[i];
                        moves.push_back(m);
                        continue;
                    }
                    int nc = sp.col, nr = sp.row;
                    switch (dir) { case UP: nr++; break; case DOWN: nr--; break; case LEFT: nc--; break; default: nc++; }
                    if (!has(nc, nr)) continue;
                    uint8_t np = board[nr][nc].piece;  // empty tile ⇒ Flat (Tile::default)
                    bool can_drop = np == FLAT ? true : np == CAP ? false : (sp.hand == 1 && tile.piece == CAP);
                    if (!can_drop) continue;
                    for (int drop = 1; drop <= sp.hand; drop++) {
                        Spread nx = sp;
                        nx.drops[nx.nd++] = (uint8_t)drop;
                        nx.col = nc; nx.row = nr; nx.hand = sp.hand - drop;
                        spreads.push_back(nx);
                    }
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------
// PTN move text (takparse `Move: FromStr + Display`; call sites game.rs:79, example.rs:95,121)
// ---------------------------------------------------------------------------------------
inline bool parse_ptn(const std::string& s_in, Move& m) {
    std::string s = s_in;
    while (!s.empty() && (s.back() == '*' || s.back() == '\'' || s.back() == '?' || s.back() == '!')) s.pop_back();
    size_t i = 0;
    int count = -1;
    m = Move();
    if (i < s.size() && s[i] >= '1' && s[i] <= '9') { count = s[i] - '0'; i++; }
    uint8_t piece = FLAT;
    bool piece_given = false;
    if (i < s.size() && (s[i] == 'F' || s[i] == 'S' || s[i] == 'C')) {
        piece = s[i] == 'F' ? FLAT : s[i] == 'S' ? WALL : CAP;
        piece_given = true;
        i++;
    }
    if (i + 2 > s.size()) return false;
    if (s[i] < 'a' || s[i] >= 'a' + MAXN || s[i + 1] < '1' || s[i + 1] >= '1' + MAXN) return false;
    m.col = (uint8_t)(s[i] - 'a');
    m.row = (uint8_t)(s[i + 1] - '1');
    i += 2;
    if (i == s.size()) {
        if (count != -1) return false;
        m.spread = false;
        m.piece = piece;
        return true;
    }
    if (piece_given) return false;
    char d = s[i++];
    m.spread = true;
    if (d == '+') m.dir = UP; else if (d == '-') m.dir = DOWN; else if (d == '<') m.dir = LEFT; else if (d == '>') m.dir = RIGHT; else return false;
    if (count == -1) count = 1;
    int total = 0;
    while (i < s.size()) {
        if (s[i] < '1' || s[i] > '8' || m.ndrops >= 8) return false;
        m.drops[m.ndrops++] = (uint8_t)(s[i] - '0');
        total += s[i] - '0';
        i++;
    }
    if (m.ndrops == 0) { m.drops[0] = (uint8_t)count; m.ndrops = 1; total = count; }
    return total == count;
}

inline std::string format_ptn(const Move& m) {
    std::string s;
    if (!m.spread) {
        if (m.piece == WALL) s += 'S'; else if (m.piece == CAP) s += 'C';
        s += (char)('a' + m.col);
        s += (char)('1' + m.row);
        return s;
    }
    int count = m.count_pieces();
    if (count > 1) s += (char)('0' + count);
    s += (char)('a' + m.col);
    s += (char)('1' + m.row);
    s += m.dir == UP ? '+' : m.dir == DOWN ? '-' : m.dir == LEFT ? '<' : '>';
    if (m.ndrops > 1) for (int i = 0; i < m.ndrops; i++) s += (char)('0' + m.drops[i]);
    return s;
}

// Game → TPS text (tak/src/tps.rs:7-35 + takparse `Tps: Display`): rows top→bottom, every
// empty square printed as its own "x" (the reference builds EmptySquares(1) per tile).
inline std::string to_tps(const Game& g) {
    std::string s;
    for (int y = g.n - 1; y >= 0; y--) {
        for (int x = 0; x < g.n; x++) {
            const Tile& t = g.board[y][x];
            if (t.empty()) s += 'x';
            else {
                for (int i = 0; i < t.len; i++) s += t.stack[i] == WHITE ? '1' : '2';
                if (t.piece == WALL) s += 'S'; else if (t.piece == CAP) s += 'C';
            }
            if (x + 1 < g.n) s += ',';
        }
        if (y > 0) s += '/';
    }
    s += ' ';
    s += g.to_move == WHITE ? '1' : '2';
    s += ' ';
    s += std::to_string(1 + g.ply / 2);
    return s;
}

// ---------------------------------------------------------------------------------------
// packed state ⇄ Game (layout: include/takgpu.h)
// ---------------------------------------------------------------------------------------
inline size_t state_bytes(int n) { return n <= 5 ? TG_STATE5_BYTES : TG_STATE6_BYTES; }

inline void pack(const Game& g, uint8_t* out) {
    size_t bytes = state_bytes(g.n);
    std::memset(out, 0, bytes);
    int slots = g.n <= 5 ? 25 : 36;
    uint64_t* stack = (uint64_t*)out;
    uint8_t* meta = out + 8 * slots;
    for (int y = 0; y < g.n; y++) for (int x = 0; x < g.n; x++) {
        const Tile& t = g.board[y][x];
        int sq = y * g.n + x;
        uint64_t bits = 0;
        for (int i = 0; i < t.len; i++) bits |= (uint64_t)t.stack[i] << i;
        stack[sq] = bits;
        meta[sq] = TG_META(t.len, t.len ? t.piece : 0);
    }
    TgHeader* h = (TgHeader*)(out + bytes - sizeof(TgHeader));
    h->n = (uint8_t)g.n; h->to_move = g.to_move; h->ply = g.ply;
    h->white_stones = g.white_stones; h->white_caps = g.white_caps;
    h->black_stones = g.black_stones; h->black_caps = g.black_caps;
    h->half_komi = g.half_komi; h->reversible_plies = g.reversible_plies;
}

inline Game unpack(const uint8_t* in, int n) {
    Game g;
    g.n = n;
    size_t bytes = state_bytes(n);
    int slots = n <= 5 ? 25 : 36;
    const uint64_t* stack = (const uint64_t*)in;
    const uint8_t* meta = in + 8 * slots;
    for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        Tile& t = g.board[y][x];
        int sq = y * n + x;
        t.len = TG_META_HEIGHT(meta[sq]);
        t.piece = t.len ? TG_META_TOP(meta[sq]) : FLAT;
        for (int i = 0; i < t.len; i++) t.stack[i] = (uint8_t)((stack[sq] >> i) & 1);
    }
    const TgHeader* h = (const TgHeader*)(in + bytes - sizeof(TgHeader));
    g.to_move = h->to_move; g.ply = h->ply;
    g.white_stones = h->white_stones; g.white_caps = h->white_caps;
    g.black_stones = h->black_stones; g.black_caps = h->black_caps;
    g.half_komi = h->half_komi; g.reversible_plies = h->reversible_plies;
    return g;
}

// perf_count of tak/tests/perft.rs:3-18
inline uint64_t perft(const Game& g, int depth) {
    if (depth == 0 || g.result() != TG_ONGOING) return 1;
    std::vector<Move> moves;
    g.possible_moves(moves);
    if (depth == 1) return moves.size();
    uint64_t total = 0;
    for (const Move& m : moves) {
        Game c = g;
        c.play(m);
        total += perft(c, depth - 1);
    }
    return total;
}

}  // namespace orc
