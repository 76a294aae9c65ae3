// notation.hip — host-side text formats at the edge of the hot path (no device code):
//   PTN moves and TPS positions (takparse 0.5.5 `Move`/`Tps` Display + FromStr, as used by reference
//   tak/src/game.rs:79, tak/src/tps.rs:7-96) and the self-play example line of
//   alpha-tak/src/example.rs:81-133:
//       "{tps};{white_stones};{white_caps};{black_stones};{black_caps};{half_komi};{result};{move:visits,…}"
// so that examples drained from the GPU can be written in the reference's `.data` format and read back.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"

namespace tg {
namespace {

struct HostState {  // unpacked view of a TgState
    int n = 0;
    std::vector<std::vector<uint8_t>> stack;  // per square, colours bottom→top
    std::vector<uint8_t> top;                 // per square piece type
    TgHeader h{};
};

int slots_of(int n) { return n <= 5 ? 25 : 36; }
size_t bytes_of(int n) { return n <= 5 ? TG_STATE5_BYTES : TG_STATE6_BYTES; }

HostState unpack(int n, const uint8_t* st) {
    HostState s;
    s.n = n;
    const uint64_t* stk = (const uint64_t*)st;
    const uint8_t* meta = st + 8 * slots_of(n);
    s.stack.resize(n * n);
    s.top.resize(n * n);
    for (int sq = 0; sq < n * n; sq++) {
        int hgt = TG_META_HEIGHT(meta[sq]);
        s.top[sq] = hgt ? TG_META_TOP(meta[sq]) : 0;
        for (int i = 0; i < hgt; i++) s.stack[sq].push_back((uint8_t)((stk[sq] >> i) & 1));
    }
    std::memcpy(&s.h, st + bytes_of(n) - sizeof(TgHeader), sizeof(TgHeader));
    return s;
}

void pack(const HostState& s, uint8_t* st) {
    std::memset(st, 0, bytes_of(s.n));
    uint64_t* stk = (uint64_t*)st;
    uint8_t* meta = st + 8 * slots_of(s.n);
    for (int sq = 0; sq < s.n * s.n; sq++) {
        uint64_t bits = 0;
        for (size_t i = 0; i < s.stack[sq].size(); i++) bits |= (uint64_t)s.stack[sq][i] << i;
        stk[sq] = bits;
        meta[sq] = TG_META(s.stack[sq].size(), s.stack[sq].empty() ? 0 : s.top[sq]);
    }
    std::memcpy(st + bytes_of(s.n) - sizeof(TgHeader), &s.h, sizeof(TgHeader));
}

std::string format_move(int n, TgMove mv) {
    int sq = mv & 63, f = (mv >> 6) & 3, pat = mv >> 8;
    int row = sq / n, col = sq % n;
    std::string s;
    if (pat == 0) {
        if (f == 1) s += 'S'; else if (f == 2) s += 'C';
        s += (char)('a' + col);
        s += (char)('1' + row);
        return s;
    }
    int count = 8 - __builtin_ctz(pat);
    if (count > 1) s += (char)('0' + count);
    s += (char)('a' + col);
    s += (char)('1' + row);
    s += f == 0 ? '+' : f == 1 ? '-' : f == 2 ? '<' : '>';
    if (__builtin_popcount(pat) > 1) {
        int run = 0;
        for (int i = 0; i < count; i++) {
            run++;
            if (pat & (0x80 >> i)) { s += (char)('0' + run); run = 0; }
        }
    }
    return s;
}

bool parse_move(int n, const std::string& in, TgMove& out) {
    std::string s = in;
    while (!s.empty() && (s.back() == '*' || s.back() == '\'' || s.back() == '?' || s.back() == '!')) s.pop_back();
    size_t i = 0;
    int count = -1, piece = 0;
    bool piece_given = false;
    if (i < s.size() && s[i] >= '1' && s[i] <= '8') count = s[i++] - '0';
    if (i < s.size() && (s[i] == 'F' || s[i] == 'S' || s[i] == 'C')) { piece = s[i] == 'F' ? 0 : s[i] == 'S' ? 1 : 2; piece_given = true; i++; }
    if (i + 2 > s.size()) return false;
    int col = s[i] - 'a', row = s[i + 1] - '1';
    if (col < 0 || col >= n || row < 0 || row >= n) return false;
    i += 2;
    int sq = row * n + col;
    if (i == s.size()) {
        if (count != -1) return false;
        out = (TgMove)(sq | (piece << 6));
        return true;
    }
    if (piece_given) return false;
    char d = s[i++];
    int dir = d == '+' ? 0 : d == '-' ? 1 : d == '<' ? 2 : d == '>' ? 3 : -1;
    if (dir < 0) return false;
    if (count == -1) count = 1;
    int pos = 0, pat = 0, total = 0;
    bool any = false;
    while (i < s.size()) {
        if (s[i] < '1' || s[i] > '8') return false;
        int dcount = s[i++] - '0';
        pos += dcount; total += dcount; any = true;
        if (pos > 8) return false;
        pat |= 1 << (8 - pos);
    }
    if (!any) { if (count > 8) return false; pat = 1 << (8 - count); total = count; }
    if (total != count) return false;
    out = (TgMove)(sq | (dir << 6) | (pat << 8));
    return true;
}

// Tps Display: rows top→bottom, every empty square as its own "x" (the reference builds EmptySquares(1) per
// tile, tak/src/tps.rs:16-18), then side to move and move number 1 + ply/2
std::string format_tps(const HostState& s) {
    std::string out;
    for (int y = s.n - 1; y >= 0; y--) {
        for (int x = 0; x < s.n; x++) {
            int sq = y * s.n + x;
            if (s.stack[sq].empty()) out += 'x';
            else {
                for (uint8_t c : s.stack[sq]) out += c ? '2' : '1';
                if (s.top[sq] == 1) out += 'S'; else if (s.top[sq] == 2) out += 'C';
            }
            if (x + 1 < s.n) out += ',';
        }
        if (y > 0) out += '/';
    }
    out += ' ';
    out += s.h.to_move ? '2' : '1';
    out += ' ';
    out += std::to_string(1 + s.h.ply / 2);
    return out;
}

// Tps FromStr + From<Tps> for Game<N> (tak/src/tps.rs:38-96): board, side to move, ply, reserves from the board
// decimal digits only (no sign, no trailing text), at most 5 of them
static bool parse_count(const std::string& t, int& out) {
    if (t.empty() || t.size() > 5) return false;
    int v = 0;
    for (char c : t) {
        if (c < '0' || c > '9') return false;
        v = v * 10 + (c - '0');
    }
    out = v;
    return true;
}

bool parse_tps(int n, const std::string& text, HostState& s) {
    s = HostState();
    s.n = n;
    s.stack.assign(n * n, {});
    s.top.assign(n * n, 0);
    size_t sp1 = text.find(' ');
    if (sp1 == std::string::npos) return false;
    size_t sp2 = text.find(' ', sp1 + 1);
    if (sp2 == std::string::npos) return false;
    std::string board = text.substr(0, sp1), color = text.substr(sp1 + 1, sp2 - sp1 - 1), number = text.substr(sp2 + 1);
    std::vector<std::string> rows;
    size_t p = 0;
    for (;;) {
        size_t q = board.find('/', p);
        rows.push_back(board.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    if ((int)rows.size() != n) return false;
    for (int r = 0; r < n; r++) {
        int y = n - 1 - r, x = 0;
        const std::string& row = rows[r];
        size_t i = 0;
        while (i <= row.size()) {
            size_t j = row.find(',', i);
            std::string cell = row.substr(i, j == std::string::npos ? std::string::npos : j - i);
            if (cell.empty()) return false;
            if (cell[0] == 'x') {
                int run = 1;
                if (cell.size() > 1 && !parse_count(cell.substr(1), run)) return false;
                if (run < 1 || run > n) return false;
                x += run;
            } else {
                if (x >= n) return false;
                int sq = y * n + x;
                uint8_t top = 0;
                for (char c : cell) {
                    if (c == '1' || c == '2') s.stack[sq].push_back(c == '2');
                    else if (c == 'S') top = 1;
                    else if (c == 'C') top = 2;
                    else return false;
                }
                if (s.stack[sq].empty() || s.stack[sq].size() > 62) return false;
                s.top[sq] = top;
                x++;
            }
            if (j == std::string::npos) break;
            i = j + 1;
        }
        if (x != n) return false;
    }
    if (color != "1" && color != "2") return false;
    int move_number = 0;
    if (!parse_count(number, move_number) || move_number < 1 || move_number > 30000) return false;
    int stones, caps;
    starting_stones(n, stones, caps);
    int ws = stones, wc = caps, bs = stones, bc = caps;
    for (int sq = 0; sq < n * n; sq++) {
        if (s.stack[sq].empty()) continue;
        if (s.top[sq] == 2) { if (s.stack[sq].back() == 0) { ws++; wc--; } else { bs++; bc--; } }
        for (uint8_t c : s.stack[sq]) { if (c == 0) ws--; else bs--; }
    }
    if (ws < 0 || wc < 0 || bs < 0 || bc < 0) return false;  // more stones on the board than a player owns
    s.h.n = (uint8_t)n;
    s.h.to_move = color == "2";
    s.h.ply = (uint16_t)((move_number - 1) * 2 + (s.h.to_move ? 1 : 0));
    s.h.white_stones = (uint8_t)ws; s.h.white_caps = (uint8_t)wc; s.h.black_stones = (uint8_t)bs; s.h.black_caps = (uint8_t)bc;
    s.h.half_komi = 0; s.h.reversible_plies = 0;
    return true;
}

// Rust `{}` of an f32 that is one of ±1, ±0
std::string format_result(float r) {
    if (r == 0.0f) return std::signbit(r) ? "-0" : "0";
    char buf[32];
    if (r == std::floor(r) && std::fabs(r) < 1e9f) std::snprintf(buf, sizeof buf, "%d", (int)r);
    else std::snprintf(buf, sizeof buf, "%g", (double)r);
    return buf;
}

int copy_out(const std::string& s, char* buf, size_t cap) {
    if (s.size() + 1 > cap) return fail(TG_ERR_INVALID_ARG, "text buffer too small");
    std::memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // namespace
}  // namespace tg

using namespace tg;

extern "C" {

int tg_format_move(int n, TgMove mv, char* buf, size_t cap) {
    if (n < 3 || n > 6 || !buf) return fail(TG_ERR_INVALID_ARG, "tg_format_move: bad arguments");
    return copy_out(format_move(n, mv), buf, cap);
}

int tg_parse_move(int n, const char* text, TgMove* out) {
    if (n < 3 || n > 6 || !text || !out) return fail(TG_ERR_INVALID_ARG, "tg_parse_move: bad arguments");
    if (!parse_move(n, text, *out)) return fail(TG_ERR_INVALID_ARG, std::string("not a PTN move: ") + text);
    return TG_OK;
}

int tg_format_tps(int n, const void* state, char* buf, size_t cap) {
    if (n < 3 || n > 6 || !state || !buf) return fail(TG_ERR_INVALID_ARG, "tg_format_tps: bad arguments");
    return copy_out(format_tps(unpack(n, (const uint8_t*)state)), buf, cap);
}

int tg_parse_tps(int n, const char* text, void* state) {
    if (n < 3 || n > 6 || !text || !state) return fail(TG_ERR_INVALID_ARG, "tg_parse_tps: bad arguments");
    HostState s;
    if (!parse_tps(n, text, s)) return fail(TG_ERR_INVALID_ARG, std::string("not a TPS position: ") + text);
    pack(s, (uint8_t*)state);
    return TG_OK;
}

int tg_format_example(int n, const void* state, int n_moves, const TgMove* moves, const uint32_t* visits, float result, char* buf,
                      size_t cap) {
    if (n < 3 || n > 6 || !state || !buf || n_moves < 0 || (n_moves && (!moves || !visits))) return fail(TG_ERR_INVALID_ARG, "tg_format_example: bad arguments");
    HostState s = unpack(n, (const uint8_t*)state);
    std::string out = format_tps(s);
    out += ';' + std::to_string(s.h.white_stones) + ';' + std::to_string(s.h.white_caps) + ';' + std::to_string(s.h.black_stones) + ';' +
           std::to_string(s.h.black_caps) + ';' + std::to_string((int)s.h.half_komi) + ';' + format_result(result) + ';';
    for (int i = 0; i < n_moves; i++) {
        if (i) out += ',';
        out += format_move(n, moves[i]) + ':' + std::to_string(visits[i]);
    }
    return copy_out(out, buf, cap);
}

int tg_parse_example(int n, const char* line, void* state, int cap_moves, TgMove* moves, uint32_t* visits, int32_t* n_moves, float* result) {
    if (n < 3 || n > 6 || !line || !state || !n_moves || !result) return fail(TG_ERR_INVALID_ARG, "tg_parse_example: bad arguments");
    std::string s(line);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
    std::vector<std::string> f;
    size_t p = 0;
    for (;;) {
        size_t q = s.find(';', p);
        f.push_back(s.substr(p, q == std::string::npos ? std::string::npos : q - p));
        if (q == std::string::npos) break;
        p = q + 1;
    }
    if (f.size() != 8) return fail(TG_ERR_INVALID_ARG, "example line needs 8 ';'-separated fields");
    HostState hs;
    if (!parse_tps(n, f[0], hs)) return fail(TG_ERR_INVALID_ARG, "example: bad tps");
    // the fields parse as Rust's u8 / i8 / f32 `FromStr` do (example.rs:108-114): digits only, in range, nothing after them
    int r[4], komi = 0;
    for (int i = 0; i < 4; i++)
        if (!parse_count(f[1 + i], r[i]) || r[i] > 255) return fail(TG_ERR_INVALID_ARG, "example: bad reserve count");
    {
        const bool neg = !f[5].empty() && f[5][0] == '-';
        if (!parse_count(neg ? f[5].substr(1) : f[5], komi) || komi > (neg ? 128 : 127)) return fail(TG_ERR_INVALID_ARG, "example: bad half komi");
        if (neg) komi = -komi;
    }
    {
        char* end = nullptr;
        *result = std::strtof(f[6].c_str(), &end);
        if (f[6].empty() || f[6].size() > 32 || end != f[6].c_str() + f[6].size() || f[6][0] == ' ' || f[6][0] == '\t')
            return fail(TG_ERR_INVALID_ARG, "example: bad result");
    }
    hs.h.white_stones = (uint8_t)r[0]; hs.h.white_caps = (uint8_t)r[1];
    hs.h.black_stones = (uint8_t)r[2]; hs.h.black_caps = (uint8_t)r[3];
    hs.h.half_komi = (int8_t)komi;
    int k = 0;
    p = 0;
    const std::string& pol = f[7];
    while (p < pol.size()) {
        size_t q = pol.find(',', p);
        std::string pair = pol.substr(p, q == std::string::npos ? std::string::npos : q - p);
        size_t c = pair.find(':');
        if (c == std::string::npos) return fail(TG_ERR_INVALID_ARG, "example: pair has missing delimiter");
        TgMove mv;
        if (!parse_move(n, pair.substr(0, c), mv)) return fail(TG_ERR_INVALID_ARG, "example: bad move " + pair);
        const std::string vs = pair.substr(c + 1);
        unsigned long long v = 0;
        if (vs.empty() || vs.size() > 10) return fail(TG_ERR_INVALID_ARG, "example: bad visit count in " + pair);
        for (char ch : vs) {
            if (ch < '0' || ch > '9') return fail(TG_ERR_INVALID_ARG, "example: bad visit count in " + pair);
            v = v * 10 + (unsigned)(ch - '0');
        }
        if (v > 0xffffffffull) return fail(TG_ERR_INVALID_ARG, "example: visit count out of range in " + pair);
        if (k < cap_moves && moves && visits) { moves[k] = mv; visits[k] = (uint32_t)v; }
        k++;
        if (q == std::string::npos) break;
        p = q + 1;
    }
    *n_moves = k;
    if (k > cap_moves) return fail(TG_ERR_INVALID_ARG, "example: more moves than cap_moves");
    pack(hs, (uint8_t*)state);
    return TG_OK;
}

}  // extern "C"
