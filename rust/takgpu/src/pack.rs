//! `Game<N>` ⇄ packed state (`TgState5` / `TgState6`), `Move` ⇄ `TgMove`, `Example<N>` ⇄ the drain / train arrays.
//! Layouts: include/takgpu.h ("Packed game state", "Move code", `tg_selfplay_drain`).
use alpha_tak::Example;
use tak::{Color, Direction, Game, Move, MoveKind, Pattern, Piece, Square, Tile};

use crate::sys;

pub const MAX_MOVES: usize = sys::TG_MAX_MOVES as usize;

pub fn state_bytes(n: usize) -> usize {
    if n <= 5 {
        sys::TG_STATE5_BYTES as usize
    } else {
        sys::TG_STATE6_BYTES as usize
    }
}

fn slots(n: usize) -> usize {
    if n <= 5 {
        25
    } else {
        36
    }
}

fn piece_code(p: Piece) -> u8 {
    match p {
        Piece::Flat => 0,
        Piece::Wall => 1,
        Piece::Cap => 2,
    }
}

fn piece_from(code: u8) -> Piece {
    match code {
        0 => Piece::Flat,
        1 => Piece::Wall,
        _ => Piece::Cap,
    }
}

/// tak/src/game.rs:24-35 → packed bytes.  Square index = row·N + col (board.rs:24-27: data[y][x]); colour bit i of a
/// stack word = colour of the i-th stone from the bottom (tile.rs:6-10: `stack` is bottom → top), 1 = black.
pub fn pack_game<const N: usize>(g: &Game<N>, out: &mut [u8]) {
    assert_eq!(out.len(), state_bytes(N));
    out.fill(0);
    let meta0 = 8 * slots(N);
    for y in 0..N {
        for x in 0..N {
            let tile = &g.board[Square::new(x as u8, y as u8)];
            let sq = y * N + x;
            let mut bits = 0u64;
            for (i, c) in tile.stack.iter().enumerate() {
                if *c == Color::Black {
                    bits |= 1u64 << i;
                }
            }
            out[8 * sq..8 * sq + 8].copy_from_slice(&bits.to_le_bytes());
            let h = tile.stack.len() as u8;
            out[meta0 + sq] = if h == 0 { 0 } else { h | (piece_code(tile.piece) << 6) };
        }
    }
    let h = out.len() - 16; // TgHeader
    out[h] = N as u8;
    out[h + 1] = (g.to_move == Color::Black) as u8;
    out[h + 2..h + 4].copy_from_slice(&g.ply.to_le_bytes());
    out[h + 4] = g.white_stones;
    out[h + 5] = g.white_caps;
    out[h + 6] = g.black_stones;
    out[h + 7] = g.black_caps;
    out[h + 8] = g.half_komi as u8;
    out[h + 9] = g.reversible_plies;
}

pub fn unpack_game<const N: usize>(st: &[u8]) -> Game<N> {
    assert_eq!(st.len(), state_bytes(N));
    let meta0 = 8 * slots(N);
    let mut g = Game::<N>::default();
    for y in 0..N {
        for x in 0..N {
            let sq = y * N + x;
            let bits = u64::from_le_bytes(st[8 * sq..8 * sq + 8].try_into().unwrap());
            let m = st[meta0 + sq];
            let height = (m & 63) as usize;
            let stack = (0..height)
                .map(|i| if (bits >> i) & 1 == 1 { Color::Black } else { Color::White })
                .collect();
            g.board[Square::new(x as u8, y as u8)] = Tile {
                piece: if height == 0 { Piece::Flat } else { piece_from(m >> 6) },
                stack,
            };
        }
    }
    let h = st.len() - 16;
    g.to_move = if st[h + 1] == 0 { Color::White } else { Color::Black };
    g.ply = u16::from_le_bytes([st[h + 2], st[h + 3]]);
    g.white_stones = st[h + 4];
    g.white_caps = st[h + 5];
    g.black_stones = st[h + 6];
    g.black_caps = st[h + 7];
    g.half_komi = st[h + 8] as i8;
    g.reversible_plies = st[h + 9];
    g
}

/// takparse `Move` → `TgMove`: square | piece-or-direction << 6 | `Pattern::mask()` << 8
/// (directions in the order of tak/src/move_gen.rs:64: Up 0, Down 1, Left 2, Right 3)
pub fn move_code<const N: usize>(m: &Move) -> sys::TgMove {
    let sq = (m.square().row() as u16) * N as u16 + m.square().column() as u16;
    match m.kind() {
        MoveKind::Place(p) => sq | (piece_code(p) as u16) << 6,
        MoveKind::Spread(d, pattern) => {
            let dir = match d {
                Direction::Up => 0u16,
                Direction::Down => 1,
                Direction::Left => 2,
                Direction::Right => 3,
            };
            sq | dir << 6 | (pattern.mask() as u16) << 8
        }
    }
}

pub fn move_from_code<const N: usize>(code: sys::TgMove) -> Move {
    let sq = (code & 63) as usize;
    let square = Square::new((sq % N) as u8, (sq / N) as u8);
    let pat = (code >> 8) as u8;
    if pat == 0 {
        return Move::new(square, MoveKind::Place(piece_from(((code >> 6) & 3) as u8)));
    }
    let dir = match (code >> 6) & 3 {
        0 => Direction::Up,
        1 => Direction::Down,
        2 => Direction::Left,
        _ => Direction::Right,
    };
    // MSB first, one bit per carried stone, a set bit closes a drop group: 0b0110_0000 → drops [2, 1]
    let total = 8 - pat.trailing_zeros();
    let mut drops: Vec<u32> = Vec::new();
    let mut run = 0u32;
    for i in 0..total {
        run += 1;
        if pat & (0x80 >> i) != 0 {
            drops.push(run);
            run = 0;
        }
    }
    let pattern: Pattern = drops.into_iter().collect(); // as tak/src/move_gen.rs:75 builds it
    Move::new(square, MoveKind::Spread(dir, pattern))
}

/// The arrays `tg_train` / `tg_train_chunk` take (and `tg_selfplay_drain` returns), for a slice of examples
pub struct ExampleArrays {
    pub states: Vec<u8>,
    pub n_moves: Vec<i32>,
    pub moves: Vec<sys::TgMove>,
    pub visits: Vec<u32>,
    pub results: Vec<f32>,
}

pub fn pack_examples<const N: usize>(examples: &[&Example<N>]) -> ExampleArrays {
    let sb = state_bytes(N);
    let k = examples.len();
    let mut a = ExampleArrays {
        states: vec![0u8; k * sb],
        n_moves: vec![0; k],
        moves: vec![0; k * MAX_MOVES],
        visits: vec![0; k * MAX_MOVES],
        results: vec![0.0; k],
    };
    for (i, ex) in examples.iter().enumerate() {
        pack_game(&ex.game, &mut a.states[i * sb..(i + 1) * sb]);
        assert!(ex.policy.len() <= MAX_MOVES, "more than TG_MAX_MOVES moves in an example");
        a.n_moves[i] = ex.policy.len() as i32;
        for (j, (m, v)) in ex.policy.iter().enumerate() {
            a.moves[i * MAX_MOVES + j] = move_code::<N>(m);
            a.visits[i * MAX_MOVES + j] = *v;
        }
        a.results[i] = ex.result;
    }
    a
}

/// One drained record → `Example { game, policy: Vec<(Move, u32)>, result }` (alpha-tak/src/example.rs:29-33)
pub fn unpack_example<const N: usize>(header: &sys::TgExampleHeader, state: &[u8], moves: &[sys::TgMove], visits: &[u32]) -> Example<N> {
    let k = header.n_moves as usize;
    Example {
        game: unpack_game::<N>(state),
        policy: (0..k).map(|j| (move_from_code::<N>(moves[j]), visits[j])).collect(),
        result: header.result,
    }
}
