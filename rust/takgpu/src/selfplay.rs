//! `self_play_parallel` (train/src/self_play.rs:96-262) as one call sequence on the engine: every phase of the
//! reference's loop — opening, instant-win scan, Dirichlet noise, ROLLOUTS lock-step iterations, move choice, tree
//! reuse, game recycling, example emission — runs on the GPU for thousands of games at once (tg_selfplay_*).
use std::{fs::File, io::Write};

use alpha_tak::{sys_time, Example, Network};

use crate::{check, net::GpuNet, pack, sys};

/// The reference's compile-time constants (self_play.rs:10-19,94), runtime here
#[derive(Clone, Copy, Debug)]
pub struct SelfPlaySettings {
    pub games: i32,         // concurrent games on this GPU (WORKERS = 32 in the reference)
    pub self_play_games: i32, // SELF_PLAY_GAMES
    pub rollouts: i32,      // ROLLOUTS
    pub noise_alpha: f32,
    pub noise_ratio: f32,
    pub noise_plies: i32,
    pub exploit_plies: i32,
    pub komi: i32,
    pub arena_nodes: i32,   // average node budget per game; 0 = sized from the free device memory
    pub seed: u64,
    pub slot_base: u32,     // rank · games when several GPUs share a run (no collective: SURVEY.md §8e)
    pub example_dir: Option<&'static str>,
}

impl Default for SelfPlaySettings {
    fn default() -> Self {
        SelfPlaySettings {
            games: 4096,
            self_play_games: 8192,
            rollouts: 400,
            noise_alpha: 0.2,
            noise_ratio: 0.3,
            noise_plies: 80,
            exploit_plies: 40,
            komi: 2,
            arena_nodes: 0,
            seed: 0,
            slot_base: 0,
            example_dir: Some("_examples"),
        }
    }
}

/// What the run did besides producing examples.  `aborted_games` is the one divergence from self_play.rs a caller must be
/// able to see: the reference keeps a game of any length, the engine retires a game that exceeds a fixed capacity (512 plies,
/// selection depth 256, 2²² visits of one node: `TG_LIMIT_*`) and discards its examples — a bias against very long games if it
/// ever happens (it has not in any soak run: `profiles/*soak*`).
#[derive(Clone, Copy, Debug, Default)]
pub struct SelfPlayReport {
    pub games_finished: u64,
    pub aborted_games: u64,
    pub examples: u64,
}

/// Drop-in for `self_play_parallel(&network)` at train/src/main.rs:120.  Games the engine had to retire are reported on
/// stderr (the reference has no such case); use `self_play_with_report` to receive the count.
pub fn self_play_parallel_gpu<const N: usize>(network: &GpuNet<N>) -> Vec<Example<N>> {
    let (examples, report) =
        self_play_with_report(network, SelfPlaySettings { seed: rand::random(), ..Default::default() }).expect("self-play on the GPU");
    if report.aborted_games > 0 {
        eprintln!("self-play: {} of {} games exceeded an engine capacity (TG_LIMIT_*) and were retired without examples",
                  report.aborted_games, report.aborted_games + report.games_finished);
    }
    examples
}

pub fn self_play_with<const N: usize>(network: &GpuNet<N>, s: SelfPlaySettings) -> Result<Vec<Example<N>>, crate::TgError> {
    self_play_with_report(network, s).map(|(examples, _)| examples)
}

pub fn self_play_with_report<const N: usize>(network: &GpuNet<N>, s: SelfPlaySettings)
                                             -> Result<(Vec<Example<N>>, SelfPlayReport), crate::TgError> {
    let scfg = sys::TgSearchConfig {
        games: s.games,
        arena_nodes: s.arena_nodes,
        exploration_base: 500.0, // EXPLORATION_BASE, alpha-tak/src/search/mcts.rs:7
        exploration_init: 4.0,   // EXPLORATION_INIT, mcts.rs:8
        seed: s.seed,
        slot_base: s.slot_base,
        batch: 1, // one leaf per game and iteration (self_play.rs:181-210)
        visit_limit: 0, // TG_LIMIT_VISITS
        reserved: 0,
    };
    let drain_cap = (s.games as usize) * 64;
    let cfg = sys::TgSelfPlayConfig {
        rollouts: s.rollouts,
        noise_plies: s.noise_plies,
        exploit_plies: s.exploit_plies,
        noise_alpha: s.noise_alpha,
        noise_ratio: s.noise_ratio,
        komi: s.komi,
        total_games: s.self_play_games,
        max_examples: (drain_cap * 4) as i32, // four drain intervals; an overrun is reported below, never waited for
        max_game_plies: 0, // TG_LIMIT_GAME_PLIES: a longer game is retired alone (TgSelfPlayStats.aborted_games)
        reserved: 0,
    };
    check(unsafe { sys::tg_selfplay_create(network.e, &scfg, &cfg) })?;
    let mut file = s.example_dir.map(|d| File::create(format!("{d}/{}.data", sys_time())).unwrap()); // self_play.rs:98
    let sb = pack::state_bytes(N);
    let mut headers = vec![sys::TgExampleHeader { game_id: 0, n_moves: 0, result: 0.0, reserved: 0 }; drain_cap];
    let mut states = vec![0u8; drain_cap * sb];
    let mut moves = vec![0u16; drain_cap * pack::MAX_MOVES];
    let mut visits = vec![0u32; drain_cap * pack::MAX_MOVES];
    let mut examples = Vec::new();
    loop {
        check(unsafe { sys::tg_selfplay_step(network.e, 4) })?; // asynchronous; the drain below synchronises
        let mut n_out = 0i32;
        check(unsafe {
            sys::tg_selfplay_drain(network.e, drain_cap as i32, headers.as_mut_ptr(), states.as_mut_ptr() as *mut _,
                                   moves.as_mut_ptr(), visits.as_mut_ptr(), &mut n_out)
        })?;
        for i in 0..n_out as usize {
            let ex = pack::unpack_example::<N>(&headers[i], &states[i * sb..(i + 1) * sb],
                                               &moves[i * pack::MAX_MOVES..(i + 1) * pack::MAX_MOVES],
                                               &visits[i * pack::MAX_MOVES..(i + 1) * pack::MAX_MOVES]);
            if let Some(f) = file.as_mut() {
                writeln!(f, "{ex}").unwrap(); // the reference's text format (alpha-tak/src/example.rs:81-102)
            }
            examples.push(ex);
        }
        let mut st = std::mem::MaybeUninit::<sys::TgSelfPlayStats>::zeroed();
        check(unsafe { sys::tg_selfplay_stats(network.e, st.as_mut_ptr()) })?;
        let st = unsafe { st.assume_init() };
        // Examples the ring overwrote before this loop fetched them can never arrive: say so instead of waiting for them
        // (st.examples counts what was emitted, examples.len() what was received).
        if st.dropped_examples > 0 {
            return Err(crate::TgError {
                code: sys::TG_ERR_LIMIT,
                message: format!("self-play: {} examples were overwritten in the device ring before they were drained (max_examples too small)",
                                 st.dropped_examples),
            });
        }
        // The reference's loop runs `while games.iter().any(Option::is_some)` (self_play.rs:107): a slot retires when its game
        // ends with completed + WORKERS ≥ SELF_PLAY_GAMES (:151,237), and up to `games` games are still in flight when the first
        // slot retires.  So: until NO slot is alive, and every emitted example has been received.
        if st.alive_games == 0 && n_out == 0 && st.examples == examples.len() as u64 {
            let report = SelfPlayReport { games_finished: st.games_finished, aborted_games: st.aborted_games, examples: st.examples };
            return Ok((examples, report));
        }
    }
}

/// `Network` is only needed as a bound by callers that stay generic over the network type
pub fn _assert_network<const N: usize>() {
    fn takes<const M: usize, T: Network<M>>() {}
    takes::<N, GpuNet<N>>();
}
