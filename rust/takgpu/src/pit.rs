//! `pit` (train/src/pit.rs:15-96): the new network against the old one — all 2 × PIT_GAMES games at once on the GPU.
use crate::{check, net::GpuNet, sys};

/// PitResult of pit.rs:98-110
#[derive(Debug, Default, Clone, Copy)]
pub struct PitResult {
    pub wins: u32,
    pub draws: u32,
    pub losses: u32,
    pub unfinished: u32,
}

impl PitResult {
    pub fn win_rate(&self) -> f64 {
        self.wins as f64 / (self.wins + self.losses) as f64 // pit.rs:105-110
    }
}

/// Drop-in for `pit(&new_network, &network)` at train/src/main.rs:100.  PIT_GAMES 128, ROLLOUTS 50, BATCH_SIZE 16,
/// RANDOM_PLIES 2, komi 2 (pit.rs:5-9,27).  All games run concurrently; the counts returned are the ones `pit` would return,
/// early exit of pit.rs:20-23 included (the library tallies the openings in the reference's order: `ref_*`).
pub fn pit_gpu<const N: usize>(new: &GpuNet<N>, old: &GpuNet<N>) -> PitResult {
    let cfg = sys::TgPitConfig {
        pairs: 128,
        rollouts: 50,
        idle_rollouts: 1,
        random_plies: 2,
        komi: 2,
        max_plies: 0,
        arena_nodes: 0,
        batch: 16,
        seed: rand::random(),
    };
    let mut out = std::mem::MaybeUninit::<sys::TgPitResult>::zeroed();
    check(unsafe { sys::tg_pit(new.e, old.e, &cfg, out.as_mut_ptr()) }).expect("tg_pit");
    let r = unsafe { out.assume_init() };
    PitResult { wins: r.ref_wins, draws: r.ref_draws, losses: r.ref_losses, unfinished: r.unfinished }
}
