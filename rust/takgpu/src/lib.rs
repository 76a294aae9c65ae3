//! `takgpu` — the batched MI355X self-play engine (libtakgpu.so) behind the surface of ViliamVadocz/tak.
//!
//! NOT COMPILED IN THE BUILD IMAGE of the takgpu repository (no cargo / rustc there): written against
//! `include/takgpu.h` (through `takgpu-sys`, which is generated from it) and against the reference sources cited below.
//!
//! | reference (path:line)                                        | here                                   |
//! |--------------------------------------------------------------|----------------------------------------|
//! | `Network<N>` — alpha-tak/src/model/network.rs:26-35          | [`net::GpuNet`]                        |
//! | `Network::train` — network.rs:37-97                          | [`net::GpuNet::train`], [`dp`]         |
//! | `self_play_parallel` — train/src/self_play.rs:96-262         | [`selfplay::self_play_parallel_gpu`]   |
//! | `pit` — train/src/pit.rs:15-96                               | [`pit::pit_gpu`]                       |
//! | `Game<N>` / `Move` / `Example<N>` — tak/src/game.rs:24-35, takparse, alpha-tak/src/example.rs:29-33 | [`pack`] |
//!
//! `train/src/main.rs` then reads `train::<5, GpuNet<5>>(args)` and calls `self_play_parallel_gpu(&network)` at :120 and
//! `pit_gpu(&new_network, &network)` at :100 — everything else of the training loop is unchanged.
pub mod dp;
pub mod net;
pub mod pack;
pub mod pit;
pub mod selfplay;

use std::ffi::CStr;

pub use net::GpuNet;
pub use pit::pit_gpu;
pub use selfplay::{self_play_parallel_gpu, SelfPlaySettings};
pub use takgpu_sys as sys;

/// A failed C-ABI call: the negative `TgStatus` and the library's thread-local message.
#[derive(Debug, Clone)]
pub struct TgError {
    pub code: i32,
    pub message: String,
}

impl std::fmt::Display for TgError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "takgpu error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for TgError {}

/// `TG_OK` → `Ok(())`, anything else → the library's message (takgpu.h: "every function returns TG_OK or a negative TgStatus")
pub fn check(rc: i32) -> Result<(), TgError> {
    if rc == sys::TG_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::tg_last_error()) }.to_string_lossy().into_owned();
    Err(TgError { code: rc, message })
}

/// Policy head of a board size, as the reference chooses it: 5×5 keeps the legacy 1575-entry FC head
/// (alpha-tak/src/search/move_map.rs:21-24), every other size the convolutional head (net6.rs:56).
pub const fn policy_head(n: usize) -> i32 {
    if n == 5 {
        sys::TG_HEAD_FC5
    } else {
        sys::TG_HEAD_CONV
    }
}
