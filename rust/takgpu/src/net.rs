//! `GpuNet<N>`: the reference's `Network<N>` (alpha-tak/src/model/network.rs:26-35) on libtakgpu.so.
//!
//! The trait names tch types in four signatures (`vs`, `forward_mcts`, `forward_training`, `save`/`load` → `TchError`).
//! MCTS, self-play and the pit never call the tensor methods (the reference's own `DummyNet` leaves them
//! `unimplemented!()`: alpha-tak/src/search/tests.rs:9-27); here they are answered from the engine where that is
//! possible (`forward_mcts`) and left `unimplemented!()` where a tch `VarStore` would be required (`vs`).
use std::{ffi::CString, marker::PhantomData, path::Path};

use alpha_tak::{Example, Network};
use tak::Game;
use tch::{nn::VarStore, Device, Kind, TchError, Tensor};

use crate::{check, pack, policy_head, sys};

/// RES_BLOCKS / FILTERS of the reference (net5.rs:16-17, net6.rs:16-17); runtime parameters of the engine.
#[derive(Clone, Copy, Debug)]
pub struct Topology {
    pub res_blocks: i32,
    pub filters: i32,
    pub max_batch: i32,
    pub device: i32,
}

impl Topology {
    pub const fn reference(n: usize) -> Self {
        Topology { res_blocks: if n == 5 { 8 } else { 16 }, filters: 128, max_batch: 4096, device: 0 }
    }
}

pub struct GpuNet<const N: usize> {
    pub(crate) e: *mut sys::TgEngine,
    pub(crate) topo: Topology,
    policy_len: usize,
    trainer: bool,
    _not_sync: PhantomData<*mut ()>, // one handle per GPU; calls on a handle are serialised by its owner (takgpu.h)
}

unsafe impl<const N: usize> Send for GpuNet<N> {}

impl<const N: usize> Drop for GpuNet<N> {
    fn drop(&mut self) {
        unsafe { sys::tg_engine_destroy(self.e) }
    }
}

impl<const N: usize> GpuNet<N> {
    /// tg_engine_create; the tensors are missing until `init_random`, `set_tensors` or `load`.
    pub fn with_topology(topo: Topology) -> Result<Self, crate::TgError> {
        let cfg = sys::TgConfig {
            abi_version: sys::TG_ABI_VERSION,
            device: topo.device,
            board_size: N as i32,
            res_blocks: topo.res_blocks,
            filters: topo.filters,
            policy_head: policy_head(N),
            evaluator: sys::TG_EVAL_RESNET,
            max_batch: topo.max_batch,
        };
        let mut e = std::ptr::null_mut();
        check(unsafe { sys::tg_engine_create(&cfg, &mut e) })?;
        let policy_len = unsafe { sys::tg_policy_size(N as i32, policy_head(N)) } as usize;
        Ok(GpuNet { e, topo, policy_len, trainer: false, _not_sync: PhantomData })
    }

    /// Network::default(): tch's default initialisers, drawn from Philox(seed) (tg_net_init_random)
    pub fn init_random(&mut self, seed: u64) -> Result<(), crate::TgError> {
        check(unsafe { sys::tg_net_init_random(self.e, seed) })?;
        check(unsafe { sys::tg_net_finalize(self.e) })
    }

    /// Names and element counts of every tensor, in the creation order of net5.rs:29-62 / net6.rs:29-57 (takgpu.h)
    pub fn tensor_names(&self) -> Vec<(String, usize)> {
        let f = self.topo.filters as usize;
        let cin = unsafe { sys::tg_input_channels(N as i32) } as usize;
        let bn = |v: &mut Vec<(String, usize)>, name: &str| {
            for leaf in ["weight", "bias", "running_mean", "running_var"] {
                v.push((format!("{name}.{leaf}"), f));
            }
        };
        let mut out: Vec<(String, usize)> = Vec::new();
        out.push(("conv0.weight".into(), f * cin * 9));
        out.push(("conv0.bias".into(), f));
        bn(&mut out, "bn0");
        for i in 0..self.topo.res_blocks {
            for c in ["conv1", "conv2"] {
                out.push((format!("res{i}.{c}.weight"), f * f * 9));
                out.push((format!("res{i}.{c}.bias"), f));
            }
            bn(&mut out, &format!("res{i}.bn1"));
            bn(&mut out, &format!("res{i}.bn2"));
        }
        if N == 5 {
            out.push(("policy.weight".into(), self.policy_len * f * N * N));
            out.push(("policy.bias".into(), self.policy_len));
        } else {
            let ch = self.policy_len / (N * N);
            out.push(("policy.weight".into(), ch * f * 9));
            out.push(("policy.bias".into(), ch));
        }
        out.push(("value.weight".into(), f * N * N));
        out.push(("value.bias".into(), 1));
        out
    }

    /// Hand the engine one tensor in tch layout (conv OIHW, linear [out, in], BN vectors) — tg_net_set_tensor
    pub fn set_tensor(&mut self, name: &str, data: &[f32]) -> Result<(), crate::TgError> {
        let c = CString::new(name).unwrap();
        check(unsafe { sys::tg_net_set_tensor(self.e, c.as_ptr(), data.as_ptr(), data.len()) })
    }

    pub fn get_tensor(&self, name: &str, count: usize) -> Result<Vec<f32>, crate::TgError> {
        let c = CString::new(name).unwrap();
        let mut out = vec![0f32; count];
        check(unsafe { sys::tg_net_get_tensor(self.e, c.as_ptr(), out.as_mut_ptr(), count) })?;
        Ok(out)
    }

    /// Fold BatchNorm, re-lay-out for the MFMA kernels, upload — after the last `set_tensor`
    pub fn finalize(&mut self) -> Result<(), crate::TgError> {
        check(unsafe { sys::tg_net_finalize(self.e) })
    }

    /// 0 = exact f32 MFMA (default, the parity path), 1 = split-bf16 tower (≤ 1e-5 relative off the f32 forward)
    pub fn set_precision(&mut self, precision: i32) -> Result<(), crate::TgError> {
        check(unsafe { sys::tg_net_set_precision(self.e, precision) })
    }

    fn ensure_trainer(&mut self) -> Result<(), crate::TgError> {
        if !self.trainer {
            // LEARNING_RATE / WEIGHT_DECAY / CHUNK_SIZE / CHUNKS_IN_STEP of network.rs:14-21, tch's Adam and BatchNorm defaults
            let cfg = sys::TgTrainConfig {
                learning_rate: 1e-4,
                weight_decay: 1e-4,
                beta1: 0.9,
                beta2: 0.999,
                eps: 1e-8,
                bn_momentum: 0.1,
                bn_eps: 1e-5,
                chunk_size: 500,
                chunks_in_step: 20,
                reserved: 0,
            };
            check(unsafe { sys::tg_train_create(self.e, &cfg) })?;
            self.trainer = true;
        }
        Ok(())
    }

    pub(crate) fn trainer_handle(&mut self) -> Result<*mut sys::TgEngine, crate::TgError> {
        self.ensure_trainer()?;
        Ok(self.e)
    }

    /// `Network::train` with the error reported instead of printed; returns (mean loss_p, mean loss_z, optimiser steps)
    pub fn try_train(&mut self, examples: &[Example<N>], seed: u64) -> Result<(f32, f32, i32), crate::TgError> {
        self.ensure_trainer()?;
        let refs: Vec<&Example<N>> = examples.iter().collect();
        let a = pack::pack_examples::<N>(&refs);
        let (mut lp, mut lz, mut steps) = (0f32, 0f32, 0i32);
        check(unsafe {
            sys::tg_train(self.e, refs.len() as i32, a.states.as_ptr() as *const _, a.n_moves.as_ptr(), a.moves.as_ptr(),
                          a.visits.as_ptr(), a.results.as_ptr(), seed, &mut lp, &mut lz, &mut steps)
        })?;
        check(unsafe { sys::tg_train_commit(self.e) })?; // policy_eval / self-play / pit now use the trained weights
        Ok((lp, lz, steps))
    }
}

impl<const N: usize> Default for GpuNet<N> {
    /// Network::default() (net5.rs:29-73): the reference's topology, random initialisation
    fn default() -> Self {
        let mut net = Self::with_topology(Topology::reference(N)).expect("tg_engine_create");
        net.init_random(rand::random()).expect("tg_net_init_random");
        net
    }
}

impl<const N: usize> Network<N> for GpuNet<N> {
    fn vs(&self) -> &VarStore {
        unimplemented!("GpuNet keeps its parameters on the MI355X, not in a tch VarStore (see save / load / get_tensor)")
    }

    /// `vs.save(path)`: the named tensors leave through tg_net_get_tensor into a CPU VarStore that tch writes
    fn save<T: AsRef<Path>>(&self, path: T) -> Result<(), TchError> {
        let named: Vec<(String, Tensor)> = self
            .tensor_names()
            .into_iter()
            .map(|(name, count)| {
                let data = self.get_tensor(&name, count).expect("tg_net_get_tensor");
                (name, Tensor::of_slice(&data))
            })
            .collect();
        Tensor::save_multi(&named, path)
    }

    /// `Network::load`: a file written by `save` above (tensor names of takgpu.h).  A `.model` written by the reference's
    /// tch networks keeps tch's auto-names (`weight__7`, …): convert it once with tak_amd/checkpoint.py.
    fn load<T: AsRef<Path>>(path: T) -> Result<Self, TchError> {
        let mut net = Self::with_topology(Topology::reference(N)).expect("tg_engine_create");
        for (name, t) in Tensor::load_multi_with_device(path, Device::Cpu)? {
            let data: Vec<f32> = Vec::from(&t.to_kind(Kind::Float).contiguous().view([-1]));
            net.set_tensor(&name, &data).expect("tg_net_set_tensor");
        }
        net.finalize().expect("tg_net_finalize");
        Ok(net)
    }

    /// net5.rs:106-111 on already encoded planes `[B, C_in, N, N]`: tg_forward_mcts
    fn forward_mcts(&self, input: Tensor) -> (Tensor, Tensor) {
        let b = input.size()[0] as usize;
        let planes: Vec<f32> = Vec::from(&input.to_device(Device::Cpu).to_kind(Kind::Float).contiguous().view([-1]));
        let mut policy = vec![0f32; b * self.policy_len];
        let mut eval = vec![0f32; b];
        check(unsafe { sys::tg_forward_mcts(self.e, b as i32, planes.as_ptr(), policy.as_mut_ptr(), eval.as_mut_ptr()) })
            .expect("tg_forward_mcts");
        (Tensor::of_slice(&policy).view([b as i64, self.policy_len as i64]), Tensor::of_slice(&eval).view([b as i64, 1]))
    }

    fn forward_training(&self, _input: Tensor) -> (Tensor, Tensor) {
        unimplemented!("training runs inside the engine (tg_train_chunk); tg_train_forward gives log-softmax / eval for packed states")
    }

    /// net5.rs:120-130: `n` games → (full softmax policy, tanh eval) each; one forward on the GPU
    fn policy_eval(&self, games: &[Game<N>]) -> Vec<(Vec<f32>, f32)> {
        if games.is_empty() {
            return Vec::new(); // net5.rs:121-123
        }
        let sb = pack::state_bytes(N);
        let mut states = vec![0u8; sb * games.len()];
        for (g, chunk) in games.iter().zip(states.chunks_mut(sb)) {
            pack::pack_game(g, chunk);
        }
        let mut policy = vec![0f32; self.policy_len * games.len()];
        let mut eval = vec![0f32; games.len()];
        check(unsafe {
            sys::tg_policy_eval(self.e, games.len() as i32, states.as_ptr() as *const _, policy.as_mut_ptr(), eval.as_mut_ptr())
        })
        .expect("tg_policy_eval");
        policy.chunks(self.policy_len).map(<[f32]>::to_vec).zip(eval).collect()
    }

    /// network.rs:37-56 — fresh Adam, shuffle, chunks_exact(500), a step every 20 chunks; all on the GPU (tg_train)
    fn train(&mut self, examples: &[Example<N>]) {
        println!("starting training with {} examples", examples.len());
        let (lp, lz, steps) = self.try_train(examples, rand::random()).expect("tg_train");
        println!("p={lp}\t z={lz}\t steps={steps}");
    }
}
