//! Data-parallel training (BASELINE config C5): one process per GPU, every rank trains on its own examples, the flat
//! gradient buffer is summed over the ranks once per optimiser step and divided by the world size (the only collective
//! of the whole system; self-play shards by `slot_base` without any).
use std::os::raw::{c_int, c_void};

use crate::{check, net::GpuNet, sys};

/// Rank 0: create the 128-byte RCCL unique id and ship it to the other ranks over any host transport
pub fn unique_id() -> Result<[u8; 128], crate::TgError> {
    let mut id = [0u8; 128];
    check(unsafe { sys::tg_comm_unique_id(id.as_mut_ptr() as *mut c_void) })?;
    Ok(id)
}

/// Every rank, once before training: from then on each optimiser step all-reduces the gradients over RCCL / xGMI and
/// `tg_train_commit` averages the BatchNorm running statistics.  All ranks must feed the same number of chunks.
pub fn init_rccl<const N: usize>(net: &mut GpuNet<N>, rank: i32, world: i32, id: &[u8; 128]) -> Result<(), crate::TgError> {
    let e = net.trainer_handle()?;
    check(unsafe { sys::tg_train_comm_init(e, rank, world, id.as_ptr() as *const c_void) })
}

/// The same reduction through a caller-supplied function (ranks sharing one GPU, a host transport, tests):
/// `reduce(device_buffer, floats, hip_stream)` must leave the SUM over all ranks in the buffer and return 0.
pub fn init_hook<const N: usize, F>(net: &mut GpuNet<N>, world: i32, reduce: F) -> Result<(), crate::TgError>
where
    F: FnMut(*mut f32, usize, *mut c_void) -> c_int + 'static,
{
    unsafe extern "C" fn trampoline<F: FnMut(*mut f32, usize, *mut c_void) -> c_int>(ctx: *mut c_void, d_buf: *mut f32, count: usize,
                                                                                     stream: *mut c_void) -> c_int {
        (*(ctx as *mut F))(d_buf, count, stream)
    }
    let e = net.trainer_handle()?;
    let ctx = Box::into_raw(Box::new(reduce)) as *mut c_void; // lives as long as the engine uses it
    check(unsafe { sys::tg_train_set_allreduce(e, Some(trampoline::<F>), ctx, world) })
}

/// What is attached to the optimiser step's reduction, as the library sees it (`ncclCommCount`, `ncclGetVersion`, the file
/// `ncclAllReduce` was bound from): log it once per rank after `init_rccl`, and refuse to train if `nccl_count != world`.
pub fn comm_info<const N: usize>(net: &mut GpuNet<N>) -> Result<sys::TgCommInfo, crate::TgError> {
    let e = net.trainer_handle()?;
    let mut info = std::mem::MaybeUninit::<sys::TgCommInfo>::zeroed();
    check(unsafe { sys::tg_train_comm_info(e, info.as_mut_ptr()) })?;
    Ok(unsafe { info.assume_init() })
}

/// One float through the reduction the optimiser step will use — RCCL's first collective on the communicator or the caller's
/// function — before anything of a training step is enqueued: a launch that cannot form a ring stops here.  Collective: every
/// rank calls it after `init_rccl` / `init_hook`.  Returns the wall-clock milliseconds of the round trip (0.0 on a single rank).
pub fn preflight<const N: usize>(net: &mut GpuNet<N>) -> Result<f64, crate::TgError> {
    let e = net.trainer_handle()?;
    let mut ms = 0.0f64;
    check(unsafe { sys::tg_train_comm_preflight(e, &mut ms) })?;
    Ok(ms)
}

/// The card this rank's engine runs on, as the HIP runtime names it (PCI bus id, name, CU count): gather it over the ranks and
/// refuse to run if two ranks report the same bus id — one shard per process means one GPU per process
/// (train/src/self_play.rs:98,102-104).
pub fn device_info<const N: usize>(net: &mut GpuNet<N>) -> Result<sys::TgDeviceInfo, crate::TgError> {
    let e = net.trainer_handle()?;
    let mut info = std::mem::MaybeUninit::<sys::TgDeviceInfo>::zeroed();
    check(unsafe { sys::tg_device_info(e, info.as_mut_ptr()) })?;
    Ok(unsafe { info.assume_init() })
}

/// The permutation `tg_train(seed)` visits `n` examples in (the reference shuffles with `thread_rng`, network.rs:49-50): chunk k of
/// `Network::train` = examples `order[k * 500 .. (k + 1) * 500]`.
pub fn train_order(seed: u64, n: usize) -> Result<Vec<i32>, crate::TgError> {
    let mut order = vec![0i32; n];
    check(unsafe { sys::tg_train_order(seed, n as c_int, order.as_mut_ptr()) })?;
    Ok(order)
}

/// The `TG_*` A/B switches that are ON in this process's environment, as the library reads them ("NAME=value NAME=value"); empty
/// in a measured run.
pub fn debug_switches() -> String {
    let mut buf = vec![0u8; 4096];
    unsafe { sys::tg_debug_switches(buf.as_mut_ptr() as *mut std::os::raw::c_char, buf.len()) };
    let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
    String::from_utf8_lossy(&buf[..end]).into_owned()
}
