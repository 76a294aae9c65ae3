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
