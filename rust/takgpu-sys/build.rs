// Tell rustc where libtakgpu.so lives: TAKGPU_LIB_DIR, or ../../tak_amd relative to this crate (the in-tree build
// of `make -C tak_amd/csrc`).  The library itself needs libamdhip64 (ROCm ≥ 7.0) at run time.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("TAKGPU_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../tak_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=takgpu");
    println!("cargo:rerun-if-env-changed=TAKGPU_LIB_DIR");
}
