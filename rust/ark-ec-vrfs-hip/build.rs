// Links libvrfhip.so (built by `make -C ark_ec_vrfs_amd/csrc`, or `python -c 'import __graft_entry__ as g; g.build()'`).
// VRFHIP_LIB_DIR names the directory that holds it; default: the in-tree build directory relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("VRFHIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../ark_ec_vrfs_amd/csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=vrfhip");
    // the HIP runtime the library itself depends on
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rerun-if-env-changed=VRFHIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-changed=../../include/vrfhip.h");
}
