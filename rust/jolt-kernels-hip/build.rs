//! Locates `libjolt_hip.so`: `JOLT_HIP_LIB_DIR` (the directory holding the shared object built by `python -m jolt_amd.build`),
//! falling back to `../../jolt_amd` relative to this crate when it lives inside the jolt_amd repository.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=JOLT_HIP_LIB_DIR");
    let dir = env::var_os("JOLT_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap_or_default()).join("../../jolt_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=jolt_hip");
    // the library is found at run time next to the prover binary or through LD_LIBRARY_PATH; embed the build-time location too
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
