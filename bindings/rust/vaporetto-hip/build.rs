// Links libvaporetto_hip.so (built by `python -m vaporetto_amd.build` into vaporetto_amd/lib/ of the MI355X repository).
// VAPORETTO_HIP_LIB_DIR names the directory that holds it; the HIP runtime it needs comes with ROCm (/opt/rocm/lib).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=VAPORETTO_HIP_LIB_DIR");
    let dir = env::var("VAPORETTO_HIP_LIB_DIR").unwrap_or_else(|_| "../../../vaporetto_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=vaporetto_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
