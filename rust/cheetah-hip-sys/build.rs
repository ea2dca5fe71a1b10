// Links libcheetah_hip.so (built by `make -C cheetah_amd/csrc`, which needs hipcc and --offload-arch=gfx950).
// CHEETAH_HIP_LIB_DIR names the directory that holds it; by default the in-tree build output is used.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=CHEETAH_HIP_LIB_DIR");
    let dir = match env::var("CHEETAH_HIP_LIB_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../cheetah_amd/lib"),
    };
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=cheetah_hip");
    // the HIP runtime the library itself is linked against
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
