// Links libbioscan.so (built by `make -C datafusion-bio-formats_amd/csrc`).  BIOSCAN_LIB_DIR names the directory that
// holds it; the default is the in-tree location relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("BIOSCAN_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../datafusion-bio-formats_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=bioscan");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BIOSCAN_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/bioscan.h");
}
