// build.rs of the crate when it is built against libtfhe_hip.so.
// SOURCE ONLY: no Rust toolchain exists in the image this repository is built in;
// this file has never been compiled.
//
// libtfhe_hip.so is produced by `python -m tfhe_amd.build`
// (hipcc --offload-arch=gfx950); point TFHE_HIP_LIB_DIR at its directory.
fn main() {
    let dir = std::env::var("TFHE_HIP_LIB_DIR")
        .expect("set TFHE_HIP_LIB_DIR to the directory that holds libtfhe_hip.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=tfhe_hip");
    println!("cargo:rerun-if-env-changed=TFHE_HIP_LIB_DIR");
}
