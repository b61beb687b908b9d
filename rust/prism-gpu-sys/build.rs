// Links libprism_gpu.so (built by `make -C prism_amd/csrc`).  PRISM_GPU_LIB_DIR points at
// the directory that holds it; the default is this repository's prism_amd/lib.
fn main() {
    let dir = std::env::var("PRISM_GPU_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../prism_amd/lib", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=prism_gpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=PRISM_GPU_LIB_DIR");
}
