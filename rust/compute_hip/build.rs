//! Links against libgs_hip.so.  GS_HIP_LIB_DIR points at the directory that holds it
//! (the `grayscott_amd/` directory of the grayscott-mi355x checkout after
//! `python __graft_entry__.py`).
fn main() {
    println!("cargo:rerun-if-env-changed=GS_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("GS_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=gs_hip");
}
