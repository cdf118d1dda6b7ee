// engine/build.rs -- link librama_hip.so when the `hip` feature is on.
fn main() {
    if std::env::var("CARGO_FEATURE_HIP").is_ok() {
        let dir = std::env::var("RAMA_HIP_LIB_DIR").expect("set RAMA_HIP_LIB_DIR to the directory holding librama_hip.so");
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=rama_hip");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
        println!("cargo:rerun-if-env-changed=RAMA_HIP_LIB_DIR");
    }
}
