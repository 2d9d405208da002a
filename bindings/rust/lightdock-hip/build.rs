// Points rustc at lightdock-rust_amd/lib/liblightdock_hip.so (override with LIGHTDOCK_HIP_LIB_DIR).
fn main() {
    let dir = std::env::var("LIGHTDOCK_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../../lightdock-rust_amd/lib", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=lightdock_hip");
    println!("cargo:rerun-if-env-changed=LIGHTDOCK_HIP_LIB_DIR");
}
