// UNBUILT SOURCE (no cargo/rustc in the build image).  Links librt_amd.so built by `make -C ray-tracing_amd`.
fn main() {
    let dir = std::env::var("RT_AMD_LIB_DIR").unwrap_or_else(|_| "../../ray-tracing_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-search=native=/opt/rocm/lib");
    println!("cargo:rustc-link-lib=dylib=rt_amd");
    println!("cargo:rerun-if-changed=../../include/rt_amd.h");
}
