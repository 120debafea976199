// Links libsumcheck_hip.so.  SUMCHECK_HIP_LIB_DIR points at the directory that holds it
// (thaler-study_amd/ after `make -C thaler-study_amd/csrc`).
fn main() {
    let dir = std::env::var("SUMCHECK_HIP_LIB_DIR")
        .unwrap_or_else(|_| "../../thaler-study_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sumcheck_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=SUMCHECK_HIP_LIB_DIR");
}
