//! `multilinear-extensions` on the GPU path: the reference crate's two public functions
//! (multilinear-extensions/src/lib.rs:6 and :29) with their names and argument order, for the field type
//! every reference crate instantiates (`Fp64<MontBackend<T, 1>>`).  Both evaluate the multilinear extension
//! of `evals` at `r` with r[0] on the most significant index bit; on the device that is ONE streaming pass
//! over the table (`sc_table_evaluate(.., SC_ORDER_BE)`), so the two differ in name only.
//! Source only: not compiled in the build image (no Rust toolchain there).
use ark_ff::MontConfig;
pub use sumcheck_hip::F64;

/// O(2^n) eq-table + dot product in the reference (:6-24).
pub fn vsbw_multilinear_from_evaluations<T: MontConfig<1>>(evals: &[F64<T>], r: &[F64<T>]) -> F64<T> {
    sumcheck_hip::vsbw_multilinear_from_evaluations::<T>(evals, r)
}

/// O(n 2^n) per-index Lagrange basis in the reference (:29-48).
pub fn cti_multilinear_from_evaluations<T: MontConfig<1>>(evals: &[F64<T>], r: &[F64<T>]) -> F64<T> {
    sumcheck_hip::cti_multilinear_from_evaluations::<T>(evals, r)
}

#[cfg(test)]
mod tests {
    //! the reference's own test (multilinear-extensions/src/lib.rs:76-120), unchanged in substance
    use super::*;
    use ark_ff::{Fp64, MontBackend, PrimeField};

    #[derive(ark_ff::MontConfig)]
    #[modulus = "5"]
    #[generator = "2"]
    struct FrConfig;
    type Fp5 = Fp64<MontBackend<FrConfig, 1>>;

    #[test]
    fn example_from_book() {
        let evals: Vec<Fp5> = [1u32, 2, 1, 4].iter().map(|e| Fp5::from_bigint((*e).into()).unwrap()).collect();
        let expected = [[1, 2, 3, 4, 0], [1, 4, 2, 0, 3], [1, 1, 1, 1, 1], [1, 3, 0, 2, 4], [1, 0, 4, 3, 2]];
        for i in 0..5u32 {
            for j in 0..5u32 {
                let r = [Fp5::from_bigint(i.into()).unwrap(), Fp5::from_bigint(j.into()).unwrap()];
                let want = Fp5::from_bigint((expected[i as usize][j as usize] as u32).into()).unwrap();
                assert_eq!(cti_multilinear_from_evaluations::<FrConfig>(&evals, &r), want);
                assert_eq!(vsbw_multilinear_from_evaluations::<FrConfig>(&evals, &r), want);
            }
        }
    }
}
