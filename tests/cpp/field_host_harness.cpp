// Host build of thaler-study_amd/csrc/field.hpp so that the exact arithmetic the kernels
// use (same header, same expressions) is checked against Python big integers on CPU.
#include "../../thaler-study_amd/csrc/field.hpp"
using namespace sc;
extern "C" {
// op: 0 add, 1 sub, 2 mul, 3 dbl(a), 4 to_mont(a), 5 from_mont(a), 6 redc(hi=a, lo=b)
void fh_gold_binop(int op, const u64* a, const u64* b, u64* out, size_t n) {
  GoldilocksMont F;
  for (size_t i = 0; i < n; ++i) {
    switch (op) {
      case 0: out[i] = F.add(a[i], b[i]); break;
      case 1: out[i] = F.sub(a[i], b[i]); break;
      case 2: out[i] = F.mul(a[i], b[i]); break;
      case 3: out[i] = F.dbl(a[i]); break;
      case 4: out[i] = F.to_mont(a[i]); break;
      case 5: out[i] = F.from_mont(a[i]); break;
      case 6: out[i] = F.redc(a[i], b[i]); break;
    }
  }
}
void fh_gen_binop(u64 p, int op, const u64* a, const u64* b, u64* out, size_t n) {
  FieldParams fp;
  field_params_from_modulus(p, &fp);
  MontGeneric F(fp);
  for (size_t i = 0; i < n; ++i) {
    switch (op) {
      case 0: out[i] = F.add(a[i], b[i]); break;
      case 1: out[i] = F.sub(a[i], b[i]); break;
      case 2: out[i] = F.mul(a[i], b[i]); break;
      case 3: out[i] = F.dbl(a[i]); break;
      case 4: out[i] = F.to_mont(a[i]); break;
      case 5: out[i] = F.from_mont(a[i]); break;
      case 6: out[i] = F.redc(a[i], b[i]); break;
    }
  }
}
// lazy accumulator: residue of sum_i a[i]*b[i] (Montgomery product sum), chunked so the
// 160-bit accumulator is exercised with many terms
u64 fh_gold_dot(const u64* a, const u64* b, size_t n) {
  GoldilocksMont F;
  GoldilocksMont::Acc acc;
  F.acc_zero(acc);
  for (size_t i = 0; i < n; ++i) F.acc_mac(acc, a[i], b[i]);
  return F.acc_get(acc);
}
u64 fh_gold_dot3(const u64* a, const u64* b, size_t n) {
  GoldilocksMont F;
  GoldilocksMont::Acc3 acc;
  F.acc3_zero(acc);
  for (size_t i = 0; i < n; ++i) F.acc3_mac(acc, a[i], b[i]);
  return F.acc3_get(acc);
}
u64 fh_gen_dot(u64 p, const u64* a, const u64* b, size_t n) {
  FieldParams fp;
  field_params_from_modulus(p, &fp);
  MontGeneric F(fp);
  MontGeneric::Acc acc;
  F.acc_zero(acc);
  for (size_t i = 0; i < n; ++i) F.acc_mac(acc, a[i], b[i]);
  return F.acc_get(acc);
}
// the three-class accumulator of the folds, and a sum split over two accumulators joined by acc_add (block reductions)
u64 fh_gen_dot3(u64 p, const u64* a, const u64* b, size_t n) {
  FieldParams fp;
  field_params_from_modulus(p, &fp);
  MontGeneric F(fp);
  MontGeneric::Acc3 acc;
  F.acc3_zero(acc);
  for (size_t i = 0; i < n; ++i) F.acc3_mac(acc, a[i], b[i]);
  return F.acc3_get(acc);
}
u64 fh_gen_dot_split(u64 p, const u64* a, const u64* b, size_t n) {
  FieldParams fp;
  field_params_from_modulus(p, &fp);
  MontGeneric F(fp);
  MontGeneric::Acc lo, hi;
  F.acc_zero(lo);
  F.acc_zero(hi);
  for (size_t i = 0; i < n; ++i) F.acc_mac((i & 1) ? hi : lo, a[i], b[i]);
  F.acc_add(lo, hi);
  return F.acc_get(lo);
}
u64 fh_gold_dot_split(const u64* a, const u64* b, size_t n) {
  GoldilocksMont F;
  GoldilocksMont::Acc lo, hi;
  F.acc_zero(lo);
  F.acc_zero(hi);
  for (size_t i = 0; i < n; ++i) F.acc_mac((i & 1) ? hi : lo, a[i], b[i]);
  F.acc_add(lo, hi);
  return F.acc_get(lo);
}
void fh_params(u64 p, u64* out4) {
  FieldParams fp;
  field_params_from_modulus(p, &fp);
  out4[0] = fp.p; out4[1] = fp.p_inv_neg; out4[2] = fp.r_mod_p; out4[3] = fp.r2_mod_p;
}
u64 fh_splitmix64(u64 x) { return splitmix64(x); }
}
