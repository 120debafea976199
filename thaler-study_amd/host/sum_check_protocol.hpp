// C++ host-side mirror of the reference crate `sum-check-protocol` (src/lib.rs) above the
// C ABI (include/sumcheck_hip.h).  The reference is Rust and this image has no Rust
// toolchain, so the compiled host layer is C++ with the same names, argument meaning and
// error behaviour; the Rust binding itself is under rust/ (source only).
//
//   RngF :13-21 | Error :24-31 | BooleanHypercube :34-70 | Prover :73-117
//   SumCheckPolynomial :121-156 | Verifier :227-331 | VerifierRoundResult :246-253
//
// Field elements are Montgomery words (uint64_t), exactly ark-ff's Fp64<MontBackend<_,1>>.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/sumcheck_hip.h"

namespace sum_check_protocol {

typedef uint64_t F;  // Montgomery word

// Fp64<MontBackend<_,1>> arithmetic on the host (O(1) per round work only).
struct Field {
  sc_field c;
  explicit Field(uint64_t p) {
    if (sc_field_from_modulus(p, &c) != SC_OK) throw std::invalid_argument("modulus must be an odd prime < 2^64");
  }
  F zero() const { return 0; }
  F one() const { return c.r_mod_p; }
  F from_int(uint64_t x) const { return sc_field_to_mont(&c, x); }
  uint64_t to_int(F m) const { return sc_field_from_mont(&c, m); }
  F add(F a, F b) const { unsigned __int128 s = (unsigned __int128)a + b; return (F)(s >= c.p ? s - c.p : s); }
  F sub(F a, F b) const { return a >= b ? a - b : a + (c.p - b); }
  F neg(F a) const { return a ? c.p - a : 0; }
  F mul(F a, F b) const {
    unsigned __int128 t = (unsigned __int128)a * b;
    uint64_t m = (uint64_t)t * c.p_inv_neg;
    unsigned __int128 mp = (unsigned __int128)m * c.p;
    unsigned __int128 s = (t >> 64) + (mp >> 64) + ((((unsigned __int128)(uint64_t)t) + (uint64_t)mp) >> 64);
    return (F)(s >= c.p ? s - c.p : s);
  }
  F pow(F a, uint64_t e) const { F r = one(); while (e) { if (e & 1) r = mul(r, a); a = mul(a, a); e >>= 1; } return r; }
  F inv(F a) const { if (a == 0) throw std::domain_error("inverse of zero"); return pow(a, c.p - 2); }
};

// :13-15
struct RngF {
  virtual ~RngF() {}
  virtual F draw() = 0;
};

// :24-31
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
struct ProverClaimMismatch : Error {
  ProverClaimMismatch(const std::string& a, const std::string& b) : Error("prover claim mismatches evaluation " + a + " " + b) {}
};
struct NoPolySet : Error { NoPolySet() : Error("verifier has no oracle access to the polynomial") {} };

// :34-70 - {0,1}^n, index bit 0 first
class BooleanHypercube {
 public:
  BooleanHypercube(const Field& f, uint32_t n) : f_(f), n_(n), current_(0) {}
  bool next(std::vector<F>* out) {
    if (current_ == ((uint64_t)1 << n_)) return false;
    out->clear();
    for (uint32_t i = 0; i < n_; ++i) out->push_back(((current_ >> i) & 1) ? f_.one() : f_.zero());
    ++current_;
    return true;
  }
 private:
  const Field& f_;
  uint32_t n_;
  uint64_t current_;
};

// ark_poly::univariate::SparsePolynomial<F>: (degree, coeff) terms sorted by degree, in arkworks' canonical form (what
// serialize_uncompressed puts on the wire, fiat-shamir/src/lib.rs:45-61): from_coefficients_vec pops the TRAILING zero terms
// of the vector as given, then sorts (zero terms elsewhere stay); add returns the other operand as is when one is zero, else
// merges, dropping a term of both whose sum is zero and copying a term of one as is; from_dense (`From<DensePolynomial>`)
// keeps the non-zero coefficients.
struct SparsePolynomial {
  std::vector<std::pair<size_t, F>> coeffs;
  static SparsePolynomial from_coefficients_vec(std::vector<std::pair<size_t, F>> v) {
    while (!v.empty() && v.back().second == 0) v.pop_back();
    std::stable_sort(v.begin(), v.end(), [](const std::pair<size_t, F>& a, const std::pair<size_t, F>& b) { return a.first < b.first; });
    if (!v.empty() && v.back().second == 0) throw std::logic_error("from_coefficients_vec: the highest term is zero");
    SparsePolynomial p;
    p.coeffs = std::move(v);
    return p;
  }
  static SparsePolynomial from_dense(const std::vector<F>& dense) {
    SparsePolynomial p;
    for (size_t d = 0; d < dense.size(); ++d) if (dense[d] != 0) p.coeffs.push_back({d, dense[d]});
    return p;
  }
  bool is_zero() const { for (auto& t : coeffs) if (t.second != 0) return false; return true; }
  SparsePolynomial add(const Field& f, const SparsePolynomial& o) const {
    if (is_zero()) return o;
    if (o.is_zero()) return *this;
    SparsePolynomial out;
    size_t i = 0, k = 0;
    while (i < coeffs.size() && k < o.coeffs.size()) {
      if (coeffs[i].first < o.coeffs[k].first) out.coeffs.push_back(coeffs[i++]);
      else if (coeffs[i].first > o.coeffs[k].first) out.coeffs.push_back(o.coeffs[k++]);
      else {
        F c = f.add(coeffs[i].second, o.coeffs[k].second);
        if (c != 0) out.coeffs.push_back({coeffs[i].first, c});
        ++i, ++k;
      }
    }
    for (; i < coeffs.size(); ++i) out.coeffs.push_back(coeffs[i]);
    for (; k < o.coeffs.size(); ++k) out.coeffs.push_back(o.coeffs[k]);
    return out;
  }
  F evaluate(const Field& f, F x) const {
    F acc = 0;
    for (auto& t : coeffs) { F term = t.second; for (size_t i = 0; i < t.first; ++i) term = f.mul(term, x); acc = f.add(acc, term); }
    return acc;
  }
};

// :121-156
struct SumCheckPolynomial {
  virtual ~SumCheckPolynomial() {}
  virtual std::optional<F> evaluate(const std::vector<F>& point) const = 0;
  virtual std::unique_ptr<SumCheckPolynomial> fix_variables(const std::vector<F>& partial_point) const = 0;
  virtual SparsePolynomial to_univariate() const = 0;
  virtual size_t num_vars() const = 0;
  virtual std::vector<F> to_evaluations() const = 0;
  virtual const Field& field() const = 0;
  // provided methods (additive): keep Prover::new / round on the device
  virtual F hypercube_sum() const { F s = 0; for (F v : to_evaluations()) s = field().add(s, v); return s; }
  struct RoundEngine { virtual ~RoundEngine() {} virtual F c_1() const = 0; virtual SparsePolynomial round(F r_prev, size_t j) = 0; };
  virtual std::unique_ptr<RoundEngine> native_engine() const { return nullptr; }
  virtual std::unique_ptr<SumCheckPolynomial> clone() const = 0;
};

// :73-117
class Prover {
 public:
  explicit Prover(std::unique_ptr<SumCheckPolynomial> g) : g_(std::move(g)) {
    engine_ = g_->native_engine();
    c_1_ = engine_ ? engine_->c_1() : g_->hypercube_sum();   // :89
    num_vars_ = g_->num_vars();
    r_.reserve(num_vars_);
  }
  F c_1() const { return c_1_; }
  SparsePolynomial round(F r_prev, size_t j) {               // :105-112
    if (j != 0) r_.push_back(r_prev);
    if (engine_) return engine_->round(r_prev, j);
    if (j != 0) g_ = g_->fix_variables({r_prev});
    return g_->to_univariate();
  }
  size_t num_vars() const { return num_vars_; }
 private:
  std::unique_ptr<SumCheckPolynomial> g_;
  std::unique_ptr<SumCheckPolynomial::RoundEngine> engine_;
  F c_1_ = 0;
  std::vector<F> r_;
  size_t num_vars_ = 0;
};

// :246-253
struct VerifierRoundResult {
  enum Kind { JthRound, FinalRound } kind;
  F r = 0;         // JthRound(r)
  bool ok = false; // FinalRound(ok)
};

// :227-331
class Verifier {
 public:
  Verifier(size_t n, std::unique_ptr<SumCheckPolynomial> g, const Field& f) : n_(n), g_(std::move(g)), f_(f) {}
  void set_c_1(F c_1) { c_1_ = c_1; }
  VerifierRoundResult round(const SparsePolynomial& g_j, RngF& rng) {
    F r_j = rng.draw();                                                         // :283
    if (r_.empty()) {                                                           // :284-297
      F evaluation = f_.add(g_j.evaluate(f_, f_.zero()), g_j.evaluate(f_, f_.one()));
      if (c_1_ != evaluation)
        throw ProverClaimMismatch("start " + std::to_string(f_.to_int(c_1_)), std::to_string(f_.to_int(evaluation)));
      g_part_.push_back(g_j);
      r_.push_back(r_j);
      return {VerifierRoundResult::JthRound, r_j, false};
    } else if (r_.size() == n_ - 1) {                                           // :298-310
      r_.push_back(r_j);
      if (!g_) throw NoPolySet();
      F lhs = g_j.evaluate(f_, r_j);
      F rhs = g_->evaluate(r_).value();
      if (lhs != rhs) throw std::logic_error("assert_eq!(g_j.evaluate(&r_j), g.evaluate(&self.r).unwrap()) failed");  // :303
      return {VerifierRoundResult::FinalRound, 0, lhs == rhs};
    }
    F prev_evaluation = g_part_.back().evaluate(f_, r_.back());                 // :313-328
    F evaluation = f_.add(g_j.evaluate(f_, f_.zero()), g_j.evaluate(f_, f_.one()));
    if (prev_evaluation != evaluation)
      throw ProverClaimMismatch(std::to_string(f_.to_int(prev_evaluation)), std::to_string(f_.to_int(evaluation)));
    g_part_.push_back(g_j);
    r_.push_back(r_j);
    return {VerifierRoundResult::JthRound, r_j, false};
  }
 private:
  size_t n_;
  F c_1_ = 0;
  std::vector<SparsePolynomial> g_part_;
  std::vector<F> r_;
  std::unique_ptr<SumCheckPolynomial> g_;
  const Field& f_;
};

}  // namespace sum_check_protocol
