// C++ host-side mirror of the reference crate `matrix-multiplication` (src/lib.rs): `G` as two
// device tables (:12-15), G::new (:77-92), the SumCheckPolynomial impl (:95-147) and
// interpolate_quadratic_poly (:17-60), all through the C ABI.  multilinear_extensions::
// {vsbw_,cti_}multilinear_from_evaluations (multilinear-extensions/src/lib.rs:6-48) ride along.
#pragma once
#include <array>
#include <string>

#include "sum_check_protocol.hpp"

namespace sumcheck_hip {

using sum_check_protocol::F;
using sum_check_protocol::Field;

// One GPU + one stream + one field (sc_ctx).  Panics (throws) on a non-zero status, like the
// reference's infallible prover methods panic on misuse.
class Context {
 public:
  Context(const Field& f, int device = 0) : field_(f) {
    if (sc_ctx_create(&f.c, device, &h_) != SC_OK) throw std::runtime_error(std::string("sc_ctx_create: ") + sc_last_error(nullptr));
  }
  ~Context() { sc_ctx_destroy(h_); }
  Context(const Context&) = delete;
  sc_ctx* raw() const { return h_; }
  const Field& field() const { return field_; }
  void check(int rc, const char* what) const {
    if (rc != SC_OK) throw std::runtime_error(std::string(what) + " failed: " + sc_last_error(h_));
  }
 private:
  sc_ctx* h_ = nullptr;
  const Field& field_;
};

// Device-resident DenseMultilinearExtension<F> (shared, immutable once built).
class DeviceMle {
 public:
  DeviceMle(const Context& ctx, sc_table* h) : ctx_(ctx), h_(h) {}
  ~DeviceMle() { sc_table_free(ctx_.raw(), h_); }
  DeviceMle(const DeviceMle&) = delete;
  static std::shared_ptr<DeviceMle> from_evaluations_vec(const Context& ctx, size_t num_vars, const std::vector<F>& ev) {
    if (ev.size() != ((size_t)1 << num_vars)) throw std::invalid_argument("The size of evaluations should be 2^num_vars.");
    sc_table* h = nullptr;
    ctx.check(sc_table_upload(ctx.raw(), ev.data(), ev.size(), &h), "sc_table_upload");
    return std::make_shared<DeviceMle>(ctx, h);
  }
  size_t num_vars() const { size_t l = sc_table_len(h_), n = 0; while (((size_t)1 << n) < l) ++n; return n; }
  std::shared_ptr<DeviceMle> fix_variables(const std::vector<F>& r, int order = SC_ORDER_LE) const {
    sc_table* h = nullptr;
    ctx_.check(sc_table_fix_variables(ctx_.raw(), h_, r.data(), r.size(), order, &h), "sc_table_fix_variables");
    return std::make_shared<DeviceMle>(ctx_, h);
  }
  F evaluate(const std::vector<F>& point, int order = SC_ORDER_LE) const {
    F out = 0;
    ctx_.check(sc_table_evaluate(ctx_.raw(), h_, point.data(), point.size(), order, &out), "sc_table_evaluate");
    return out;
  }
  std::vector<F> to_evaluations() const {
    std::vector<F> v(sc_table_len(h_));
    ctx_.check(sc_table_download(ctx_.raw(), h_, v.data(), v.size()), "sc_table_download");
    return v;
  }
  sc_table* raw() const { return h_; }
  const Context& ctx() const { return ctx_; }
 private:
  const Context& ctx_;
  sc_table* h_;
};

}  // namespace sumcheck_hip

namespace multilinear_extensions {
using sum_check_protocol::F;
// multilinear-extensions/src/lib.rs:6-24 and :29-48: both are the BE evaluate of the table
inline F vsbw_multilinear_from_evaluations(const sumcheck_hip::Context& ctx, const std::vector<F>& evals, const std::vector<F>& r) {
  return sumcheck_hip::DeviceMle::from_evaluations_vec(ctx, r.size(), evals)->evaluate(r, SC_ORDER_BE);
}
inline F cti_multilinear_from_evaluations(const sumcheck_hip::Context& ctx, const std::vector<F>& evals, const std::vector<F>& r) {
  return sumcheck_hip::DeviceMle::from_evaluations_vec(ctx, r.size(), evals)->evaluate(r, SC_ORDER_BE);
}
}  // namespace multilinear_extensions

namespace matrix_multiplication {

using sum_check_protocol::F;
using sum_check_protocol::Field;
using sum_check_protocol::SparsePolynomial;
using sum_check_protocol::SumCheckPolynomial;
using sumcheck_hip::Context;
using sumcheck_hip::DeviceMle;

// :17-60 - three Lagrange terms (one division each), each a SparsePolynomial::from_coefficients_vec, summed with
// SparsePolynomial's add: arkworks' canonical form of the sum, explicit zero constant term included where it arises
inline SparsePolynomial interpolate_quadratic_poly(const Field& f, const std::array<std::pair<F, F>, 3>& pts) {
  SparsePolynomial sum;
  for (int i = 0; i < 3; ++i) {
    int j = (i + 1) % 3, k = (i + 2) % 3;
    F den = f.mul(f.sub(pts[i].first, pts[j].first), f.sub(pts[i].first, pts[k].first));
    F w = f.mul(pts[i].second, f.inv(den));
    SparsePolynomial term = SparsePolynomial::from_coefficients_vec(
        {{0, f.mul(f.mul(pts[j].first, pts[k].first), w)}, {1, f.mul(f.sub(f.neg(pts[j].first), pts[k].first), w)}, {2, w}});
    sum = i == 0 ? term : sum.add(f, term);
  }
  return sum;
}

// the round polynomial as triangle_counting::G and W hand it out (`p.into()` of a DensePolynomial): non-zero terms only
inline SparsePolynomial round_poly_from_evals(const Field& f, const F e[3]) {
  F c[3];
  if (sc_interpolate_quadratic(&f.c, e, c) != SC_OK) throw std::runtime_error("sc_interpolate_quadratic");
  return SparsePolynomial::from_dense({c[0], c[1], c[2]});
}

// the round polynomial as matrix_multiplication::G hands it out (:124-130): the three-term sum over the points 0, 1, 2
inline SparsePolynomial round_poly_lagrange(const Field& f, const F e[3]) {
  return interpolate_quadratic_poly(f, {{{f.from_int(0), e[0]}, {f.from_int(1), e[1]}, {f.from_int(2), e[2]}}});
}

// :12-15
class G : public SumCheckPolynomial {
 public:
  G(std::shared_ptr<DeviceMle> f_a, std::shared_ptr<DeviceMle> f_b) : f_a_(std::move(f_a)), f_b_(std::move(f_b)) {}

  // :77-92 - a, b: 2^n x 2^n matrices flattened row-major
  static G create(const Context& ctx, size_t n, const std::vector<F>& a, const std::vector<F>& b, const std::vector<F>& point) {
    auto A = DeviceMle::from_evaluations_vec(ctx, 2 * n, a);
    auto B = DeviceMle::from_evaluations_vec(ctx, 2 * n, b);
    if (point.size() != 2 * n) throw std::invalid_argument("point must have 2n entries");
    sc_table *ha = nullptr, *hb = nullptr;
    ctx.check(sc_matmul_g_new(ctx.raw(), A->raw(), B->raw(), n, point.data(), &ha, &hb), "sc_matmul_g_new");
    return G(std::make_shared<DeviceMle>(ctx, ha), std::make_shared<DeviceMle>(ctx, hb));
  }

  // ---- SumCheckPolynomial (:95-147)
  std::optional<F> evaluate(const std::vector<F>& point) const override {
    if (point.size() != num_vars()) return std::nullopt;
    F out = 0;
    ctx().check(sc_prod2_evaluate(ctx().raw(), f_a_->raw(), f_b_->raw(), point.data(), point.size(), &out), "sc_prod2_evaluate");
    return out;
  }
  std::unique_ptr<SumCheckPolynomial> fix_variables(const std::vector<F>& pp) const override {
    return std::make_unique<G>(f_a_->fix_variables(pp), f_b_->fix_variables(pp));
  }
  SparsePolynomial to_univariate() const override {
    F e[3];
    ctx().check(sc_prod2_round_sums(ctx().raw(), f_a_->raw(), f_b_->raw(), e), "sc_prod2_round_sums");
    return round_poly_lagrange(field(), e);
  }
  size_t num_vars() const override { return f_a_->num_vars(); }
  std::vector<F> to_evaluations() const override {
    sc_table* h = nullptr;
    ctx().check(sc_prod2_to_evaluations(ctx().raw(), f_a_->raw(), f_b_->raw(), &h), "sc_prod2_to_evaluations");
    return DeviceMle(ctx(), h).to_evaluations();
  }
  const Field& field() const override { return ctx().field(); }
  F hypercube_sum() const override {
    F out = 0;
    ctx().check(sc_prod2_sum(ctx().raw(), f_a_->raw(), f_b_->raw(), &out), "sc_prod2_sum");
    return out;
  }
  std::unique_ptr<SumCheckPolynomial> clone() const override { return std::make_unique<G>(f_a_, f_b_); }  // #[derive(Clone)] :11

  // sc_prover: the fused fold + round-sum engine behind Prover::round
  class Engine : public SumCheckPolynomial::RoundEngine {
   public:
    explicit Engine(const G& g) : f_a_(g.f_a_), f_b_(g.f_b_) {  // shares (keeps alive) the borrowed tables
      ctx().check(sc_prover_create(ctx().raw(), f_a_->raw(), f_b_->raw(), &h_), "sc_prover_create");
    }
    ~Engine() override { sc_prover_destroy(h_); }
    F c_1() const override { F out = 0; ctx().check(sc_prover_c1(h_, &out), "sc_prover_c1"); return out; }
    SparsePolynomial round(F r_prev, size_t j) override {
      F e[3];
      ctx().check(sc_prover_round(h_, r_prev, j, e), "sc_prover_round");
      return round_poly_lagrange(ctx().field(), e);
    }
   private:
    const Context& ctx() const { return f_a_->ctx(); }
    std::shared_ptr<DeviceMle> f_a_, f_b_;
    sc_prover* h_ = nullptr;
  };
  std::unique_ptr<RoundEngine> native_engine() const override { return std::make_unique<Engine>(*this); }

 private:
  const Context& ctx() const { return f_a_->ctx(); }
  std::shared_ptr<DeviceMle> f_a_, f_b_;
};

}  // namespace matrix_multiplication
