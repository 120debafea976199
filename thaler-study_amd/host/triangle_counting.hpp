// C++ host-side mirror of the reference crate `triangle-counting` (src/lib.rs): G = f(X,Y) f(Y,Z) f(X,Z) over
// three copies of the adjacency MLE (:22-27), G::new_adj_matrix (:32-51) and the SumCheckPolynomial impl
// (:70-166), all through the C ABI (sc_tri_*).  The same call sequence the Rust `GpuTriangleG` makes.
#pragma once
#include "matrix_multiplication.hpp"

namespace triangle_counting {

using matrix_multiplication::round_poly_from_evals;
using sum_check_protocol::F;
using sum_check_protocol::Field;
using sum_check_protocol::SparsePolynomial;
using sum_check_protocol::SumCheckPolynomial;
using sumcheck_hip::Context;
using sumcheck_hip::DeviceMle;

class G : public SumCheckPolynomial {
 public:
  G(std::shared_ptr<DeviceMle> f1, std::shared_ptr<DeviceMle> f2, std::shared_ptr<DeviceMle> f3, size_t var_len)
      : f1_(std::move(f1)), f2_(std::move(f2)), f3_(std::move(f3)), var_len_(var_len) {}

  // :32-51 - matrix: row-major booleans of a 2^(num_vars/2)-vertex graph
  static G new_adj_matrix(const Context& ctx, size_t num_vars, const std::vector<bool>& matrix) {
    std::vector<F> ev(matrix.size());
    for (size_t i = 0; i < matrix.size(); ++i) ev[i] = matrix[i] ? ctx.field().one() : ctx.field().zero();
    auto g = DeviceMle::from_evaluations_vec(ctx, num_vars, ev);
    return G(g, g, g, num_vars / 2);
  }

  // :53-67
  size_t x_vars_num() const { return f1_->num_vars() > var_len_ ? f1_->num_vars() - var_len_ : 0; }
  size_t y_vars_num() const { return f2_->num_vars() > var_len_ ? f2_->num_vars() - var_len_ : 0; }
  size_t z_vars_num() const { return f3_->num_vars() < var_len_ ? f3_->num_vars() : var_len_; }

  // ---- SumCheckPolynomial (:70-166)
  std::optional<F> evaluate(const std::vector<F>& point) const override {
    if (point.size() != num_vars()) return std::nullopt;
    F out = 0;
    ctx().check(sc_tri_evaluate(ctx().raw(), f1_->raw(), f2_->raw(), f3_->raw(), var_len_, point.data(), point.size(), &out), "sc_tri_evaluate");
    return out;
  }
  std::unique_ptr<SumCheckPolynomial> fix_variables(const std::vector<F>& pp) const override {
    sc_table *h1 = nullptr, *h2 = nullptr, *h3 = nullptr;
    ctx().check(sc_tri_fix_variables(ctx().raw(), f1_->raw(), f2_->raw(), f3_->raw(), var_len_, pp.data(), pp.size(), &h1, &h2, &h3),
                "sc_tri_fix_variables");
    return std::make_unique<G>(std::make_shared<DeviceMle>(ctx(), h1), std::make_shared<DeviceMle>(ctx(), h2),
                               std::make_shared<DeviceMle>(ctx(), h3), var_len_);
  }
  SparsePolynomial to_univariate() const override {
    F e[3];
    ctx().check(sc_tri_round_sums(ctx().raw(), f1_->raw(), f2_->raw(), f3_->raw(), var_len_, e), "sc_tri_round_sums");
    return round_poly_from_evals(field(), e);
  }
  size_t num_vars() const override { return x_vars_num() + y_vars_num() + z_vars_num(); }
  std::vector<F> to_evaluations() const override {
    sc_table* h = nullptr;
    ctx().check(sc_tri_to_evaluations(ctx().raw(), f1_->raw(), f2_->raw(), f3_->raw(), var_len_, &h), "sc_tri_to_evaluations");
    return DeviceMle(ctx(), h).to_evaluations();
  }
  const Field& field() const override { return ctx().field(); }
  std::unique_ptr<SumCheckPolynomial> clone() const override { return std::make_unique<G>(f1_, f2_, f3_, var_len_); }

  // sc_tri_prover: one n^3 pass + three product-of-two-tables sumchecks behind Prover::round
  class Engine : public SumCheckPolynomial::RoundEngine {
   public:
    explicit Engine(const G& g) : adj_(g.f1_) {
      ctx().check(sc_tri_prover_create(ctx().raw(), adj_->raw(), g.var_len_, &h_), "sc_tri_prover_create");
    }
    ~Engine() override { sc_tri_prover_destroy(h_); }
    F c_1() const override { F out = 0; ctx().check(sc_tri_prover_c1(h_, &out), "sc_tri_prover_c1"); return out; }
    SparsePolynomial round(F r_prev, size_t j) override {
      F e[3];
      ctx().check(sc_tri_prover_round(h_, r_prev, j, e), "sc_tri_prover_round");
      return round_poly_from_evals(ctx().field(), e);
    }
   private:
    const Context& ctx() const { return adj_->ctx(); }
    std::shared_ptr<DeviceMle> adj_;
    sc_tri_prover* h_ = nullptr;
  };
  // the fast engine applies to the polynomial as new_adj_matrix builds it (three views of one table, nothing fixed)
  std::unique_ptr<RoundEngine> native_engine() const override {
    if (f1_ == f2_ && f2_ == f3_ && var_len_ >= 1 && f1_->num_vars() == 2 * var_len_) return std::make_unique<Engine>(*this);
    return nullptr;
  }

 private:
  const Context& ctx() const { return f1_->ctx(); }
  std::shared_ptr<DeviceMle> f1_, f2_, f3_;
  size_t var_len_;
};

}  // namespace triangle_counting
