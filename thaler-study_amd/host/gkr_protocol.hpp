// C++ host-side mirror of the reference crate `gkr-protocol`:
//   circuit.rs:7-213            Gate, CircuitLayer, Circuit (evaluate, num_vars_at, add_i, mul_i)
//   round_polynomial.rs:23-119  W = add_i(r_i,b,c)(W(b)+W(c)) + mul_i(r_i,b,c) W(b) W(c) as a SumCheckPolynomial
//   lib.rs:278-321              line, restrict_poly
//   lib.rs:38-218, :324-474     the Verifier / Prover message state machines
// every table operation through the C ABI (sc_gkr_*, sc_table_*): the same call sequences the Rust `GpuW` makes.
#pragma once
#include <algorithm>

#include "matrix_multiplication.hpp"

namespace gkr_protocol {

using matrix_multiplication::round_poly_from_evals;
using sum_check_protocol::F;
using sum_check_protocol::Field;
using sum_check_protocol::RngF;
using sum_check_protocol::SparsePolynomial;
using sum_check_protocol::SumCheckPolynomial;
using sum_check_protocol::VerifierRoundResult;
using sumcheck_hip::Context;
using sumcheck_hip::DeviceMle;

enum class GateType { Add, Mul };
struct Gate { GateType ttype; size_t inputs[2]; };          // circuit.rs:17-31
typedef std::vector<Gate> CircuitLayer;                     // circuit.rs:34-52

struct Circuit {                                            // circuit.rs:72-213; layers[0] = outputs
  std::vector<CircuitLayer> layers;
  size_t num_inputs;
  std::optional<size_t> num_vars_at(size_t layer) const {
    size_t n;
    if (layer < layers.size()) n = layers[layer].size();
    else if (layer == layers.size()) n = num_inputs;
    else return std::nullopt;
    size_t k = 0;
    while (n > 1 && (n & 1) == 0) { n >>= 1; ++k; }         // trailing_zeros
    return k;
  }
  std::vector<std::vector<F>> evaluate(const Field& f, const std::vector<F>& input) const {   // :99-124
    std::vector<std::vector<F>> out{input};
    std::vector<F> cur = input;
    for (size_t i = layers.size(); i-- > 0;) {
      std::vector<F> nxt;
      for (const Gate& g : layers[i])
        nxt.push_back(g.ttype == GateType::Add ? f.add(cur[g.inputs[0]], cur[g.inputs[1]]) : f.mul(cur[g.inputs[0]], cur[g.inputs[1]]));
      out.push_back(nxt);
      cur = nxt;
    }
    std::reverse(out.begin(), out.end());
    return out;
  }
};

// add_i(r_i,.,.) and mul_i(r_i,.,.) (lib.rs:388-416; circuit.rs:156-213) straight from the gate list
inline std::pair<std::shared_ptr<DeviceMle>, std::shared_ptr<DeviceMle>> wiring(const Context& ctx, const Circuit& c, size_t i,
                                                                                const std::vector<F>& r_i) {
  const CircuitLayer& layer = c.layers[i];
  std::vector<int32_t> gt;
  std::vector<uint32_t> i0, i1;
  for (const Gate& g : layer) { gt.push_back(g.ttype == GateType::Add ? 0 : 1); i0.push_back((uint32_t)g.inputs[0]); i1.push_back((uint32_t)g.inputs[1]); }
  sc_table *ha = nullptr, *hm = nullptr;
  ctx.check(sc_gkr_wiring(ctx.raw(), gt.data(), i0.data(), i1.data(), *c.num_vars_at(i), *c.num_vars_at(i + 1), r_i.data(), &ha, &hm), "sc_gkr_wiring");
  return {std::make_shared<DeviceMle>(ctx, ha), std::make_shared<DeviceMle>(ctx, hm)};
}

// round_polynomial.rs:23-44
class W : public SumCheckPolynomial {
 public:
  W(std::shared_ptr<DeviceMle> add_i, std::shared_ptr<DeviceMle> mul_i, std::shared_ptr<DeviceMle> w_b, std::shared_ptr<DeviceMle> w_c)
      : add_(std::move(add_i)), mul_(std::move(mul_i)), wb_(std::move(w_b)), wc_(std::move(w_c)) {}
  std::optional<F> evaluate(const std::vector<F>& point) const override {          // :48-57
    if (point.size() != num_vars()) return std::nullopt;
    F out = 0;
    ctx().check(sc_gkr_w_evaluate(ctx().raw(), add_->raw(), mul_->raw(), wb_->raw(), wc_->raw(), point.data(), point.size(), &out), "sc_gkr_w_evaluate");
    return out;
  }
  std::unique_ptr<SumCheckPolynomial> fix_variables(const std::vector<F>& pp) const override {   // :59-76
    sc_table* h[4] = {nullptr, nullptr, nullptr, nullptr};
    ctx().check(sc_gkr_w_fix_variables(ctx().raw(), add_->raw(), mul_->raw(), wb_->raw(), wc_->raw(), pp.data(), pp.size(), &h[0], &h[1], &h[2], &h[3]),
                "sc_gkr_w_fix_variables");
    return std::make_unique<W>(std::make_shared<DeviceMle>(ctx(), h[0]), std::make_shared<DeviceMle>(ctx(), h[1]),
                               std::make_shared<DeviceMle>(ctx(), h[2]), std::make_shared<DeviceMle>(ctx(), h[3]));
  }
  SparsePolynomial to_univariate() const override {                                 // :78-90
    F e[3];
    ctx().check(sc_gkr_w_round_sums(ctx().raw(), add_->raw(), mul_->raw(), wb_->raw(), wc_->raw(), e), "sc_gkr_w_round_sums");
    return round_poly_from_evals(field(), e);
  }
  size_t num_vars() const override { return add_->num_vars(); }                     // :92-94
  std::vector<F> to_evaluations() const override {                                  // :96-118
    sc_table* h = nullptr;
    ctx().check(sc_gkr_w_to_evaluations(ctx().raw(), add_->raw(), mul_->raw(), wb_->raw(), wc_->raw(), &h), "sc_gkr_w_to_evaluations");
    return DeviceMle(ctx(), h).to_evaluations();
  }
  const Field& field() const override { return ctx().field(); }
  std::unique_ptr<SumCheckPolynomial> clone() const override { return std::make_unique<W>(add_, mul_, wb_, wc_); }

  class Engine : public SumCheckPolynomial::RoundEngine {   // sc_gkr_prover: the two-phase W prover
   public:
    explicit Engine(const W& w) : add_(w.add_), mul_(w.mul_), wb_(w.wb_), wc_(w.wc_) {
      ctx().check(sc_gkr_prover_create(ctx().raw(), add_->raw(), mul_->raw(), wb_->raw(), wc_->raw(), &h_), "sc_gkr_prover_create");
    }
    ~Engine() override { sc_gkr_prover_destroy(h_); }
    F c_1() const override { F out = 0; ctx().check(sc_gkr_prover_c1(h_, &out), "sc_gkr_prover_c1"); return out; }
    SparsePolynomial round(F r_prev, size_t j) override {
      F e[3];
      ctx().check(sc_gkr_prover_round(h_, r_prev, j, e), "sc_gkr_prover_round");
      return round_poly_from_evals(ctx().field(), e);
    }
   private:
    const Context& ctx() const { return add_->ctx(); }
    std::shared_ptr<DeviceMle> add_, mul_, wb_, wc_;
    sc_gkr_prover* h_ = nullptr;
  };
  std::unique_ptr<RoundEngine> native_engine() const override { return std::make_unique<Engine>(*this); }
  const std::shared_ptr<DeviceMle>& w_b() const { return wb_; }

 private:
  const Context& ctx() const { return add_->ctx(); }
  std::shared_ptr<DeviceMle> add_, mul_, wb_, wc_;
};

// lib.rs:278-289
inline std::vector<SparsePolynomial> line(const Field& f, const std::vector<F>& b, const std::vector<F>& c) {
  std::vector<SparsePolynomial> out;
  for (size_t i = 0; i < b.size(); ++i) out.push_back(SparsePolynomial::from_coefficients_vec({{0, b[i]}, {1, f.sub(c[i], b[i])}}));
  return out;
}
// lib.rs:291-321
inline SparsePolynomial restrict_poly(const std::vector<F>& b, const std::vector<F>& c, const DeviceMle& mle) {
  const size_t k = mle.num_vars();
  std::vector<F> coeffs(k + 1);
  mle.ctx().check(sc_table_restrict_to_line(mle.ctx().raw(), mle.raw(), b.data(), c.data(), k, coeffs.data()), "sc_table_restrict_to_line");
  return SparsePolynomial::from_dense(coeffs);
}

// lib.rs:255-289 / :231-252
struct ProverMessage {
  enum Kind { Begin, SumCheckProverMessage, FinalRoundMessage, StartSumCheck } kind;
  std::vector<F> circuit_outputs;   // Begin
  SparsePolynomial p, q;            // SumCheckProverMessage { p } / FinalRoundMessage { p, q }
  F c_1 = 0;                        // StartSumCheck
  size_t round = 0, num_vars = 0;
};
struct VerifierMessage {
  enum Kind { SumCheckRoundResult, RoundStarted, R } kind;
  VerifierRoundResult res{VerifierRoundResult::JthRound, 0, false};
  size_t round = 0;
  std::vector<F> r;
};
struct WrongVerifierState : std::runtime_error { WrongVerifierState() : std::runtime_error("Verifier is in the wrong state.") {} };

// lib.rs:38-218
class Verifier {
 public:
  Verifier(const Context& ctx, const Circuit& circuit) : ctx_(ctx), circuit_(circuit) {}
  VerifierMessage final_random_point(RngF& rng) {                                   // :110-121
    if (!running_) throw WrongVerifierState();
    F p = rng.draw();
    bc_.push_back(p);
    VerifierMessage m{VerifierMessage::SumCheckRoundResult};
    m.res = {VerifierRoundResult::JthRound, p, false};
    return m;
  }
  VerifierMessage receive_prover_msg(const ProverMessage& msg, RngF& rng) {         // :177-207
    const Field& f = ctx_.field();
    switch (msg.kind) {
      case ProverMessage::SumCheckProverMessage: {                                  // :123-139
        if (!running_) throw WrongVerifierState();
        VerifierMessage m{VerifierMessage::SumCheckRoundResult};
        m.res = sc_verifier_->round(msg.p, rng);
        if (m.res.kind == VerifierRoundResult::JthRound) bc_.push_back(m.res.r);
        return m;
      }
      case ProverMessage::StartSumCheck: {                                          // :89-107
        auto am = wiring(ctx_, circuit_, msg.round, r_.back());
        add_i_ = am.first;
        mul_i_ = am.second;
        sc_verifier_ = std::make_unique<sum_check_protocol::Verifier>(msg.num_vars, nullptr, f);
        sc_verifier_->set_c_1(msg.c_1);
        bc_.clear();
        running_ = true;
        VerifierMessage m{VerifierMessage::RoundStarted};
        m.round = msg.round;
        return m;
      }
      case ProverMessage::FinalRoundMessage: {                                      // :141-174
        if (!running_) throw WrongVerifierState();
        F q_0 = msg.q.evaluate(f, f.zero()), q_1 = msg.q.evaluate(f, f.one());
        F eval = f.add(f.mul(add_i_->evaluate(bc_), f.add(q_0, q_1)), f.mul(f.mul(mul_i_->evaluate(bc_), q_0), q_1));
        if (eval != msg.p.evaluate(f, bc_.back())) throw std::logic_error("assert_eq!(eval, p.evaluate(bc.last().unwrap())) failed");
        F r = rng.draw();
        size_t half = bc_.size() / 2;
        std::vector<F> b(bc_.begin(), bc_.begin() + half), c(bc_.begin() + half, bc_.end()), r_next;
        for (const SparsePolynomial& l : line(f, b, c)) r_next.push_back(l.evaluate(f, r));
        r_.push_back(r_next);
        m_.push_back(msg.q.evaluate(f, r));
        VerifierMessage m{VerifierMessage::R};
        m.r = r_next;
        return m;
      }
      default: {                                                                    // Begin, :186-205
        size_t k0 = *circuit_.num_vars_at(0);
        auto d = DeviceMle::from_evaluations_vec(ctx_, k0, msg.circuit_outputs);
        std::vector<F> r_zero;
        for (size_t i = 0; i < k0; ++i) r_zero.push_back(rng.draw());
        r_ = {r_zero};
        m_ = {d->evaluate(r_zero)};
        VerifierMessage m{VerifierMessage::R};
        m.r = r_zero;
        return m;
      }
    }
  }
  bool check_input(const std::vector<F>& input) const {                             // :210-217
    size_t k = 0;
    while (((size_t)1 << k) < input.size()) ++k;
    return DeviceMle::from_evaluations_vec(ctx_, k, input)->evaluate(r_.back()) == m_.back();
  }
 private:
  const Context& ctx_;
  const Circuit& circuit_;
  std::vector<std::vector<F>> r_;
  std::vector<F> m_, bc_;
  bool running_ = false;
  std::unique_ptr<sum_check_protocol::Verifier> sc_verifier_;
  std::shared_ptr<DeviceMle> add_i_, mul_i_;
};

// lib.rs:324-474
class Prover {
 public:
  Prover(const Context& ctx, const Circuit& circuit, const std::vector<F>& input)
      : ctx_(ctx), circuit_(circuit), evaluation_(circuit.evaluate(ctx.field(), input)) {}
  ProverMessage start_protocol() const {                                            // :363-367
    ProverMessage m{ProverMessage::Begin};
    m.circuit_outputs = evaluation_.front();
    return m;
  }
  ProverMessage start_round(size_t i, const std::vector<F>& r_i) {                  // :373-436
    size_t k_next = *circuit_.num_vars_at(i + 1);
    auto w_b = DeviceMle::from_evaluations_vec(ctx_, k_next, evaluation_[i + 1]);
    w_ = w_b;
    auto am = wiring(ctx_, circuit_, i, r_i);
    if (am.first->num_vars() != 2 * w_b->num_vars()) throw std::logic_error("assert_eq!(add_i.num_vars(), 2 * w_b.num_vars())");
    i_ = i;
    prover_ = std::make_unique<sum_check_protocol::Prover>(std::make_unique<W>(am.first, am.second, w_b, w_b));
    r_.clear();
    ProverMessage m{ProverMessage::StartSumCheck};
    m.c_1 = prover_->c_1();
    m.round = i;
    m.num_vars = am.first->num_vars();
    return m;
  }
  ProverMessage round_msg(size_t j) {                                               // :439-456
    const Field& f = ctx_.field();
    if (j == 2 * *circuit_.num_vars_at(i_ + 1) - 1) {
      size_t half = r_.size() / 2;
      std::vector<F> b(r_.begin(), r_.begin() + half), c(r_.begin() + half, r_.end());
      ProverMessage m{ProverMessage::FinalRoundMessage};
      m.q = restrict_poly(b, c, *w_);
      m.p = prover_->round(j ? r_[j - 1] : f.one(), j);
      return m;
    }
    ProverMessage m{ProverMessage::SumCheckProverMessage};
    m.p = prover_->round(j == 0 ? f.one() : r_[j - 1], j);
    return m;
  }
  void receive_verifier_msg(const VerifierMessage& vm) {                            // :459-468
    if (vm.kind == VerifierMessage::SumCheckRoundResult) {
      if (vm.res.kind != VerifierRoundResult::JthRound) throw std::logic_error("panic!()");
      r_.push_back(vm.res.r);
    }
  }
  F c_1() const { return prover_->c_1(); }
 private:
  const Context& ctx_;
  const Circuit& circuit_;
  std::vector<std::vector<F>> evaluation_;
  size_t i_ = 0;
  std::unique_ptr<sum_check_protocol::Prover> prover_;
  std::shared_ptr<DeviceMle> w_;
  std::vector<F> r_;
};

}  // namespace gkr_protocol
