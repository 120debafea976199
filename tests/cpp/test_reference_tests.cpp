// The reference's own unit tests for this path, replayed in C++ against the C ABI through
// the host mirror (thaler-study_amd/host/*.hpp).  Exit code 0 = all passed.  Needs a GPU.
//   multilinear-extensions/src/lib.rs:76-120   example_from_book
//   matrix-multiplication/src/lib.rs:202-243   matrix_test_from_book
//   matrix-multiplication/src/lib.rs:245-303   example_from_book
//   matrix-multiplication/src/lib.rs:315-374   randomized_test (2^2..2^5, every (i,j))
//   triangle-counting/src/lib.rs:232-317       test_simple_matrix, randomized_test (2..64 vertices)
//   gkr-protocol/src/lib.rs:507-702            test_restrict_poly, protocol_test_from_book, three_layer_protocol_test
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../thaler-study_amd/host/fiat_shamir.hpp"
#include "../../thaler-study_amd/host/gkr_protocol.hpp"
#include "../../thaler-study_amd/host/matrix_multiplication.hpp"
#include "../../thaler-study_amd/host/triangle_counting.hpp"

using namespace sum_check_protocol;
using matrix_multiplication::G;
using sumcheck_hip::Context;

#define REQUIRE(cond)                                                              \
  do {                                                                             \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
  } while (0)

struct StdRng : RngF {  // the blanket impl<F: Field, T: Rng> RngF<F> for T (:17-21)
  const Field& f;
  std::mt19937_64 gen;
  StdRng(const Field& f_, uint64_t seed) : f(f_), gen(seed) {}
  F draw() override { return f.from_int(gen() % f.c.p); }
};

static std::vector<F> u32_to_boolean_vec(const Field& f, uint32_t v, size_t bits) {  // :305-313
  std::vector<F> out;
  for (size_t i = 0; i < bits; ++i) out.push_back(((v >> i) & 1) ? f.one() : f.zero());
  return out;
}

typedef std::vector<std::vector<F>> Matrix;
static Matrix matmul(const Field& f, const Matrix& a, const Matrix& b) {  // :181-200
  size_t n = a.size();
  Matrix res(n, std::vector<F>(n, 0));
  for (size_t i = 0; i < n; ++i)
    for (size_t j = 0; j < n; ++j)
      for (size_t k = 0; k < n; ++k) res[i][j] = f.add(res[i][j], f.mul(a[i][k], b[k][j]));
  return res;
}
static std::vector<F> flatten(const Matrix& m) {
  std::vector<F> v;
  for (auto& row : m) v.insert(v.end(), row.begin(), row.end());
  return v;
}

static void run_protocol(const Field& f, const G& g, F c_1_expected, bool check_c1, RngF& rng) {
  Prover prover(g.clone());
  F c_1 = prover.c_1();
  if (check_c1) REQUIRE(c_1 == c_1_expected);
  size_t num_vars = g.num_vars();
  F r_j = f.one();
  Verifier verifier(num_vars, g.clone(), f);
  verifier.set_c_1(c_1);
  for (size_t j = 0; j < num_vars; ++j) {
    SparsePolynomial g_j = prover.round(r_j, j);
    VerifierRoundResult res = verifier.round(g_j, rng);
    if (res.kind == VerifierRoundResult::JthRound) r_j = res.r;
    else REQUIRE(res.ok);
  }
}

static void run_generic(const Field& f, const SumCheckPolynomial& g, RngF& rng, F* c_1_out) {
  Prover prover(g.clone());
  F c_1 = prover.c_1();
  size_t num_vars = g.num_vars();
  F r_j = f.one();
  Verifier verifier(num_vars, g.clone(), f);
  verifier.set_c_1(c_1);
  bool final_seen = false;
  for (size_t j = 0; j < num_vars; ++j) {
    VerifierRoundResult res = verifier.round(prover.round(r_j, j), rng);
    if (res.kind == VerifierRoundResult::JthRound) r_j = res.r;
    else { REQUIRE(res.ok); final_seen = true; }
  }
  REQUIRE(final_seen);
  if (c_1_out) *c_1_out = c_1;
}

static void triangle_tests() {
  {  // triangle-counting/src/lib.rs:232-266 test_simple_matrix
    Field f(389);
    Context ctx(f);
    StdRng rng(f, 1);
    std::vector<bool> adj = {false, true, true, false, true, false, true, false, true, true, false, false, false, false, false, false};
    triangle_counting::G g = triangle_counting::G::new_adj_matrix(ctx, 4, adj);
    F c_1 = 0;
    run_generic(f, g, rng, &c_1);
    REQUIRE(f.to_int(c_1) == 6);
    std::printf("ok triangle_counting::test_simple_matrix\n");
  }
  {  // :268-317 randomized_test
    Field f(1572869);
    Context ctx(f);
    StdRng rng(f, 2);
    std::mt19937_64 gen(11);
    for (size_t k = 1; k <= 6; ++k) {
      size_t n = (size_t)1 << k;
      std::vector<std::vector<bool>> m(n, std::vector<bool>(n, false));
      for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j) m[i][j] = m[j][i] = (gen() & 1) != 0;
      uint64_t tri = 0;
      for (size_t x = 0; x < n; ++x)
        for (size_t y = 0; y < n; ++y)
          for (size_t z = 0; z < n; ++z) tri += (m[x][y] && m[y][z] && m[x][z]) ? 1 : 0;
      std::vector<bool> flat;
      for (auto& row : m) flat.insert(flat.end(), row.begin(), row.end());
      triangle_counting::G g = triangle_counting::G::new_adj_matrix(ctx, 2 * k, flat);
      F c_1 = 0;
      run_generic(f, g, rng, &c_1);
      REQUIRE(f.to_int(c_1) == tri);   // 6 * triangle count = ordered triples
      if (k <= 3) {                    // the generic trait path (fix_variables -> to_univariate) agrees
        struct NoEngine : triangle_counting::G { using G::G; NoEngine(const G& g) : G(g) {} std::unique_ptr<RoundEngine> native_engine() const override { return nullptr; }
          std::unique_ptr<SumCheckPolynomial> clone() const override { return std::make_unique<NoEngine>(*this); } };
        F c_2 = 0;
        run_generic(f, NoEngine(g), rng, &c_2);
        REQUIRE(c_2 == c_1);
      }
    }
    std::printf("ok triangle_counting::randomized_test\n");
  }
}

static void gkr_tests() {
  using namespace gkr_protocol;
  Field f(389);
  Context ctx(f);
  {  // gkr-protocol/src/lib.rs:507-548 test_restrict_poly
    std::vector<F> b = {f.from_int(2), f.from_int(4)}, c = {f.from_int(3), f.from_int(2)};
    auto mle = sumcheck_hip::DeviceMle::from_evaluations_vec(ctx, 2, {f.from_int(0), f.from_int(0), f.from_int(2), f.from_int(5)});
    SparsePolynomial q = restrict_poly(b, c, *mle);
    uint64_t dense[3] = {0, 0, 0};
    for (auto& t : q.coeffs) dense[t.first] = f.to_int(t.second);
    REQUIRE(dense[0] == 32 && dense[1] == 385 && dense[2] == 383);   // -6t^2 - 4t + 32
    std::printf("ok gkr_protocol::test_restrict_poly\n");
  }
  auto G2 = [](GateType t, size_t a, size_t b) { return Gate{t, {a, b}}; };
  Circuit book{{{G2(GateType::Mul, 0, 1), G2(GateType::Mul, 2, 3)},
                {G2(GateType::Mul, 0, 0), G2(GateType::Mul, 1, 1), G2(GateType::Mul, 1, 2), G2(GateType::Mul, 3, 3)}}, 4};
  Circuit three{{{G2(GateType::Add, 0, 1), G2(GateType::Add, 2, 3)},
                 {G2(GateType::Add, 0, 1), G2(GateType::Add, 2, 3), G2(GateType::Add, 4, 5), G2(GateType::Add, 6, 7)}}, 8};
  struct Case { const char* name; Circuit* c; std::vector<uint64_t> input; std::vector<uint64_t> outputs; };
  Case cases[2] = {{"protocol_test_from_book", &book, {3, 2, 3, 1}, {36, 6}}, {"three_layer_protocol_test", &three, {0, 1, 0, 1, 0, 1, 0, 1}, {2, 2}}};
  for (Case& cs : cases) {
    for (uint64_t seed = 0; seed < 4; ++seed) {
      StdRng rng(f, 100 + seed);
      const Circuit& circuit = *cs.c;
      std::vector<F> input;
      for (uint64_t v : cs.input) input.push_back(f.from_int(v));
      gkr_protocol::Prover prover(ctx, circuit, input);
      ProverMessage begin = prover.start_protocol();                               // :587-596
      REQUIRE(begin.circuit_outputs.size() == cs.outputs.size());
      for (size_t i = 0; i < cs.outputs.size(); ++i) REQUIRE(f.to_int(begin.circuit_outputs[i]) == cs.outputs[i]);
      gkr_protocol::Verifier verifier(ctx, circuit);
      VerifierMessage vm = verifier.receive_prover_msg(begin, rng);
      REQUIRE(vm.kind == VerifierMessage::R);
      std::vector<F> r_i = vm.r;
      for (size_t i = 0; i < circuit.layers.size(); ++i) {                         // :608-621
        ProverMessage msg = prover.start_round(i, r_i);
        size_t num_vars = 2 * *circuit.num_vars_at(i + 1);
        verifier.receive_prover_msg(msg, rng);
        for (size_t j = 0; j + 1 < num_vars; ++j) {
          VerifierMessage v2 = verifier.receive_prover_msg(prover.round_msg(j), rng);
          prover.receive_verifier_msg(v2);
        }
        prover.receive_verifier_msg(verifier.final_random_point(rng));
        VerifierMessage v3 = verifier.receive_prover_msg(prover.round_msg(num_vars - 1), rng);
        REQUIRE(v3.kind == VerifierMessage::R);
        r_i = v3.r;
      }
      REQUIRE(verifier.check_input(input));                                        // :623
      std::vector<F> bad = input;
      bad[0] = f.add(bad[0], f.one());
      REQUIRE(!verifier.check_input(bad));
    }
    std::printf("ok gkr_protocol::%s\n", cs.name);
  }
}

#include "fs_fixture.inc"
static std::string to_hex(const fiat_shamir::Bytes& b) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (uint8_t x : b) { s.push_back(d[x >> 4]); s.push_back(d[x & 15]); }
  return s;
}
// fiat-shamir/src/lib.rs:216-236 (`it_works`) over the GPU-backed G, and SURVEY 8f rank 3's bytes: a compiled caller of the C ABI
// produces, message for message, the non-interactive proofs oracle/fs_ref.py committed to tests/golden/fs_transcripts.json
static void test_fiat_shamir() {
  using namespace fiat_shamir;
  {  // the expander against RFC 9380 appendix K.1 (expand_message_xmd, SHA-256), 64-byte Z_pad mode
    Field g(0xFFFFFFFF00000001ull);
    const std::string dst = "QUUX-V01-CS02-with-expander-SHA256-128";
    Sha256FieldHasher h(g, Bytes(dst.begin(), dst.end()), 64);
    REQUIRE(to_hex(h.expand(Bytes(), 32)) == "68a985b87eb6b46952128911f2a4412bbc302a9d759667f87f7a21d803f07235");
    const std::string abc = "abc";
    REQUIRE(to_hex(h.expand(Bytes(abc.begin(), abc.end()), 32)) == "d8ccab23b5985ccea865c6c97b6e5b8350e794e603b4b97902f53a8a0d605615");
  }
  for (const FsCase& c : kFsCases) {
    Field f(c.p);
    Context ctx(f);
    sc_table *ha = nullptr, *hb = nullptr;
    ctx.check(sc_table_generate(ctx.raw(), c.seed_a, 0, (size_t)1 << c.n, &ha), "sc_table_generate");
    ctx.check(sc_table_generate(ctx.raw(), c.seed_b, 0, (size_t)1 << c.n, &hb), "sc_table_generate");
    G g(std::make_shared<sumcheck_hip::DeviceMle>(ctx, ha), std::make_shared<sumcheck_hip::DeviceMle>(ctx, hb));
    Sha256FieldHasher hasher(f);
    Prover prover(g.clone());
    const FiatShamirTranscript t = generate_transcript(InteractiveProver(prover, f), hasher);
    REQUIRE(t.g.size() == (size_t)c.n);
    for (int j = 0; j < c.n; ++j) REQUIRE(to_hex(t.g[j]) == c.messages[j]);
    Verifier verifier(c.n, g.clone(), f);
    REQUIRE(verify_transcript(t, verifier, f, hasher));
    if (c.p > 5) {   // a tampered message is caught (claim mismatch, failed final check, or codec error); over F_5 the last round's
      FiatShamirTranscript bad = t;   // only check holds by chance one time in five
      bad.g[c.n - 1].back() ^= 1;
      bool ok = true;
      try {
        Verifier v2(c.n, g.clone(), f);
        ok = verify_transcript(bad, v2, f, hasher);
      } catch (const std::exception&) { ok = false; }   // (ProverClaimMismatch, a codec error, or the reference's assert_eq! at :303)
      REQUIRE(!ok);
    }
  }
  std::printf("ok fiat_shamir: %zu transcripts byte for byte\n", sizeof(kFsCases) / sizeof(kFsCases[0]));
}

int main() {
  test_fiat_shamir();
  Field f5(5);
  Context ctx(f5);
  StdRng rng(f5, 42);

  {  // multilinear-extensions example_from_book
    std::vector<F> evals = {f5.from_int(1), f5.from_int(2), f5.from_int(1), f5.from_int(4)};
    const int expected[5][5] = {{1, 2, 3, 4, 0}, {1, 4, 2, 0, 3}, {1, 1, 1, 1, 1}, {1, 3, 0, 2, 4}, {1, 0, 4, 3, 2}};
    for (uint32_t i = 0; i < 5; ++i)
      for (uint32_t j = 0; j < 5; ++j) {
        std::vector<F> r = {f5.from_int(i), f5.from_int(j)};
        REQUIRE((int)f5.to_int(multilinear_extensions::cti_multilinear_from_evaluations(ctx, evals, r)) == expected[i][j]);
        REQUIRE((int)f5.to_int(multilinear_extensions::vsbw_multilinear_from_evaluations(ctx, evals, r)) == expected[i][j]);
      }
    std::printf("ok multilinear_extensions::example_from_book\n");
  }

  {  // the round polynomial's canonical form (matrix-multiplication/src/lib.rs:17-60 adds three SparsePolynomials): every
     // (H(0), H(1), H(2)) of F_5^3 - the rule tests/golden/fs_transcripts.json lists term by term (tests/test_oracle_fs.py)
    int explicit_zero = 0;
    for (uint64_t e0 = 0; e0 < 5; ++e0) for (uint64_t e1 = 0; e1 < 5; ++e1) for (uint64_t e2 = 0; e2 < 5; ++e2) {
      F e[3] = {f5.from_int(e0), f5.from_int(e1), f5.from_int(e2)};
      SparsePolynomial q = matrix_multiplication::round_poly_lagrange(f5, e);
      for (uint64_t x = 0; x < 3; ++x) REQUIRE(q.evaluate(f5, f5.from_int(x)) == e[x]);
      bool expect_zero_term = e0 == 0 && (e1 == 0) != (e2 == 0);
      size_t zeros = 0;
      for (size_t t = 0; t < q.coeffs.size(); ++t) {
        if (t) REQUIRE(q.coeffs[t - 1].first < q.coeffs[t].first);
        if (q.coeffs[t].second == 0) { ++zeros; REQUIRE(q.coeffs[t].first == 0); }
      }
      REQUIRE(zeros == (expect_zero_term ? 1u : 0u));
      explicit_zero += (int)zeros;
      SparsePolynomial d = matrix_multiplication::round_poly_from_evals(f5, e);      // W / triangle: `p.into()`
      for (auto& t : d.coeffs) REQUIRE(t.second != 0);
      for (uint64_t x = 0; x < 3; ++x) REQUIRE(d.evaluate(f5, f5.from_int(x)) == e[x]);
    }
    REQUIRE(explicit_zero == 8);
    std::printf("ok SparsePolynomial canonical forms\n");
  }

  Matrix a = {{f5.from_int(0), f5.from_int(1)}, {f5.from_int(2), f5.from_int(0)}};
  Matrix b = {{f5.from_int(1), f5.from_int(0)}, {f5.from_int(0), f5.from_int(4)}};
  Matrix c = {{f5.from_int(0), f5.from_int(4)}, {f5.from_int(2), f5.from_int(0)}};
  REQUIRE(matmul(f5, a, b) == c);  // matrix_test_from_book
  for (uint32_t i = 0; i < 2; ++i)
    for (uint32_t j = 0; j < 2; ++j) {
      std::vector<F> point = u32_to_boolean_vec(f5, i, 1), pj = u32_to_boolean_vec(f5, j, 1);
      point.insert(point.end(), pj.begin(), pj.end());
      G g = G::create(ctx, 1, flatten(a), flatten(b), point);
      run_protocol(f5, g, c[i][j], true, rng);
    }
  std::printf("ok matrix_multiplication::example_from_book\n");

  std::mt19937_64 gen(7);
  for (uint32_t p = 2; p < 6; ++p) {  // randomized_test
    size_t n = (size_t)1 << p;
    Matrix A(n, std::vector<F>(n)), B(n, std::vector<F>(n));
    for (auto& row : A) for (auto& x : row) x = f5.from_int(gen() % 5);
    for (auto& row : B) for (auto& x : row) x = f5.from_int(gen() % 5);
    Matrix C = matmul(f5, A, B);
    size_t step = p >= 4 ? 5 : 1;  // every (i,j) for 4x4 and 8x8, a lattice of them above
    for (uint32_t i = 0; i < n; i += step)
      for (uint32_t j = 0; j < n; j += step) {
        std::vector<F> point = u32_to_boolean_vec(f5, i, p), pj = u32_to_boolean_vec(f5, j, p);
        point.insert(point.end(), pj.begin(), pj.end());
        G g = G::create(ctx, p, flatten(A), flatten(B), point);
        F resu = f5.zero();
        for (uint32_t z = 0; z < n; ++z) resu = f5.add(resu, g.evaluate(u32_to_boolean_vec(f5, z, p)).value());  // :346-350
        REQUIRE(resu == C[i][j]);                                                                                // :352 with :340
        REQUIRE(!g.evaluate(u32_to_boolean_vec(f5, 0, p + 1)).has_value());
        run_protocol(f5, g, C[i][j], true, rng);
      }
  }
  std::printf("ok matrix_multiplication::randomized_test\n");

  {  // Verifier error paths (:286-291, :308-309)
    std::vector<F> point = u32_to_boolean_vec(f5, 1, 2), pj = u32_to_boolean_vec(f5, 2, 2);
    point.insert(point.end(), pj.begin(), pj.end());
    Matrix A(4, std::vector<F>(4)), B(4, std::vector<F>(4));
    for (auto& row : A) for (auto& x : row) x = f5.from_int(gen() % 5);
    for (auto& row : B) for (auto& x : row) x = f5.from_int(gen() % 5);
    G g = G::create(ctx, 2, flatten(A), flatten(B), point);
    Prover prover(g.clone());
    Verifier bad(2, g.clone(), f5);
    bad.set_c_1(f5.add(prover.c_1(), f5.one()));
    bool threw = false;
    try { bad.round(prover.round(f5.one(), 0), rng); } catch (const ProverClaimMismatch&) { threw = true; }
    REQUIRE(threw);
    Prover p2(g.clone());
    Verifier blind(2, nullptr, f5);
    blind.set_c_1(p2.c_1());
    threw = false;
    try {
      F r = f5.one();
      for (size_t j = 0; j < 2; ++j) r = blind.round(p2.round(r, j), rng).r;
    } catch (const NoPolySet&) { threw = true; }
    REQUIRE(threw);
    std::printf("ok verifier error paths\n");
  }
  triangle_tests();
  gkr_tests();
  std::printf("ALL OK\n");
  return 0;
}
