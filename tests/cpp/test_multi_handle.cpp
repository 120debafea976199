// A compiled caller of the multi-device handle, over the C ABI alone (include/sumcheck_hip.h; no torch, no Python): what the
// reference's single-process prove / verify loop (matrix-multiplication/src/lib.rs:354-370, sum-check-protocol/src/lib.rs:
// 278-330) looks like on ONE handle over several GPUs.  The verifier below is the reference's: it draws r_j BEFORE it checks
// (:283), checks g_1(0) + g_1(1) == c_1 (:286-297), g_j(0) + g_j(1) == g_{j-1}(r_{j-1}) (:313-328) and, last,
// g_n(r_n) == g(r) through the oracle evaluation of the polynomial (:298-310).  The transcript on a handle of N entries must
// equal the one-device transcript bit for bit.   usage: test_multi_handle [n = 18] [devices = 4]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/sumcheck_hip.h"

#define SC(x)                                                                            \
  do {                                                                                   \
    int rc_ = (x);                                                                       \
    if (rc_ != SC_OK) {                                                                  \
      fprintf(stderr, "%s -> %d: %s\n", #x, rc_, sc_last_error(ctx));                    \
      return 1;                                                                          \
    }                                                                                    \
  } while (0)

typedef unsigned __int128 u128;
static sc_field fld;
// host field arithmetic on Montgomery words, through the library's own conversion helpers
static uint64_t f_add(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % fld.p); }
static uint64_t f_sub(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + fld.p - b) % fld.p); }
static uint64_t f_mul(uint64_t a, uint64_t b) {   // a b R^-1: to canonical, multiply, back
  const uint64_t x = sc_field_from_mont(&fld, a), y = sc_field_from_mont(&fld, b);
  return sc_field_to_mont(&fld, (uint64_t)(((u128)x * y) % fld.p));
}
static uint64_t poly2(const uint64_t c[3], uint64_t x) { return f_add(c[0], f_mul(x, f_add(c[1], f_mul(x, c[2])))); }

struct Verifier {   // sum-check-protocol/src/lib.rs:227-331 for a quadratic round polynomial
  uint64_t claim = 0, state = 0x9E3779B97F4A7C15ull;
  std::vector<uint64_t> r;
  bool ok = true;
  uint64_t draw() {   // the verifier's own randomness (xorshift), reduced and in Montgomery form
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return sc_field_to_mont(&fld, state % fld.p);
  }
};
static uint64_t verifier_round(void* user, size_t j, const uint64_t e[3]) {
  Verifier* v = (Verifier*)user;
  const uint64_t r_j = v->draw();                      // :283 - drawn before any check
  uint64_t c[3];
  sc_interpolate_quadratic(&fld, e, c);                // matrix-multiplication/src/lib.rs:124-130
  const uint64_t one = fld.r_mod_p;
  if (f_add(poly2(c, 0), poly2(c, one)) != v->claim) v->ok = false;   // :286-297 / :313-328
  if (e[0] != poly2(c, 0) || e[1] != poly2(c, one)) v->ok = false;
  v->claim = poly2(c, r_j);
  v->r.push_back(r_j);
  (void)j;
  return r_j;
}

static int run(sc_ctx* ctx, size_t n, std::vector<uint64_t>* transcript, const char* what) {
  sc_table *a = nullptr, *b = nullptr;
  SC(sc_table_generate(ctx, 0xA5A5000000000001ull, 0, (size_t)1 << n, &a));
  SC(sc_table_generate(ctx, 0xB6B6000000000002ull, 0, (size_t)1 << n, &b));
  // Prover::new(g.clone()), c_1, Verifier::new(n, Some(g)), set_c_1, then n x (prover.round, verifier.round)
  sc_prover* pr = nullptr;
  SC(sc_prover_create(ctx, a, b, &pr));
  Verifier v;
  SC(sc_prover_c1(pr, &v.claim));
  transcript->assign(1, v.claim);
  uint64_t r_j = fld.r_mod_p;   // callers pass F::one() for round 0
  for (size_t j = 0; j < n; ++j) {
    uint64_t e[3];
    SC(sc_prover_round(pr, r_j, j, e));
    transcript->insert(transcript->end(), e, e + 3);
    r_j = verifier_round(&v, j, e);
  }
  SC(sc_prover_destroy(pr));
  uint64_t final_eval = 0;
  SC(sc_prod2_evaluate(ctx, a, b, v.r.data(), n, &final_eval));        // :298-310: g_n(r_n) == g(r)
  if (final_eval != v.claim) v.ok = false;
  // and the whole loop inside the library, challenges through the callback
  Verifier v2;
  std::vector<uint64_t> ev(3 * n);
  uint64_t c1 = 0;
  SC(sc_prove(ctx, a, b, verifier_round, &v2, 0, &c1, ev.data(), nullptr));
  // (v2's first claim is set by hand: sc_prove returns c_1 only at the end of the call)
  if (c1 != (*transcript)[0] || memcmp(ev.data(), transcript->data() + 1, 3 * n * sizeof(uint64_t)) != 0) v.ok = false;
  sc_table_free(ctx, a);
  sc_table_free(ctx, b);
  printf("%s: n = %zu, verifier %s\n", what, n, v.ok ? "accepts" : "REJECTS");
  return v.ok ? 0 : 1;
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atoi(argv[1]) : 18;
  const int n_dev = argc > 2 ? atoi(argv[2]) : 4;
  sc_field_from_modulus(0xFFFFFFFF00000001ull, &fld);
  sc_ctx* ctx = nullptr;
  if (sc_ctx_create(&fld, 0, &ctx) != SC_OK) {
    fprintf(stderr, "sc_ctx_create: %s\n", sc_last_error(nullptr));
    return 1;
  }
  std::vector<uint64_t> one, many;
  if (run(ctx, n, &one, "one device")) return 1;
  sc_ctx_destroy(ctx);
  std::vector<int> devices(n_dev, 0);   // (a one-GPU box: every entry names device 0)
  if (sc_ctx_create_multi(&fld, devices.data(), n_dev, &ctx) != SC_OK) {
    fprintf(stderr, "sc_ctx_create_multi: %s\n", sc_last_error(nullptr));
    return 1;
  }
  if (run(ctx, n, &many, "one handle over several devices")) return 1;
  sc_ctx_destroy(ctx);
  if (one != many) {
    fprintf(stderr, "the transcripts differ\n");
    return 1;
  }
  printf("ALL OK: %d-device handle == one device, %zu words\n", n_dev, one.size());
  return 0;
}
