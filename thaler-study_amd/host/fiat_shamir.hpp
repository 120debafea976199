// C++ host-side mirror of the reference crate `fiat-shamir` (src/lib.rs) over the mirrors of sum_check_protocol::{Prover, Verifier}
// (SURVEY.md section 8f rank 3).  Host-side only: O(n) bytes per proof; every prover.round underneath is a GPU pass of the engine.
//
//   InteractiveProver :33-66 | generate_transcript :75-98 | RandNums :102-119 | verify_transcript :123-143
//   InteractiveVerifier :146-171
//
// Wire format: ark-serialize `serialize_uncompressed` of `(F, SparsePolynomial<F>)` (round 1) and of `SparsePolynomial<F>` (later
// rounds): field element = canonical integer, little-endian, ceil(modulus_bits / 8) bytes; Vec = u64-LE length, then the items;
// usize = u64 LE.  The polynomial's TERM LIST goes on the wire as it is, which is why SparsePolynomial (sum_check_protocol.hpp)
// keeps arkworks' canonical forms.  Challenges: ark-ff's DefaultFieldHasher<Sha256, 128> (RFC 9380 expand_message_xmd with
// arkworks' Z_pad of len_per_elem bytes).  The bytes are compared with tests/golden/fs_transcripts.json (made by oracle/fs_ref.py)
// in tests/cpp/test_reference_tests.cpp; byte identity with arkworks itself is unpinned (no Rust toolchain in this image).
#pragma once
#include <array>
#include <cstring>

#include "sum_check_protocol.hpp"

namespace fiat_shamir {

using sum_check_protocol::F;
using sum_check_protocol::Field;
using sum_check_protocol::Prover;
using sum_check_protocol::RngF;
using sum_check_protocol::SparsePolynomial;
using sum_check_protocol::Verifier;
using sum_check_protocol::VerifierRoundResult;
typedef std::vector<uint8_t> Bytes;

struct SerializationError : sum_check_protocol::Error { SerializationError() : Error("Codec error") {} };   // Error::Serialization :13-16

// ---- SHA-256 (FIPS 180-4), enough of it for the expander ------------------------------------------------------------------------
class Sha256 {
 public:
  Sha256() { reset(); }
  void update(const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      buf_[fill_++] = p[i];
      if (fill_ == 64) { block(buf_); fill_ = 0; }
    }
    bits_ += 8 * (uint64_t)n;
  }
  void update(const Bytes& b) { update(b.data(), b.size()); }
  std::array<uint8_t, 32> finish() {
    const uint64_t bits = bits_;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill_ != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (56 - 8 * i));
    update(len, 8);
    std::array<uint8_t, 32> out;
    for (int i = 0; i < 8; ++i)
      for (int k = 0; k < 4; ++k) out[4 * i + k] = (uint8_t)(h_[i] >> (24 - 8 * k));
    reset();
    return out;
  }
 private:
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  void reset() {
    static const uint32_t init[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    memcpy(h_, init, sizeof(h_));
    fill_ = 0;
    bits_ = 0;
  }
  void block(const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
        0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
        0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
        0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
        0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
        0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
      const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], h = h_[7];
    for (int i = 0; i < 64; ++i) {
      const uint32_t t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
      const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
      h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += h;
  }
  uint32_t h_[8];
  uint8_t buf_[64];
  size_t fill_;
  uint64_t bits_;
};

// ---- ark-serialize ---------------------------------------------------------------------------------------------------------
inline size_t field_bytes(const Field& f) {
  int bits = 0;
  for (uint64_t p = f.c.p; p; p >>= 1) ++bits;
  return (size_t)(bits + 7) / 8;
}
inline void put_u64(Bytes& out, uint64_t v, size_t n = 8) { for (size_t i = 0; i < n; ++i) out.push_back((uint8_t)(v >> (8 * i))); }
inline void serialize_field(const Field& f, F m, Bytes& out) { put_u64(out, f.to_int(m), field_bytes(f)); }
inline void serialize_poly(const Field& f, const SparsePolynomial& p, Bytes& out) {
  put_u64(out, p.coeffs.size());
  for (auto& t : p.coeffs) {
    put_u64(out, (uint64_t)t.first);
    serialize_field(f, t.second, out);
  }
}
inline uint64_t get_u64(const Bytes& in, size_t& off, size_t n = 8) {
  if (off + n > in.size()) throw SerializationError();
  uint64_t v = 0;
  for (size_t i = 0; i < n; ++i) v |= (uint64_t)in[off + i] << (8 * i);
  off += n;
  return v;
}
inline F deserialize_field(const Field& f, const Bytes& in, size_t& off) {
  const uint64_t v = get_u64(in, off, field_bytes(f));
  if (v >= f.c.p) throw SerializationError();
  return f.from_int(v);
}
inline SparsePolynomial deserialize_poly(const Field& f, const Bytes& in, size_t& off) {   // the derived CanonicalDeserialize: the Vec as written
  SparsePolynomial p;
  const uint64_t n = get_u64(in, off);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t d = get_u64(in, off);
    p.coeffs.push_back({(size_t)d, deserialize_field(f, in, off)});
  }
  return p;
}

// ---- ark_ff::field_hashers::DefaultFieldHasher<Sha256, 128> ------------------------------------------------------------------
class Sha256FieldHasher {
 public:
  // z_pad < 0: arkworks (ExpanderXmd { block_size: len_per_base_elem }); 64: RFC 9380's s_in_bytes of SHA-256
  explicit Sha256FieldHasher(const Field& f, Bytes dst = {}, int z_pad = -1) : f_(f), dst_(std::move(dst)) {
    int bits = 0;
    for (uint64_t p = f.c.p; p; p >>= 1) ++bits;
    len_per_elem_ = (size_t)(bits + 128 + 7) / 8;
    z_pad_ = z_pad < 0 ? len_per_elem_ : (size_t)z_pad;
  }
  Bytes expand(const Bytes& msg, size_t n) const {
    const size_t ell = (n + 31) / 32;
    if (ell > 255 || n >= (1u << 16)) throw std::invalid_argument("expand_message_xmd: output too long");
    Bytes dst = dst_;
    if (dst.size() > 255) {
      Sha256 h;
      const char* tag = "H2C-OVERSIZE-DST-";
      h.update((const uint8_t*)tag, 17);
      h.update(dst);
      auto d = h.finish();
      dst.assign(d.begin(), d.end());
    }
    Bytes dst_prime = dst;
    dst_prime.push_back((uint8_t)dst.size());
    Sha256 h;
    h.update(Bytes(z_pad_, 0));
    h.update(msg);
    const uint8_t lib[3] = {(uint8_t)(n >> 8), (uint8_t)n, 0};
    h.update(lib, 3);
    h.update(dst_prime);
    const auto b0 = h.finish();
    h.update(b0.data(), 32);
    const uint8_t one = 1;
    h.update(&one, 1);
    h.update(dst_prime);
    auto bi = h.finish();
    Bytes out(bi.begin(), bi.end());
    for (size_t i = 2; i <= ell; ++i) {
      uint8_t x[32];
      for (int k = 0; k < 32; ++k) x[k] = b0[k] ^ bi[k];
      h.update(x, 32);
      const uint8_t idx = (uint8_t)i;
      h.update(&idx, 1);
      h.update(dst_prime);
      bi = h.finish();
      out.insert(out.end(), bi.begin(), bi.end());
    }
    out.resize(n);
    return out;
  }
  F hash_to_field_1(const Bytes& msg) const {   // hash_to_field::<1>(msg)[0]: from_be_bytes_mod_order of len_per_elem bytes
    const Bytes u = expand(msg, len_per_elem_);
    unsigned __int128 acc = 0;
    for (uint8_t b : u) acc = ((acc << 8) | b) % f_.c.p;
    return f_.from_int((uint64_t)acc);
  }
 private:
  const Field& f_;
  Bytes dst_;
  size_t len_per_elem_, z_pad_;
};

// :69-71
struct FiatShamirTranscript { std::vector<Bytes> g; };

// the impl for sum_check_protocol::Prover (:44-66)
class InteractiveProver {
 public:
  InteractiveProver(Prover& p, const Field& f) : p_(p), f_(f) {}
  Bytes g_1() {
    Bytes out;
    serialize_field(f_, p_.c_1(), out);
    serialize_poly(f_, p_.round(f_.one(), 0), out);
    return out;
  }
  Bytes round(size_t j, F r_j) {
    Bytes out;
    serialize_poly(f_, p_.round(r_j, j), out);
    return out;
  }
  size_t num_rounds() const { return p_.num_vars(); }
 private:
  Prover& p_;
  const Field& f_;
};

// :75-98
inline FiatShamirTranscript generate_transcript(InteractiveProver prover, const Sha256FieldHasher& hasher, std::vector<F>* challenges = nullptr) {
  FiatShamirTranscript t;
  Bytes hash_input = prover.g_1();
  t.g.push_back(hash_input);
  for (size_t j = 1; j < prover.num_rounds(); ++j) {
    const F r_j = hasher.hash_to_field_1(hash_input);
    if (challenges) challenges->push_back(r_j);
    Bytes g_j = prover.round(j, r_j);
    hash_input.insert(hash_input.end(), g_j.begin(), g_j.end());
    t.g.push_back(std::move(g_j));
  }
  return t;
}

// :102-119
class RandNums : public RngF {
 public:
  explicit RandNums(std::vector<F> nums) : nums_(std::move(nums)) {}
  F draw() override { return nums_.at(current_++); }
 private:
  std::vector<F> nums_;
  size_t current_ = 0;
};

// :123-171 (the impl of InteractiveVerifier for sum_check_protocol::Verifier folded in)
inline bool verify_transcript(const FiatShamirTranscript& t, Verifier& verifier, const Field& f, const Sha256FieldHasher& hasher) {
  Bytes hash_input;
  for (size_t j = 0; j < t.g.size(); ++j) {
    hash_input.insert(hash_input.end(), t.g[j].begin(), t.g[j].end());
    RandNums rng({hasher.hash_to_field_1(hash_input)});
    size_t off = 0;
    if (j == 0) verifier.set_c_1(deserialize_field(f, t.g[j], off));
    const SparsePolynomial g_j = deserialize_poly(f, t.g[j], off);
    const VerifierRoundResult res = verifier.round(g_j, rng);
    if (res.kind == VerifierRoundResult::FinalRound && !res.ok) return false;
  }
  return true;
}

}  // namespace fiat_shamir
