// 64-bit prime-field arithmetic in Montgomery form (R = 2^64), host + gfx950 device.
//
// Table elements are exactly the in-memory words of ark-ff's
// `Fp64<MontBackend<_, 1>>` (the field type every reference crate instantiates, e.g.
// /root/reference/sum-check-protocol/src/lib.rs:349-354): one u64 holding x*2^64 mod p,
// fully reduced to [0, p).  Two arithmetic policies share one interface:
//
//   GoldilocksMont - p = 2^64 - 2^32 + 1.  p^-1 = 2^32 + 1 (mod 2^64), so the Montgomery
//                    reduction is shifts/adds only, and sums of products are accumulated
//                    unreduced in a signed 128-bit word (2^96 = -1 mod p; one reduction per thread, not per product).
//   MontGeneric    - any odd p < 2^64 with runtime constants (the reference's field type, Fp64<MontBackend<T,1>> with any
//                    modulus: toy moduli 5 / 389 / 1572869 in its tests); sums of products are accumulated unreduced in
//                    160 bits and reduced once per thread (round 4; before, every product was reduced).
//
// Interface (F = policy object, passed by value to kernels):
//   F.add(a,b) F.sub(a,b) F.dbl(a) F.mul(a,b)         residues in, residue out
//   F::Acc acc; F.acc_zero(acc); F.acc_mac(acc,a,b); F.acc_get(acc) -> residue of sum(a_i*b_i)
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SC_HD __host__ __device__ __forceinline__
#else
#define SC_HD inline
#endif

namespace sc {

typedef uint64_t u64;
typedef uint32_t u32;

// A residue kept as two 32-bit halves.  The hand-scheduled Goldilocks sequences below work on
// halves; passing them as separate registers (instead of re-packing into an aligned 64-bit
// pair between a subtraction and the product that consumes it) saves ~80 v_mov per wave tile
// of the 27-cell grid.
struct X64 {
  u32 lo, hi;
};
SC_HD X64 split64(u64 v) { return X64{(u32)v, (u32)(v >> 32)}; }
SC_HD u64 join64(X64 v) { return ((u64)v.hi << 32) | v.lo; }

// 64 x 64 -> 128 product.  Device: four 32x32+64 multiply-adds (v_mad_u64_u32); asking the
// compiler for `a*b` and `__umul64hi(a,b)` separately costs seven quarter-rate multiplies.
SC_HD void mul_wide(u64 a, u64 b, u64& hi, u64& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 p00 = (u64)a0 * b0;
  const u64 mid = (u64)a0 * b1 + (p00 >> 32);     // < 2^64: (2^32-1)^2 + 2^32-1
  const u64 mid2 = (u64)a1 * b0 + (u32)mid;
  hi = (u64)a1 * b1 + (mid >> 32) + (mid2 >> 32);  // < 2^64: (2^32-1)^2 + 2(2^32-1)
  lo = (mid2 << 32) | (u32)p00;
#else
  unsigned __int128 t = (unsigned __int128)a * b;
  lo = (u64)t;
  hi = (u64)(t >> 64);
#endif
}

// splitmix64 output function on state x (x is "seed + index" in the synthetic instance,
// BASELINE.md section 3): z = x + gamma, then the two xor-shift-multiply rounds.
SC_HD u64 splitmix64(u64 x) {
  u64 z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Runtime field description crossing the C ABI (include/sumcheck_hip.h: sc_field).
struct FieldParams {
  u64 p;          // odd modulus, 2 < p < 2^64
  u64 p_inv_neg;  // -p^-1 mod 2^64
  u64 r_mod_p;    // 2^64 mod p   (Montgomery form of 1)
  u64 r2_mod_p;   // 2^128 mod p  (to_mont multiplier)
};

// ---------------------------------------------------------------------------------
struct MontGeneric {
  u64 p, p_inv_neg, r1, r2;

  MontGeneric() = default;
  SC_HD explicit MontGeneric(const FieldParams& f)
      : p(f.p), p_inv_neg(f.p_inv_neg), r1(f.r_mod_p), r2(f.r2_mod_p) {}

  SC_HD u64 modulus() const { return p; }
  SC_HD u64 one() const { return r1; }
  SC_HD u64 r_squared() const { return r2; }

  SC_HD u64 add(u64 a, u64 b) const {
    u64 s = a + b;
    bool carry = s < a;
    return (carry || s >= p) ? s - p : s;
  }
  SC_HD u64 sub(u64 a, u64 b) const {
    u64 d = a - b;
    return (a < b) ? d + p : d;
  }
  SC_HD u64 dbl(u64 a) const { return add(a, a); }
  // Four independent differences d[k] = a[k] - b[k] (mod p), the four borrow chains interleaved by hand as in
  // GoldilocksMont::sub4: per chain a 64-bit subtract (borrow in an SGPR pair), p masked by the borrow (two v_cndmask),
  // a 64-bit add - 6 VALU per difference, every carry consumer four slots behind its producer, no s_nop; the compiler's
  // own sequence for `a < b ? a - b + p : a - b` is 7 instructions + 2-4 wait states.  (p sits in two VGPRs: a VOP3
  // instruction of this ISA reads one scalar operand, and the mask is it.)
  SC_HD void sub4(X64 (&d)[4], const X64 (&a)[4], const X64 (&b)[4]) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 p0 = (u32)p, p1 = (u32)(p >> 32);
    u32 t0, t1, t2, t3, t4, t5, t6, t7;
    asm("v_sub_co_u32_e64 %0, vcc, %16, %24\n\t"
        "v_sub_co_u32_e64 %1, s[72:73], %17, %25\n\t"
        "v_sub_co_u32_e64 %2, s[74:75], %18, %26\n\t"
        "v_sub_co_u32_e64 %3, s[76:77], %19, %27\n\t"
        "v_subb_co_u32_e64 %4, vcc, %20, %28, vcc\n\t"
        "v_subb_co_u32_e64 %5, s[72:73], %21, %29, s[72:73]\n\t"
        "v_subb_co_u32_e64 %6, s[74:75], %22, %30, s[74:75]\n\t"
        "v_subb_co_u32_e64 %7, s[76:77], %23, %31, s[76:77]\n\t"
        "v_cndmask_b32_e64 %8, 0, %32, vcc\n\t"
        "v_cndmask_b32_e64 %9, 0, %32, s[72:73]\n\t"
        "v_cndmask_b32_e64 %10, 0, %32, s[74:75]\n\t"
        "v_cndmask_b32_e64 %11, 0, %32, s[76:77]\n\t"
        "v_cndmask_b32_e64 %12, 0, %33, vcc\n\t"
        "v_cndmask_b32_e64 %13, 0, %33, s[72:73]\n\t"
        "v_cndmask_b32_e64 %14, 0, %33, s[74:75]\n\t"
        "v_cndmask_b32_e64 %15, 0, %33, s[76:77]\n\t"
        "v_add_co_u32_e64 %0, vcc, %0, %8\n\t"
        "v_add_co_u32_e64 %1, s[72:73], %1, %9\n\t"
        "v_add_co_u32_e64 %2, s[74:75], %2, %10\n\t"
        "v_add_co_u32_e64 %3, s[76:77], %3, %11\n\t"
        "v_addc_co_u32_e64 %4, vcc, %4, %12, vcc\n\t"
        "v_addc_co_u32_e64 %5, s[72:73], %5, %13, s[72:73]\n\t"
        "v_addc_co_u32_e64 %6, s[74:75], %6, %14, s[74:75]\n\t"
        "v_addc_co_u32_e64 %7, s[76:77], %7, %15, s[76:77]"
        : "=&v"(d[0].lo), "=&v"(d[1].lo), "=&v"(d[2].lo), "=&v"(d[3].lo), "=&v"(d[0].hi), "=&v"(d[1].hi),
          "=&v"(d[2].hi), "=&v"(d[3].hi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
        : "v"(a[0].lo), "v"(a[1].lo), "v"(a[2].lo), "v"(a[3].lo), "v"(a[0].hi), "v"(a[1].hi), "v"(a[2].hi),
          "v"(a[3].hi), "v"(b[0].lo), "v"(b[1].lo), "v"(b[2].lo), "v"(b[3].lo), "v"(b[0].hi), "v"(b[1].hi),
          "v"(b[2].hi), "v"(b[3].hi), "v"(p0), "v"(p1)
        : "vcc", "s72", "s73", "s74", "s75", "s76", "s77");
#else
    for (int k = 0; k < 4; ++k) d[k] = split64(sub(join64(a[k]), join64(b[k])));
#endif
  }
  SC_HD X64 sub(X64 a, X64 b) const { return split64(sub(join64(a), join64(b))); }
  SC_HD void sub2(X64 (&d)[2], const X64 (&a)[2], const X64 (&b)[2]) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 p0 = (u32)p, p1 = (u32)(p >> 32);
    u32 t0, t1, t2, t3;
    asm("v_sub_co_u32_e64 %0, vcc, %8, %12\n\t"
        "v_sub_co_u32_e64 %1, s[72:73], %9, %13\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %2, vcc, %10, %14, vcc\n\t"
        "v_subb_co_u32_e64 %3, s[72:73], %11, %15, s[72:73]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %4, 0, %16, vcc\n\t"
        "v_cndmask_b32_e64 %5, 0, %16, s[72:73]\n\t"
        "v_cndmask_b32_e64 %6, 0, %17, vcc\n\t"
        "v_cndmask_b32_e64 %7, 0, %17, s[72:73]\n\t"
        "v_add_co_u32_e64 %0, vcc, %0, %4\n\t"
        "v_add_co_u32_e64 %1, s[72:73], %1, %5\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %2, vcc, %2, %6, vcc\n\t"
        "v_addc_co_u32_e64 %3, s[72:73], %3, %7, s[72:73]"
        : "=&v"(d[0].lo), "=&v"(d[1].lo), "=&v"(d[0].hi), "=&v"(d[1].hi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(a[0].lo), "v"(a[1].lo), "v"(a[0].hi), "v"(a[1].hi), "v"(b[0].lo), "v"(b[1].lo), "v"(b[0].hi), "v"(b[1].hi), "v"(p0), "v"(p1)
        : "vcc", "s72", "s73");
#else
    for (int k = 0; k < 2; ++k) d[k] = split64(sub(join64(a[k]), join64(b[k])));
#endif
  }

  // (hi:lo) < p * 2^64  ->  (hi:lo) * 2^-64 mod p
  SC_HD u64 redc(u64 hi, u64 lo) const {
    u64 m = lo * p_inv_neg;
    u64 mh, ml;
    mul_wide(m, p, mh, ml);
    (void)ml;                       // lo + ml == 0 (mod 2^64); carry out iff lo != 0
    u64 t = hi + mh;
    bool c1 = t < hi;
    u64 t2 = t + (lo != 0 ? 1u : 0u);
    bool c2 = t2 < t;
    return (c1 || c2 || t2 >= p) ? t2 - p : t2;
  }
  SC_HD u64 mul(u64 a, u64 b) const {
    u64 hi, lo;
    mul_wide(a, b, hi, lo);
    return redc(hi, lo);
  }
  SC_HD u64 to_mont(u64 canonical) const { return mul(canonical, r2); }
  SC_HD u64 from_mont(u64 m) const { return redc(0, m); }
  // canonical value of an arbitrary 64-bit word
  SC_HD u64 reduce_word(u64 z) const { return z % p; }

  // (w2*2^128 + w1*2^64 + w0) * 2^-64 mod p for ANY three words: w0 * R^-1 + (w1 mod p) + w2 * R, each term one
  // Montgomery product - mul(w1, R mod p) = w1 mod p and mul(w2, R^2 mod p) = w2 * R mod p are valid because one factor
  // of each is below p.  Once per accumulator, not per product.
  SC_HD u64 wide_get(u64 w0, u64 w1, u64 w2) const { return add(add(redc(0, w0), mul(w1, r1)), mul(w2, r2)); }

  // Unreduced sum of products (round 4; until then every product of the generic field was reduced on its own - a full
  // Montgomery REDC, ~35 instructions, per multiply-accumulate).  A product of two residues is below 2^128 whatever the
  // modulus, so a 160-bit unsigned accumulator - FIVE registers - takes 2^32 of them; kAccMaxTerms (the bound the host
  // checks caller-sized sums against) is the Goldilocks accumulator's 2^28.  Device: four v_mad_u64_u32 (t = x0 y0,
  // m = x0 y1 + x1 y0 with its carry sC, q = x1 y1) and eleven add-with-carry in three interleaved chains (A adds t and q,
  // B adds m, C adds sC at 2^96), every carry consumer three slots behind its producer as gfx950 wants (two wait states
  // between a VALU that writes VCC / an SGPR pair and the VALU that reads it; GoldilocksMont::acc_mac has the same shape),
  // two s_nop where only two chains are left: 17 issue slots, the same order as the 15 of the Goldilocks sequence.
  static constexpr u64 kAccMaxTerms = (u64)1 << 28;
  struct Acc {
    u32 l0, l1, l2, l3, l4;   // little-endian 32-bit limbs
  };
  SC_HD void acc_zero(Acc& a) const { a.l0 = a.l1 = a.l2 = a.l3 = a.l4 = 0; }
  SC_HD void acc_add(Acc& a, const Acc& b) const {
    const u64 s0 = (u64)a.l0 + b.l0;
    const u64 s1 = (u64)a.l1 + b.l1 + (s0 >> 32);
    const u64 s2 = (u64)a.l2 + b.l2 + (s1 >> 32);
    const u64 s3 = (u64)a.l3 + b.l3 + (s2 >> 32);
    a.l0 = (u32)s0;
    a.l1 = (u32)s1;
    a.l2 = (u32)s2;
    a.l3 = (u32)s3;
    a.l4 = a.l4 + b.l4 + (u32)(s3 >> 32);
  }
  SC_HD void acc_mac(Acc& a, u64 x, u64 y) const { acc_mac(a, split64(x), split64(y)); }
  SC_HD void acc_mac(Acc& a, X64 xs, X64 ys) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 x0 = xs.lo, x1 = xs.hi, y0 = ys.lo, y1 = ys.hi;
    u64 t, m, q, sC, sB, sD;
    asm("v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %7, 0"
        : "=&v"(t), "=&v"(m), "=&v"(q), "=&s"(sC)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1)
        : "vcc");
    const u32 t0 = (u32)t, t1 = (u32)(t >> 32), m0 = (u32)m, m1 = (u32)(m >> 32), q0 = (u32)q,
              q1 = (u32)(q >> 32);
    asm("v_add_co_u32_e32 %0, vcc, %0, %7\n\t"           // A1  l0 += t0
        "v_add_co_u32_e64 %1, %5, %1, %9\n\t"            // B1  l1 += m0
        "v_addc_co_u32_e64 %3, %6, %3, 0, %13\n\t"       // C1  l3 += sC
        "v_addc_co_u32_e32 %1, vcc, %1, %8, vcc\n\t"     // A2  l1 += t1 + c
        "v_addc_co_u32_e64 %2, %5, %2, %10, %5\n\t"      // B2  l2 += m1 + c
        "v_addc_co_u32_e64 %4, %6, %4, 0, %6\n\t"        // C2  l4 += c
        "v_addc_co_u32_e32 %2, vcc, %2, %11, vcc\n\t"    // A3  l2 += q0 + c
        "v_addc_co_u32_e64 %3, %5, %3, 0, %5\n\t"        // B3  l3 += c
        "s_nop 0\n\t"
        "v_addc_co_u32_e32 %3, vcc, %3, %12, vcc\n\t"    // A4  l3 += q1 + c
        "v_addc_co_u32_e64 %4, %5, %4, 0, %5\n\t"        // B4  l4 += c
        "s_nop 0\n\t"
        "v_addc_co_u32_e32 %4, vcc, 0, %4, vcc"            // A5  l4 += c
        : "+v"(a.l0), "+v"(a.l1), "+v"(a.l2), "+v"(a.l3), "+v"(a.l4), "=&s"(sB), "=&s"(sD)
        : "v"(t0), "v"(t1), "v"(m0), "v"(m1), "v"(q0), "v"(q1), "s"(sC)
        : "vcc");
#else
    u64 hi, lo;
    mul_wide(join64(xs), join64(ys), hi, lo);
    typedef unsigned __int128 u128;
    u128 s = ((u128)a.l3 << 96) | ((u128)a.l2 << 64) | ((u128)a.l1 << 32) | a.l0;
    const u128 s2 = s + (((u128)hi << 64) | lo);
    a.l4 += (s2 < s) ? 1u : 0u;
    a.l0 = (u32)s2;
    a.l1 = (u32)(s2 >> 32);
    a.l2 = (u32)(s2 >> 64);
    a.l3 = (u32)(s2 >> 96);
#endif
  }
  SC_HD u64 acc_get(const Acc& a) const {
    u32 l0 = a.l0, l1 = a.l1, l2 = a.l2, l3 = a.l3, l4 = a.l4;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3), "+v"(l4));   // see GoldilocksMont::acc_get: keeps the limbs unpaired in the loops
#endif
    return wide_get(((u64)l1 << 32) | l0, ((u64)l3 << 32) | l2, (u64)l4);
  }

  // Three-class accumulator for short sums (folds): the four 32x32 partial products are added into three 64-bit words by
  // v_mad_u64_u32 itself, each with a carry counter - 4 multiply-adds + 4 add-with-carry per product, nine registers
  // (GoldilocksMont::Acc3: the arithmetic does not depend on the modulus, only the final reduction does)
  struct Acc3 {
    u64 a00, a01, a11;
    u32 c00, c01, c11;
  };
  SC_HD void acc3_zero(Acc3& A) const { A.a00 = A.a01 = A.a11 = 0; A.c00 = A.c01 = A.c11 = 0; }
  SC_HD void acc3_mac(Acc3& A, u64 x, u64 y) const {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u64 s0, s1, s2, s3;
    asm volatile(
        "v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"
        "v_addc_co_u32_e64 %3, %6, %3, 0, %6\n\t"
        "v_mad_u64_u32 %1, %9, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %4, %7, %4, 0, %7\n\t"
        "v_addc_co_u32_e64 %5, %8, %5, 0, %8\n\t"
        "v_addc_co_u32_e64 %4, %9, %4, 0, %9\n\t"
        : "+v"(A.a00), "+v"(A.a01), "+v"(A.a11), "+v"(A.c00), "+v"(A.c01), "+v"(A.c11), "=&s"(s0), "=&s"(s1),
          "=&s"(s2), "=&s"(s3)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
#else
    u64 t;
    bool c;
    c = __builtin_add_overflow((u64)x0 * y0, A.a00, &t); A.a00 = t; A.c00 += c ? 1u : 0u;
    c = __builtin_add_overflow((u64)x0 * y1, A.a01, &t); A.a01 = t; A.c01 += c ? 1u : 0u;
    c = __builtin_add_overflow((u64)x1 * y1, A.a11, &t); A.a11 = t; A.c11 += c ? 1u : 0u;
    c = __builtin_add_overflow((u64)x1 * y0, A.a01, &t); A.a01 = t; A.c01 += c ? 1u : 0u;
#endif
  }
  SC_HD u64 acc3_get(const Acc3& A) const {
    // w0 + w1*2^64 + w2*2^128 = a00 + c00*2^64 + (a01 + c01*2^64)*2^32 + (a11 + c11*2^64)*2^64
    u64 w0, w1, t;
    bool k = __builtin_add_overflow(A.a00, A.a01 << 32, &w0);
    u32 carry = k ? 1u : 0u;
    t = (A.a01 >> 32) + ((u64)A.c01 << 32);          // < 2^64: c01 < 2^31 for any sum used here
    k = __builtin_add_overflow(t, (u64)A.c00 + carry, &t);
    carry = k ? 1u : 0u;
    k = __builtin_add_overflow(t, A.a11, &w1);
    carry += k ? 1u : 0u;
    return wide_get(w0, w1, (u64)A.c11 + carry);
  }
};

// ---------------------------------------------------------------------------------
struct GoldilocksMont {
  static constexpr u64 P = 0xFFFFFFFF00000001ull;
  static constexpr u64 EPS = 0x00000000FFFFFFFFull;  // 2^64 mod p
  // 2^64 mod p and 2^128 mod p
  static constexpr u64 R1 = 0x00000000FFFFFFFFull;
  static constexpr u64 R2 = 0xFFFFFFFE00000001ull;

  GoldilocksMont() = default;
  SC_HD explicit GoldilocksMont(const FieldParams&) {}

  SC_HD u64 modulus() const { return P; }
  SC_HD u64 one() const { return R1; }
  SC_HD u64 r_squared() const { return R2; }

  // a + b >= p  <=>  a + b + (2^64 - p) carries out of 64 bits, and 2^64 - p = EPS:
  // two add-with-carry pairs and a select, no 64-bit compare against p.
  SC_HD u64 add(u64 a, u64 b) const {
    u64 s, u;
    const bool c1 = __builtin_add_overflow(a, b, &s);
    const bool c2 = __builtin_add_overflow(s, EPS, &u);
    return (c1 | c2) ? u : s;
  }
  // a - b + p on borrow; +p == -EPS (mod 2^64)
  SC_HD u64 sub(u64 a, u64 b) const {
    u64 d;
    const bool bw = __builtin_sub_overflow(a, b, &d);
    return d - (bw ? EPS : (u64)0);
  }
  SC_HD u64 dbl(u64 a) const { return add(a, a); }
  // Four independent differences d[k] = a[k] - b[k] (mod p).  On the device the four borrow
  // chains are interleaved by hand: gfx950 needs two wait states between a VALU that writes a
  // carry (SGPR pair / VCC) and the VALU that consumes it, so one subtraction on its own is
  // 6 instructions + ~4 s_nop from the compiler; four together are 16 VALU + 4 SALU, no s_nop.
  // Per chain: d = a - b (borrow bw); on borrow subtract EPS = 2^32 - 1, i.e. d0 += bw (carry c)
  // and d1 -= (bw & ~c); that mask is formed on the scalar unit (a VALU-written carry can be read
  // by the next SALU instruction, and SALU issue does not take a VALU slot): 4 VALU per subtraction.
  // s_andn2 writes SCC: the clobber list says so (without it a loop counter compare kept in SCC
  // across the block is destroyed and the kernel never ends - found the hard way).
  SC_HD void sub4(X64 (&d)[4], const X64 (&a)[4], const X64 (&b)[4]) const {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_sub_co_u32_e64 %0, vcc, %8, %16\n\t"
        "v_sub_co_u32_e64 %1, s[72:73], %9, %17\n\t"
        "v_sub_co_u32_e64 %2, s[74:75], %10, %18\n\t"
        "v_sub_co_u32_e64 %3, s[76:77], %11, %19\n\t"
        "v_subb_co_u32_e64 %4, vcc, %12, %20, vcc\n\t"
        "v_subb_co_u32_e64 %5, s[72:73], %13, %21, s[72:73]\n\t"
        "v_subb_co_u32_e64 %6, s[74:75], %14, %22, s[74:75]\n\t"
        "v_subb_co_u32_e64 %7, s[76:77], %15, %23, s[76:77]\n\t"
        "v_addc_co_u32_e64 %0, s[78:79], %0, 0, vcc\n\t"
        "v_addc_co_u32_e64 %1, s[80:81], %1, 0, s[72:73]\n\t"
        "v_addc_co_u32_e64 %2, s[82:83], %2, 0, s[74:75]\n\t"
        "v_addc_co_u32_e64 %3, s[84:85], %3, 0, s[76:77]\n\t"
        "s_andn2_b64 vcc, vcc, s[78:79]\n\t"
        "s_andn2_b64 s[72:73], s[72:73], s[80:81]\n\t"
        "s_andn2_b64 s[74:75], s[74:75], s[82:83]\n\t"
        "s_andn2_b64 s[76:77], s[76:77], s[84:85]\n\t"
        "v_subb_co_u32_e64 %4, s[78:79], %4, 0, vcc\n\t"
        "v_subb_co_u32_e64 %5, s[80:81], %5, 0, s[72:73]\n\t"
        "v_subb_co_u32_e64 %6, s[82:83], %6, 0, s[74:75]\n\t"
        "v_subb_co_u32_e64 %7, s[84:85], %7, 0, s[76:77]"
        : "=&v"(d[0].lo), "=&v"(d[1].lo), "=&v"(d[2].lo), "=&v"(d[3].lo), "=&v"(d[0].hi), "=&v"(d[1].hi),
          "=&v"(d[2].hi), "=&v"(d[3].hi)
        : "v"(a[0].lo), "v"(a[1].lo), "v"(a[2].lo), "v"(a[3].lo), "v"(a[0].hi), "v"(a[1].hi), "v"(a[2].hi),
          "v"(a[3].hi), "v"(b[0].lo), "v"(b[1].lo), "v"(b[2].lo), "v"(b[3].lo), "v"(b[0].hi), "v"(b[1].hi),
          "v"(b[2].hi), "v"(b[3].hi)
        : "vcc", "scc", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85");
#else
    for (int k = 0; k < 4; ++k) d[k] = split64(sub(join64(a[k]), join64(b[k])));
#endif
  }
  SC_HD X64 sub(X64 a, X64 b) const { return split64(sub(join64(a), join64(b))); }
  // two independent differences, same scheme (the two-way interleave leaves one idle slot after
  // each carry producer; the scalar instructions fill two of them)
  SC_HD void sub2(X64 (&d)[2], const X64 (&a)[2], const X64 (&b)[2]) const {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_sub_co_u32_e64 %0, vcc, %4, %8\n\t"
        "v_sub_co_u32_e64 %1, s[72:73], %5, %9\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %2, vcc, %6, %10, vcc\n\t"
        "v_subb_co_u32_e64 %3, s[72:73], %7, %11, s[72:73]\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %0, s[74:75], %0, 0, vcc\n\t"
        "v_addc_co_u32_e64 %1, s[76:77], %1, 0, s[72:73]\n\t"
        "s_andn2_b64 vcc, vcc, s[74:75]\n\t"
        "s_andn2_b64 s[72:73], s[72:73], s[76:77]\n\t"
        "s_nop 0\n\t"
        "v_subb_co_u32_e64 %2, s[74:75], %2, 0, vcc\n\t"
        "v_subb_co_u32_e64 %3, s[76:77], %3, 0, s[72:73]"
        : "=&v"(d[0].lo), "=&v"(d[1].lo), "=&v"(d[0].hi), "=&v"(d[1].hi)
        : "v"(a[0].lo), "v"(a[1].lo), "v"(a[0].hi), "v"(a[1].hi), "v"(b[0].lo), "v"(b[1].lo), "v"(b[0].hi), "v"(b[1].hi)
        : "vcc", "scc", "s72", "s73", "s74", "s75", "s76", "s77");
#else
    for (int k = 0; k < 2; ++k) d[k] = split64(sub(join64(a[k]), join64(b[k])));
#endif
  }

  // floor(m * p / 2^64) for the m with m*p == lo (mod 2^64), i.e. m = lo * (2^32+1).
  // m*p = (m - (m>>32)) * 2^64 + (m - (m<<32)); the low word borrows iff m < (m<<32).
  SC_HD static u64 mp_high(u64 lo) {
    u64 m = lo + (lo << 32);
    u64 borrow = (m < (m << 32)) ? 1u : 0u;
    return m - (m >> 32) - borrow;
  }
  // (hi:lo) with hi < p  ->  (hi:lo) * 2^-64 mod p
  SC_HD u64 redc(u64 hi, u64 lo) const { return sub(hi, mp_high(lo)); }
  SC_HD u64 mul(u64 a, u64 b) const {
    u64 hi, lo;
    mul_wide(a, b, hi, lo);
    return redc(hi, lo);
  }
  SC_HD u64 to_mont(u64 canonical) const { return mul(canonical, R2); }
  SC_HD u64 from_mont(u64 m) const { return redc(0, m); }
  SC_HD u64 reduce_word(u64 z) const { return z >= P ? z - P : z; }

  // (w2*2^128 + w1*2^64 + w0) * 2^-64 mod p  =  w1 - floor(m p / 2^64) + w2 * 2^64  (mod p)
  SC_HD u64 wide_get(u64 w0, u64 w1, u32 w2) const {
    u64 x = reduce_word(w1);
    u64 y = mp_high(w0);                     // < p
    u64 z = ((u64)w2 << 32) - (u64)w2;       // w2 * (2^32-1) < p
    return add(sub(x, y), z);
  }

  // Unreduced sum of products in FOUR registers.  2^96 == -1 (mod p), so a 128-bit product
  // x*y = L + H*2^96 (L = its low 96 bits, H < 2^32) is congruent to L - H: a 128-bit
  // two's-complement accumulator needs no carry word - one register and about a third of the
  // add-with-carry work less than a 160-bit sum.  Capacity: the device sequence adds, per product,
  // t + m*2^32 + q0*2^64 - (q1 + sC) with t, q0*2^64 < 2^96 and m*2^32 < 2^97, i.e. less than 3*2^96;
  // the sum stays inside (-2^127, 2^127) for kAccMaxTerms = 2^28 terms with a wide margin (the exact
  // bound is about 2^29.4).  Beyond that it would wrap mod 2^128, which is not 0 mod p.  Every
  // kernel's per-thread accumulation count is far below (tiles per wave, rows per chunk, 2^k <= 2^17
  // entries per dot product); the host checks the caller-sized ones against kAccMaxTerms.
  static constexpr u64 kAccMaxTerms = (u64)1 << 28;
  struct Acc {
    u32 l0, l1, l2, l3;  // little-endian 32-bit limbs of the two's-complement sum
  };
  SC_HD void acc_zero(Acc& a) const { a.l0 = a.l1 = a.l2 = a.l3 = 0; }
  // sum of two accumulators: a plain 128-bit two's-complement add (the term bound covers the sum of their terms)
  SC_HD void acc_add(Acc& a, const Acc& b) const {
    const u64 s0 = (u64)a.l0 + b.l0;
    const u64 s1 = (u64)a.l1 + b.l1 + (s0 >> 32);
    const u64 s2 = (u64)a.l2 + b.l2 + (s1 >> 32);
    a.l0 = (u32)s0;
    a.l1 = (u32)s1;
    a.l2 = (u32)s2;
    a.l3 = a.l3 + b.l3 + (u32)(s2 >> 32);
  }
  SC_HD void acc_mac(Acc& a, u64 x, u64 y) const { acc_mac(a, split64(x), split64(y)); }
  SC_HD void acc_mac(Acc& a, X64 xs, X64 ys) const {
#if defined(__HIP_DEVICE_COMPILE__)
    // 4 multiplies + 11 add/subtract-with-carry, scheduled by hand: with x = (x1:x0), y = (y1:y0),
    //   t = x0*y0 (weight 1), m = x0*y1 + x1*y0 (weight 2^32, carry sC at 2^96), q = x1*y1 (2^64);
    //   chain A adds t and q0, chain B adds m, chain C subtracts H = q1 + sC at weight 1.
    // gfx950 wants two wait states between a VALU writing an SGPR/VCC and a VALU reading it; the
    // three carry chains (each limb update is an independent wrap-around add or subtract, so
    // they commute) are interleaved so that every consumer sits three slots after its producer:
    // 15 instructions, no s_nop.  The compiler's own sequence for the same arithmetic is ~17
    // instructions + ~6 s_nop.
    const u32 x0 = xs.lo, x1 = xs.hi, y0 = ys.lo, y1 = ys.hi;
    u64 t, m, q, sC, sB, sD;
    asm("v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1\n\t"
        "v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
        "v_mad_u64_u32 %2, vcc, %5, %7, 0"
        : "=&v"(t), "=&v"(m), "=&v"(q), "=&s"(sC)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1)
        : "vcc");
    const u32 t0 = (u32)t, t1 = (u32)(t >> 32), m0 = (u32)m, m1 = (u32)(m >> 32), q0 = (u32)q,
              q1 = (u32)(q >> 32);
    asm("v_subb_co_u32_e64 %0, %5, %0, %11, %12\n\t"    // C1  l0 -= q1 + sC
        "v_add_co_u32_e32 %0, vcc, %0, %6\n\t"          // A1  l0 += t0
        "v_add_co_u32_e64 %1, %4, %1, %8\n\t"           // B1  l1 += m0
        "v_subb_co_u32_e64 %1, %5, %1, 0, %5\n\t"       // C2  l1 -= b
        "v_addc_co_u32_e32 %1, vcc, %1, %7, vcc\n\t"    // A2  l1 += t1 + c
        "v_addc_co_u32_e64 %2, %4, %2, %9, %4\n\t"      // B2  l2 += m1 + c
        "v_subb_co_u32_e64 %2, %5, %2, 0, %5\n\t"       // C3  l2 -= b
        "v_addc_co_u32_e32 %2, vcc, %2, %10, vcc\n\t"   // A3  l2 += q0 + c
        "v_addc_co_u32_e64 %3, %4, %3, 0, %4\n\t"       // B3  l3 += c
        "v_subb_co_u32_e64 %3, %5, %3, 0, %5\n\t"       // C4  l3 -= b
        "v_addc_co_u32_e32 %3, vcc, 0, %3, vcc"          // A4  l3 += c
        : "+v"(a.l0), "+v"(a.l1), "+v"(a.l2), "+v"(a.l3), "=&s"(sB), "=&s"(sD)
        : "v"(t0), "v"(t1), "v"(m0), "v"(m1), "v"(q0), "v"(q1), "s"(sC)
        : "vcc");
#else
    u64 hi, lo;
    mul_wide(join64(xs), join64(ys), hi, lo);
    typedef unsigned __int128 u128;
    u128 s = ((u128)a.l3 << 96) | ((u128)a.l2 << 64) | ((u128)a.l1 << 32) | a.l0;
    s += ((u128)(hi & 0xFFFFFFFFull) << 64) | lo;  // + L
    s -= (u128)(hi >> 32);                         // - H
    a.l0 = (u32)s;
    a.l1 = (u32)(s >> 32);
    a.l2 = (u32)(s >> 64);
    a.l3 = (u32)(s >> 96);
#endif
  }
  // residue of (accumulated value) * 2^-64
  SC_HD u64 acc_get(const Acc& a) const {
    u32 l0 = a.l0, l1 = a.l1, l2 = a.l2, l3 = a.l3;
#if defined(__HIP_DEVICE_COMPILE__)
    // Copy the limbs out through an opaque statement: joining l0/l1 into a 64-bit value below
    // would otherwise make the register allocator keep them in an aligned pair for the whole
    // accumulation loop and re-pack them after every acc_mac (three v_mov per product).
    asm("" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3));
#endif
    // value = L' + T*2^96 == L' - T, L' = low 96 bits, T = top limb as a signed integer
    const int64_t T = (int64_t)(int32_t)l3;
    const u64 t = (T > 0) ? P - (u64)T : (u64)(-T);  // -T mod p, in [0, p)
    u64 w0;
    const bool c = __builtin_add_overflow(((u64)l1 << 32) | l0, t, &w0);
    const u64 w1 = (u64)l2 + (c ? 1u : 0u);
    return wide_get(w0, w1, 0);
  }

  // Three-class accumulator for short sums of products (folds): the four 32x32 partial
  // products of x*y are added into three 64-bit words by v_mad_u64_u32 itself
  // (value = A00 + A01*2^32 + A11*2^64), each with a carry counter:
  // 4 multiply-adds + 4 add-with-carry per product, no operand shuffling and no carry
  // chains.  On the host (tests) the same arithmetic in plain C.
  struct Acc3 {
    u64 a00, a01, a11;
    u32 c00, c01, c11;
  };
  SC_HD void acc3_zero(Acc3& A) const { A.a00 = A.a01 = A.a11 = 0; A.c00 = A.c01 = A.c11 = 0; }
  SC_HD void acc3_mac(Acc3& A, u64 x, u64 y) const {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
    u64 s0, s1, s2, s3;  // SGPR pairs receiving the carries; every reader is >= 2 VALU instructions
                         // behind its writer (the VALU-writes-SGPR -> VALU-reads-as-carry hazard)
    asm volatile(
        "v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"
        "v_addc_co_u32_e64 %3, %6, %3, 0, %6\n\t"
        "v_mad_u64_u32 %1, %9, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %4, %7, %4, 0, %7\n\t"
        "v_addc_co_u32_e64 %5, %8, %5, 0, %8\n\t"
        "v_addc_co_u32_e64 %4, %9, %4, 0, %9\n\t"
        : "+v"(A.a00), "+v"(A.a01), "+v"(A.a11), "+v"(A.c00), "+v"(A.c01), "+v"(A.c11), "=&s"(s0), "=&s"(s1),
          "=&s"(s2), "=&s"(s3)
        : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
#else
    u64 t;
    bool c;
    c = __builtin_add_overflow((u64)x0 * y0, A.a00, &t); A.a00 = t; A.c00 += c ? 1u : 0u;
    c = __builtin_add_overflow((u64)x0 * y1, A.a01, &t); A.a01 = t; A.c01 += c ? 1u : 0u;
    c = __builtin_add_overflow((u64)x1 * y1, A.a11, &t); A.a11 = t; A.c11 += c ? 1u : 0u;
    c = __builtin_add_overflow((u64)x1 * y0, A.a01, &t); A.a01 = t; A.c01 += c ? 1u : 0u;
#endif
  }
  // residue of the accumulated sum of products (same value acc_get returns for the same products)
  SC_HD u64 acc3_get(const Acc3& A) const {
    // w0 + w1*2^64 + w2*2^128 = a00 + c00*2^64 + (a01 + c01*2^64)*2^32 + (a11 + c11*2^64)*2^64
    u64 w0, w1;
    u64 t;
    bool k = __builtin_add_overflow(A.a00, A.a01 << 32, &w0);
    u32 carry = k ? 1u : 0u;
    t = (A.a01 >> 32) + ((u64)A.c01 << 32);          // < 2^64: c01 < 2^31 for any sum used here
    k = __builtin_add_overflow(t, (u64)A.c00 + carry, &t);
    carry = k ? 1u : 0u;
    k = __builtin_add_overflow(t, A.a11, &w1);
    carry += k ? 1u : 0u;
    return wide_get(w0, w1, A.c11 + carry);
  }
};

// Host-side helper: derive the Montgomery constants of an odd modulus.
inline bool field_params_from_modulus(u64 p, FieldParams* out) {
  if (p < 3 || (p & 1) == 0) return false;
  u64 inv = 1;  // Newton iteration for p^-1 mod 2^64
  for (int i = 0; i < 7; ++i) inv *= 2 - p * inv;
  out->p = p;
  out->p_inv_neg = (u64)0 - inv;
#if !defined(__HIP_DEVICE_COMPILE__)
  unsigned __int128 r = ((unsigned __int128)1 << 64) % p;
  out->r_mod_p = (u64)r;
  out->r2_mod_p = (u64)((r * r) % p);
#endif
  return true;
}

}  // namespace sc
