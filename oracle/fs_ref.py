"""fs_ref.py - oracle-side restatement of the reference's non-interactive transform.

TEST INFRASTRUCTURE ONLY.  Nothing under thaler-study_amd/ imports
this file and this file imports nothing from thaler-study_amd/: it is written from the reference's
source and from the published layout of the arkworks crates the reference calls, on canonical
integers, so that a transcript produced over the GPU prover can be compared BYTE FOR BYTE with one
produced here.

What it follows
  fiat-shamir/src/lib.rs:44-66    InteractiveProver for sum_check_protocol::Prover
                                  (g_1 = serialize_uncompressed of (c_1, round(1, 0)); later rounds = the polynomial alone)
  fiat-shamir/src/lib.rs:75-98    generate_transcript (r_j = H(g_1 || ... || g_j))
  fiat-shamir/src/lib.rs:123-143  verify_transcript
  fiat-shamir/src/lib.rs:151-171  InteractiveVerifier for sum_check_protocol::Verifier
  matrix-multiplication/src/lib.rs:17-60, :124-130   the round polynomial of G: three Lagrange terms, each a
                                  SparsePolynomial::from_coefficients_vec, added with SparsePolynomial's `+`
  triangle-counting/src/lib.rs:120-132, gkr-protocol/src/round_polynomial.rs:78-90
                                  the round polynomial of triangle G and of W: Evaluations::interpolate over the
                                  size-4 radix-2 domain, then `DensePolynomial -> SparsePolynomial` (`.into()`)

Third-party behaviour restated (crates are NOT under /root/reference; workspace requirement "0.6",
Cargo.toml:19-30, exact patch unpinned; the text below is the behaviour of the published 0.4 / 0.5 sources,
which is what the reference's `hash_to_field::<1>` call shape matches):
  ark-poly  univariate::SparsePolynomial::from_coefficients_vec - pops TRAILING zero terms of the vector as given,
            then sorts by degree; interior zero terms stay.
            `&a + &b` - if a.is_zero() return b.clone(); if b.is_zero() return a.clone(); else a merge of the two
            sorted term lists in which a term present in both is dropped when the sum is zero and a term present in
            one is copied AS IS (also when its coefficient is zero).
            From<DensePolynomial> - keeps the non-zero coefficients only.
  ark-serialize  Fp: canonical integer, little-endian, ceil(MODULUS_BIT_SIZE / 8) bytes; usize: u64 LE;
            Vec<T>: u64 LE length, then the items; tuples and derived structs: the fields in order.
  ark-ff    field_hashers::DefaultFieldHasher<Sha256, 128>: RFC 9380 expand_message_xmd with
            len_per_elem = ceil((MODULUS_BIT_SIZE + 128) / 8), and - arkworks' own quirk - Z_pad of
            block_size = len_per_elem bytes (RFC 9380 says the hash's input block, 64 for SHA-256);
            the element is from_be_bytes_mod_order of its len_per_elem bytes.

PARITY: the expander is pinned against RFC 9380 appendix K.1 in its 64-byte Z_pad mode
(tests/golden/rfc9380_k1_xmd_sha256.json).  Byte identity with arkworks itself stays UNPINNED: the
reference's only assertion on this path is accept / reject (fiat-shamir/src/lib.rs:231-234), it holds no byte
vector, and no Rust toolchain exists in this image to produce one.
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref as R  # noqa: E402


# ---- ark_poly::univariate::SparsePolynomial on canonical integers ------------------------------------------
def sparse_from_coefficients_vec(terms):
    """terms: [(degree, coeff)] as handed to from_coefficients_vec"""
    terms = list(terms)
    while terms and terms[-1][1] == 0:
        terms.pop()
    terms.sort(key=lambda t: t[0])                 # a stable sort, like slice::sort_by
    assert not terms or terms[-1][1] != 0
    return terms


def sparse_is_zero(a):
    return all(c == 0 for _, c in a)


def sparse_add(a, b, p):
    if sparse_is_zero(a):
        return list(b)
    if sparse_is_zero(b):
        return list(a)
    out, i, k = [], 0, 0
    while i < len(a) or k < len(b):
        if i == len(a):
            out += b[k:]
            break
        if k == len(b):
            out += a[i:]
            break
        (da, ca), (db, cb) = a[i], b[k]
        if da < db:
            out.append((da, ca))
            i += 1
        elif da > db:
            out.append((db, cb))
            k += 1
        else:
            s = (ca + cb) % p
            if s:
                out.append((da, s))
            i, k = i + 1, k + 1
    return out


def sparse_eval(a, x, p):
    return sum(c * pow(x, d, p) for d, c in a) % p


def sparse_from_dense(dense):
    """From<DensePolynomial<F>> for SparsePolynomial<F>"""
    return sparse_from_coefficients_vec([(d, c) for d, c in enumerate(dense) if c != 0])


def lagrange_quadratic(points, p):
    """matrix-multiplication/src/lib.rs:17-60: one from_coefficients_vec per point, then poly_1 + poly_2 + poly_3"""
    polys = []
    for i in range(3):
        xi, yi = points[i]
        xj, xk = [points[m][0] for m in range(3) if m != i]
        den_inv = pow((xi - xj) * (xi - xk) % p, p - 2, p)
        raw = [(0, xj * xk % p), (1, (-xj - xk) % p), (2, 1 % p)]
        polys.append(sparse_from_coefficients_vec([(d, c * yi % p * den_inv % p) for d, c in raw]))
    return sparse_add(sparse_add(polys[0], polys[1], p), polys[2], p)


# ---- ark-serialize ------------------------------------------------------------------------------------------
def field_len(p):
    return (p.bit_length() + 7) // 8


def ser_field(x, p):
    assert 0 <= x < p
    return x.to_bytes(field_len(p), "little")


def ser_sparse(a, p):
    out = len(a).to_bytes(8, "little")
    for d, c in a:
        out += d.to_bytes(8, "little") + ser_field(c, p)
    return out


def de_field(data, off, p):
    n = field_len(p)
    if off + n > len(data):
        raise ValueError("codec")
    v = int.from_bytes(data[off:off + n], "little")
    if v >= p:
        raise ValueError("codec")
    return v, off + n


def de_sparse(data, off, p):
    if off + 8 > len(data):
        raise ValueError("codec")
    cnt, off = int.from_bytes(data[off:off + 8], "little"), off + 8
    terms = []
    for _ in range(cnt):
        if off + 8 > len(data):
            raise ValueError("codec")
        d, off = int.from_bytes(data[off:off + 8], "little"), off + 8
        c, off = de_field(data, off, p)
        terms.append((d, c))
    return terms, off


# ---- ark_ff::field_hashers::DefaultFieldHasher<Sha256, 128> ---------------------------------------------------
def expand_message_xmd_sha256(msg, dst, n, z_pad_len):
    ell = -(-n // 32)
    if ell > 255 or n >= 1 << 16:
        raise ValueError("requested output too long")
    if len(dst) > 255:
        dst = hashlib.sha256(b"H2C-OVERSIZE-DST-" + dst).digest()
    dst_prime = dst + len(dst).to_bytes(1, "big")
    b_0 = hashlib.sha256(b"\x00" * z_pad_len + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
    blocks = [hashlib.sha256(b_0 + b"\x01" + dst_prime).digest()]
    for i in range(2, ell + 1):
        mixed = bytes(u ^ v for u, v in zip(b_0, blocks[-1]))
        blocks.append(hashlib.sha256(mixed + i.to_bytes(1, "big") + dst_prime).digest())
    return b"".join(blocks)[:n]


def hash_to_field_1(msg, p, dst=b"", z_pad="arkworks"):
    """hasher.hash_to_field::<1>(msg)[0] for a prime field (extension degree 1)"""
    per = (p.bit_length() + 128 + 7) // 8
    z = per if z_pad == "arkworks" else 64
    return int.from_bytes(expand_message_xmd_sha256(bytes(msg), bytes(dst), per, z), "big") % p


# ---- provers of the three table-backed polynomials, as sum_check_protocol::Prover sees them -------------------
class MatMulProver:
    """Prover<F, matrix_multiplication::G>"""

    def __init__(self, a, b, p):
        self.p, self.inner = p, R.ProverRef(a, b, p)
        self.c_1, self.num_vars = self.inner.c_1, self.inner.num_vars

    def round(self, r_prev, j):
        self.inner.round(r_prev, j)
        e = R.g_round_evals(self.inner.a, self.inner.b, self.p)            # :110-122
        p = self.p
        return lagrange_quadratic([(0, e[0]), (1 % p, e[1]), (2 % p, e[2])], p)   # :124-130


class TriangleProver:
    """Prover<F, triangle_counting::G> on G::new_adj_matrix(adj)"""

    def __init__(self, adj, var_len, p):
        self.p, self.var_len = p, var_len
        self.cur = (list(adj), list(adj), list(adj))
        self.num_vars = 3 * var_len
        self.c_1 = sum(R.tri_to_evaluations(*self.cur, var_len, p)) % p

    def round(self, r_prev, j):
        if j:
            self.cur = R.tri_fix_variables(*self.cur, self.var_len, [r_prev], self.p)
        return sparse_from_dense(R.tri_to_univariate_domain(*self.cur, self.var_len, self.p))


class WProver:
    """Prover<F, gkr_protocol::round_polynomial::W>"""

    def __init__(self, add, mul, w_b, w_c, p):
        self.p = p
        self.cur = (list(add), list(mul), list(w_b), list(w_c))
        self.num_vars = (len(add) - 1).bit_length()
        self.c_1 = sum(R.w_to_evaluations(*self.cur, p)) % p

    def round(self, r_prev, j):
        if j:
            self.cur = R.w_fix_variables(*self.cur, [r_prev], self.p)
        return sparse_from_dense(R.w_to_univariate_domain(*self.cur, self.p))


# ---- the transform ------------------------------------------------------------------------------------------
def generate_transcript(prover, dst=b""):
    """fiat-shamir/src/lib.rs:75-98.  Returns (messages, challenges): challenges[j-1] = r_j handed to round j."""
    p = prover.p
    g_1 = ser_field(prover.c_1, p) + ser_sparse(prover.round(1 % p, 0), p)      # :45-53
    hash_input, g, rs = bytearray(g_1), [g_1], []
    for j in range(1, prover.num_vars):
        r_j = hash_to_field_1(hash_input, p, dst)
        g_j = ser_sparse(prover.round(r_j, j), p)                              # :55-61
        hash_input += g_j
        g.append(g_j)
        rs.append(r_j)
    return g, rs


def verify_transcript(g, n, evaluate, p, dst=b""):
    """fiat-shamir/src/lib.rs:123-171 over sum_check_protocol::Verifier (sum-check-protocol/src/lib.rs:278-330).
    `evaluate(point)` is the verifier's oracle access to the polynomial."""
    hash_input, rs, prev, c_1 = bytearray(), [], None, None
    for j, g_j in enumerate(g):
        hash_input += g_j
        r_j = hash_to_field_1(hash_input, p, dst)
        off = 0
        if j == 0:
            c_1, off = de_field(g_j, 0, p)
        poly, off = de_sparse(g_j, off, p)
        s01 = (sparse_eval(poly, 0, p) + sparse_eval(poly, 1 % p, p)) % p
        if not rs:
            if s01 != c_1:
                return False
        elif len(rs) == n - 1:
            rs.append(r_j)
            return sparse_eval(poly, r_j, p) == evaluate(rs)
        elif s01 != sparse_eval(prev, rs[-1], p):
            return False
        prev = poly
        rs.append(r_j)
    return True
