"""pyref.py - independent big-integer restatement of the reference's sumcheck hot path.

TEST INFRASTRUCTURE ONLY (see oracle/sc_oracle.c header): imported by tests/, by
oracle/gen_golden.py (which writes tests/golden/*.json) and by nothing in the product.

Everything here works on *canonical* Python ints in [0, p) - no Montgomery form, no
64-bit tricks - so it shares no arithmetic code with oracle/sc_oracle.c or with the HIP
kernels; the three must agree value-for-value.  Each function cites the reference
file:line it follows (paths relative to /root/reference).  The dense-MLE operations are
those of the un-vendored crate ark-poly = "0.6" (`DenseMultilinearExtension`), restated
from its published algorithm and anchored on the reference's call sites.
"""

GOLDILOCKS = 2**64 - 2**32 + 1
MASK64 = 2**64 - 1

SEED_A = 0xA5A5000000000001
SEED_B = 0xB6B6000000000002
SEED_R = 0xC7C7000000000003
SEED_PT = 0xD8D8000000000004


def splitmix64(x):
    z = (x + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def synth_table(seed, log_len, p, start=0):
    """t[i] = splitmix64(seed + i) mod p."""
    return [splitmix64((seed + start + i) & MASK64) % p for i in range(1 << log_len)]


def synth_challenge(seed, j, p):
    return splitmix64((seed + j) & MASK64) % p


# ---- ark_poly::DenseMultilinearExtension (LE: variable 0 = index bit 0) -------------

def mle_fix_variables(t, r, p):
    """fix the first len(r) variables; call sites matrix-multiplication/src/lib.rs:83,86,104,105."""
    poly = list(t)
    nv = (len(poly) - 1).bit_length()
    assert len(poly) == 1 << nv
    for i, ri in enumerate(r, start=1):
        for b in range(1 << (nv - i)):
            left, right = poly[2 * b], poly[2 * b + 1]
            poly[b] = (left + ri * (right - left)) % p
    return poly[: 1 << (nv - len(r))]


def mle_evaluate(t, point, p):
    """matrix-multiplication/src/lib.rs:97-98."""
    out = mle_fix_variables(t, point, p)
    assert len(out) == 1
    return out[0]


def mle_relabel(t, a, b, k):
    """swap variables [a,a+k) and [b,b+k); matrix-multiplication/src/lib.rs:82."""
    mask = (1 << k) - 1
    out = [0] * len(t)
    for i, v in enumerate(t):
        fa, fb = (i >> a) & mask, (i >> b) & mask
        j = i & ~((mask << a) | (mask << b))
        out[j | (fb << a) | (fa << b)] = v
    return out


# ---- matrix_multiplication::G -------------------------------------------------------

def g_new(n, A_flat, B_flat, point, p):
    """matrix-multiplication/src/lib.rs:77-92."""
    f_a = mle_fix_variables(mle_relabel(list(A_flat), 0, n, n), point[:n], p)
    f_b = mle_fix_variables(list(B_flat), point[n:], p)
    assert len(f_a) == 1 << n and len(f_b) == 1 << n
    return f_a, f_b


def g_round_evals(a, b, p):
    """the three running sums of G::to_univariate, matrix-multiplication/src/lib.rs:110-122."""
    e = [0, 0, 0]
    for i in range(len(a)):
        if i & 1:
            e[1] += a[i] * b[i]
            e[2] += (2 * a[i] - a[i - 1]) * (2 * b[i] - b[i - 1])
        else:
            e[0] += a[i] * b[i]
    return [x % p for x in e]


def interpolate_quadratic(points, p):
    """matrix-multiplication/src/lib.rs:17-60: dense (c0,c1,c2) through three points."""
    c = [0, 0, 0]
    for i in range(3):
        (xi, yi), (xj, _), (xk, _) = points[i], points[(i + 1) % 3], points[(i + 2) % 3]
        den_inv = pow((xi - xj) * (xi - xk) % p, p - 2, p)
        c[0] += xj * xk * yi * den_inv
        c[1] += (-xj - xk) * yi * den_inv
        c[2] += yi * den_inv
    return [x % p for x in c]


def g_to_univariate(a, b, p):
    """matrix-multiplication/src/lib.rs:110-131 -> dense coefficients."""
    e = g_round_evals(a, b, p)
    return interpolate_quadratic([(0, e[0]), (1 % p, e[1]), (2 % p, e[2])], p)


def poly_eval(c, x, p):
    acc = 0
    for coef in reversed(c):
        acc = (acc * x + coef) % p
    return acc


def g_to_evaluations(a, b, p):
    """matrix-multiplication/src/lib.rs:137-146."""
    return [(x * y) % p for x, y in zip(a, b)]


def g_evaluate(a, b, point, p):
    """matrix-multiplication/src/lib.rs:96-101."""
    return mle_evaluate(a, point, p) * mle_evaluate(b, point, p) % p


# ---- sum_check_protocol::{Prover, Verifier} -----------------------------------------

class ProverRef:
    """sum-check-protocol/src/lib.rs:73-117 specialised to the table-backed G."""

    def __init__(self, a, b, p):
        self.a, self.b, self.p = list(a), list(b), p
        self.c_1 = sum(g_to_evaluations(a, b, p)) % p          # :89
        self.num_vars = (len(a) - 1).bit_length()
        self.r = []

    def round(self, r_prev, j):                                # :105-112
        if j != 0:
            self.r.append(r_prev)
            self.a = mle_fix_variables(self.a, [r_prev], self.p)
            self.b = mle_fix_variables(self.b, [r_prev], self.p)
        return g_to_univariate(self.a, self.b, self.p)


class VerifierRef:
    """sum-check-protocol/src/lib.rs:227-331.  round() returns ('jth', r) / ('final', bool)
    or raises ValueError for ProverClaimMismatch."""

    def __init__(self, n, oracle, p):
        self.n, self.oracle, self.p = n, oracle, p
        self.c_1 = 0
        self.g_part, self.r = [], []

    def set_c_1(self, c_1):
        self.c_1 = c_1

    def round(self, g_j, r_j):
        p = self.p
        if not self.r:                                         # :284-297
            ev = (poly_eval(g_j, 0, p) + poly_eval(g_j, 1 % p, p)) % p
            if self.c_1 != ev:
                raise ValueError("prover claim mismatches evaluation start %d %d" % (self.c_1, ev))
            self.g_part.append(g_j)
            self.r.append(r_j)
            return ("jth", r_j)
        if len(self.r) == self.n - 1:                          # :298-310
            self.r.append(r_j)
            if self.oracle is None:
                raise LookupError("verifier has no oracle access to the polynomial")
            return ("final", poly_eval(g_j, r_j, p) == self.oracle(self.r))
        prev = poly_eval(self.g_part[-1], self.r[-1], p)       # :313-328
        ev = (poly_eval(g_j, 0, p) + poly_eval(g_j, 1 % p, p)) % p
        if prev != ev:
            raise ValueError("prover claim mismatches evaluation %d %d" % (prev, ev))
        self.g_part.append(g_j)
        self.r.append(r_j)
        return ("jth", r_j)


def transcript(a, b, challenges, p):
    """Full interactive run as in matrix-multiplication/src/lib.rs:337-370 with the
    verifier's draws scripted by `challenges` (challenges[j] drawn in round j).
    n = 1 follows the reference literally: Verifier::round takes the 'first round' branch
    (r is empty, :284) and never reaches FinalRound."""
    n = (len(a) - 1).bit_length()
    prover = ProverRef(a, b, p)
    verifier = VerifierRef(n, lambda pt: g_evaluate(a, b, pt, p), p)
    verifier.set_c_1(prover.c_1)
    r_j = 1 % p
    evals, coeffs, accepted = [], [], None
    pa, pb = list(a), list(b)
    for j in range(n):
        if j:
            pa = mle_fix_variables(pa, [r_j], p)
            pb = mle_fix_variables(pb, [r_j], p)
        evals.append(g_round_evals(pa, pb, p))
        g_j = prover.round(r_j, j)
        coeffs.append(g_j)
        kind, val = verifier.round(g_j, challenges[j])
        if kind == "jth":
            r_j = val
        else:
            accepted = val
    return {
        "c_1": prover.c_1,
        "evals": evals,
        "coeffs": coeffs,
        "final_eval": g_evaluate(a, b, challenges[:n], p),
        "accepted": accepted,
    }


# ---- multilinear-extensions crate (BE: r[0] <-> index MSB) --------------------------

def vsbw(evals, r, p):
    """multilinear-extensions/src/lib.rs:6-24."""
    table = [1 % p]
    for r_j in r:
        new = []
        for e in table:
            new.append(e * (1 - r_j) % p)
            new.append(e * r_j % p)
        table = new
    return sum(w * v for w, v in zip(table, evals)) % p


def cti(evals, r, p):
    """multilinear-extensions/src/lib.rs:29-60."""
    n = len(r)
    res = 0
    for i, ev in enumerate(evals):
        w = [1 if i & (1 << j) else 0 for j in reversed(range(n))]
        basis = 1
        for x_i, w_i in zip(r, w):
            basis = basis * (x_i * w_i + (1 - x_i) * (1 - w_i)) % p
        res += ev * basis
    return res % p


def mle_fix_variables_be(t, r, p):
    """stride-half pairing: fixes index bits nv-1, nv-2, ... (the crate's variable order)."""
    poly = list(t)
    for ri in r:
        half = len(poly) // 2
        poly = [(poly[b] + ri * (poly[b + half] - poly[b])) % p for b in range(half)]
    return poly


# ---- two-variables-per-pass identities (checker for the GPU schedule) ---------------

def g_grid_sums(a, b, p):
    """S[u][v] = sum over quads of a(u,v)*b(u,v), (u,v) in {0,1,inf}^2 on index bits (0,1);
    'inf' is the leading coefficient t1 - t0 of the (bi)linear extension."""
    S = [[0] * 3 for _ in range(3)]
    for q in range(len(a) // 4):
        def ext(t):
            g = [[t[4 * q + 2 * h], t[4 * q + 2 * h + 1],
                  t[4 * q + 2 * h + 1] - t[4 * q + 2 * h]] for h in range(2)]
            return [[g[0][u], g[1][u], g[1][u] - g[0][u]] for u in range(3)]
        ea, eb = ext(a), ext(b)
        for u in range(3):
            for v in range(3):
                S[u][v] += ea[u][v] * eb[u][v]
    return [[x % p for x in row] for row in S]


def evals_from_inf(h0, h1, hinf, p):
    """(H(0), H(1), H(inf)) -> (H(0), H(1), H(2)) for a quadratic H"""
    return [h0 % p, h1 % p, (2 * h1 - h0 + 2 * hinf) % p]


def grid_round_evals(S, r, p):
    """the two rounds one grid serves: round j, and round j+1 once r_j = r is known"""
    first = evals_from_inf(S[0][0] + S[0][1], S[1][0] + S[1][1], S[2][0] + S[2][1], p)
    q = [(S[0][v] + r * (S[1][v] - S[0][v] - S[2][v]) + r * r * S[2][v]) % p for v in range(3)]
    return first, evals_from_inf(q[0], q[1], q[2], p)


# ---- other reference call sites that pin the LE index convention --------------------

def poly_mul(a, b, p):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % p
    return out


def restrict_poly(b, c, evals, p):
    """gkr-protocol/src/lib.rs:291-321 -> dense coefficients (trailing zeros trimmed)."""
    k = [(ci - bi) % p for bi, ci in zip(b, c)]
    nv = (len(evals) - 1).bit_length()
    res = [0] * (nv + 1)
    for i, ev in enumerate(evals):
        poly = [ev % p]
        for bit in range(nv):
            lin = [b[bit] % p, k[bit]]
            if i & (1 << bit) == 0:
                lin = [(1 - lin[0]) % p, (-lin[1]) % p]
            poly = poly_mul(poly, lin, p)
        for d, x in enumerate(poly):
            res[d] = (res[d] + x) % p
    while len(res) > 1 and res[-1] == 0:
        res.pop()
    return res


def triangle_c1(adj_flat, k, p):
    """Prover::new on triangle_counting::G: sum of G::to_evaluations,
    triangle-counting/src/lib.rs:138-172 with idx(i,j,k) = (i << k) | j."""
    n = 1 << k
    total = 0
    for x in range(n):
        for y in range(n):
            for z in range(n):
                total += adj_flat[(y << k) | x] * adj_flat[(z << k) | y] * adj_flat[(z << k) | x]
    return total % p


def matmul(A, B, p):
    n = len(A)
    return [[sum(A[i][k] * B[k][j] for k in range(n)) % p for j in range(n)] for i in range(n)]


def bits_le(v, nbits):
    """u32_to_boolean_vec, matrix-multiplication/src/lib.rs:305-313."""
    return [(v >> i) & 1 for i in range(nbits)]


# ---- gkr_protocol::round_polynomial::W and the wiring predicates ----------------------
# f(b, c) = add(b,c) (W(b) + W(c)) + mul(b,c) W(b) W(c); tables indexed (c << k) | b.

def w_to_evaluations(add, mul, w_b, w_c, p):
    """gkr-protocol/src/round_polynomial.rs:96-118 - note the push order: b outer, c inner,
    while the tables are read at idx(c, b, num_vars(w_b)) = (c << kb) | b."""
    kb = (len(w_b) - 1).bit_length()
    res = []
    for b_idx, wb in enumerate(w_b):
        for c_idx, wc in enumerate(w_c):
            bc = (c_idx << kb) | b_idx
            res.append((add[bc] * (wb + wc) + mul[bc] * (wb * wc)) % p)
    return res


def w_fix_variables(add, mul, w_b, w_c, partial, p):
    """gkr-protocol/src/round_polynomial.rs:59-76"""
    kb = (len(w_b) - 1).bit_length()
    b_part = partial[:min(kb, len(partial))]
    c_part = partial[kb:]
    return (mle_fix_variables(add, partial, p), mle_fix_variables(mul, partial, p),
            mle_fix_variables(w_b, b_part, p), mle_fix_variables(w_c, c_part, p))


def w_evaluate(add, mul, w_b, w_c, point, p):
    """gkr-protocol/src/round_polynomial.rs:48-57"""
    kb = (len(w_b) - 1).bit_length()
    b, c = point[:kb], point[kb:]
    wb, wc = mle_evaluate(w_b, b, p), mle_evaluate(w_c, c, p)
    return (mle_evaluate(add, point, p) * (wb + wc) + mle_evaluate(mul, point, p) * wb * wc) % p


def primitive_root_of_unity(order, p):
    assert (p - 1) % order == 0
    for g in range(2, p):
        w = pow(g, (p - 1) // order, p)
        if all(pow(w, order // q, p) != 1 for q in (2,) if order % q == 0) and pow(w, order, p) == 1:
            if len({pow(w, i, p) for i in range(order)}) == order:
                return w
    raise ValueError("no root of unity")


def w_to_univariate_domain(add, mul, w_b, w_c, p):
    """gkr-protocol/src/round_polynomial.rs:78-90, literally: sum the polynomial with the
    first variable fixed at each element of the size-4 radix-2 domain, then interpolate
    (inverse DFT).  Any primitive 4th root gives the same coefficient vector."""
    w = primitive_root_of_unity(4, p)
    dom = [pow(w, i, p) for i in range(4)]
    evals = [sum(w_to_evaluations(*w_fix_variables(add, mul, w_b, w_c, [e], p), p)) % p for e in dom]
    inv4 = pow(4, p - 2, p)
    coeffs = [sum(evals[i] * pow(w, (-i * d) % 4, p) for i in range(4)) * inv4 % p for d in range(4)]
    while coeffs and coeffs[-1] == 0:
        coeffs.pop()
    return coeffs


def w_round_evals(add, mul, w_b, w_c, p):
    """(H(0), H(1), H(2)) of the round polynomial by direct substitution"""
    return [sum(w_to_evaluations(*w_fix_variables(add, mul, w_b, w_c, [x % p], p), p)) % p for x in (0, 1, 2)]


def w_transcript(add, mul, w_b, w_c, challenges, p):
    """Prover::new + rounds on W with the verifier's identities checked"""
    n = (len(add) - 1).bit_length()
    c1 = sum(w_to_evaluations(add, mul, w_b, w_c, p)) % p
    cur = (list(add), list(mul), list(w_b), list(w_c))
    evals, coeffs = [], []
    claim = c1
    for j in range(n):
        if j:
            cur = w_fix_variables(*cur, [challenges[j - 1]], p)
        e = w_round_evals(*cur, p)
        c = interpolate_quadratic([(0, e[0]), (1 % p, e[1]), (2 % p, e[2])], p)
        assert (e[0] + e[1]) % p == claim, "round %d" % j
        claim = poly_eval(c, challenges[j], p)
        evals.append(e)
        coeffs.append(c)
    final = w_evaluate(add, mul, w_b, w_c, challenges, p)
    assert claim == final
    return {"c_1": c1, "evals": evals, "coeffs": coeffs, "final_eval": final}


def circuit_evaluate(layers, inputs, p):
    """gkr-protocol/src/circuit.rs:99-124; layers[0] is the output layer; a gate is
    ('add'|'mul', in0, in1).  Returns the per-layer values, outputs first."""
    vals = [[x % p for x in inputs]]
    cur = vals[0]
    for layer in reversed(layers):
        cur = [(cur[i0] + cur[i1]) % p if t == "add" else (cur[i0] * cur[i1]) % p for (t, i0, i1) in layer]
        vals.append(cur)
    vals.reverse()
    return vals


def wiring_tables(layer, k_next, p):
    """the dense add_i / mul_i tables of start_round, gkr-protocol/src/lib.rs:388-404:
    index ((c << k_next) | b) << k_i | a"""
    k_i = (len(layer) - 1).bit_length()
    n = 1 << k_next
    add_t, mul_t = [], []
    for c in range(n):
        for b in range(n):
            for a in range(1 << k_i):
                t, i0, i1 = layer[a]
                add_t.append(1 if (t == "add" and i0 == b and i1 == c) else 0)
                mul_t.append(1 if (t == "mul" and i0 == b and i1 == c) else 0)
    return add_t, mul_t


def wiring_fixed(layer, k_next, r_i, p):
    """add_i(r_i, ., .), mul_i(r_i, ., .): the tables above with the a-variables fixed at r_i
    (gkr-protocol/src/lib.rs:406-416)"""
    add_t, mul_t = wiring_tables(layer, k_next, p)
    return mle_fix_variables(add_t, r_i, p), mle_fix_variables(mul_t, r_i, p)


# ---- triangle_counting::G: g(X,Y,Z) = f(X,Y) f(Y,Z) f(X,Z) --------------------------------
# three copies of the adjacency MLE; idx(i, j, nv) = (i << nv) | j; variables x, then y, then z.

def tri_var_counts(f1, f2, f3, var_len):
    """triangle-counting/src/lib.rs:53-67"""
    nv = lambda t: (len(t) - 1).bit_length()
    xv = max(nv(f1) - var_len, 0)
    yv = max(nv(f2) - var_len, 0)
    zv = nv(f3) if nv(f3) < var_len else var_len
    return xv, yv, zv


def tri_to_evaluations(f1, f2, f3, var_len, p):
    """triangle-counting/src/lib.rs:138-165"""
    xv, yv, zv = tri_var_counts(f1, f2, f3, var_len)
    res = []
    for x in range(1 << xv):
        for y in range(1 << yv):
            for z in range(1 << zv):
                res.append(f1[(y << xv) | x] * f2[(z << yv) | y] * f3[(z << xv) | x] % p)
    return res


def tri_fix_variables(f1, f2, f3, var_len, pp, p):
    """triangle-counting/src/lib.rs:89-118"""
    xv, yv, zv = tri_var_counts(f1, f2, f3, var_len)
    x_y = pp[:min(xv + yv, len(pp))]
    y_z = pp[xv:]
    x_z = pp[:min(xv, len(pp))] + pp[xv + yv:]
    return mle_fix_variables(f1, x_y, p), mle_fix_variables(f2, y_z, p), mle_fix_variables(f3, x_z, p)


def tri_evaluate(f1, f2, f3, var_len, point, p):
    """triangle-counting/src/lib.rs:71-87"""
    xv, yv, zv = tri_var_counts(f1, f2, f3, var_len)
    x_y = point[:xv + yv]
    y_z = point[xv:]
    x_z = point[:xv] + point[xv + yv:]
    return mle_evaluate(f1, x_y, p) * mle_evaluate(f2, y_z, p) * mle_evaluate(f3, x_z, p) % p


def tri_round_evals(f1, f2, f3, var_len, p):
    """(H(0), H(1), H(2)) by direct substitution (the reference uses the 4-point domain, :120-132)"""
    return [sum(tri_to_evaluations(*tri_fix_variables(f1, f2, f3, var_len, [x % p], p), var_len, p)) % p
            for x in (0, 1, 2)]


def tri_to_univariate_domain(f1, f2, f3, var_len, p):
    """triangle-counting/src/lib.rs:120-132 literally (size-4 domain + inverse DFT)"""
    w = primitive_root_of_unity(4, p)
    dom = [pow(w, i, p) for i in range(4)]
    evals = [sum(tri_to_evaluations(*tri_fix_variables(f1, f2, f3, var_len, [e], p), var_len, p)) % p for e in dom]
    inv4 = pow(4, p - 2, p)
    coeffs = [sum(evals[i] * pow(w, (-i * d) % 4, p) for i in range(4)) * inv4 % p for d in range(4)]
    while coeffs and coeffs[-1] == 0:
        coeffs.pop()
    return coeffs


def tri_transcript(adj, var_len, challenges, p):
    """Prover::new + all 3*var_len rounds on G::new_adj_matrix(adj), verifier identities checked"""
    cur = (list(adj), list(adj), list(adj))
    n = 3 * var_len
    c1 = sum(tri_to_evaluations(*cur, var_len, p)) % p
    evals, claim = [], c1
    for j in range(n):
        if j:
            cur = tri_fix_variables(*cur, var_len, [challenges[j - 1]], p)
        e = tri_round_evals(*cur, var_len, p)
        c = interpolate_quadratic([(0, e[0]), (1 % p, e[1]), (2 % p, e[2])], p)
        assert (e[0] + e[1]) % p == claim, "round %d" % j
        claim = poly_eval(c, challenges[j], p)
        evals.append(e)
    final = tri_evaluate(adj, adj, adj, var_len, challenges, p)
    assert claim == final
    return {"c_1": c1, "evals": evals, "final_eval": final}


# ---- gkr_protocol::{Prover, Verifier}: the message loop of the reference's protocol tests ---------
# gkr-protocol/src/lib.rs:38-218 (Verifier), :324-474 (Prover), driven as in
# protocol_test_from_book (:550-624) / three_layer_protocol_test (:626-702).  Randomness is a
# scripted list consumed in the order the reference draws it: k_0 values for r_0 (Begin, :193),
# then per layer one value per sumcheck round (verifier.round draws before it checks, :283 of
# sum-check-protocol; the last one by final_random_point, :108), then the line parameter (:157).

def gkr_transcript(layers, num_inputs, inputs, draws, p):
    """Runs the whole GKR protocol on canonical ints and returns every message.  Raises
    AssertionError where the reference's verifier would reject (unwrap / assert_eq!)."""
    draws = list(draws)
    vals = circuit_evaluate(layers, inputs, p)                     # Prover::new, :346-357
    k = [(len(l) - 1).bit_length() for l in layers] + [(num_inputs - 1).bit_length()]
    out = {"circuit_outputs": vals[0], "layers": []}              # Begin, :363-367
    r_i = [draws.pop(0) for _ in range(k[0])]                      # :193
    m_i = mle_evaluate(vals[0], r_i, p)                            # :195
    out["r_0"], out["m_0"] = list(r_i), m_i
    for i in range(len(layers)):
        k_next = k[i + 1]
        add_f, mul_f = wiring_fixed(layers[i], k_next, r_i, p)     # :388-416 (prover), :90-91 (verifier)
        w = vals[i + 1]
        n = 2 * k_next
        ch = [draws.pop(0) for _ in range(n)]                      # n-1 by verifier.round, 1 by final_random_point
        tr = w_transcript(add_f, mul_f, w, w, ch, p)               # rounds + the sumcheck verifier's identities
        assert tr["c_1"] == m_i, "layer %d: c_1 != m_i" % i        # the claim the layer reduces
        b, c = ch[:k_next], ch[k_next:]                            # :442, :155
        q = restrict_poly(b, c, w, p)                              # :444
        q0, q1 = poly_eval(q, 0, p), poly_eval(q, 1 % p, p)        # :146-147
        ev = (mle_evaluate(add_f, ch, p) * (q0 + q1) + mle_evaluate(mul_f, ch, p) * q0 * q1) % p   # :149
        assert ev == poly_eval(tr["coeffs"][-1], ch[-1], p), "layer %d: final round message" % i   # :151
        r_line = draws.pop(0)                                      # :153
        r_next = [(bi + r_line * (ci - bi)) % p for bi, ci in zip(b, c)]       # :156-158
        m_next = poly_eval(q, r_line, p)                           # :159
        out["layers"].append({"c_1": tr["c_1"], "num_vars": n, "evals": tr["evals"], "coeffs": tr["coeffs"],
                              "challenges": ch, "q": q, "r_line": r_line, "r_next": r_next, "m_next": m_next})
        r_i, m_i = r_next, m_next
    out["check_input"] = mle_evaluate([x % p for x in inputs], r_i, p) == m_i   # :210-217
    assert not draws, "unused draws"
    return out
