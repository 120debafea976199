"""ctypes binding of oracle/libsc_oracle.so (TEST INFRASTRUCTURE ONLY - see sc_oracle.c).

Values crossing this binding are Montgomery-form u64 words (numpy uint64 arrays), the
same words the C ABI of the product uses; `to_mont` / `from_mont` convert from/to the
canonical integers that oracle/pyref.py and the JSON fixtures use.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsc_oracle.so")

u64 = ctypes.c_uint64
u64p = ctypes.POINTER(ctypes.c_uint64)


class Field(ctypes.Structure):
    _fields_ = [("p", u64), ("p_inv_neg", u64), ("r_mod_p", u64), ("r2_mod_p", u64)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "sc_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "libsc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _ptr(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


class Oracle:
    def __init__(self, p):
        self.lib = ctypes.CDLL(build())
        L = self.lib
        FP = ctypes.POINTER(Field)
        L.sco_field_init.argtypes = [u64, FP]
        L.sco_field_init.restype = ctypes.c_int
        for name in ("sco_add", "sco_sub", "sco_mul"):
            getattr(L, name).argtypes = [FP, u64, u64]
            getattr(L, name).restype = u64
        for name in ("sco_inv", "sco_to_mont", "sco_from_mont"):
            getattr(L, name).argtypes = [FP, u64]
            getattr(L, name).restype = u64
        L.sco_to_mont_vec.argtypes = [FP, u64p, ctypes.c_size_t, u64p]
        L.sco_from_mont_vec.argtypes = [FP, u64p, ctypes.c_size_t, u64p]
        L.sco_generate.argtypes = [FP, u64, u64, ctypes.c_size_t, u64p]
        L.sco_challenge.argtypes = [FP, u64, u64]
        L.sco_challenge.restype = u64
        L.sco_mle_fix_variables.argtypes = [FP, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
        L.sco_mle_fix_variables_be.argtypes = [FP, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
        L.sco_mle_evaluate.argtypes = [FP, u64p, ctypes.c_size_t, u64p]
        L.sco_mle_evaluate.restype = u64
        L.sco_mle_relabel.argtypes = [u64p, ctypes.c_size_t] + [ctypes.c_size_t] * 3 + [u64p]
        L.sco_g_new.argtypes = [FP, ctypes.c_size_t, u64p, u64p, u64p, u64p, u64p]
        L.sco_g_round_evals.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p]
        L.sco_g_grid_sums.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p]
        L.sco_g_gridk_sums.argtypes = [FP, u64p, u64p, ctypes.c_size_t, ctypes.c_int, u64p]
        L.sco_interpolate_quadratic.argtypes = [FP, u64p, u64p]
        L.sco_poly2_eval.argtypes = [FP, u64p, u64]
        L.sco_poly2_eval.restype = u64
        L.sco_g_to_evaluations.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p]
        L.sco_g_evaluate.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p]
        L.sco_g_evaluate.restype = u64
        L.sco_prover_c1.argtypes = [FP, u64p, u64p, ctypes.c_size_t]
        L.sco_prover_c1.restype = u64
        L.sco_prove.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p, u64p, u64p, u64p, u64p]
        L.sco_prove.restype = ctypes.c_int
        L.sco_prover_run.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p, u64p, u64p]
        L.sco_prover_run.restype = None
        L.sco_w_to_evaluations.argtypes = [FP, u64p, u64p, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
        L.sco_w_round_evals.argtypes = [FP, u64p, u64p, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
        L.sco_w_evaluate.argtypes = [FP, u64p, u64p, u64p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
        L.sco_w_evaluate.restype = u64
        L.sco_w_prove.argtypes = [FP, u64p, u64p, u64p, u64p, ctypes.c_size_t, u64p, u64p, u64p, u64p]
        L.sco_w_prove.restype = ctypes.c_int
        L.sco_wiring_fixed.argtypes = [FP, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint32),
                                       ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, ctypes.c_size_t, u64p, u64p, u64p]
        sz = ctypes.c_size_t
        L.sco_tri_to_evaluations.argtypes = [FP, u64p, sz, u64p, sz, u64p, sz, sz, u64p]
        L.sco_tri_round_evals.argtypes = [FP, u64p, sz, u64p, sz, u64p, sz, sz, u64p]
        L.sco_tri_evaluate.argtypes = [FP, u64p, sz, u64p]
        L.sco_tri_evaluate.restype = u64
        L.sco_tri_prove.argtypes = [FP, u64p, sz, u64p, u64p, u64p, u64p]
        L.sco_tri_prove.restype = ctypes.c_int
        L.sco_prover_run_mt.argtypes = [FP, u64p, u64p, ctypes.c_size_t, u64p, u64p, u64p]
        L.sco_prover_run_mt.restype = None
        L.sco_vsbw.argtypes = [FP, u64p, u64p, ctypes.c_size_t]
        L.sco_vsbw.restype = u64
        L.sco_cti.argtypes = [FP, u64p, u64p, ctypes.c_size_t]
        L.sco_cti.restype = u64
        self.f = Field()
        if L.sco_field_init(p, ctypes.byref(self.f)) != 0:
            raise ValueError("modulus must be odd and > 2")
        self.p = p
        self.fp = ctypes.byref(self.f)

    # -- conversions ------------------------------------------------------------------
    def to_mont(self, xs):
        if isinstance(xs, (int, np.integer)):
            xs = [xs]
        a = np.array([int(x) % self.p for x in xs], dtype=np.uint64)
        out = np.empty_like(a)
        self.lib.sco_to_mont_vec(self.fp, _ptr(a), a.size, _ptr(out))
        return out

    def from_mont(self, ms):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(ms, dtype=np.uint64)))
        out = np.empty_like(a)
        self.lib.sco_from_mont_vec(self.fp, _ptr(a), a.size, _ptr(out))
        return [int(x) for x in out]

    def to_mont1(self, x):
        return int(self.lib.sco_to_mont(self.fp, int(x) % self.p))

    def from_mont1(self, m):
        return int(self.lib.sco_from_mont(self.fp, int(m)))

    # -- synthetic instance -----------------------------------------------------------
    def generate(self, seed, log_len, start=0):
        out = np.empty(1 << log_len, dtype=np.uint64)
        self.lib.sco_generate(self.fp, seed, start, out.size, _ptr(out))
        return out

    def generate_range(self, seed, start, length):
        out = np.empty(length, dtype=np.uint64)
        self.lib.sco_generate(self.fp, seed, start, out.size, _ptr(out))
        return out

    def challenge(self, seed, j):
        return int(self.lib.sco_challenge(self.fp, seed, j))

    # -- dense MLE --------------------------------------------------------------------
    @staticmethod
    def _nv(t):
        nv = int(t.size - 1).bit_length()
        assert t.size == 1 << nv
        return nv

    def fix_variables(self, t, r, order=0):
        nv = self._nv(t)
        r = np.ascontiguousarray(np.asarray(r, dtype=np.uint64))
        out = np.empty(1 << (nv - r.size), dtype=np.uint64)
        fn = self.lib.sco_mle_fix_variables_be if order else self.lib.sco_mle_fix_variables
        fn(self.fp, _ptr(t), nv, _ptr(r), r.size, _ptr(out))
        return out

    def evaluate(self, t, point):
        pt = np.ascontiguousarray(np.asarray(point, dtype=np.uint64))
        return int(self.lib.sco_mle_evaluate(self.fp, _ptr(t), self._nv(t), _ptr(pt)))

    def relabel(self, t, a, b, k):
        out = np.empty_like(t)
        self.lib.sco_mle_relabel(_ptr(t), self._nv(t), a, b, k, _ptr(out))
        return out

    def vsbw(self, evals, r):
        r = np.ascontiguousarray(np.asarray(r, dtype=np.uint64))
        return int(self.lib.sco_vsbw(self.fp, _ptr(evals), _ptr(r), r.size))

    def cti(self, evals, r):
        r = np.ascontiguousarray(np.asarray(r, dtype=np.uint64))
        return int(self.lib.sco_cti(self.fp, _ptr(evals), _ptr(r), r.size))

    # -- matrix_multiplication::G + Prover --------------------------------------------
    def g_new(self, n, A, B, point):
        fa = np.empty(1 << n, dtype=np.uint64)
        fb = np.empty(1 << n, dtype=np.uint64)
        pt = np.ascontiguousarray(np.asarray(point, dtype=np.uint64))
        self.lib.sco_g_new(self.fp, n, _ptr(A), _ptr(B), _ptr(pt), _ptr(fa), _ptr(fb))
        return fa, fb

    def round_evals(self, a, b):
        e = np.empty(3, dtype=np.uint64)
        self.lib.sco_g_round_evals(self.fp, _ptr(a), _ptr(b), self._nv(a), _ptr(e))
        return e

    def grid_sums(self, a, b):
        s = np.empty(9, dtype=np.uint64)
        self.lib.sco_g_grid_sums(self.fp, _ptr(a), _ptr(b), self._nv(a), _ptr(s))
        return s

    def gridk_sums(self, a, b, k):
        """3^k-cell grid in the {0,1,inf} basis, first variable on the slowest axis (k = 1 .. 5)"""
        s = np.empty(3 ** k, dtype=np.uint64)
        self.lib.sco_g_gridk_sums(self.fp, _ptr(a), _ptr(b), self._nv(a), k, _ptr(s))
        return s

    def interpolate(self, e):
        e = np.ascontiguousarray(np.asarray(e, dtype=np.uint64))
        c = np.empty(3, dtype=np.uint64)
        self.lib.sco_interpolate_quadratic(self.fp, _ptr(e), _ptr(c))
        return c

    def poly2_eval(self, c, x):
        c = np.ascontiguousarray(np.asarray(c, dtype=np.uint64))
        return int(self.lib.sco_poly2_eval(self.fp, _ptr(c), int(x)))

    def to_evaluations(self, a, b):
        out = np.empty_like(a)
        self.lib.sco_g_to_evaluations(self.fp, _ptr(a), _ptr(b), self._nv(a), _ptr(out))
        return out

    def g_evaluate(self, a, b, point):
        pt = np.ascontiguousarray(np.asarray(point, dtype=np.uint64))
        return int(self.lib.sco_g_evaluate(self.fp, _ptr(a), _ptr(b), self._nv(a), _ptr(pt)))

    def c1(self, a, b):
        return int(self.lib.sco_prover_c1(self.fp, _ptr(a), _ptr(b), self._nv(a)))

    def prover_run(self, a, b, challenges):
        """the criterion bench's timed region (no verifier): returns (c_1, evals[n,3])"""
        nv = self._nv(a)
        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64))
        c1 = u64(0)
        ev = np.empty((nv, 3), dtype=np.uint64)
        self.lib.sco_prover_run(self.fp, _ptr(a), _ptr(b), nv, _ptr(ch), ctypes.byref(c1), _ptr(ev))
        return int(c1.value), ev

    def prover_run_mt(self, a, b, challenges):
        """all-cores (OpenMP) variant of prover_run; identical outputs"""
        nv = self._nv(a)
        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64))
        c1 = u64(0)
        ev = np.empty((nv, 3), dtype=np.uint64)
        self.lib.sco_prover_run_mt(self.fp, _ptr(a), _ptr(b), nv, _ptr(ch), ctypes.byref(c1), _ptr(ev))
        return int(c1.value), ev

    def prove(self, a, b, challenges):
        """returns dict(status, c_1, evals[n,3], coeffs[n,3], final_eval) in Montgomery form"""
        nv = self._nv(a)
        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64))
        assert ch.size == nv
        c1 = u64(0)
        fin = u64(0)
        ev = np.empty((nv, 3), dtype=np.uint64)
        co = np.empty((nv, 3), dtype=np.uint64)
        st = self.lib.sco_prove(self.fp, _ptr(a), _ptr(b), nv, _ptr(ch), ctypes.byref(c1),
                                _ptr(ev), _ptr(co), ctypes.byref(fin))
        return {"status": st, "c_1": int(c1.value), "evals": ev, "coeffs": co,
                "final_eval": int(fin.value)}

    # -- gkr_protocol::round_polynomial::W -------------------------------------------------
    def w_to_evaluations(self, add, mul, w_b, w_c):
        out = np.empty(add.size, dtype=np.uint64)
        self.lib.sco_w_to_evaluations(self.fp, _ptr(add), _ptr(mul), _ptr(w_b), self._nv(w_b), _ptr(w_c),
                                      self._nv(w_c), _ptr(out))
        return out

    def w_round_evals(self, add, mul, w_b, w_c):
        e = np.empty(3, dtype=np.uint64)
        self.lib.sco_w_round_evals(self.fp, _ptr(add), _ptr(mul), _ptr(w_b), self._nv(w_b), _ptr(w_c),
                                   self._nv(w_c), _ptr(e))
        return e

    def w_evaluate(self, add, mul, w_b, w_c, point):
        pt = np.ascontiguousarray(np.asarray(point, dtype=np.uint64))
        return int(self.lib.sco_w_evaluate(self.fp, _ptr(add), _ptr(mul), _ptr(w_b), self._nv(w_b), _ptr(w_c),
                                           self._nv(w_c), _ptr(pt)))

    def w_prove(self, add, mul, w_b, w_c, challenges):
        k = self._nv(w_b)
        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64))
        assert ch.size == 2 * k
        c1, fin = u64(0), u64(0)
        ev = np.empty((2 * k, 3), dtype=np.uint64)
        st = self.lib.sco_w_prove(self.fp, _ptr(add), _ptr(mul), _ptr(w_b), _ptr(w_c), k, _ptr(ch),
                                  ctypes.byref(c1), _ptr(ev), ctypes.byref(fin))
        return {"status": st, "c_1": int(c1.value), "evals": ev, "final_eval": int(fin.value)}

    def wiring_fixed(self, layer, k_next, r_i):
        """layer: list of ('add'|'mul', in0, in1); returns add_i(r_i,.,.), mul_i(r_i,.,.)"""
        k_i = (len(layer) - 1).bit_length()
        gt = (ctypes.c_int * len(layer))(*[0 if t == "add" else 1 for t, _, _ in layer])
        i0 = (ctypes.c_uint32 * len(layer))(*[a for _, a, _ in layer])
        i1 = (ctypes.c_uint32 * len(layer))(*[b for _, _, b in layer])
        r = np.ascontiguousarray(np.asarray(r_i, dtype=np.uint64))
        add = np.empty(1 << (2 * k_next), dtype=np.uint64)
        mul = np.empty(1 << (2 * k_next), dtype=np.uint64)
        self.lib.sco_wiring_fixed(self.fp, gt, i0, i1, k_i, k_next, _ptr(r), _ptr(add), _ptr(mul))
        return add, mul

    # -- triangle_counting::G -------------------------------------------------------------------
    def tri_to_evaluations(self, f1, f2, f3, k):
        n1, n2, n3 = self._nv(f1), self._nv(f2), self._nv(f3)
        xv, yv, zv = max(n1 - k, 0), max(n2 - k, 0), (n3 if n3 < k else k)
        out = np.empty(1 << (xv + yv + zv), dtype=np.uint64)
        self.lib.sco_tri_to_evaluations(self.fp, _ptr(f1), n1, _ptr(f2), n2, _ptr(f3), n3, k, _ptr(out))
        return out

    def tri_round_evals(self, f1, f2, f3, k):
        e = np.empty(3, dtype=np.uint64)
        self.lib.sco_tri_round_evals(self.fp, _ptr(f1), self._nv(f1), _ptr(f2), self._nv(f2), _ptr(f3), self._nv(f3),
                                     k, _ptr(e))
        return e

    def tri_evaluate(self, adj, k, point):
        pt = np.ascontiguousarray(np.asarray(point, dtype=np.uint64))
        return int(self.lib.sco_tri_evaluate(self.fp, _ptr(adj), k, _ptr(pt)))

    def tri_prove(self, adj, k, challenges):
        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64))
        assert ch.size == 3 * k
        c1, fin = u64(0), u64(0)
        ev = np.empty((3 * k, 3), dtype=np.uint64)
        st = self.lib.sco_tri_prove(self.fp, _ptr(adj), k, _ptr(ch), ctypes.byref(c1), _ptr(ev), ctypes.byref(fin))
        return {"status": st, "c_1": int(c1.value), "evals": ev, "final_eval": int(fin.value)}
