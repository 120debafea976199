"""Python mirror of the reference crate `fiat-shamir` (src/lib.rs): the non-interactive transform
around `sum_check_protocol::{Prover, Verifier}`.  Host-side only: O(n)
bytes per proof; every `prover.round` underneath is a GPU pass of the engine.

  InteractiveProver (:33-66)   generate_transcript (:75-98)   RandNums (:102-119)
  verify_transcript (:123-143) InteractiveVerifier (:146-171)

Wire format: `ark-serialize` `serialize_uncompressed` of `(F, SparsePolynomial<F>)` (round 1) and
of `SparsePolynomial<F>` (later rounds), restated from the published layout - field element =
canonical integer, little-endian, ceil(modulus_bits / 8) bytes; Vec = u64-LE length then items;
`usize` = u64 LE.  Challenges: any `HashToField`; `Sha256FieldHasher` restates ark-ff's
`DefaultFieldHasher<Sha256, 128>` (RFC 9380 expand_message_xmd, empty DST unless given).  The
expander is pinned against RFC 9380 appendix K.1 (tests/golden/rfc9380_k1_xmd_sha256.json) in
its "rfc9380" mode (Z_pad = the SHA-256 block, 64 bytes); the default "arkworks" mode follows
ark-ff's `DefaultFieldHasher::new`, which builds its `ExpanderXmd` with
`block_size = len_per_base_elem` = ceil((modulus_bits + 128) / 8) - so Z_pad is 17 bytes for p = 5,
24 for Goldilocks, and 64 only by coincidence for BLS12-381.
PARITY UNPINNED for arkworks byte identity: the reference's only assertion here is accept/reject
(fiat-shamir/src/lib.rs:231-234) and the Rust cannot be run in this image, so no ark-ff known answer
exists to pin the arkworks mode or the serializer against.
"""
import hashlib

from .sum_check_protocol import Error, RngF, SparsePolynomial, VerifierRoundResult


class SerializationError(Error):
    """Error::Serialization (:13-16)"""


def _field_bytes(field):
    return (field.p.bit_length() + 7) // 8


def serialize_field(field, m):
    return field.to_int(m).to_bytes(_field_bytes(field), "little")


def deserialize_field(field, data, off):
    n = _field_bytes(field)
    if off + n > len(data):
        raise SerializationError("Codec error")
    v = int.from_bytes(data[off:off + n], "little")
    if v >= field.p:
        raise SerializationError("Codec error")
    return field.from_int(v), off + n


def serialize_poly(poly):
    out = len(poly.coeffs).to_bytes(8, "little")
    for d, c in poly.coeffs:
        out += int(d).to_bytes(8, "little") + serialize_field(poly.field, c)
    return out


def deserialize_poly(field, data, off=0):
    if off + 8 > len(data):
        raise SerializationError("Codec error")
    n = int.from_bytes(data[off:off + 8], "little")
    off += 8
    coeffs = []
    for _ in range(n):
        if off + 8 > len(data):
            raise SerializationError("Codec error")
        d = int.from_bytes(data[off:off + 8], "little")
        c, off = deserialize_field(field, data, off + 8)
        coeffs.append((d, c))
    return SparsePolynomial(field, coeffs), off        # the derived CanonicalDeserialize: the Vec as it was written


class Sha256FieldHasher:
    """ark_ff::field_hashers::DefaultFieldHasher<Sha256, 128> restated (RFC 9380, section 5.3.1)"""

    def __init__(self, field, dst=b"", z_pad="arkworks"):
        self.field, self.dst = field, bytes(dst)
        self.len_per_elem = (field.p.bit_length() + 128 + 7) // 8
        # ark-ff: ExpanderXmd { block_size: len_per_base_elem }; RFC 9380: s_in_bytes of SHA-256
        if z_pad == "arkworks":
            self.z_pad = self.len_per_elem
        elif z_pad == "rfc9380":
            self.z_pad = 64
        else:
            self.z_pad = int(z_pad)

    def _expand(self, msg, n):
        ell = (n + 31) // 32
        if ell > 255 or n >= 1 << 16:
            raise ValueError("expand_message_xmd: output too long")
        dst = self.dst
        if len(dst) > 255:                      # construct_dst_prime / RFC 9380 section 5.3.3
            dst = hashlib.sha256(b"H2C-OVERSIZE-DST-" + dst).digest()
        dst_prime = dst + bytes([len(dst)])
        b0 = hashlib.sha256(bytes(self.z_pad) + msg + n.to_bytes(2, "big") + b"\x00" + dst_prime).digest()
        out, bi = b"", hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
        out += bi
        for i in range(2, ell + 1):
            bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
            out += bi
        return out[:n]

    def hash_to_field(self, msg, count=1):
        data = self._expand(bytes(msg), count * self.len_per_elem)
        return [self.field.from_int(int.from_bytes(data[i * self.len_per_elem:(i + 1) * self.len_per_elem], "big"))
                for i in range(count)]


class InteractiveProver:
    """the impl for sum_check_protocol::Prover (:44-66)"""

    def __init__(self, prover):
        self.prover = prover

    def g_1(self):
        f = self.prover.field
        c_1 = self.prover.c_1()
        return serialize_field(f, c_1) + serialize_poly(self.prover.round(f.one, 0))

    def round(self, j, r_j):
        return serialize_poly(self.prover.round(r_j, j))

    def num_rounds(self):
        return self.prover.num_vars()


class FiatShamirTranscript:
    """:69-71"""

    def __init__(self, g):
        self.g = list(g)


def generate_transcript(prover, hasher):
    """:75-98"""
    prover = prover if isinstance(prover, InteractiveProver) else InteractiveProver(prover)
    g_1 = prover.g_1()
    hash_input = bytearray(g_1)
    g = [g_1]
    for j in range(1, prover.num_rounds()):
        r_j = hasher.hash_to_field(hash_input, 1)[0]
        g_j = prover.round(j, r_j)
        hash_input += g_j
        g.append(g_j)
    return FiatShamirTranscript(g)


class RandNums(RngF):
    """:102-119"""

    def __init__(self, nums):
        self.nums, self.current = list(nums), 0

    def draw(self):
        res = self.nums[self.current]
        self.current += 1
        return res


class InteractiveVerifier:
    """the impl for sum_check_protocol::Verifier (:151-171)"""

    def __init__(self, verifier):
        self.verifier = verifier

    def round(self, j, g_j, rng):
        f = self.verifier.field
        if j == 0:
            c_1, off = deserialize_field(f, g_j, 0)
            self.verifier.set_c_1(c_1)
            poly, _ = deserialize_poly(f, g_j, off)
        else:
            poly, _ = deserialize_poly(f, g_j, 0)
        res = self.verifier.round(poly, rng)
        if res.is_final():
            return bool(res.value)
        return True


def verify_transcript(transcript, verifier, hasher):
    """:123-143"""
    verifier = verifier if isinstance(verifier, InteractiveVerifier) else InteractiveVerifier(verifier)
    hash_input = bytearray()
    for j, g_j in enumerate(transcript.g):
        hash_input += g_j
        r_j = hasher.hash_to_field(hash_input, 1)[0]
        if not verifier.round(j, g_j, RandNums([r_j])):
            return False
    return True
