"""Python mirror of the reference crate `multilinear-extensions` (src/lib.rs:6-60): both
routines evaluate the multilinear extension of `evals` at `r` with r[0] <-> index MSB
("BE").  On the GPU they are the same streaming evaluate; the two names are kept for API
parity (the reference's doc comments on them are swapped, :3-5 vs :26-28)."""
from . import _lib
from .dense_mle import DenseMultilinearExtension


def _evaluate_be(ctx, evals, r):
    n = len(r)
    if isinstance(evals, DenseMultilinearExtension):
        t = evals
    else:
        t = DenseMultilinearExtension.from_evaluations_vec(ctx, n, evals)
    return t.evaluate(r, order=_lib.ORDER_BE)


def vsbw_multilinear_from_evaluations(ctx, evals, r):
    """:6-24"""
    return _evaluate_be(ctx, evals, r)


def cti_multilinear_from_evaluations(ctx, evals, r):
    """:29-48"""
    return _evaluate_be(ctx, evals, r)
