"""The synthetic benchmark instance of BASELINE.md section 3, generated on
the device: a[i] = splitmix64(SEED_A + i) mod p, b[i] = splitmix64(SEED_B + i) mod p, and the
synthetic challenger r_j = splitmix64(SEED_R + j) mod p (inside sc_prove)."""
from .dense_mle import DenseMultilinearExtension

SEED_A = 0xA5A5000000000001
SEED_B = 0xB6B6000000000002
SEED_R = 0xC7C7000000000003
SEED_PT = 0xD8D8000000000004   # evaluation points of the single-table (config 2) runs


def tables(ctx, num_vars_local, start=0):
    """this rank's shard (2^num_vars_local entries from global index `start`) of a and b"""
    a = DenseMultilinearExtension.generate(ctx, SEED_A, num_vars_local, start=start)
    b = DenseMultilinearExtension.generate(ctx, SEED_B, num_vars_local, start=start)
    return a, b
