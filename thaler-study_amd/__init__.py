"""MI355X-native sumcheck prover hot path.

The directory is `thaler-study_amd/`; import it as `thaler_study_amd` through
`__graft_entry__.load_package()` / tests/conftest.py (a hyphen is not importable).
"""
from . import _lib  # noqa: F401
from ._lib import ORDER_BE, ORDER_LE, SumcheckHipError, build, load  # noqa: F401
from .field import GOLDILOCKS, Field  # noqa: F401
from .dense_mle import Context, DenseMultilinearExtension  # noqa: F401
from . import sum_check_protocol, matrix_multiplication, multilinear_extensions, distributed, gkr_protocol, triangle_counting, fiat_shamir, synthetic, schedule  # noqa: F401
