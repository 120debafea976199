"""ctypes binding of libsumcheck_hip.so (C ABI: include/sumcheck_hip.h).

The library is built in-tree by `make -C thaler-study_amd/csrc` (hipcc, --offload-arch=gfx950)
and must be present: there is no Python or CPU fallback for any computing call.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# SUMCHECK_HIP_LIB: another build of the same ABI (A/B measurements: tools/generic_ab.sh); default: the in-tree library
LIB_PATH = os.environ.get("SUMCHECK_HIP_LIB") or os.path.join(_HERE, "libsumcheck_hip.so")
CSRC = os.path.join(_HERE, "csrc")

u64 = ctypes.c_uint64
u64p = ctypes.POINTER(ctypes.c_uint64)
voidp = ctypes.c_void_p
size_t = ctypes.c_size_t

SC_OK = 0
ERR_NAMES = {1: "SC_ERR_ARG", 2: "SC_ERR_HIP", 3: "SC_ERR_RCCL", 4: "SC_ERR_OOM", 5: "SC_ERR_STATE",
             6: "SC_ERR_UNSUPPORTED"}
ORDER_LE, ORDER_BE = 0, 1


class ScField(ctypes.Structure):
    _fields_ = [("p", u64), ("p_inv_neg", u64), ("r_mod_p", u64), ("r2_mod_p", u64)]


ABI_VERSION = 6   # SC_ABI_VERSION of include/sumcheck_hip.h as this binding was written


class ScPlanOptions(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32)] + [
        (k, ctypes.c_int32) for k in ("vars_per_pass", "first_pass_vars", "grid_pass", "grid_log", "grid_max_vars", "grid_sharded",
                                      "tail_log", "use_mailbox", "gram_log", "host_tail_log", "wfold_log", "wfold_min_log", "wfold_always", "wfold5_min_log")]


class ScPlanStep(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("action", "kf", "ks", "log_in", "sharded")]


PLAN_ACTIONS = {0: "pass", 1: "grid_pass", 2: "rank_pass", 3: "gather", 4: "host_tail", 5: "gram_pass", 6: "wfold_pass"}
TRANSPORTS = {"none": 0, "rccl": 1, "host": 2, "peer": 3, "local": 4}


class ScLaunchRecord(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("kf", ctypes.c_int32), ("ks", ctypes.c_int32), ("log_in", ctypes.c_int32),
                ("bytes_read", u64), ("bytes_written", u64), ("ms", ctypes.c_double)]


KIND_NAMES = {0: "pass", 2: "evaluate", 3: "fold", 4: "fix_low", 5: "fold_be", 6: "coldot", 7: "gkr", 8: "matsq", 10: "grid_pass", 11: "gram_pass", 13: "wfold_pass"}

ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, voidp, u64p, size_t)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, voidp, u64p, u64p, size_t)
DRAW_FN = ctypes.CFUNCTYPE(u64, voidp, size_t, u64p)

# name -> (restype, argtypes); every symbol include/sumcheck_hip.h declares
SIGNATURES = {
    "sc_field_from_modulus": (ctypes.c_int, [u64, ctypes.POINTER(ScField)]),
    "sc_field_to_mont": (u64, [ctypes.POINTER(ScField), u64]),
    "sc_field_from_mont": (u64, [ctypes.POINTER(ScField), u64]),
    "sc_interpolate_quadratic": (ctypes.c_int, [ctypes.POINTER(ScField), u64p, u64p]),
    "sc_ctx_create": (ctypes.c_int, [ctypes.POINTER(ScField), ctypes.c_int, ctypes.POINTER(voidp)]),
    "sc_ctx_create_multi": (ctypes.c_int, [ctypes.POINTER(ScField), ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.POINTER(voidp)]),
    "sc_ctx_destroy": (ctypes.c_int, [voidp]),
    "sc_last_error": (ctypes.c_char_p, [voidp]),
    "sc_ctx_set_option": (ctypes.c_int, [voidp, ctypes.c_char_p, ctypes.c_int64]),
    "sc_ctx_get_option": (ctypes.c_int, [voidp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int64)]),
    "sc_ctx_synchronize": (ctypes.c_int, [voidp]),
    "sc_ctx_stream": (voidp, [voidp]),
    "sc_ctx_kernel_time": (ctypes.c_int, [voidp, ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    "sc_ctx_launch_log": (ctypes.c_int, [voidp, ctypes.POINTER(ScLaunchRecord), size_t, ctypes.POINTER(size_t), ctypes.c_int]),
    "sc_comm_unique_id": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint8)]),
    "sc_ctx_comm_init_rccl": (ctypes.c_int, [voidp, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int, ctypes.c_int]),
    "sc_ctx_comm_init_host": (ctypes.c_int, [voidp, ctypes.c_int, ctypes.c_int, ALLREDUCE_FN, ALLGATHER_FN, voidp]),
    "sc_ctx_comm_peer_export": (ctypes.c_int, [voidp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)]),
    "sc_ctx_comm_peer_connect": (ctypes.c_int, [voidp, ctypes.POINTER(ctypes.c_uint8)]),
    "sc_ctx_comm_peer_connect_local": (ctypes.c_int, [voidp, ctypes.POINTER(voidp)]),
    "sc_ctx_comm_rank": (ctypes.c_int, [voidp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "sc_table_upload": (ctypes.c_int, [voidp, u64p, size_t, ctypes.POINTER(voidp)]),
    "sc_table_from_device": (ctypes.c_int, [voidp, voidp, size_t, ctypes.POINTER(voidp)]),
    "sc_table_generate": (ctypes.c_int, [voidp, u64, u64, size_t, ctypes.POINTER(voidp)]),
    "sc_table_clone": (ctypes.c_int, [voidp, voidp, ctypes.POINTER(voidp)]),
    "sc_table_download": (ctypes.c_int, [voidp, voidp, u64p, size_t]),
    "sc_table_len": (size_t, [voidp]),
    "sc_table_device_ptr": (voidp, [voidp]),
    "sc_table_free": (ctypes.c_int, [voidp, voidp]),
    "sc_table_fix_variables": (ctypes.c_int, [voidp, voidp, u64p, size_t, ctypes.c_int, ctypes.POINTER(voidp)]),
    "sc_table_evaluate": (ctypes.c_int, [voidp, voidp, u64p, size_t, ctypes.c_int, u64p]),
    "sc_table_evaluate_many": (ctypes.c_int, [voidp, voidp, u64p, size_t, size_t, ctypes.c_int, u64p]),
    "sc_table_relabel": (ctypes.c_int, [voidp, voidp, size_t, size_t, size_t, ctypes.POINTER(voidp)]),
    "sc_matmul_g_new": (ctypes.c_int, [voidp, voidp, voidp, size_t, u64p, ctypes.POINTER(voidp), ctypes.POINTER(voidp)]),
    "sc_prod2_to_evaluations": (ctypes.c_int, [voidp, voidp, voidp, ctypes.POINTER(voidp)]),
    "sc_prod2_sum": (ctypes.c_int, [voidp, voidp, voidp, u64p]),
    "sc_prod2_round_sums": (ctypes.c_int, [voidp, voidp, voidp, u64p]),
    "sc_prod2_fold_and_sums": (ctypes.c_int, [voidp, voidp, voidp, u64p, ctypes.POINTER(voidp), ctypes.POINTER(voidp), u64p]),
    "sc_prod2_evaluate": (ctypes.c_int, [voidp, voidp, voidp, u64p, size_t, u64p]),
    "sc_abi_version": (ctypes.c_int, []),
    "sc_plan_options_init": (None, [ctypes.POINTER(ScPlanOptions), size_t]),
    "sc_plan_proof": (ctypes.c_int, [ctypes.POINTER(ScPlanOptions), size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ScPlanStep), size_t,
                                     ctypes.POINTER(size_t)]),
    "sc_prover_create": (ctypes.c_int, [voidp, voidp, voidp, ctypes.POINTER(voidp)]),
    "sc_prover_c1": (ctypes.c_int, [voidp, u64p]),
    "sc_prover_num_vars": (ctypes.c_int, [voidp, ctypes.POINTER(size_t)]),
    "sc_prover_round": (ctypes.c_int, [voidp, u64, size_t, u64p]),
    "sc_prover_destroy": (ctypes.c_int, [voidp]),
    "sc_prove": (ctypes.c_int, [voidp, voidp, voidp, DRAW_FN, voidp, u64, u64p, u64p, u64p]),
    "sc_gkr_wiring": (ctypes.c_int, [voidp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint32),
                                      ctypes.POINTER(ctypes.c_uint32), size_t, size_t, u64p, ctypes.POINTER(voidp),
                                      ctypes.POINTER(voidp)]),
    "sc_gkr_w_to_evaluations": (ctypes.c_int, [voidp, voidp, voidp, voidp, voidp, ctypes.POINTER(voidp)]),
    "sc_gkr_w_round_sums": (ctypes.c_int, [voidp, voidp, voidp, voidp, voidp, u64p]),
    "sc_gkr_w_fix_variables": (ctypes.c_int, [voidp, voidp, voidp, voidp, voidp, u64p, size_t] + [ctypes.POINTER(voidp)] * 4),
    "sc_gkr_w_evaluate": (ctypes.c_int, [voidp, voidp, voidp, voidp, voidp, u64p, size_t, u64p]),
    "sc_gkr_prover_create": (ctypes.c_int, [voidp, voidp, voidp, voidp, voidp, ctypes.POINTER(voidp)]),
    "sc_gkr_prover_create_sparse": (ctypes.c_int, [voidp, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint32),
                                                    ctypes.POINTER(ctypes.c_uint32), size_t, size_t, u64p, voidp,
                                                    ctypes.POINTER(voidp)]),
    "sc_gkr_prove": (ctypes.c_int, [voidp, voidp, voidp, voidp, voidp, DRAW_FN, voidp, u64, u64p, u64p, u64p]),
    "sc_gkr_prover_c1": (ctypes.c_int, [voidp, u64p]),
    "sc_gkr_prover_round": (ctypes.c_int, [voidp, u64, size_t, u64p]),
    "sc_gkr_prover_destroy": (ctypes.c_int, [voidp]),
    "sc_table_restrict_to_line": (ctypes.c_int, [voidp, voidp, u64p, u64p, size_t, u64p]),
    "sc_tri_to_evaluations": (ctypes.c_int, [voidp, voidp, voidp, voidp, size_t, ctypes.POINTER(voidp)]),
    "sc_tri_round_sums": (ctypes.c_int, [voidp, voidp, voidp, voidp, size_t, u64p]),
    "sc_tri_fix_variables": (ctypes.c_int, [voidp, voidp, voidp, voidp, size_t, u64p, size_t] + [ctypes.POINTER(voidp)] * 3),
    "sc_tri_evaluate": (ctypes.c_int, [voidp, voidp, voidp, voidp, size_t, u64p, size_t, u64p]),
    "sc_tri_prover_create": (ctypes.c_int, [voidp, voidp, size_t, ctypes.POINTER(voidp)]),
    "sc_tri_prove": (ctypes.c_int, [voidp, voidp, size_t, DRAW_FN, voidp, u64, u64p, u64p, u64p]),
    "sc_tri_prover_c1": (ctypes.c_int, [voidp, u64p]),
    "sc_tri_prover_round": (ctypes.c_int, [voidp, u64, size_t, u64p]),
    "sc_tri_prover_destroy": (ctypes.c_int, [voidp]),
}


class SumcheckHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, "status %d" % code), msg))
        self.code = code


def build(force=False):
    """compile libsumcheck_hip.so for gfx950 (hipcc cross-compiles without a GPU)"""
    srcs = [os.path.join(CSRC, f) for f in ("sumcheck_hip.hip", "kernels.hpp", "field.hpp")]
    for sub, ext in (("engine", ".inc"), ("kernels", ".hpp")):
        srcs += [os.path.join(CSRC, sub, f) for f in sorted(os.listdir(os.path.join(CSRC, sub))) if f.endswith(ext)]
    srcs.append(os.path.join(_HERE, "..", "include", "sumcheck_hip.h"))
    stale = (not os.path.exists(LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs if os.path.exists(s)))
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load():
    """dlopen the product library; raises if it is missing (never falls back to anything)"""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: run `make -C thaler-study_amd/csrc` (or __graft_entry__.build()); "
                          "there is no CPU fallback" % LIB_PATH)
    # torch bundles its own libamdhip64.so.7; import it first so that one HIP runtime (and
    # one RCCL) serves the whole process whenever torch is part of it.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    if not hasattr(lib, "sc_abi_version"):      # (a library older than the version check itself)
        raise ImportError("%s exports no sc_abi_version: it predates this binding (ABI %d); rebuild (`make -C thaler-study_amd/csrc`)"
                          % (LIB_PATH, ABI_VERSION))
    lib.sc_abi_version.restype = ctypes.c_int
    if lib.sc_abi_version() != ABI_VERSION:
        raise ImportError("%s speaks ABI version %d, this binding %d: rebuild (`make -C thaler-study_amd/csrc`)"
                          % (LIB_PATH, lib.sc_abi_version(), ABI_VERSION))
    missing = [name for name in SIGNATURES if not hasattr(lib, name)]
    if missing:
        raise ImportError("%s lacks %s: rebuild (`make -C thaler-study_amd/csrc`)" % (LIB_PATH, ", ".join(missing)))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
