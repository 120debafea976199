"""The prover's launch schedule without a device: `sc_plan_proof` runs the planner the engine itself runs at every pass
(pure host logic), so the schedule of any (num_vars, world, transport, options) can be inspected - and tested - on a
machine without a GPU."""
import ctypes

from . import _lib


def plan_proof(num_vars, world=1, transport="none", **options):
    """list of dicts (action, kf, ks, log_in, sharded), in launch order; options: the sc_ctx_set_option names the
    schedule depends on (vars_per_pass, first_pass_vars, grid_pass, grid_log, grid_max_vars, grid_sharded, tail_log,
    use_mailbox, gram_log, host_tail_log); unknown names raise"""
    lib = _lib.load()
    opt = _lib.ScPlanOptions()
    lib.sc_plan_options_init(ctypes.byref(opt), ctypes.sizeof(opt))
    for k, v in options.items():
        if k == "struct_size" or k not in dict(opt._fields_):
            raise KeyError("not a schedule option: %s" % k)
        setattr(opt, k, int(v))
    cap = 128
    steps = (_lib.ScPlanStep * cap)()
    n = ctypes.c_size_t()
    rc = lib.sc_plan_proof(ctypes.byref(opt), num_vars, world, _lib.TRANSPORTS[transport], steps, cap, ctypes.byref(n))
    if rc != _lib.SC_OK:
        raise _lib.SumcheckHipError(rc, "sc_plan_proof(num_vars=%d, world=%d, transport=%s, %r)" % (num_vars, world, transport, options))
    return [{"action": _lib.PLAN_ACTIONS[s.action], "kf": s.kf, "ks": s.ks, "log_in": s.log_in, "sharded": bool(s.sharded)}
            for s in steps[:n.value]]
