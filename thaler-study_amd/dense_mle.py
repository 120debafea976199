"""Device-resident `ark_poly::DenseMultilinearExtension<F>` and the GPU context.

Mirrors the methods the reference's hot path calls: from_evaluations_vec
(matrix-multiplication/src/lib.rs:81), relabel (:82), fix_variables (:83,:104), evaluate (:97),
to_evaluations (:138), num_vars (:88), Clone.  Variable 0 is index bit 0.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import SC_OK, SumcheckHipError, u64, u64p, voidp
from .field import Field


def _u64p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def _words(xs):
    return np.ascontiguousarray(np.array([int(x) for x in xs], dtype=np.uint64))


class Context:
    """one GPU, one stream, one field (sc_ctx).  Not thread-safe, like `&mut Prover`.

    devices=[d0, d1, ...] (a power of two of them, up to 8; entries may repeat): ONE handle over several GPUs of this
    process (sc_ctx_create_multi) - every table made on it is one table split over the devices by its top index bits,
    and Prover / evaluate / fix_variables work on the whole table with no launcher and no communicator."""

    def __init__(self, field, device=0, devices=None):
        self.lib = _lib.load()
        self.field = field if isinstance(field, Field) else Field(field)
        h = voidp()
        if devices is not None:
            devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
            rc = self.lib.sc_ctx_create_multi(self.field.ref(), devs, len(devices), ctypes.byref(h))
        else:
            rc = self.lib.sc_ctx_create(self.field.ref(), device, ctypes.byref(h))
        if rc != SC_OK:
            raise SumcheckHipError(rc, self.lib.sc_last_error(None).decode())
        self.h = h
        self._keep = []

    def check(self, rc):
        if rc != SC_OK:
            raise SumcheckHipError(rc, self.lib.sc_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.sc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self.check(self.lib.sc_ctx_set_option(self.h, key.encode(), int(value)))

    def get_option(self, key):
        v = ctypes.c_int64()
        self.check(self.lib.sc_ctx_get_option(self.h, key.encode(), ctypes.byref(v)))
        return v.value

    def synchronize(self):
        self.check(self.lib.sc_ctx_synchronize(self.h))

    def kernel_time(self, reset=True):
        out = (ctypes.c_double * 2)()
        self.check(self.lib.sc_ctx_kernel_time(self.h, out, 1 if reset else 0))
        return int(out[0]), float(out[1])

    def launch_log(self, reset=True, cap=65536):
        """per-launch records of the timed launches: list of dicts (kind, kf, ks, log_in, bytes_read,
        bytes_written, ms)"""
        buf = (_lib.ScLaunchRecord * cap)()
        n = ctypes.c_size_t()
        self.check(self.lib.sc_ctx_launch_log(self.h, buf, cap, ctypes.byref(n), 1 if reset else 0))
        return [{"kind": _lib.KIND_NAMES.get(r.kind, str(r.kind)), "kf": r.kf, "ks": r.ks, "log_in": r.log_in,
                 "bytes_read": int(r.bytes_read), "bytes_written": int(r.bytes_written), "ms": float(r.ms)}
                for r in buf[:min(n.value, cap)]]

    # ---- sharding -----------------------------------------------------------------------
    def comm_init_rccl(self, unique_id, rank, world):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self.check(self.lib.sc_ctx_comm_init_rccl(self.h, buf, rank, world))

    def comm_init_host(self, rank, world, allreduce, allgather):
        """allreduce(np.uint64 array) sums in place across ranks; allgather(send) -> concatenation"""
        def _ar(_user, buf, count):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                allreduce(a)
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        def _ag(_user, send, recv, count):
            try:
                s = np.ctypeslib.as_array(send, shape=(count,))
                r = np.ctypeslib.as_array(recv, shape=(count * world,))
                r[:] = allgather(s.copy())
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1

        ar, ag = _lib.ALLREDUCE_FN(_ar), _lib.ALLGATHER_FN(_ag)
        self._keep += [ar, ag]
        self.check(self.lib.sc_ctx_comm_init_host(self.h, rank, world, ar, ag, None))

    def comm_peer_export(self, rank, world):
        """allocate this rank's peer region; returns its 64-byte HIP IPC handle"""
        buf = (ctypes.c_uint8 * 64)()
        self.check(self.lib.sc_ctx_comm_peer_export(self.h, rank, world, buf))
        return bytes(buf)

    def comm_peer_connect(self, handles):
        """handles: the world 64-byte handles in rank order (other processes' regions are mapped over IPC)"""
        world = self.rank_world()[1]
        if len(handles) != world or any(len(bytes(h)) != 64 for h in handles):   # the C side reads 64 * world bytes
            raise ValueError("comm_peer_connect: expected %d handles of 64 bytes, got %d" % (world, len(handles)))
        blob = b"".join(bytes(h) for h in handles)
        buf = (ctypes.c_uint8 * len(blob)).from_buffer_copy(blob)
        self.check(self.lib.sc_ctx_comm_peer_connect(self.h, buf))

    def comm_peer_connect_local(self, peers):
        """peers: the world Context objects of this process, in rank order"""
        if len(peers) != self.rank_world()[1]:   # the C side reads world pointers
            raise ValueError("comm_peer_connect_local: expected %d contexts, got %d" % (self.rank_world()[1], len(peers)))
        arr = (voidp * len(peers))(*[p.h for p in peers])
        self.check(self.lib.sc_ctx_comm_peer_connect_local(self.h, arr))

    def rank_world(self):
        r, w = ctypes.c_int(), ctypes.c_int()
        self.check(self.lib.sc_ctx_comm_rank(self.h, ctypes.byref(r), ctypes.byref(w)))
        return r.value, w.value

    @staticmethod
    def rccl_unique_id():
        lib = _lib.load()
        buf = (ctypes.c_uint8 * 128)()
        rc = lib.sc_comm_unique_id(buf)
        if rc != SC_OK:
            raise SumcheckHipError(rc, lib.sc_last_error(None).decode())
        return bytes(buf)


class DenseMultilinearExtension:
    """evaluation table of a multilinear polynomial, resident in HBM (sc_table)"""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.h = handle

    # constructors ------------------------------------------------------------------------
    @classmethod
    def from_evaluations_vec(cls, ctx, num_vars, evaluations):
        ev = np.ascontiguousarray(np.asarray(evaluations, dtype=np.uint64))
        if ev.size != 1 << num_vars:
            raise ValueError("The size of evaluations should be 2^num_vars.")
        h = voidp()
        ctx.check(ctx.lib.sc_table_upload(ctx.h, _u64p(ev), ev.size, ctypes.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx, device_ptr, num_vars, keep=None):
        """a table over device memory the caller owns (borrowed; `keep`: whatever must stay alive with it, e.g. a torch tensor)"""
        h = voidp()
        ctx.check(ctx.lib.sc_table_from_device(ctx.h, voidp(int(device_ptr)), 1 << num_vars, ctypes.byref(h)))
        t = cls(ctx, h)
        t._keep = keep
        return t

    @classmethod
    def generate(cls, ctx, seed, num_vars, start=0):
        h = voidp()
        ctx.check(ctx.lib.sc_table_generate(ctx.h, seed, start, 1 << num_vars, ctypes.byref(h)))
        return cls(ctx, h)

    def clone(self):
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_table_clone(self.ctx.h, self.h, ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h)

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                self.ctx.lib.sc_table_free(self.ctx.h, self.h)
                self.h = None
        except Exception:
            pass

    # accessors ---------------------------------------------------------------------------
    def __len__(self):
        return int(self.ctx.lib.sc_table_len(self.h))

    def num_vars(self):
        return len(self).bit_length() - 1

    def to_evaluations(self):
        out = np.empty(len(self), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.sc_table_download(self.ctx.h, self.h, _u64p(out), out.size))
        return out

    # operations --------------------------------------------------------------------------
    def fix_variables(self, partial_point, order=_lib.ORDER_LE):
        r = _words(partial_point)
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_table_fix_variables(self.ctx.h, self.h, _u64p(r), r.size, order,
                                                          ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h)

    def evaluate(self, point, order=_lib.ORDER_LE):
        r = _words(point)
        out = u64()
        self.ctx.check(self.ctx.lib.sc_table_evaluate(self.ctx.h, self.h, _u64p(r), r.size, order,
                                                     ctypes.byref(out)))
        return int(out.value)

    def evaluate_many(self, points, order=_lib.ORDER_LE):
        """evaluate at every point of `points` (m rows of num_vars words) in one pass over the table"""
        pts = np.ascontiguousarray(np.array([[int(x) for x in row] for row in points], dtype=np.uint64))
        m = pts.shape[0]
        n = pts.shape[1] if m else 0
        out = np.zeros(max(m, 1), dtype=np.uint64)
        self.ctx.check(self.ctx.lib.sc_table_evaluate_many(self.ctx.h, self.h, _u64p(pts.reshape(-1)) if pts.size else None, m, n, order,
                                                          _u64p(out)))
        return [int(x) for x in out[:m]]

    def relabel(self, a, b, k):
        h = voidp()
        self.ctx.check(self.ctx.lib.sc_table_relabel(self.ctx.h, self.h, a, b, k, ctypes.byref(h)))
        return DenseMultilinearExtension(self.ctx, h)
