"""Scoped changes of the library's process-wide state.

The Linear scheduling knobs (``tad_linear_tuning``), the operand format (``set_precision``) and the q pre-scale contract of the
attention route are ONE copy per process: a change made for one model is seen by every other model and thread
of the process.  ``TuningScope`` makes such a change a scope: it records what it replaces (the library's own getter, not a shadow
table), applies the new values, and puts the old ones back on exit, also when the body raises.  Scopes nest.  While a scope is open
it holds a process-wide re-entrant lock, so a second THREAD that wants another plan waits instead of interleaving its launches with
the first one's plan (``lock=False`` for an owner that keeps its scope for its whole life, e.g. ``parallel.DataParallel``).

    with TuningScope(precision="half", persistent=0):
        loss = model_a(x).sum()          # half operands, one workgroup per tile
    model_b(x)                           # whatever was in force before
"""
from __future__ import annotations

import threading

from . import kernels as K

_LOCK = threading.RLock()


class TuningScope:
    def __init__(self, precision: str | None = None, attn_q_prescale: bool | None = None, lock: bool = True, **linear_knobs):
        unknown = [k for k in linear_knobs if k not in K.LINEAR_TUNING_DEFAULTS and k != "debug"]
        if unknown:
            raise ValueError(f"TuningScope: unknown tad_linear_tuning knob(s) {unknown}; known: {sorted(K.LINEAR_TUNING_DEFAULTS)}")
        self.precision, self.attn_q_prescale, self.knobs, self.lock = precision, attn_q_prescale, dict(linear_knobs), bool(lock)
        self._saved = None

    def __enter__(self):
        from . import ops
        if self._saved is not None:
            raise RuntimeError("TuningScope is not re-entrant: build a new one per `with`")
        if self.lock:
            _LOCK.acquire()
        saved = {"knobs": {}, "precision": None, "qpre": None}
        try:
            for k, v in self.knobs.items():
                saved["knobs"][k] = K.linear_tuning_get(k)
                K.linear_tuning(**{k: v})
            if self.precision is not None:
                saved["precision"] = ops.get_precision()
                ops.set_precision(self.precision)
            if self.attn_q_prescale is not None:
                saved["qpre"] = ops.get_attn_q_prescale()
                ops.set_attn_q_prescale(self.attn_q_prescale)
        except Exception:
            self._saved = saved
            self._restore()
            if self.lock:
                _LOCK.release()
            raise
        self._saved = saved
        return self

    def _restore(self):
        from . import ops
        saved, self._saved = self._saved, None
        if saved is None:
            return
        if saved["qpre"] is not None:
            ops.set_attn_q_prescale(saved["qpre"])
        if saved["precision"] is not None:
            ops.set_precision(saved["precision"])
        for k, v in reversed(list(saved["knobs"].items())):
            K.linear_tuning(**{k: v})

    def __exit__(self, *exc):
        try:
            self._restore()
        finally:
            if self.lock:
                _LOCK.release()
        return False

    close = lambda self: self.__exit__(None, None, None)  # noqa: E731  (for owners that open the scope without `with`)
