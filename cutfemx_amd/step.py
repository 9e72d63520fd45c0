"""Sync-free steps of a moving-domain loop (cfx_step_begin / cfx_step_end, include/cutfemx_amd.h).

The loop of python/demo/demo_moving_poisson.py:53-67 -- cut.update(), runtime rules, forms, create_matrix, assemble --
produces a dozen data-dependent sizes per time step.  Outside a step the engine reads each of them back where it is
produced; inside one it sizes buffers and grids by the same site's count in the previous step of the loop, leaves the
exact lengths in HBM for its kernels and fetches everything in ONE read-back when the step ends:

    for it in range(nsteps):
        phi.values[:] = ...                       # move the level set
        out = cutfemx_amd.run_step(lambda: one_step(phi), key="moving-poisson")

`run_step` repeats the body when a count did not fit the capacity taken from the previous step (`redo`: the results of
that pass are void; the repeat reads sizes back and always fits).  Sizes read from engine objects (`A.nnz`,
`rules.total_points`, ...) while a step is open are capacities; after it they are exact.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


class step:
    """Context manager around one step; `redo`, `published`, `read_back` are set on exit."""

    def __init__(self, key: str = "default"):
        self.key = str(key)
        self.redo = False
        self.published = 0      # sites whose count stayed in HBM
        self.read_back = 0      # sites read back at once (first step of a loop, or a site the engine does not defer)

    def __enter__(self):
        _lib.check(_lib.lib().cfx_step_begin(self.key.encode()))
        _lib._step_open = True
        return self

    def __exit__(self, exc_type, exc, tb):
        _lib._step_open = False
        if exc_type is not None and issubclass(exc_type, _lib.StepVoid):
            # a call met the void state of the step before its end: close it (redo = 1) and let the caller repeat
            redo = C.c_int()
            _lib.lib().cfx_step_end(C.byref(redo), None, None)
            self.redo = True
            return True
        if exc_type is not None:
            _lib.lib().cfx_step_abort()
            return False
        redo, pub, rb = C.c_int(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().cfx_step_end(C.byref(redo), C.byref(pub), C.byref(rb)))
        self.redo, self.published, self.read_back = bool(redo.value), pub.value, rb.value
        return False


def run_step(body, key: str = "default", max_passes: int = 3, info: dict | None = None):
    """Run `body()` as one sync-free step of the loop `key`; repeat it while the engine reports `redo`."""
    for attempt in range(max_passes):
        out = None
        with step(key) as s:
            out = body()
        if info is not None:
            info.update(passes=attempt + 1, published=s.published, read_back=s.read_back)
        if not s.redo:
            return out
        del out
    raise RuntimeError("cutfemx_amd.run_step: the step still does not fit its capacities after a sized repeat")


def set_margin(factor: float = 1.03125, slack: int = 256):
    """Capacity of a list = its count in the previous step x factor + slack."""
    _lib.check(_lib.lib().cfx_step_set_margin(C.c_double(factor), C.c_int64(slack)))


def forget(key: str | None = None):
    """Drop the size history of loop `key` (None: of all loops): its next step reads sizes back."""
    _lib.check(_lib.lib().cfx_step_forget(None if key is None else key.encode()))
