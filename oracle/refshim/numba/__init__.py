"""Stand-in for `numba`, used ONLY by oracle/gen_golden.py in the build container.

The reference's hot-path kernels are plain-Python loops decorated with
`@njit(...)`; numba is not installable here (no network).  This shim turns the
decorators into identities so the reference's *own* source runs under CPython
in strict IEEE-754 order (no fastmath reassociation, no FMA contraction).
It is test infrastructure: nothing in the product imports it.
"""


def _identity_decorator(*dargs, **dkwargs):
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


njit = jit = vectorize = guvectorize = _identity_decorator
prange = range
