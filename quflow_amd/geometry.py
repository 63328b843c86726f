"""Host scalars of the hot path (quflow/geometry.py:7-9)."""
import numpy as np


def hbar(N):
    """hbar(N) = 2/sqrt(N^2-1)  (quflow/geometry.py:7-9)."""
    return 2.0 / np.sqrt(N ** 2 - 1)
