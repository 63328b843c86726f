"""quflow_amd -- MI355X-native implementation of quflow's isospectral hot path.

Drop-in for the reference's stepper and Laplacian-backend protocols
(SURVEY.md section 8b):

    import quflow_amd as qfa
    W = qfa.isomp(W, dt, steps=100, stats=stats)        # quflow.integrators.isomp
    P = qfa.solve_poisson(W); W2 = qfa.laplace(P)       # quflow.laplacian
    qfa.laplacian  -> module with solve_poisson / laplace / laplacian / select_skewherm
    qfa.IsompHIP(N, dtype), qfa.PoissonHIP(N, dtype)    # device-object form (runfile selection)

All compute runs in hand-written HIP kernels for gfx950 behind the C ABI of
include/quflow_hip.h; there is no CPU fallback.
"""
import os as _os

# The HIP runtime multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and gives a
# new stream the least-used one: with one trajectory alive, a DeviceEnsemble of four more put two of its
# replicas on ONE queue, where their kernels serialise (measured: sum rate of 4 replicas at N=512 1.43x
# instead of 1.78x the single-trajectory rate).  Eight queues keep up to seven concurrent trajectories
# apart.  Read by the runtime when it initialises: set before anything touches HIP; a user's own setting wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import laplacian
from . import integrators
from . import physics
from . import geometry
from . import ensemble
from . import quantization
from . import simulation
from .simulation import Simulation, solve, create_runfile
QuSimulation = Simulation          # the reference's name (quflow/simulation.py:60): scripts that say qf.QuSimulation run unchanged
from .quantization import (shr2mat, mat2shr, shc2mat, mat2shc, get_basis, compute_basis, basis_break_index, elm2ind, ind2elm,
                           berezin_multipliers)
from .geometry import hbar, bracket, norm_L2, inner_L2, norm_Linf, norm_L1, integral, qtime2seconds, seconds2qtime
from .laplacian import (solve_poisson, laplace, PoissonHIP, solve_heat, solve_helmholtz, solve_viscdamp,
                        solve_globalqg, ViscDampStep)
from .integrators import (isomp, isomp_fixedpoint, IsompHIP, DeviceTrajectory, DeviceEnsemble, euler, heun, rk4,
                          isomp_simple, isomp_quasinewton, magmp, magmp_fixedpoint, solve_mhd,
                          commutator, commutator_generic, commutator_skewherm, estimate_stepsize, project_skewherm)
from .physics import energy_euler, enstrophy, inner_Hm1, norm_Hm1, inner_H1, norm_H1
from .context import get_context, set_device, release_contexts, guard_report
from ._lib import QuflowHipError, device_count, device_info

__version__ = "0.1.0"
