"""MI355X-native PINN Schroedinger-residual training path (H2+ ion).

Host side mirrors the reference's Python surface (poc/main.py, train.py); the
compute is the HIP library behind include/pinn.h.
"""
from . import capi, flat, formats  # noqa: F401
from .analysis import (calculate_E_R, dEdR_int, energy_from_psi, energy_from_psi_LCAO,  # noqa: F401
                       force_curves, psi_cut)
from .capi import PinnError  # noqa: F401
from .model import (NN_ion, V, exactE, radial, sampling, set_params)  # noqa: F401
from .parallel import DataParallelPlan  # noqa: F401
from .trainer import Trainer, saveLoss, train  # noqa: F401
