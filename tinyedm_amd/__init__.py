"""tinyedm_amd -- MI355X-native EDM/EDM2 training + sampling hot path: hand-written HIP kernels
(tinyedm_amd/csrc, C-ABI in include/tinyedm_hip.h) behind the tinyedm Python API.  The ``tinyedm``
alias package re-exports these names so reference configs (``_target_: tinyedm.EDM``) resolve."""
__version__ = "0.1.0"

from . import _runtime_env  # noqa: F401  (first: sets HIP runtime flags before anything can initialise the GPU)
from .edm import EDM, Diffuser
from .ema import EMA, EMAOptimizer, FusedAdam, sigma_rel_to_gamma
from .metric import WeightedMeanSquaredError
from .networks import Conv2d, Denoiser, DenoiserWrapper, Embedding, Linear, manual_seed
from .solvers import DeterministicSolver
from .trainer import LightningModule, Trainer
from . import config, networks, ops

__all__ = ["EDM", "Diffuser", "DeterministicSolver", "WeightedMeanSquaredError", "Denoiser", "Linear", "Conv2d",
           "Embedding", "DenoiserWrapper", "EMA", "EMAOptimizer", "FusedAdam", "Trainer", "LightningModule",
           "sigma_rel_to_gamma", "manual_seed", "config", "networks", "ops"]
