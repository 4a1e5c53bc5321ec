"""shivae_amd: the Shi-VAE train step as hand-written HIP kernels for MI355X (gfx950), behind the
reference's Python API (``ShiVAE``, ``Trainer``).  See DESIGN.md / INTEGRATION.md."""
from . import _native  # noqa: F401
from . import utils  # noqa: F401
from .shivae import ShiVAE, LOSS_KEYS  # noqa: F401
from .optim import FusedClipAdam  # noqa: F401
from .trainers import Trainer, KL_Annealing, LinearTempScheduler  # noqa: F401
from .scalers import HeterogeneousScaler  # noqa: F401
from . import datasets, metrics  # noqa: F401

__all__ = ["ShiVAE", "FusedClipAdam", "Trainer", "HeterogeneousScaler", "LOSS_KEYS", "utils"]
