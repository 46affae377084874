"""MI355X-native GP posterior + acquisition hot path: GP fit -> posterior mean / variance -> acquisition -> argmax over large
candidate sets, behind the notebooks' own function signatures (notebook_api) and a C ABI (include/gpbo.h)."""
from ._lib import (ACQ_EI, ACQ_PI, ACQ_UCB, INCUMBENT_GIVEN, INCUMBENT_MAX_MU, INCUMBENT_MAX_Y, KERNEL_MATERN52,
                   KERNEL_RBF, VARIANT_M3, VARIANT_SINGLE, VARIANT_V4)
from .gp import GP, GpboError, MultiGP, device_count, make_gp

__all__ = ['GP', 'MultiGP', 'make_gp', 'device_count', 'GpboError', 'KERNEL_RBF', 'KERNEL_MATERN52', 'VARIANT_V4', 'VARIANT_SINGLE', 'VARIANT_M3', 'ACQ_EI', 'ACQ_UCB',
           'ACQ_PI', 'INCUMBENT_MAX_MU', 'INCUMBENT_MAX_Y', 'INCUMBENT_GIVEN']
