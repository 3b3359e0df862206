from .repository import Repository  # noqa: F401
from .meters import (AverageMeter, PerformanceMeter, accuracy, Timer, set_random_seed,  # noqa: F401
                     TqdmHandler, ScalarWriter)
