from .MAMC_loss import MAMCLoss, NPairsLoss  # noqa: F401
from .CIN_loss import CINLoss  # noqa: F401
