from .MAMC_loss import MAMCLoss, NPairsLoss  # noqa: F401
