from .BCNN import BCNN, BilinearPooling  # noqa: F401
from .CBCNN import CBCNN, CompactBilinearPooling  # noqa: F401
from .MPNCOV import MPN, MPNCOV  # noqa: F401
from .APCNN import APCNN  # noqa: F401
from .OSME import OSMENet, OSME, OSME_block  # noqa: F401
from .CIN import CIN, CINClassifier, ChannelInteractionModule  # noqa: F401
