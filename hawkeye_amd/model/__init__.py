"""Import side effect = registration, like the reference's model/__init__.py:1-2."""
from .registry import MODEL, BACKBONE  # noqa: F401
from .backbone import *  # noqa: F401,F403
from .methods import *  # noqa: F401,F403
