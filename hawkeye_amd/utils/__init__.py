from .repository import Repository  # noqa: F401
