from .model import get_model  # noqa: F401
