from .auto_model import AutoModelLM  # noqa: F401
