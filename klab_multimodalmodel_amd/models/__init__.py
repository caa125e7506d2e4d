from .model import MyModel  # noqa: F401
