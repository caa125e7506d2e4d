"""MI355X-native Swin-V2 -> T5 caption-training hot path (drop-in for the reference's models/model.py)."""
__all__ = ["MyModel"]


def __getattr__(name):
    if name == "MyModel":
        from .models.model import MyModel
        return MyModel
    raise AttributeError(name)
