"""Drop-in shim: with this repository on PYTHONPATH, `from models.model import MyModel` (ref/train.py:9)
resolves to the MI355X-native implementation."""
from klab_multimodalmodel_amd.models.model import MyModel  # noqa: F401
