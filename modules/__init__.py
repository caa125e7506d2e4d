"""Drop-in shim for `from modules import *` (ref/train.py:8)."""
from klab_multimodalmodel_amd.modules import *  # noqa: F401,F403
from klab_multimodalmodel_amd.modules import __all__  # noqa: F401
