"""`from modules import *` surface of the reference (ref/modules/__init__.py:1-4)."""
from .config import parse_arguments
from .loader import get_dataloader
from .logger import get_logger
from .losses import LossCounter

__all__ = ["parse_arguments", "get_dataloader", "LossCounter", "get_logger"]
