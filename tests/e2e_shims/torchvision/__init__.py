"""`import torchvision` stand-in (train.py:17): utils.save_image only (train.py:235, behind `is_save_images = False`)."""
from . import utils  # noqa: F401
