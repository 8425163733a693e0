"""Drop-in for the reference's ``pointnet2_ops`` package."""
from . import pointnet2_utils  # noqa: F401

__version__ = "3.0.0"
