"""
pygho_amd -- MI355X-native backend for PygHO's sparse / masked high-order-GNN operator path.

Same public surface as the reference package root (``pygho/__init__.py:1-2``).
"""
from .backend.SpTensor import SparseTensor  # noqa: F401
from .backend.MaTensor import MaskedTensor  # noqa: F401

__version__ = "0.1.0"
