"""COALA_GNN -- the reference's Python package surface (COALA-GNN-Setup/COALA_GNN/__init__.py:1-3) on the MI355X path."""
from .Shared_Tensor import *  # noqa: F401,F403
from .Training_node_distributor import *  # noqa: F401,F403
from .COALA_GNN_DataLoader import *  # noqa: F401,F403
