"""MI355X-native GatedGCN edge-logit engine (drop-in for lvrcek/GNNome-assembly's
models.GraphGatedGCNModel / layers.* forward-backward).  See DESIGN.md."""
from .graph import AssemblyGraph, from_dgl  # noqa: F401
from . import layers, models, synth, features, cluster, decode, io  # noqa: F401
from .models import GraphGatedGCNModel, BCEWithLogitsLoss  # noqa: F401

__version__ = "0.1.0"
