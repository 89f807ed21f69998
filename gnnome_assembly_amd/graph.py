"""AssemblyGraph: the graph argument of the model, a duck-type of the DGLGraph surface the
reference touches on the hot path:
``num_nodes() num_edges() edges() ndata edata to(device) device in_degrees() out_degrees()
local_scope()``.  ``edges()`` and every per-edge tensor at the model boundary are in the
caller's edge-id order (graph_parser.py:297 fixes it; inference.py:46-49,454 relies on it).

Internally the engine keeps all [E,H] tensors sorted by destination; the index
(gnm_graph_build_index, replaces DGL's lazy CSR/CSC build and dgl.reverse,
gated_gcn_full.py:115) is built once per graph on the host and cached on the device.
"""
from __future__ import annotations

import contextlib
import ctypes as C

import numpy as np
import torch

from . import _lib

__all__ = ["AssemblyGraph", "from_dgl"]

_INDEX_KEYS = ("perm", "isrc", "idst", "in_ptr", "out_ptr", "out_pos", "out_dst")


class AssemblyGraph:
    def __init__(self, src, dst, num_nodes=None):
        src = np.ascontiguousarray(_to_numpy(src), dtype=np.int32)
        dst = np.ascontiguousarray(_to_numpy(dst), dtype=np.int32)
        if src.shape != dst.shape or src.ndim != 1:
            raise ValueError("src and dst must be 1-D arrays of equal length")
        if num_nodes is None:
            num_nodes = int(max(src.max(initial=-1), dst.max(initial=-1)) + 1)
        self._n = int(num_nodes)
        self._src = src
        self._dst = dst
        self._host_index = None
        self._dev_index = {}      # device -> dict of int32 tensors
        self._dev_edges = {}      # device -> (src, dst) tensors
        self.device = torch.device("cpu")
        self.ndata = {}
        self.edata = {}

    # ---- DGLGraph surface ------------------------------------------------------------
    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src.size)

    number_of_nodes = num_nodes
    number_of_edges = num_edges

    def edges(self):
        if self.device not in self._dev_edges:
            self._dev_edges[self.device] = (torch.from_numpy(self._src).to(self.device),
                                            torch.from_numpy(self._dst).to(self.device))
        return self._dev_edges[self.device]

    def in_degrees(self):
        return torch.from_numpy(np.bincount(self._dst, minlength=self._n)).to(self.device)

    def out_degrees(self):
        return torch.from_numpy(np.bincount(self._src, minlength=self._n)).to(self.device)

    def to(self, device):
        """Like DGLGraph.to: returns a graph on `device` sharing the cached index; features move."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        g = AssemblyGraph.__new__(AssemblyGraph)
        g.__dict__.update(self.__dict__)
        g.device = device
        g.ndata = {k: v.to(device) for k, v in self.ndata.items()}
        g.edata = {k: v.to(device) for k, v in self.edata.items()}
        return g

    def int(self):
        return self

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata, self.edata = nd, ed

    # ---- engine side -----------------------------------------------------------------
    def host_index(self):
        """dict of int32 numpy arrays: perm isrc idst in_ptr out_ptr out_pos out_dst."""
        if self._host_index is None:
            lib = _lib.load()
            n, e = self._n, self.num_edges()
            idx = {
                "perm": np.empty(e, np.int32), "isrc": np.empty(e, np.int32), "idst": np.empty(e, np.int32),
                "in_ptr": np.empty(n + 1, np.int32), "out_ptr": np.empty(n + 1, np.int32),
                "out_pos": np.empty(e, np.int32), "out_dst": np.empty(e, np.int32),
            }
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            _lib.check(lib.gnm_graph_build_index(ptr(self._src), ptr(self._dst), n, e,
                                                 *[ptr(idx[k]) for k in _INDEX_KEYS]),
                       "gnm_graph_build_index")
            self._host_index = idx
        return self._host_index

    def index(self, device=None):
        """The index as int32 tensors on `device` (default: the graph's device)."""
        device = torch.device(device) if device is not None else self.device
        if device not in self._dev_index:
            h = self.host_index()
            self._dev_index[device] = {k: torch.from_numpy(v).to(device) for k, v in h.items()}
        return self._dev_index[device]


def _to_numpy(a):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def from_dgl(g):
    """Adapter for environments where DGL exists: copies structure and ndata/edata."""
    s, d = g.edges()
    ag = AssemblyGraph(s, d, g.num_nodes())
    ag.ndata = dict(g.ndata)
    ag.edata = dict(g.edata)
    return ag
