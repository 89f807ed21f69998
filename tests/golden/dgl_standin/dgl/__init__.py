"""Test-only stand-in for the five DGL builtins the reference hot path calls.

DGL is not installed in the build container (no wheel, no network) and has no ROCm
build.  This module exists ONLY so that tests/golden/make_golden.py can import the
reference's own layers/ and models/ packages unmodified from /root/reference and run
them on torch-CPU to generate golden vectors.  It never ships in the product path and
is never imported by gnnome_assembly_amd.

Semantics follow DGL's documented builtin definitions (call sites:
gated_gcn_full.py:115,120,128,129,133,141,142; score_predictor.py:21,24):
  apply_edges(u_add_v(a,b,o))      edata[o][k] = ndata[a][src k] + ndata[b][dst k]
  update_all(u_mul_e(a,b,m), sum)  ndata[o][v] = sum_{k: dst k = v} ndata[a][src k] * edata[b][k]
  update_all(copy_e(a,m), sum)     ndata[o][v] = sum_{k: dst k = v} edata[a][k]
  reverse(g)                       same edge ids, src and dst swapped, features shared
  apply_edges(python_udf)          udf(EdgeBatch with .src/.dst/.data) -> dict into edata
Only the floating-point summation order inside a destination is unspecified by DGL; this
stand-in uses index_add_ in edge-id order.
"""
import contextlib
import torch
from . import function  # noqa: F401


class _EdgeBatch:
    def __init__(self, g):
        self.src = {k: v.index_select(0, g._src) for k, v in g.ndata.items()}
        self.dst = {k: v.index_select(0, g._dst) for k, v in g.ndata.items()}
        self.data = g.edata


class DGLGraph:
    def __init__(self, src, dst, num_nodes):
        self._src = torch.as_tensor(src, dtype=torch.int64)
        self._dst = torch.as_tensor(dst, dtype=torch.int64)
        self._n = int(num_nodes)
        self.ndata = {}
        self.edata = {}

    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src.numel())

    def edges(self):
        return self._src, self._dst

    def to(self, device):
        return self

    def in_degrees(self):
        return torch.bincount(self._dst, minlength=self._n)

    def out_degrees(self):
        return torch.bincount(self._src, minlength=self._n)

    def adjacency_matrix(self, scipy_fmt="csr"):
        # A[src, dst] = multiplicity (utils.py:124 sums rows for the out-degree)
        import numpy as np
        from scipy import sparse as sp
        a = sp.coo_matrix((np.ones(self.num_edges()), (self._src.numpy(), self._dst.numpy())),
                          shape=(self._n, self._n))
        return a.asformat(scipy_fmt)

    @contextlib.contextmanager
    def local_scope(self):
        nd, ed = dict(self.ndata), dict(self.edata)
        try:
            yield
        finally:
            self.ndata, self.edata = nd, ed

    def apply_edges(self, f):
        if isinstance(f, function._Msg):
            if f.kind != "u_add_v":
                raise NotImplementedError(f.kind)
            self.edata[f.out] = (self.ndata[f.a].index_select(0, self._src)
                                 + self.ndata[f.b].index_select(0, self._dst))
        else:
            self.edata.update(f(_EdgeBatch(self)))

    def update_all(self, msg, red):
        if red.kind != "sum" or red.msg != msg.out:
            raise NotImplementedError((msg.kind, red.kind))
        if msg.kind == "u_mul_e":
            m = self.ndata[msg.a].index_select(0, self._src) * self.edata[msg.b]
        elif msg.kind == "copy_e":
            m = self.edata[msg.a]
        else:
            raise NotImplementedError(msg.kind)
        out = torch.zeros((self._n,) + tuple(m.shape[1:]), dtype=m.dtype)
        self.ndata[red.out] = out.index_add(0, self._dst, m)


def graph(data, num_nodes=None):
    src, dst = data
    return DGLGraph(src, dst, num_nodes)


def reverse(g, copy_ndata=True, copy_edata=False):
    r = DGLGraph(g._dst, g._src, g._n)
    if copy_ndata:
        r.ndata = dict(g.ndata)
    if copy_edata:
        r.edata = dict(g.edata)
    return r


def seed(val):
    """dgl.seed (utils.set_seed, utils.py:34): DGL's own RNG is not used by anything the stand-in covers."""
    return None
