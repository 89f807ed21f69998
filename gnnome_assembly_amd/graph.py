"""AssemblyGraph: the graph argument of the model, a duck-type of the DGLGraph surface the
reference touches on the hot path:
``num_nodes() num_edges() edges() ndata edata to(device) device in_degrees() out_degrees()
local_scope()``.  ``edges()`` and every per-edge tensor at the model boundary are in the
caller's edge-id order (graph_parser.py:297 fixes it; inference.py:46-49,454 relies on it).

Internally the engine keeps all [E,H] tensors sorted by destination; the index
(gnm_graph_build_index, replaces DGL's lazy CSR/CSC build and dgl.reverse,
gated_gcn_full.py:115) is built once per graph on the host and cached on the device.

Node numbering.  The gather kernels are fast when node ids follow the genome (the node rows of the edges
in flight then sit in the 4 MB per-XCD L2).  The reference's graphs do NOT come that way: reads keep the
simulator's / sequencer's order (pipeline.py:46-61,160-169) and graph_parser.py:297-304 numbers nodes by read
id.  So the index is built over an INTERNAL node numbering (gnm_graph_locality_order: breadth-first over the
triangle-supported overlaps) whenever the caller's numbering is not already local; index()['nperm'] (internal
-> caller id) / ['nrank'] (caller -> internal) exist exactly then.  Nothing node-shaped leaves the engine in
the internal numbering: the model gathers `pe` rows in, the stand-alone layers permute h in and out.
`node_order`: 'auto' (default; GNM_NODE_ORDER overrides), 'keep' (trust the caller), 'bfs' (always renumber).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import collections
import os
import time

import numpy as np
import torch

from . import _lib

__all__ = ["AssemblyGraph", "from_dgl", "as_assembly_graph"]

_INDEX_KEYS = ("perm", "isrc", "idst", "in_ptr", "out_ptr", "out_pos", "out_dst")

NODE_ORDER = os.environ.get("GNM_NODE_ORDER", "auto").strip().lower()
# 'auto' keeps the caller's numbering when at least LOCAL_FRAC of the edges join nodes whose ids differ by at most
# LOCAL_WINDOW (2048 rows x 512 B = 1 MB of every gathered [N,128] tensor: inside one XCD's L2 with room to spare)
LOCAL_WINDOW = 2048
LOCAL_FRAC = 0.9


def set_node_order(mode: str) -> None:
    """Default node-order policy of graphs built from now on: 'auto' | 'keep' | 'bfs'."""
    global NODE_ORDER
    if mode not in ("auto", "keep", "bfs"):
        raise ValueError(f"node order {mode!r}: expected 'auto', 'keep' or 'bfs'")
    NODE_ORDER = mode


def tensor_index(src: torch.Tensor, dst: torch.Tensor, n: int, nrank: torch.Tensor = None):
    """The same seven index arrays as gnm_graph_build_index, built with tensor ops on the device the edge
    list lives on (two stable sorts, two bincounts): for graphs that are born on the GPU, e.g. the induced
    sub-graphs of the mini-batch mode, so that no edge list crosses PCIe.  Bit-identical to the host
    builder (tests/test_host_cpu.py::test_tensor_index_equals_host_index).  With `nrank` (caller node id ->
    internal node id) the index is built over the internal numbering and carries nperm / nrank."""
    s64, d64 = src.long(), dst.long()
    if nrank is not None:
        r64 = nrank.long()
        s64, d64 = r64[s64], r64[d64]
    perm = torch.sort(d64, stable=True).indices                       # stable by destination
    isrc, idst = s64[perm], d64[perm]
    zero = torch.zeros(1, dtype=torch.int64, device=src.device)
    in_ptr = torch.cat((zero, torch.cumsum(torch.bincount(d64, minlength=n), 0)))
    out_pos = torch.sort(isrc, stable=True).indices                   # by source, ascending internal position
    out_ptr = torch.cat((zero, torch.cumsum(torch.bincount(s64, minlength=n), 0)))
    i32 = lambda t: t.to(torch.int32).contiguous()  # noqa: E731
    idx = {"perm": i32(perm), "isrc": i32(isrc), "idst": i32(idst), "in_ptr": i32(in_ptr), "out_ptr": i32(out_ptr),
           "out_pos": i32(out_pos), "out_dst": i32(idst[out_pos])}
    if nrank is not None:
        idx["nrank"] = i32(nrank)
        idx["nperm"] = i32(torch.argsort(nrank.long()))
    return idx


class AssemblyGraph:
    @classmethod
    def from_tensors(cls, src: torch.Tensor, dst: torch.Tensor, num_nodes: int, nrank: torch.Tensor = None) -> "AssemblyGraph":
        """A graph whose edge list already lives on a device: edges and index stay there (tensor_index);
        the host copy is only made if something asks for it.  `nrank` (optional, [num_nodes] on the same device):
        the internal node numbering to use (caller id -> internal id), e.g. the parent graph's order restricted
        to a sub-graph; without it the caller's numbering is kept (no host round trip for a breadth-first sweep)."""
        if src.shape != dst.shape or src.dim() != 1:
            raise ValueError("src and dst must be 1-D tensors of equal length")
        g = cls.__new__(cls)
        g._n = int(num_nodes)
        g._src_t, g._dst_t = src.to(torch.int32).contiguous(), dst.to(torch.int32).contiguous()
        g._src_np = g._dst_np = None
        g._host_index = None
        g._dev_index = {}
        g._plans = {}
        g._dev_edges = {src.device: (g._src_t, g._dst_t)}
        g.device = src.device
        g.ndata = {}
        g.edata = {}
        g._node_order_mode = "keep"
        g._nrank_t = nrank
        g.relabel_info = {"mode": "given" if nrank is not None else "keep", "relabelled": nrank is not None}
        return g

    @property
    def _src(self):
        if self._src_np is None:
            self._src_np = np.ascontiguousarray(self._src_t.cpu().numpy(), dtype=np.int32)
        return self._src_np

    @property
    def _dst(self):
        if self._dst_np is None:
            self._dst_np = np.ascontiguousarray(self._dst_t.cpu().numpy(), dtype=np.int32)
        return self._dst_np

    def __init__(self, src, dst, num_nodes=None, node_order=None):
        if node_order not in (None, "auto", "keep", "bfs"):
            raise ValueError(f"node_order {node_order!r}: expected None, 'auto', 'keep' or 'bfs'")
        src = np.ascontiguousarray(_to_numpy(src), dtype=np.int32)
        dst = np.ascontiguousarray(_to_numpy(dst), dtype=np.int32)
        if src.shape != dst.shape or src.ndim != 1:
            raise ValueError("src and dst must be 1-D arrays of equal length")
        if num_nodes is None:
            num_nodes = int(max(src.max(initial=-1), dst.max(initial=-1)) + 1)
        self._n = int(num_nodes)
        self._src_np = src
        self._dst_np = dst
        self._src_t = self._dst_t = None
        self._host_index = None
        self._dev_index = {}      # device -> dict of int32 tensors
        self._plans = {}          # device -> sweep plan (or None)
        self._dev_edges = {}      # device -> (src, dst) tensors
        self.device = torch.device("cpu")
        self.ndata = {}
        self.edata = {}
        self._node_order_mode = node_order      # None: the module default at the time the index is built
        self._nrank_t = None
        self.relabel_info = {}             # filled by host_index(): what was decided, on what evidence, how long it took

    # ---- DGLGraph surface ------------------------------------------------------------
    def num_nodes(self):
        return self._n

    def num_edges(self):
        return int(self._src_np.size if self._src_np is not None else self._src_t.numel())

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
            ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
            src, dst = self._src, self._dst
            mode = self._node_order_mode or NODE_ORDER
            info = {"mode": mode, "relabelled": False}
            order = rank = None
            if self._nrank_t is not None:
                # born on a device with a GIVEN internal numbering (from_tensors(..., nrank): e.g. the parent's order of a
                # mini-batch sub-graph): the host index follows it too, so that every device's copy numbers the nodes alike
                rank = np.ascontiguousarray(self._nrank_t.cpu().numpy(), dtype=np.int32)
                order = np.ascontiguousarray(np.argsort(rank, kind="stable").astype(np.int32))
                src, dst = np.ascontiguousarray(rank[src]), np.ascontiguousarray(rank[dst])
                info = {"mode": "given", "relabelled": True}
                relabel = False
            elif mode == "auto":
                frac = C.c_double(1.0)
                _lib.check(lib.gnm_graph_edge_locality(ptr(src), ptr(dst), n, e, LOCAL_WINDOW, C.byref(frac)),
                           "gnm_graph_edge_locality")
                info["local_edge_fraction"] = frac.value
                relabel = n > LOCAL_WINDOW and frac.value < LOCAL_FRAC
            else:
                relabel = mode == "bfs"
            if relabel:
                t0 = time.perf_counter()
                order, rank = np.empty(n, np.int32), np.empty(n, np.int32)
                core = C.c_double(0.0)
                _lib.check(lib.gnm_graph_locality_order(ptr(src), ptr(dst), n, e, ptr(order), ptr(rank), C.byref(core)),
                           "gnm_graph_locality_order")
                src, dst = np.ascontiguousarray(rank[src]), np.ascontiguousarray(rank[dst])
                frac = C.c_double(1.0)
                lib.gnm_graph_edge_locality(ptr(src), ptr(dst), n, e, LOCAL_WINDOW, C.byref(frac))
                info.update(relabelled=True, seconds=time.perf_counter() - t0, triangle_edge_fraction=core.value,
                            local_edge_fraction_after=frac.value)
            self.relabel_info.clear()          # in place: the .to(device) copies share this dict
            self.relabel_info.update(info)
            idx = {
                "perm": np.empty(e, np.int32), "isrc": np.empty(e, np.int32), "idst": np.empty(e, np.int32),
                "in_ptr": np.empty(n + 1, np.int32), "out_ptr": np.empty(n + 1, np.int32),
                "out_pos": np.empty(e, np.int32), "out_dst": np.empty(e, np.int32),
            }
            _lib.check(lib.gnm_graph_build_index(ptr(src), ptr(dst), n, e, *[ptr(idx[k]) for k in _INDEX_KEYS]),
                       "gnm_graph_build_index")
            if order is not None:
                idx["nperm"], idx["nrank"] = order, rank
            self._host_index = idx
        return self._host_index

    def index(self, device=None):
        """The index as int32 tensors on `device` (default: the graph's device)."""
        device = torch.device(device) if device is not None else self.device
        if device not in self._dev_index:
            if self._src_t is not None and self._host_index is None:       # born on a device: build it there
                idx = tensor_index(self._src_t, self._dst_t, self._n, self._nrank_t)
                self._dev_index[self._src_t.device] = idx
                if device != self._src_t.device:
                    self._dev_index[device] = {k: v.to(device) for k, v in idx.items()}
            else:
                h = self.host_index()
                self._dev_index[device] = {k: torch.from_numpy(v).to(device) for k, v in h.items()}
        return self._dev_index[device]


    def sweep_plan(self, device=None, wg_per_cu: int = 1):
        """The sweep plan of this graph on `device` (gnm_graph_build_sweep_plan over the partition the sweep kernel with
        `wg_per_cu` workgroups per CU uses there -- 1: the chained backward, 2: the two-sided forward gate): dict(sinfo, dinfo [E] int32 tensors holding the plan words, fix_nodes [nfix] int32, nodes_per_block,
        nfix, peak_live).  A graph that was born on a device (its index never visits the host: mini-batch sub-graphs, from_tensors)
        gets the same plan from gnm_graph_build_sweep_plan_device (fix_nodes then is [N] with -1 for the served nodes and nfix = N);
        None only with GNM_DEVICE_PLANS=0 -- the engine then keeps the separate by-source passes."""
        device = torch.device(device) if device is not None else self.device
        key = (device, wg_per_cu)
        if key in self._plans:
            return self._plans[key]
        plan = None
        if (self._src_t is None or self._host_index is not None) and self.num_edges() > 0 and device.type == "cuda":
            lib = _lib.load()
            h = self.host_index()
            n, e = self._n, self.num_edges()
            npb, grid = C.c_int64(0), C.c_int(0)
            with torch.cuda.device(device):
                _lib.check(lib.gnm_sweep_partition(n, wg_per_cu, C.byref(npb), C.byref(grid)), "gnm_sweep_partition")
            plan = build_sweep_plan(h, n, npb.value)
            if plan is not None:
                plan = {k: (torch.from_numpy(v).to(device) if isinstance(v, np.ndarray) else v) for k, v in plan.items()}
        elif self.num_edges() > 0 and device.type == "cuda" and DEVICE_PLANS:
            plan = build_sweep_plan_device(self.index(device), self._n, device, wg_per_cu)
        self._plans[key] = plan
        return plan


SWEEP_TILE_ROWS, SWEEP_SLOTS, SWEEP_MARGIN = 16, 32, 1 << 16      # = kSweepTileRows / kSweepSlots / kSweepMargin (gnm_tr.h)


def build_sweep_plan(host_index, n: int, nodes_per_block: int, nslots: int = SWEEP_SLOTS, margin: int = SWEEP_MARGIN):
    """gnm_graph_build_sweep_plan on a host index (dict of int32 numpy arrays): numpy arrays sinfo / dinfo (the plan
    words, viewed as int32) and fix_nodes, plus nodes_per_block, nfix, peak_live."""
    lib = _lib.load()
    e = int(host_index["isrc"].size)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    sinfo, dinfo = np.zeros(e, np.uint32), np.zeros(e, np.uint32)
    fix = np.empty(max(n, 1), np.int32)
    nfix, peak = C.c_int64(0), C.c_int(0)
    rc = lib.gnm_graph_build_sweep_plan(ptr(host_index["isrc"]), ptr(host_index["idst"]), ptr(host_index["in_ptr"]), n, e,
                                        int(nodes_per_block), SWEEP_TILE_ROWS, nslots, margin, ptr(sinfo), ptr(dinfo),
                                        ptr(fix), C.byref(nfix), C.byref(peak))
    if rc == 3:         # one workgroup's rows exceed the sweep kernels' 32-bit offsets (a hub-heavy graph): no plan, separate passes
        return None
    _lib.check(rc, "gnm_graph_build_sweep_plan")
    return {"sinfo": sinfo.view(np.int32), "dinfo": dinfo.view(np.int32), "fix_nodes": fix[:nfix.value].copy(),
            "nodes_per_block": int(nodes_per_block), "nfix": int(nfix.value), "peak_live": int(peak.value)}


# graphs born on a device (mini-batch sub-graphs, from_tensors on device tensors): the plan is built there too
# (gnm_graph_build_sweep_plan_device); GNM_DEVICE_PLANS=0 leaves them on the separate by-source passes
DEVICE_PLANS = os.environ.get("GNM_DEVICE_PLANS", "1") != "0"


def build_sweep_plan_device(idx, n: int, device, wg_per_cu: int, nslots: int = SWEEP_SLOTS, margin: int = SWEEP_MARGIN):
    """The plan of build_sweep_plan from a DEVICE index, without a host round trip and without a synchronisation: the plan
    words and the fix list come from gnm_graph_build_sweep_plan_device; the list is [N] with -1 for the served nodes (the fix-up
    kernels skip negative entries), so `nfix` is N, not the count."""
    lib = _lib.load()
    e = int(idx["isrc"].numel())
    i32 = dict(dtype=torch.int32, device=device)
    npb, grid = C.c_int64(0), C.c_int(0)
    with torch.cuda.device(device):
        _lib.check(lib.gnm_sweep_partition(n, wg_per_cu, C.byref(npb), C.byref(grid)), "gnm_sweep_partition")
        if (e + 64) * 1024 >= 2 ** 31:
            # more rows than ANY workgroup could address through the sweep kernels' 32-bit row offsets if it owned them all: look at
            # the real shares (one synchronisation; the mini-batch sub-graphs this path exists for are far below the bound)
            b = torch.arange(0, n + npb.value, npb.value, device=device).clamp_(max=n)
            rows = idx["in_ptr"].long()[b]
            if int((rows[1:] - rows[:-1]).max()) + 64 >= 2 ** 21:
                return None
        sinfo, dinfo = torch.empty(e, **i32), torch.empty(e, **i32)
        served = torch.empty(n, dtype=torch.uint8, device=device)
        fix = torch.empty(n, **i32)
        first, last, peak = torch.empty(n, **i32), torch.empty(n, **i32), torch.empty(1, **i32)
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        _lib.check(lib.gnm_graph_build_sweep_plan_device(p(idx["isrc"]), p(idx["idst"]), p(idx["in_ptr"]), n, e, npb.value,
                                                         SWEEP_TILE_ROWS, nslots, margin, p(sinfo), p(dinfo), p(served), p(fix),
                                                         p(first), p(last), p(peak),
                                                         C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                   "gnm_graph_build_sweep_plan_device")
    return {"sinfo": sinfo, "dinfo": dinfo, "fix_nodes": fix, "nodes_per_block": int(npb.value), "nfix": int(n),
            "peak_live": None, "served": served, "peak_dev": peak}


def _to_numpy(a):
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy()
    return np.asarray(a)


def from_dgl(g):
    """An AssemblyGraph with the structure (edge-id order kept) and the ndata / edata of a DGLGraph -- or of anything with
    that surface: edges() -> (src, dst), num_nodes(), ndata, edata, device (train.py:245,252 / inference.py:446,453 pass
    such an object to the model).  The result lives on the source graph's device."""
    s, d = g.edges()
    ag = AssemblyGraph(s, d, g.num_nodes())
    ag.ndata = dict(getattr(g, "ndata", {}))
    ag.edata = dict(getattr(g, "edata", {}))
    dev = getattr(g, "device", None)
    if dev is None and torch.is_tensor(s):
        dev = s.device
    return ag.to(dev) if dev is not None and torch.device(dev).type != "cpu" else ag


_WRAPPED = {}      # id(foreign graph) -> AssemblyGraph, for graph objects that refuse new attributes (dropped by a weakref finalizer)
# Structure by CONTENT.  The reference's loops call g = g.to(device) on every step (train.py:244,297; inference.py:446), and DGL
# returns a NEW graph object each time: a cache on the object alone would rebuild the host index, the locality order and both
# sweep plans (~1.5 s for a chr19-scale graph) per step.  A foreign graph is therefore also looked up by a fingerprint of its edge
# list (node / edge counts + two position-weighted 64-bit sums over src and dst: three tiny reductions and one synchronisation).
# What is cached is the IMMUTABLE part only -- edge list, host index, device indices, sweep plans -- as a feature-less
# AssemblyGraph; a hit is CONFIRMED by comparing the edge lists element for element (a fingerprint collision must never hand a
# different graph a stale index) and returns a shallow per-caller copy that carries the caller's own ndata / edata (two live
# graphs of equal structure never see each other's features).  The cache is bounded by entries (GNM_GRAPH_CACHE, default 16;
# 0 = off) AND by an estimate of the device bytes the entries pin (GNM_GRAPH_CACHE_BYTES, default 8 GiB: index + two plans are
# ~64 B per edge); graphs below GNM_GRAPH_CACHE_MIN_EDGES (default 65,536: rebuilding is cheaper than a fingerprint's host
# synchronisation, and never-repeating graphs -- DGL mini-batch sub-graphs -- are of that kind) are neither fingerprinted nor kept.
_BY_CONTENT = collections.OrderedDict()      # fingerprint -> (structure-only AssemblyGraph, estimated bytes)
GRAPH_CACHE = int(os.environ.get("GNM_GRAPH_CACHE", "16"))
GRAPH_CACHE_BYTES = int(os.environ.get("GNM_GRAPH_CACHE_BYTES", str(8 << 30)))
GRAPH_CACHE_MIN_EDGES = int(os.environ.get("GNM_GRAPH_CACHE_MIN_EDGES", "65536"))


def _edge_tensors(g):
    s, d = g.edges()
    if not (torch.is_tensor(s) and torch.is_tensor(d)):
        s, d = torch.as_tensor(np.asarray(s)), torch.as_tensor(np.asarray(d))
    return s, d


def _fingerprint(g):
    s, d = _edge_tensors(g)
    w = torch.arange(1, 2 * s.numel() + 1, 2, dtype=torch.int64, device=s.device)       # odd weights: order-sensitive
    h = torch.stack(((s.long() * w).sum(), (d.long() * w).sum()))
    return (int(g.num_nodes()), int(s.numel())) + tuple(int(x) for x in h.cpu())


def _same_edges(base: "AssemblyGraph", g) -> bool:
    """Element-for-element comparison of a cached structure's edge list with a foreign graph's (on the device the foreign edge
    list lives on when the cached graph has a copy there, else on the host)."""
    s, d = _edge_tensors(g)
    if s.numel() != base.num_edges():
        return False
    have = base._dev_edges.get(s.device)
    if have is None and s.device.type != "cpu":
        have = (torch.from_numpy(base._src).to(s.device), torch.from_numpy(base._dst).to(s.device))
        base._dev_edges[s.device] = have
    if have is None:
        have = (torch.from_numpy(base._src), torch.from_numpy(base._dst))
    return bool(torch.equal(have[0].to(torch.int64), s.to(torch.int64)) and torch.equal(have[1].to(torch.int64), d.to(torch.int64)))


def _caller_view(base: "AssemblyGraph", g, device=None) -> "AssemblyGraph":
    """A shallow copy of a cached structure (every index / plan dict shared) with the CALLER's feature dicts."""
    ag = AssemblyGraph.__new__(AssemblyGraph)
    ag.__dict__.update(base.__dict__)
    ag.ndata = dict(getattr(g, "ndata", {}))
    ag.edata = dict(getattr(g, "edata", {}))
    dev = getattr(g, "device", None)
    if dev is None:
        s = g.edges()[0]
        dev = s.device if torch.is_tensor(s) else None
    if dev is not None and torch.device(dev).type != "cpu":
        dev = torch.device(dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        ag.device = dev
    return ag


def _cache_put(key, base: "AssemblyGraph") -> None:
    nbytes = 64 * base.num_edges() + 16 * base.num_nodes()
    _BY_CONTENT[key] = (base, nbytes)
    while len(_BY_CONTENT) > 1 and (len(_BY_CONTENT) > GRAPH_CACHE or sum(b for _, b in _BY_CONTENT.values()) > GRAPH_CACHE_BYTES):
        _BY_CONTENT.popitem(last=False)


def as_assembly_graph(g, device=None):
    """What the modules call on their `graph` argument: an AssemblyGraph is returned as it is; any other object with the
    DGLGraph surface (edges(), num_nodes()) is wrapped (from_dgl: the index is built then) and the wrapper is cached on the
    object; its STRUCTURE is also cached by the content of its edge list (see _BY_CONTENT: the reference's per-step
    g.to(device) makes a new object every time) -- so the reference's call sites -- model(g, x, e, pe) with a DGLGraph,
    train.py:252 -- need only the import change.  `device`: where the features of this call live (a foreign graph may not
    carry a device)."""
    if isinstance(g, AssemblyGraph):
        return g
    ag = getattr(g, "_gnm_graph", None) or _WRAPPED.get(id(g))
    if ag is not None and (ag.num_edges() != int(g.num_edges()) or ag.num_nodes() != int(g.num_nodes())):
        ag = None                   # the object changed under the cached wrapper
    if ag is None:
        if not (hasattr(g, "edges") and hasattr(g, "num_nodes")):
            raise TypeError(f"graph argument of type {type(g).__name__} has no edges() / num_nodes()")
        cached = GRAPH_CACHE > 0 and int(g.num_edges()) >= GRAPH_CACHE_MIN_EDGES
        key = _fingerprint(g) if cached else None
        hit = _BY_CONTENT.get(key) if key is not None else None
        if hit is not None and not _same_edges(hit[0], g):
            hit = None              # a fingerprint collision: another graph (the entry is replaced below)
        if hit is None:
            ag = from_dgl(g)
            if key is not None:
                ag.host_index()     # built now (the first forward needs it anyway), so that every later view shares it
                base = AssemblyGraph.__new__(AssemblyGraph)         # the structure without anybody's features
                base.__dict__.update(ag.__dict__)
                base.ndata, base.edata = {}, {}
                _cache_put(key, base)
        else:
            _BY_CONTENT.move_to_end(key)
            ag = _caller_view(hit[0], g)
        _remember(g, ag)
    if device is not None and torch.device(device) != ag.device:
        ag = ag.to(device)
        _remember(g, ag)
    return ag


def _remember(g, ag):
    """Cache the wrapper on the foreign object; objects that refuse attributes are tracked by id only while a weakref finalizer can
    remove the entry again (an id is reused after the object dies) -- otherwise not at all: the content cache still finds them."""
    try:
        g._gnm_graph = ag
        return
    except AttributeError:
        pass
    import weakref
    try:
        weakref.finalize(g, _WRAPPED.pop, id(g), None)
        _WRAPPED[id(g)] = ag
    except TypeError:
        _WRAPPED.pop(id(g), None)
