"""On-disk graph format (the reference stores `.dgl` files written by
dgl.save_graphs, graph_dataset.py:129, a private DGL binary format, plus pickled succ / pred / edges
dicts, :130-134).  Here a graph is ONE documented `.npz`:

    src[E] int32, dst[E] int32, num_nodes,  ndata/<key>[N,...],  edata/<key>[E,...]

in the caller's edge-id order (so `edata['score']`, labels `y`, `prefix_length` ... keep their meaning,
inference.py:454, graph_parser.py:55-73,309).  The adjacency dicts need no file: decode.DecodeGraph
rebuilds them from src / dst in edge-id order.  `convert_dgl` turns a reference `.dgl` file into this
format wherever DGL is installed (it is not in this image)."""
from __future__ import annotations

import numpy as np
import torch

from .graph import AssemblyGraph

__all__ = ["save_graph", "load_graph", "convert_dgl"]


def _npz_path(path: str) -> str:
    """np.savez appends '.npz' to a path that lacks it; do the same on both sides so that
    save_graph(p) / load_graph(p) always name the same file."""
    path = str(path)
    return path if path.endswith(".npz") else path + ".npz"


def save_graph(path: str, graph: AssemblyGraph) -> None:
    arrays = {"src": graph._src, "dst": graph._dst, "num_nodes": np.int64(graph.num_nodes())}
    for k, v in graph.ndata.items():
        arrays["ndata/" + k] = v.detach().cpu().numpy()
    for k, v in graph.edata.items():
        arrays["edata/" + k] = v.detach().cpu().numpy()
    np.savez_compressed(_npz_path(path), **arrays)


def load_graph(path: str, device="cpu") -> AssemblyGraph:
    import os
    path = str(path)
    # the path as given if it names a file (saved through a file object, or under another extension); else save_graph's name
    with np.load(path if os.path.isfile(path) else _npz_path(path)) as z:
        src, dst, num_nodes = z["src"], z["dst"], int(z["num_nodes"])
        for name, a in (("src", src), ("dst", dst)):
            if a.ndim != 1 or not np.issubdtype(a.dtype, np.integer):
                raise ValueError(f"{name}: expected a 1-D integer array, got {a.dtype} {a.shape}")
            if a.size and (int(a.min()) < 0 or int(a.max()) >= num_nodes):
                raise ValueError(f"{name}: node id outside [0, {num_nodes})")
        if src.shape != dst.shape:
            raise ValueError(f"src has {src.size} entries, dst {dst.size}")
        if num_nodes > np.iinfo(np.int32).max or src.size > np.iinfo(np.int32).max:
            raise ValueError("graph too large for the int32 index of the HIP path")
        g = AssemblyGraph(src, dst, num_nodes)
        n, e = g.num_nodes(), g.num_edges()
        for k in z.files:
            if k.startswith("ndata/"):
                a = torch.from_numpy(z[k])
                if a.shape[0] != n:
                    raise ValueError(f"{k}: {a.shape[0]} rows for {n} nodes")
                g.ndata[k[6:]] = a
            elif k.startswith("edata/"):
                a = torch.from_numpy(z[k])
                if a.shape[0] != e:
                    raise ValueError(f"{k}: {a.shape[0]} rows for {e} edges")
                g.edata[k[6:]] = a
    return g.to(device) if torch.device(device).type != "cpu" else g


def convert_dgl(dgl_path: str, out_path: str, index: int = 0) -> None:
    """`.dgl` (dgl.load_graphs) -> `.npz`; needs DGL, which this image does not have."""
    try:
        import dgl
    except ImportError as e:       # pragma: no cover
        raise RuntimeError("convert_dgl needs DGL; run it where the reference's environment is installed") from e
    from .graph import from_dgl
    save_graph(out_path, from_dgl(dgl.load_graphs(dgl_path)[0][index]))
