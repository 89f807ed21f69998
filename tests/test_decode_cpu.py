"""Greedy decode: the oracle against the reference's own walks (golden),
and the C++ walks of libgnm.so (host code, no GPU needed) against the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import decode_oracle as orc


def _case():
    z = np.load(os.path.join(GOLDEN, "decode_walks.npz"))
    n = int(z["n"])
    unpack = lambda a: np.unpackbits(a, axis=1)[:, :n].astype(bool)  # noqa: E731
    return z, n, unpack(z["visited_old"]), unpack(z["visited_f"]), unpack(z["visited_b"])


def test_oracle_walks_match_reference_golden():
    z, n, old, vf_want, vb_want = _case()
    succs, preds, edges = orc.build_adjacency(z["src"], z["dst"], n)
    p = torch.from_numpy(z["scores"])
    for i, k in enumerate(z["starts"]):
        o = set(np.flatnonzero(old[i]).tolist())
        wf, vf = orc.walk_forwards(int(z["dst"][k]), p, succs, edges, o)
        wb, vb = orc.walk_backwards(int(z["src"][k]), p, preds, edges, o | vf)
        want = z["walks"][z["walk_offsets"][i]:z["walk_offsets"][i + 1]].tolist()
        assert wb + wf == want, i
        assert vf == set(np.flatnonzero(vf_want[i]).tolist()) and vb == set(np.flatnonzero(vb_want[i]).tolist())
        assert orc.get_contig_length(want, z["prefix_length"], z["read_length"], edges) == int(z["contig_length"][i])


def test_oracle_sampling_matches_reference_golden():
    z, *_ = _case()
    torch.manual_seed(int(z["sample_seed"]))
    idx = orc.sample_edges(torch.from_numpy(z["sample_scores"]), int(z["sample_nb_paths"]))
    assert np.array_equal(idx.numpy(), z["sample_idx"])


def test_cxx_single_walks_match_reference_golden():
    """gnm_decode_iteration with one start edge == reversed(walk_backwards) + walk_forwards of the reference."""
    import ctypes as C
    from gnnome_assembly_amd import _lib, decode
    z, n, old, *_ = _case()
    g = decode.DecodeGraph(z["src"], z["dst"], n)
    lib = _lib.load()
    sc = np.ascontiguousarray(z["scores"])
    pl, rl = np.ascontiguousarray(z["prefix_length"]), np.ascontiguousarray(z["read_length"])
    walk = np.empty(2 * n + 2, np.int32)
    blen = C.c_int64(0)
    p = decode._p
    for i, k in enumerate(z["starts"]):
        vis = np.ascontiguousarray(old[i].astype(np.uint8))
        s0, d0 = np.array([z["src"][k]], np.int32), np.array([z["dst"][k]], np.int32)
        ln = lib.gnm_decode_iteration(n, p(sc), p(pl), p(rl), *[p(a) for a in g.succ], *[p(a) for a in g.pred], p(vis), 1,
                                      p(s0), p(d0), 10 ** 9, p(walk), walk.size, C.byref(blen))
        want = z["walks"][z["walk_offsets"][i]:z["walk_offsets"][i + 1]]
        assert ln == want.size and np.array_equal(walk[:ln], want), i
        assert blen.value == int(z["contig_length"][i])
        assert np.array_equal(vis, old[i].astype(np.uint8))          # below the threshold nothing is consumed


@pytest.mark.parametrize("seed,thr", [(0, 20), (1, 5), (2, 60)])
def test_cxx_get_contigs_matches_oracle(seed, thr):
    """Whole loop (sampling -> 50 candidate walks -> best -> visited update incl. jumped-over nodes) against
    the oracle, same seeded draws on both sides."""
    from gnnome_assembly_amd import decode, synth
    rng = np.random.default_rng(seed)
    src, dst, n = synth.make_graph(500, seed=seed, permute_edge_ids=True)
    e = src.size
    scores = (rng.standard_normal(e) * 2).astype(np.float32)
    pl = rng.integers(500, 12000, e)
    rl = rng.integers(8000, 25000, n)
    torch.manual_seed(seed)
    want = orc.get_contigs(src, dst, n, scores, pl, rl, nb_paths=20, len_threshold=thr)
    torch.manual_seed(seed)
    got = decode.get_contigs(decode.DecodeGraph(src, dst, n), scores, pl, rl, nb_paths=20, len_threshold=thr)
    assert len(want) > 0 and got == want


def test_decode_rejects_bad_input_and_reports_cycles():
    from gnnome_assembly_amd import _lib, decode
    with pytest.raises(_lib.GnmError):
        decode.DecodeGraph(np.array([0, 5], np.int32), np.array([1, 2], np.int32), 3)
    # 0 -> 2 -> 4 -> 0: every node has exactly one successor: a forced-move cycle
    g = decode.DecodeGraph(np.array([0, 2, 4], np.int32), np.array([2, 4, 0], np.int32), 6)
    with pytest.raises(_lib.GnmError, match="forced"):
        decode.get_contigs(g, np.zeros(3, np.float32), np.ones(3, np.int64), np.ones(6, np.int64), nb_paths=2, len_threshold=1)
    with pytest.raises(ValueError):
        decode.get_contigs(g, np.zeros(2, np.float32), np.ones(3, np.int64), np.ones(6, np.int64))
