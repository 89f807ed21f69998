"""The sweep plan (gnm_graph_build_sweep_plan, host C++) against a numpy emulation of what the two-sided sweep kernels do
with it: per workgroup and 16-row tile every LEADER row sums its source's (destination's) rows of the tile, joins the sum
carried in its accumulator slot and parks or emits it.  Checked: the emitted sums equal the direct by-source /
by-destination sums for every served node, served + fix_nodes = all nodes, no two leaders of a tile touch one slot, a slot
is never read before it was written by the same source.  No GPU (the plan is host code)."""
import numpy as np
import pytest

from gnnome_assembly_amd import synth
from gnnome_assembly_amd.graph import AssemblyGraph, build_sweep_plan, SWEEP_TILE_ROWS

OPEN, CLOSE = 1 << 22, 1 << 23


def emulate(info, key, in_ptr, n, npb, x, nslots):
    """Run the leaders' protocol over the plan words `info` (per row); returns (sum per node or nan, served mask)."""
    E = key.size
    out = np.full(n, np.nan)
    nblk = -(-n // npb)
    for w in range(nblk):
        v0, v1 = w * npb, min(n, (w + 1) * npb)
        rb, re = int(in_ptr[v0]), int(in_ptr[v1])
        slots = np.full(nslots, np.nan)           # nan: never written (reading it would poison the sum)
        owner = np.full(nslots, -1)
        for r0 in range(rb, re, SWEEP_TILE_ROWS):
            nv = min(SWEEP_TILE_ROWS, re - r0)
            touched = []
            writes = []
            for r in range(nv):
                wd = int(info[r0 + r]) & 0xFFFFFFFF
                mask = wd & 0xFFFF
                if not mask:
                    continue
                node = int(key[r0 + r])
                assert mask & (1 << r), "the leader's own row is in its mask"
                assert (mask & ((1 << r) - 1)) == 0, "the leader is the FIRST row of its run"
                rows = [q for q in range(nv) if mask >> q & 1]
                assert mask >> nv == 0
                assert all(key[r0 + q] == node for q in rows)
                assert sum(1 for q in range(nv) if key[r0 + q] == node) == len(rows), "the mask holds ALL rows of the node"
                acc = 0.0
                for q in rows:
                    acc += x[r0 + q]
                sl = (wd >> 16) & 63
                if not wd & OPEN:
                    assert owner[sl] == node, "the slot read was written by the same node"
                    acc += slots[sl]
                    touched.append(sl)
                if wd & CLOSE:
                    assert np.isnan(out[node]), "a node is emitted once"
                    out[node] = acc
                    if not wd & OPEN:
                        writes.append((sl, np.nan, -1))
                else:
                    assert sl < nslots
                    writes.append((sl, acc, node))
                    if wd & OPEN:
                        touched.append(sl)
            assert len(touched) == len(set(touched)), "two leaders of one tile on one slot"
            for sl, v, o in writes:                    # all leaders of a tile run concurrently: writes land after the reads
                slots[sl], owner[sl] = v, o
        assert (owner == -1).all(), "every opened node of the workgroup was closed"
    return out


def check_plan(src, dst, n, npb, nslots=32, margin=1 << 16, node_order="keep", expect_all_local=False):
    g = AssemblyGraph(src, dst, n, node_order=node_order)
    ix = g.host_index()
    plan = build_sweep_plan(ix, n, npb, nslots=nslots, margin=margin)
    isrc, idst, in_ptr = ix["isrc"], ix["idst"], ix["in_ptr"]
    E = isrc.size
    rng = np.random.default_rng(0)
    x = rng.integers(-8, 9, size=E).astype(np.float64)          # small integers: every order of additions is exact
    by_src = np.zeros(n)
    np.add.at(by_src, isrc, x)
    by_dst = np.zeros(n)
    np.add.at(by_dst, idst, x)
    out_s = emulate(plan["sinfo"], isrc, in_ptr, n, npb, x, nslots)
    served = ~np.isnan(out_s)
    fix = plan["fix_nodes"]
    assert plan["nfix"] == fix.size
    assert np.array_equal(np.sort(fix), np.nonzero(~served)[0]), "fix_nodes = exactly the nodes the sweep does not emit"
    assert np.array_equal(out_s[served], by_src[served])
    out_d = emulate(plan["dinfo"], idst, in_ptr, n, npb, x, 2)
    has_in = np.diff(in_ptr) > 0
    assert np.array_equal(~np.isnan(out_d), has_in), "every destination with rows is emitted"
    assert np.array_equal(out_d[has_in], by_dst[has_in])
    assert plan["peak_live"] <= nslots
    outdeg = np.bincount(isrc, minlength=n)
    assert (outdeg[~served] == 0).all() or not expect_all_local
    return plan, served, outdeg


def test_banded_graph_is_served_almost_everywhere():
    src, dst, n = synth.make_graph(3000, seed=1)
    plan, served, outdeg = check_plan(src, dst, n, npb=750)
    # 0.5 % repeat edges + 7 chunk boundaries: a few per cent of the sources at most
    assert served[outdeg > 0].mean() > 0.93
    assert plan["peak_live"] <= 32


@pytest.mark.parametrize("npb", [8, 33, 64, 1000])
def test_tiny_adversarial_graph(npb):
    src, dst, n = synth.tiny_edge_case_graph(seed=3)      # hubs, self loops, duplicates, isolated nodes, random ids
    check_plan(src, dst, n, npb=npb)


def test_slot_overflow_goes_to_the_fix_list():
    src, dst, n = synth.make_graph(1500, seed=2)
    plan2, served2, outdeg = check_plan(src, dst, n, npb=3000, nslots=2)
    plan32, served32, _ = check_plan(src, dst, n, npb=3000, nslots=32)
    assert served2.sum() < served32.sum()          # with two slots most sources do not fit ...
    assert plan2["peak_live"] == 2                  # ... and the plan never hands out a third


def test_margin_bounds_the_store_offsets():
    # a long-range source whose only out-edge lands in a far workgroup: all its rows are in ONE workgroup, but it is
    # farther than `margin` ids from that workgroup's node range
    n = 4000
    src = np.concatenate([np.arange(0, n - 1), [5]]).astype(np.int32)
    dst = np.concatenate([np.arange(1, n), [3900]]).astype(np.int32)
    src = np.delete(src, 5)                        # node 5 keeps only the far edge
    dst = np.delete(dst, 5)
    plan, served, outdeg = check_plan(src, dst, n, npb=500, margin=1000)
    assert not served[5] and 5 in plan["fix_nodes"]
    plan, served, outdeg = check_plan(src, dst, n, npb=500, margin=1 << 16)
    assert served[5]


def test_shuffled_ids_after_the_internal_renumbering():
    src, dst, n = synth.make_graph(2500, seed=5)
    p = np.random.default_rng(9).permutation(n).astype(np.int32)
    plan, served, outdeg = check_plan(p[src], p[dst], n, npb=625, node_order="bfs")
    assert served[outdeg > 0].mean() > 0.9


def test_high_degree_destination_crosses_tiles():
    # one destination with 100 in-edges (7 tiles), sources with out-edges into several of its tiles
    n = 300
    src = np.concatenate([np.arange(100, 200), np.arange(100, 200), [0, 1, 2]]).astype(np.int32)
    dst = np.concatenate([np.full(100, 50), np.full(100, 51), [1, 2, 3]]).astype(np.int32)
    plan, served, outdeg = check_plan(src, dst, n, npb=300)
    assert served[100:200].sum() == 32          # 100 sources are open at once: 32 slots, the rest go to the fix list
    plan, served, outdeg = check_plan(src, dst, n, npb=300, nslots=64)
    assert served[100:200].sum() == 64


def test_a_hub_beyond_the_32_bit_row_offsets_gets_no_plan():
    """The sweep kernels reach a workgroup's rows through 32-bit buffer offsets (row x up to 1024 bytes): a node partition that
    gives ONE workgroup more than 2^21 - 64 rows (a hub with millions of in-edges; the average share says nothing) cannot be
    swept.  The builder reports it (return code 3 -> build_sweep_plan returns None) and the graph keeps the separate by-source
    passes, instead of a plan whose kernels would drop or misplace stores."""
    n, e = 64, (1 << 21) + 200
    rng = np.random.default_rng(0)
    src = rng.integers(1, n, size=e).astype(np.int32)
    dst = np.zeros(e, np.int32)                       # every edge into node 0
    dst[:100] = rng.integers(1, n, size=100)
    order = np.argsort(dst, kind="stable")
    isrc, idst = src[order], dst[order]
    in_ptr = np.concatenate(([0], np.cumsum(np.bincount(idst, minlength=n)))).astype(np.int32)
    h = {"isrc": isrc, "idst": idst, "in_ptr": in_ptr}
    assert build_sweep_plan(h, n, 8) is None
    # the same graph with a quarter of the edges is fine
    keep = np.sort(rng.choice(e, e // 4, replace=False))
    isrc, idst = isrc[keep], idst[keep]
    in_ptr = np.concatenate(([0], np.cumsum(np.bincount(idst, minlength=n)))).astype(np.int32)
    assert build_sweep_plan({"isrc": isrc, "idst": idst, "in_ptr": in_ptr}, n, 8) is not None
