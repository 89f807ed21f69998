"""How local are the by-SOURCE sums in the destination-sorted sweep?  

For the metric's graph (SURVEY 8d generator), with the generator's node ids and with shuffled ids renumbered by the
index, in the destination-sorted internal edge order:
  * per source: pos(last out-edge) - pos(first out-edge)   (rows a ring of sigma rows would have to hold);
  * distinct sources per 16 / 64 / 128 / 256-row tile;
  * LIVE sources at a sweep position (first out-edge seen, last not yet): what an accumulator window has to hold,
    as id span (a ring indexed by id mod W) and as a count (slots handed out by interval colouring);
  * what a plan with C slots per workgroup and G workgroups leaves to the fix-up pass.
CPU only (numpy + the host half of libgnm.so).  usage: python tools/twosided_probe.py [reads] > profiles/r04_band_histogram.txt
"""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnnome_assembly_amd import synth          # noqa: E402
from gnnome_assembly_amd.graph import AssemblyGraph   # noqa: E402


def pct(a, qs=(50, 90, 99, 99.9, 100)):
    return "  ".join(f"p{q}={np.percentile(a, q):.0f}" for q in qs)


def analyse(name, src, dst, n, node_order):
    g = AssemblyGraph(src, dst, n, node_order=node_order)
    ix = g.host_index()
    isrc, idst, in_ptr = ix["isrc"].astype(np.int64), ix["idst"].astype(np.int64), ix["in_ptr"].astype(np.int64)
    E = isrc.size
    print(f"== {name}: N={n} E={E} relabel={g.relabel_info}")
    pos = np.arange(E, dtype=np.int64)
    first = np.full(n, E, np.int64)
    last = np.full(n, -1, np.int64)
    np.minimum.at(first, isrc, pos)
    np.maximum.at(last, isrc, pos)
    has = last >= 0
    span = (last - first)[has]
    print(f"rows between a source's first and last out-edge: {pct(span)}   mean={span.mean():.1f}")
    d = np.abs(isrc - idst)
    print(f"|src - dst| per edge (internal ids): {pct(d)}")
    for T in (16, 64, 128, 256):
        nt = E // T
        tiles = isrc[: nt * T].reshape(nt, T)
        srt = np.sort(tiles, axis=1)
        distinct = 1 + (np.diff(srt, axis=1) != 0).sum(1)
        print(f"distinct sources per {T}-row tile: mean={distinct.mean():.1f}  {pct(distinct, (50, 99, 100))}")
    # live sources along the sweep, edges of "far" sources (any out-edge more than FAR ids away) left out
    for FAR in (64, 256, 2048):
        far_edge = d > FAR
        far_src = np.zeros(n, bool)
        far_src[isrc[far_edge]] = True
        loc = has & ~far_src
        ev = np.zeros(E + 1, np.int64)
        np.add.at(ev, first[loc], 1)
        np.add.at(ev, last[loc] + 1, -1)
        live = np.cumsum(ev)[:E]
        # id span of the live set at each position: via running max of opened ids and min of not-yet-closed ids is
        # costly; sample 20000 positions instead
        rng = np.random.default_rng(0)
        samp = np.sort(rng.integers(0, E, 20000))
        order_first = np.argsort(first[loc], kind="stable")
        ids = np.nonzero(loc)[0]
        f_sorted = first[loc][order_first]
        spans = []
        for p in samp[:4000]:
            hi = np.searchsorted(f_sorted, p, side="right")
            lo = max(0, hi - 4096)
            cand = ids[order_first[lo:hi]]
            cand = cand[last[cand] >= p]
            spans.append(cand.max() - cand.min() + 1 if cand.size else 0)
        print(f"far > {FAR:5d}: far sources {far_src.sum() / n:.4%} (edges {far_src[isrc].sum() / E:.4%});  live count: "
              f"{pct(live, (50, 99, 99.9, 100))};  id span of the live set: {pct(np.array(spans), (50, 99, 100))}")
    return ix


def plan_stats(ix, n, G, C, tile=16):
    """Greedy slot plan: G workgroups over contiguous destination-node ranges, C accumulator slots each."""
    isrc, idst, in_ptr = ix["isrc"].astype(np.int64), ix["idst"].astype(np.int64), ix["in_ptr"].astype(np.int64)
    E = isrc.size
    npb = -(-n // G)
    wg_of_edge = idst // npb
    # a source is a candidate iff all its out-edges fall into one workgroup
    wmin = np.full(n, G, np.int64)
    wmax = np.full(n, -1, np.int64)
    np.minimum.at(wmin, isrc, wg_of_edge)
    np.maximum.at(wmax, isrc, wg_of_edge)
    one = (wmin == wmax) & (wmax >= 0)
    pos = np.arange(E, dtype=np.int64)
    first = np.full(n, E, np.int64)
    last = np.full(n, -1, np.int64)
    np.minimum.at(first, isrc, pos)
    np.maximum.at(last, isrc, pos)
    # slots are handed out per tile (a source opens at the tile of its first edge, frees after the tile of its last)
    rb = in_ptr[np.minimum(wg_of_edge * npb, n)]
    tfirst = (first - rb[np.minimum(first, E - 1)]) // tile
    tlast = (last - rb[np.minimum(first, E - 1)]) // tile
    overflow = 0
    cand = np.nonzero(one)[0]
    # sweep per workgroup: count live (tile granularity); overflow when more than C are live
    key = wmin[cand] * (1 << 40) + first[cand]
    cand = cand[np.argsort(key, kind="stable")]
    import heapq
    nonlocal_src = (~one) & (last >= 0)
    cur_w, heap = -1, []
    peak = 0
    for s in cand:
        w = wmin[s]
        if w != cur_w:
            cur_w, heap = w, []
        tf = tfirst[s]
        while heap and heap[0] < tf:
            heapq.heappop(heap)
        if len(heap) >= C:
            overflow += 1
            nonlocal_src[s] = True
            continue
        heapq.heappush(heap, tlast[s])
        peak = max(peak, len(heap))
    nl_edges = nonlocal_src[isrc].sum()
    print(f"plan G={G} C={C} tile={tile}: sources split over workgroups {((~one) & (last >= 0)).sum() / n:.3%}, slot overflow "
          f"{overflow / n:.4%}, peak live {peak};  fix-up: {nonlocal_src.sum() / n:.3%} of the nodes, {nl_edges / E:.3%} of the edges")


def main():
    reads = int(sys.argv[1]) if len(sys.argv) > 1 else 750000
    src, dst, n = synth.make_graph(reads, seed=0)
    ix = analyse("generator ids (position-sorted), kept", src, dst, n, "auto")
    for G in (256, 512):
        for C in (32, 40, 48, 64):
            plan_stats(ix, n, G, C)
    ix = analyse("generator ids, renumbered breadth-first", src, dst, n, "bfs")
    for G in (256,):
        for C in (32, 40, 48, 64):
            plan_stats(ix, n, G, C)
    rng = np.random.default_rng(7)
    p = rng.permutation(n).astype(np.int32)
    ix = analyse("shuffled ids, renumbered by the index (auto)", p[src], p[dst], n, "auto")
    for G in (256,):
        for C in (32, 40, 48, 64):
            plan_stats(ix, n, G, C)


if __name__ == "__main__":
    main()
