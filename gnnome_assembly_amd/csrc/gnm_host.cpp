// Host-side pieces of libgnm.so: error reporting, device query, graph index build.
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <vector>

#include "gnm_common.h"

namespace gnm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;  // MI355X
  }
  return cus;
}

static int g_occ_cap = 0;
int occupancy_cap() { return g_occ_cap; }

}  // namespace gnm

extern "C" int gnm_set_occupancy_cap(int blocks_per_cu) {
  GNM_CHECK_ARG(blocks_per_cu >= 0 && blocks_per_cu <= 8, "set_occupancy_cap: %d outside [0, 8]", blocks_per_cu);
  gnm::g_occ_cap = blocks_per_cu;
  return 0;
}
extern "C" int gnm_abi_version(void) { return GNM_ABI_VERSION; }
extern "C" const char* gnm_last_error(void) { return gnm::g_err; }
extern "C" int gnm_num_cus(void) { return gnm::num_cus(); }
extern "C" int gnm_max_partial_blocks(void) { return gnm::kMaxPartialBlocks; }

// Stable counting sort of the edge list by destination (internal order) and by source.
// Replaces DGL's lazy CSR/CSC construction and dgl.reverse (gated_gcn_full.py:115): the
// reversed graph is the same edge set indexed by source, so one index serves both passes.
extern "C" int gnm_graph_build_index(const int32_t* src, const int32_t* dst, int64_t N, int64_t E,
                                     int32_t* perm, int32_t* isrc, int32_t* idst, int32_t* in_ptr,
                                     int32_t* out_ptr, int32_t* out_pos, int32_t* out_dst) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && N < INT32_MAX && E < INT32_MAX, "graph_build_index: N/E out of int32 range");
  GNM_CHECK_ARG((E == 0 || (src && dst)) && perm && isrc && idst && in_ptr && out_ptr && out_pos && out_dst,
                "graph_build_index: null argument");
  for (int64_t k = 0; k < E; ++k) {
    if (src[k] < 0 || src[k] >= N || dst[k] < 0 || dst[k] >= N) {
      gnm::set_error("graph_build_index: edge %lld = (%d -> %d) outside [0, %lld)", (long long)k, src[k],
                     dst[k], (long long)N);
      return -2;
    }
  }
  // by destination, stable in edge id
  std::memset(in_ptr, 0, sizeof(int32_t) * (size_t)(N + 1));
  for (int64_t k = 0; k < E; ++k) in_ptr[dst[k] + 1]++;
  for (int64_t v = 0; v < N; ++v) in_ptr[v + 1] += in_ptr[v];
  {
    std::vector<int32_t> cur(in_ptr, in_ptr + N);
    for (int64_t k = 0; k < E; ++k) {
      const int32_t j = cur[dst[k]]++;
      perm[j] = (int32_t)k;
      isrc[j] = src[k];
      idst[j] = dst[k];
    }
  }
  // by source, stable in internal position
  std::memset(out_ptr, 0, sizeof(int32_t) * (size_t)(N + 1));
  for (int64_t j = 0; j < E; ++j) out_ptr[isrc[j] + 1]++;
  for (int64_t v = 0; v < N; ++v) out_ptr[v + 1] += out_ptr[v];
  {
    std::vector<int32_t> cur(out_ptr, out_ptr + N);
    for (int64_t j = 0; j < E; ++j) {
      const int32_t m = cur[isrc[j]]++;
      out_pos[m] = (int32_t)j;
      out_dst[m] = idst[j];
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------
// Locality order of the nodes.  The gather kernels live on the 4 MB per-XCD L2 holding the node rows of the edges
// in flight, i.e. on node ids that follow the genome: the reference never gives that (pipeline.py:46-61,160-169
// keep the simulator's read order, graph_parser.py:297-304 numbers nodes by read id), so the engine renumbers the
// nodes internally, once per graph, and nothing node-shaped leaves it in that numbering.
//
// Order = breadth-first (Cuthill-McKee without the degree sort) over the symmetrised graph, every component
// started from a pseudo-peripheral node (the last node of a first sweep).  True overlaps are transitive -- reads
// that overlap share neighbours -- while repeat-induced edges join reads with disjoint neighbourhoods and act as
// shortcuts that fold a breadth-first order (every shortcut seeds a new front: the levels become unions of
// thousands of short runs).  So the sweep runs on the triangle-supported edges only, as long as those are at least
// half of the edges; the rest still count as edges of the graph, just not of the ordering.
// ------------------------------------------------------------------------------------------
extern "C" int gnm_graph_edge_locality(const int32_t* src, const int32_t* dst, int64_t N, int64_t E, int64_t window,
                                       double* frac_out) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && window >= 0 && (E == 0 || (src && dst)) && frac_out,
                "graph_edge_locality: bad argument");
  int64_t local = 0;
  for (int64_t k = 0; k < E; ++k) {
    const int64_t d = (int64_t)src[k] - (int64_t)dst[k];
    local += (d < 0 ? -d : d) <= window;
  }
  *frac_out = E ? (double)local / (double)E : 1.0;
  return 0;
}

extern "C" int gnm_graph_locality_order(const int32_t* src, const int32_t* dst, int64_t N, int64_t E,
                                        int32_t* order, int32_t* rank, double* core_frac_out) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && N < INT32_MAX && 2 * E < INT32_MAX, "graph_locality_order: N/E out of int32 range");
  GNM_CHECK_ARG((E == 0 || (src && dst)) && (N == 0 || (order && rank)), "graph_locality_order: null argument");
  for (int64_t k = 0; k < E; ++k)
    if (src[k] < 0 || src[k] >= N || dst[k] < 0 || dst[k] >= N) {
      gnm::set_error("graph_locality_order: edge %lld = (%d -> %d) outside [0, %lld)", (long long)k, src[k], dst[k],
                     (long long)N);
      return -2;
    }
  // symmetrised adjacency without self loops, every list sorted by neighbour id (duplicates stay: harmless)
  std::vector<int32_t> ptr((size_t)N + 1, 0);
  for (int64_t k = 0; k < E; ++k)
    if (src[k] != dst[k]) { ++ptr[(size_t)src[k] + 1]; ++ptr[(size_t)dst[k] + 1]; }
  for (int64_t v = 0; v < N; ++v) ptr[(size_t)v + 1] += ptr[(size_t)v];
  std::vector<int32_t> adj((size_t)ptr[(size_t)N]);
  {
    std::vector<int32_t> cur(ptr.begin(), ptr.end() - 1);
    for (int64_t k = 0; k < E; ++k)
      if (src[k] != dst[k]) {
        adj[(size_t)cur[(size_t)src[k]]++] = dst[k];
        adj[(size_t)cur[(size_t)dst[k]]++] = src[k];
      }
  }
  for (int64_t v = 0; v < N; ++v) {           // short lists: insertion sort (a hub's list: std::sort)
    int32_t* a = adj.data() + ptr[(size_t)v];
    const int32_t n = ptr[(size_t)v + 1] - ptr[(size_t)v];
    if (n > 64) { std::sort(a, a + n); continue; }
    for (int32_t i = 1; i < n; ++i) {
      const int32_t x = a[i];
      int32_t j = i;
      for (; j > 0 && a[j - 1] > x; --j) a[j] = a[j - 1];
      a[j] = x;
    }
  }
  // core[p] = the edge (v, adj[p]) closes a triangle: v and adj[p] have a common neighbour
  std::vector<uint8_t> core(adj.size(), 0);
  int64_t ncore = 0;
  for (int64_t v = 0; v < N; ++v)
    for (int32_t p = ptr[(size_t)v]; p < ptr[(size_t)v + 1]; ++p) {
      const int32_t u = adj[(size_t)p];
      if (u < v) continue;                    // decided from the smaller end, mirrored below
      int32_t i = ptr[(size_t)v], ie = ptr[(size_t)v + 1], j = ptr[(size_t)u], je = ptr[(size_t)u + 1];
      if ((int64_t)(ie - i) + (je - j) > 16384) continue;   // a hub (collapsed repeat): its edges stay out of the ordering, and the
                                                            // merge below stays linear in E instead of quadratic in the hub's degree
      bool tri = false;
      while (i < ie && j < je) {
        const int32_t x = adj[(size_t)i], y = adj[(size_t)j];
        if (x == y) { if (x != u && x != v) { tri = true; break; } ++i; ++j; }
        else if (x < y) ++i;
        else ++j;
      }
      if (tri) { core[(size_t)p] = 1; ++ncore; }
    }
  for (int64_t v = 0; v < N; ++v)             // mirror onto the larger end's entries (binary search in the sorted list)
    for (int32_t p = ptr[(size_t)v]; p < ptr[(size_t)v + 1]; ++p) {
      const int32_t u = adj[(size_t)p];
      if (u >= v) continue;
      int32_t lo = ptr[(size_t)u], hi = ptr[(size_t)u + 1];
      while (lo < hi) { const int32_t m = (lo + hi) >> 1; if (adj[(size_t)m] < (int32_t)v) lo = m + 1; else hi = m; }
      core[(size_t)p] = core[(size_t)lo];     // the first copy of (u, v): duplicates share the flag
    }
  const int64_t nund = (int64_t)adj.size() / 2;
  const bool use_core = nund > 0 && 2 * ncore >= nund;
  if (core_frac_out) *core_frac_out = nund ? (double)ncore / (double)nund : 0.0;
  auto sweep = [&](int32_t start, std::vector<uint8_t>& mark, std::vector<int32_t>& q) {
    q.clear();
    q.push_back(start);
    mark[(size_t)start] = 1;
    for (size_t h = 0; h < q.size(); ++h) {
      const int32_t v = q[h];
      for (int32_t p = ptr[(size_t)v]; p < ptr[(size_t)v + 1]; ++p) {
        if (use_core && !core[(size_t)p]) continue;
        const int32_t u = adj[(size_t)p];
        if (!mark[(size_t)u]) { mark[(size_t)u] = 1; q.push_back(u); }
      }
    }
  };
  std::vector<uint8_t> seen1((size_t)N, 0), seen2((size_t)N, 0);
  std::vector<int32_t> q1, q2;
  int64_t pos = 0;
  for (int64_t s = 0; s < N; ++s) {
    if (seen1[(size_t)s]) continue;
    sweep((int32_t)s, seen1, q1);             // first sweep: its last node is (pseudo-)peripheral
    sweep(q1.back(), seen2, q2);              // second sweep from there = the order of this component
    for (int32_t v : q2) order[pos++] = v;
  }
  for (int64_t i = 0; i < N; ++i) rank[(size_t)order[i]] = (int32_t)i;
  return 0;
}

// ------------------------------------------------------------------------------------------
// Greedy decode on the host (inference.py:31-77,182-253): sequential walks over an adjacency in
// edge-id order.  The reference keeps dict-of-list successors / predecessors and a
// (src, dst) -> edge id dict (graph_parser.py:13-73); here both directions are CSR arrays whose
// `eid` column already holds that dict's answer (the LAST edge id of a duplicated pair).
// ------------------------------------------------------------------------------------------
// Sweep plan: what lets ONE destination-sorted sweep also form the by-SOURCE sums (the reference aggregates the same
// gate on g and on dgl.reverse(g), gated_gcn_full.py:128-129,141-142; the backward duals likewise).  A workgroup of the
// sweep owns the rows of a contiguous range of destination nodes and walks them in tiles of `tile_rows`; the out-edges
// of a source land in a few nearby tiles (overlap graphs are banded once the nodes are numbered along the genome:
// <= 28 sources are open at any time on the chr19-scale graph, profiles/r04_band_histogram.txt), so a source's sum is
// carried from tile to tile in one of `nslots` accumulator slots of the workgroup's LDS.  Everything data dependent
// is decided HERE, once per graph, so that the kernels stay branch-free around their memory operations and
// deterministic (one owner per source and tile, fixed order of additions, no atomics):
//   sinfo[j] != 0  iff row j is the FIRST row of its source inside its tile (the "leader") and the source is local:
//     bits 0-15  the tile rows that share the source (bit r = row r of the tile), always including the leader's;
//     bits 16-21 the accumulator slot;  bit 22 OPEN: first tile of the source (the slot is not read);
//     bit 23     CLOSE: last tile of the source (the sum goes to memory, the slot is not written).
//   dinfo[j]: the same for the destination of row j (its rows are contiguous; two slots alternate).
//   fix_nodes: the sources the sweep does NOT serve -- out-edges in more than one workgroup (boundaries, repeat
//     edges), no free slot, or outside [first node - margin, last node + margin) of the workgroup (the sums are
//     stored through 32-bit buffer offsets) -- plus the nodes without out-edges; a small gather kernel covers them.
// A slot freed by a CLOSE is handed out again from the NEXT tile on (inside a tile every leader runs concurrently).
// ------------------------------------------------------------------------------------------
extern "C" int gnm_graph_build_sweep_plan(const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, int64_t N,
                                          int64_t E, int64_t nodes_per_block, int tile_rows, int nslots, int64_t margin,
                                          uint32_t* sinfo, uint32_t* dinfo, int32_t* fix_nodes, int64_t* nfix_out,
                                          int32_t* peak_live_out) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && N < INT32_MAX && E < INT32_MAX && nodes_per_block > 0 && tile_rows > 0 &&
                    tile_rows <= 16 && nslots > 0 && nslots <= 64 && margin >= 0,
                "graph_build_sweep_plan: bad extent (tile_rows <= 16, nslots <= 64)");
  GNM_CHECK_ARG((E == 0 || (isrc && idst && sinfo && dinfo)) && in_ptr && (N == 0 || fix_nodes) && nfix_out,
                "graph_build_sweep_plan: null argument");
  constexpr uint32_t kOpen = 1u << 22, kClose = 1u << 23;
  std::memset(sinfo, 0, sizeof(uint32_t) * (size_t)E);
  std::memset(dinfo, 0, sizeof(uint32_t) * (size_t)E);
  std::vector<int32_t> first((size_t)N, -1), last((size_t)N, -1);
  for (int64_t j = 0; j < E; ++j) {
    const int32_t s = isrc[j];
    if (s < 0 || s >= N || idst[j] < 0 || idst[j] >= N) {
      gnm::set_error("graph_build_sweep_plan: row %lld = (%d -> %d) outside [0, %lld)", (long long)j, s, idst[j], (long long)N);
      return -2;
    }
    if (first[(size_t)s] < 0) first[(size_t)s] = (int32_t)j;
    last[(size_t)s] = (int32_t)j;
  }
  // state per source: -1 = not served by the sweep, -2 = served without a slot so far, >= 0 = its slot
  std::vector<int8_t> slot_of((size_t)N, -1);
  std::vector<uint8_t> served((size_t)N, 0);
  int peak = 0;
  const int64_t nblk = (N + nodes_per_block - 1) / nodes_per_block;
  std::vector<int> free_slots, pending;
  // The sweep kernels address a workgroup's rows through 32-bit buffer offsets (row index x up to 1024 bytes for a
  // 256-wide layer): a hub-heavy graph whose node partition gives one workgroup more rows than that cannot be swept --
  // report it (return code 3, nothing written) and the caller keeps the separate by-source passes.
  for (int64_t w = 0; w < nblk; ++w) {
    const int64_t v0 = w * nodes_per_block, v1 = std::min<int64_t>(N, v0 + nodes_per_block);
    if (((int64_t)in_ptr[v1] - in_ptr[v0] + 64) * 1024 >= (int64_t)INT32_MAX) {
      gnm::set_error("graph_build_sweep_plan: workgroup %lld of the node partition owns %lld rows: beyond the sweep kernels' 32-bit "
                     "row offsets (the graph runs the separate by-source passes)", (long long)w, (long long)(in_ptr[v1] - in_ptr[v0]));
      return 3;
    }
  }
  for (int64_t w = 0; w < nblk; ++w) {
    const int64_t v0 = w * nodes_per_block, v1 = std::min<int64_t>(N, v0 + nodes_per_block);
    const int64_t rb = in_ptr[v0], re = in_ptr[v1];
    free_slots.clear();
    pending.clear();
    for (int q = nslots - 1; q >= 0; --q) free_slots.push_back(q);
    int live = 0, dslot = 0;
    for (int64_t r0 = rb; r0 < re; r0 += tile_rows) {
      const int nv = (int)std::min<int64_t>(tile_rows, re - r0);
      for (int q : pending) free_slots.push_back(q);       // closed in the previous tile
      live -= (int)pending.size();
      pending.clear();
      // ---- sources ----
      for (int r = 0; r < nv; ++r) {
        const int64_t j = r0 + r;
        const int32_t s = isrc[j];
        bool leader = true;
        for (int q = 0; q < r; ++q) leader = leader && isrc[r0 + q] != s;
        if (!leader) continue;
        const int64_t f = first[(size_t)s], l = last[(size_t)s];
        const bool opens = f >= r0;                           // the source's first row is in this tile (it is a leader's)
        const bool closes = l < r0 + nv;
        if (opens) {
          // all out-edges inside this workgroup's rows, and close enough for 32-bit store offsets
          const bool inside = f >= rb && l < re && s >= v0 - margin && s < v1 + margin;
          if (!inside) continue;
          if (!closes) {
            if (free_slots.empty()) continue;                 // no slot: left to the fix-up pass
            slot_of[(size_t)s] = (int8_t)free_slots.back();
            free_slots.pop_back();
            ++live;
            peak = std::max(peak, live);
          } else {
            slot_of[(size_t)s] = -2;
          }
          served[(size_t)s] = 1;
        }
        if (!served[(size_t)s]) continue;
        uint32_t mask = 0;
        for (int q = r; q < nv; ++q) mask |= (isrc[r0 + q] == s) ? (1u << q) : 0u;
        uint32_t wd = mask;
        const int sl = slot_of[(size_t)s];
        if (sl >= 0) wd |= (uint32_t)sl << 16;
        if (opens) wd |= kOpen;
        if (closes) {
          wd |= kClose;
          if (sl >= 0) pending.push_back(sl);
        }
        sinfo[j] = wd;
      }
      // ---- destinations (contiguous runs; a run that crosses the tile boundary carries its sum in one of two slots) ----
      for (int r = 0; r < nv;) {
        const int32_t d = idst[r0 + r];
        int q = r;
        uint32_t mask = 0;
        while (q < nv && idst[r0 + q] == d) mask |= 1u << q++;
        const bool opens = r0 + r == in_ptr[d];
        const bool closes = r0 + q == in_ptr[d + 1];
        uint32_t wd = mask;
        if (!opens) wd |= (uint32_t)dslot << 16;              // continues the run of the previous tile: its slot
        if (opens) wd |= kOpen;
        if (closes) wd |= kClose;
        if (!closes) {                                        // carried into the next tile
          if (opens) dslot ^= 1;                              // a fresh slot (the other one may still be read in this tile)
          wd = (wd & ~(63u << 16)) | ((uint32_t)dslot << 16);
        }
        dinfo[r0 + r] = wd;
        r = q;
      }
    }
  }
  int64_t nfix = 0;
  for (int64_t v = 0; v < N; ++v)
    if (!served[(size_t)v]) fix_nodes[nfix++] = (int32_t)v;
  *nfix_out = nfix;
  if (peak_live_out) *peak_live_out = peak;
  return 0;
}

// ------------------------------------------------------------------------------------------
namespace {

void csr_by(const int32_t* key, const int32_t* other, int64_t N, int64_t E, int32_t* ptr, int32_t* nbr, int32_t* eid) {
  std::vector<int32_t> cnt((size_t)N + 1, 0);
  for (int64_t k = 0; k < E; ++k) ++cnt[(size_t)key[k] + 1];
  ptr[0] = 0;
  for (int64_t v = 0; v < N; ++v) ptr[v + 1] = ptr[v] + cnt[(size_t)v + 1];
  std::vector<int32_t> fill(ptr, ptr + N);
  for (int64_t k = 0; k < E; ++k) {          // ascending edge id inside a node: the dict-of-list order
    const int32_t p = fill[(size_t)key[k]]++;
    nbr[p] = other[k];
    eid[p] = (int32_t)k;
  }
  for (int64_t v = 0; v < N; ++v)            // duplicate (v, nbr) pairs: every copy answers with the last id
    for (int32_t a = ptr[v]; a < ptr[v + 1]; ++a)
      for (int32_t b = a + 1; b < ptr[v + 1]; ++b)
        if (nbr[b] == nbr[a]) eid[a] = eid[b];
}

// inference.py:31-52 / :55-76.  Marks seen[] (node and node^1), returns the walk in walking order.
// A forced move (exactly one neighbour) is taken even into consumed territory, as in the reference;
// a walk longer than `cap` can only be a forced-move cycle (the reference would not terminate).
bool greedy_walk(int32_t start, const float* scores, const int32_t* ptr, const int32_t* nbr, const int32_t* eid,
                 int64_t N, const uint8_t* old1, const uint8_t* old2, uint8_t* seen, std::vector<int32_t>& walk,
                 std::vector<int32_t>& touched, int64_t cap) {
  int32_t cur = start;
  for (;;) {
    if ((int64_t)walk.size() >= cap) return false;
    walk.push_back(cur);
    if (!seen[cur]) { seen[cur] = 1; touched.push_back(cur); }
    const int32_t rc = cur ^ 1;
    if (rc < N && !seen[rc]) { seen[rc] = 1; touched.push_back(rc); }
    const int32_t a = ptr[cur], b = ptr[cur + 1];
    if (a == b) break;
    if (b - a == 1) { cur = nbr[a]; continue; }
    int32_t best = -1;
    float bs = 0.f;
    for (int32_t p = a; p < b; ++p) {
      const int32_t v = nbr[p];
      if (old1[v] || (old2 && old2[v]) || seen[v]) continue;
      const float s = scores[eid[p]];
      if (best < 0 || s > bs) { best = v; bs = s; }   // first maximum, as argmax / topk(k=1)
    }
    if (best < 0) break;
    cur = best;
  }
  return true;
}

}  // namespace

extern "C" int gnm_decode_build_adjacency(const int32_t* src, const int32_t* dst, int64_t N, int64_t E,
                                          int32_t* succ_ptr, int32_t* succ_nbr, int32_t* succ_eid,
                                          int32_t* pred_ptr, int32_t* pred_nbr, int32_t* pred_eid) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && N < INT32_MAX && E < INT32_MAX, "decode_build_adjacency: N/E out of int32 range");
  GNM_CHECK_ARG((E == 0 || (src && dst && succ_nbr && succ_eid && pred_nbr && pred_eid)) && succ_ptr && pred_ptr,
                "decode_build_adjacency: null argument");
  for (int64_t k = 0; k < E; ++k)
    if (src[k] < 0 || src[k] >= N || dst[k] < 0 || dst[k] >= N) {
      gnm::set_error("decode_build_adjacency: edge %lld = (%d -> %d) outside [0, %lld)", (long long)k, src[k], dst[k],
                     (long long)N);
      return -2;
    }
  csr_by(src, dst, N, E, succ_ptr, succ_nbr, succ_eid);
  csr_by(dst, src, N, E, pred_ptr, pred_nbr, pred_eid);
  return 0;
}

// One iteration of get_contigs (inference.py:203-250) for nb start edges (start_src[i] -> start_dst[i]):
// forward walk from the head, backward walk from the tail under visited | seen_forward, the walk of the
// greatest reconstructed length wins (first one on ties), the nodes it jumped over are added, and --
// if it has at least len_threshold nodes -- `visited` is updated.  Returns the winning walk's length
// (its nodes in walk_out[0..len)), the caller stops when that is below len_threshold; < 0 on error.
extern "C" int64_t gnm_decode_iteration(int64_t N, const float* scores, const int64_t* prefix_length,
                                        const int64_t* read_length, const int32_t* succ_ptr, const int32_t* succ_nbr,
                                        const int32_t* succ_eid, const int32_t* pred_ptr, const int32_t* pred_nbr,
                                        const int32_t* pred_eid, uint8_t* visited, int nb, const int32_t* start_src,
                                        const int32_t* start_dst, int len_threshold, int32_t* walk_out,
                                        int64_t walk_cap, int64_t* best_length_out) {
  if (!(N > 0 && scores && prefix_length && read_length && succ_ptr && succ_nbr && succ_eid && pred_ptr && pred_nbr &&
        pred_eid && visited && nb > 0 && start_src && start_dst && walk_out && walk_cap > 0)) {
    gnm::set_error("decode_iteration: bad argument");
    return -1;
  }
  const int64_t cap = 2 * N + 2;
  std::vector<uint8_t> seen_f((size_t)N, 0), seen_b((size_t)N, 0);
  std::vector<int32_t> wf, wb, tf, tb, best_walk, best_seen;
  int64_t best_len = -1;
  for (int i = 0; i < nb; ++i) {
    if (start_src[i] < 0 || start_src[i] >= N || start_dst[i] < 0 || start_dst[i] >= N) {
      gnm::set_error("decode_iteration: start edge %d outside the graph", i);
      return -2;
    }
    wf.clear(); wb.clear(); tf.clear(); tb.clear();
    const bool ok = greedy_walk(start_dst[i], scores, succ_ptr, succ_nbr, succ_eid, N, visited, nullptr, seen_f.data(),
                                wf, tf, cap) &&
                    greedy_walk(start_src[i], scores, pred_ptr, pred_nbr, pred_eid, N, visited, seen_f.data(),
                                seen_b.data(), wb, tb, cap);
    if (!ok) {
      gnm::set_error("decode_iteration: walk %d exceeds 2N nodes (a cycle of forced moves; the reference does not "
                     "terminate on this input)", i);
      return -3;
    }
    // walk = reversed(backward) + forward; its length in bases (inference.py:20-28)
    int64_t bases = 0;
    auto edge_of = [&](int32_t a, int32_t b) -> int32_t {
      for (int32_t p = succ_ptr[a]; p < succ_ptr[a + 1]; ++p)
        if (succ_nbr[p] == b) return succ_eid[p];
      return -1;
    };
    bool bad = false;
    auto add = [&](int32_t a, int32_t b) {
      const int32_t k = edge_of(a, b);
      if (k < 0) bad = true; else bases += prefix_length[k];
    };
    for (size_t j = wb.size(); j-- > 1;) add(wb[j], wb[j - 1]);
    if (!wb.empty() && !wf.empty()) add(wb[0], wf[0]);
    for (size_t j = 0; j + 1 < wf.size(); ++j) add(wf[j], wf[j + 1]);
    if (bad) {
      gnm::set_error("decode_iteration: start edge %d (%d -> %d) is not an edge of the graph", i, start_src[i],
                     start_dst[i]);
      return -4;
    }
    bases += read_length[wf.back()];
    if (bases > best_len) {                       // max(): the first of equally long walks
      best_len = bases;
      best_walk.assign(wb.rbegin(), wb.rend());
      best_walk.insert(best_walk.end(), wf.begin(), wf.end());
      best_seen = tf;
      best_seen.insert(best_seen.end(), tb.begin(), tb.end());
    }
    for (int32_t v : tf) seen_f[(size_t)v] = 0;
    for (int32_t v : tb) seen_b[(size_t)v] = 0;
  }
  const int64_t len = (int64_t)best_walk.size();
  if (len > walk_cap) {
    gnm::set_error("decode_iteration: walk of %lld nodes does not fit walk_out (%lld)", (long long)len, (long long)walk_cap);
    return -5;
  }
  std::memcpy(walk_out, best_walk.data(), (size_t)len * sizeof(int32_t));
  if (best_length_out) *best_length_out = best_len;
  if (len >= len_threshold) {
    for (int32_t v : best_seen) visited[(size_t)v] = 1;
    // nodes the walk jumped over: succs[a] & preds[b] for consecutive (a, b), and their complements (:231-239)
    for (int64_t j = 0; j + 1 < len; ++j) {
      const int32_t a = best_walk[(size_t)j], b = best_walk[(size_t)j + 1];
      for (int32_t p = succ_ptr[a]; p < succ_ptr[a + 1]; ++p) {
        const int32_t t = succ_nbr[p];
        bool is_pred = false;
        for (int32_t q = pred_ptr[b]; q < pred_ptr[b + 1] && !is_pred; ++q) is_pred = pred_nbr[q] == t;
        if (is_pred) {
          visited[(size_t)t] = 1;
          if ((t ^ 1) < N) visited[(size_t)(t ^ 1)] = 1;
        }
      }
    }
  }
  return len;
}
