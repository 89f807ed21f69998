// The sweep plan (gnm_graph_build_sweep_plan, gnm_host.cpp) built ON THE DEVICE, for graphs whose index never visits the
// host: the induced sub-graphs of the ClusterGCN mini-batch mode (train.py:288-343 counterpart, cluster.py) and graphs
// built from device tensors.  Same protocol, same words, bit for bit (tests/test_gpu_parity.py compares the two):
//   plan_first_last_k   first / last row of every source (atomic min / max: order-independent)
//   plan_build_k        one WAVE per workgroup of the sweep: lanes 0-15 own the rows of a 16-row tile (who shares my source /
//                       destination: 16 shuffles), the slot allocator -- the only sequential part -- runs wave-uniformly over
//                       the tile's leaders in row order on a stack and a slot -> source table in LDS (LIFO like the host's)
//   plan_fix_list_k     fix_nodes[v] = v for the nodes the plan does not serve, -1 for the others (the *_fix kernels skip
//                       negative entries: no compaction, no count, no host synchronisation)
#include "gnm_tr.h"

namespace gnm {

__global__ void plan_first_last_k(int64_t E, const int32_t* __restrict__ isrc, int32_t* __restrict__ first,
                                  int32_t* __restrict__ last) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < E; j += (int64_t)gridDim.x * blockDim.x) {
    const int32_t s = isrc[j];
    atomicMin(first + s, (int32_t)j);
    atomicMax(last + s, (int32_t)j);
  }
}

constexpr int PW = 4;             // waves (= sweep workgroups planned) per workgroup of this kernel
constexpr int PSL = 64;           // slot capacity of the LDS structures (nslots <= 64 as on the host)

// The allocator state (stk / tab / pend / res) lives in LDS and is written by lane 0 and read by every lane of the SAME wave.  The
// lanes of a wave run in lock step, so this needs no hardware barrier -- but the compiler must not move such a read above the write
// it depends on: a wavefront-scope release / acquire pair around a wave barrier pins the order.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64 * PW) void plan_build_k(
    int64_t N, const int32_t* __restrict__ isrc, const int32_t* __restrict__ idst, const int32_t* __restrict__ in_ptr,
    const int32_t* __restrict__ first, const int32_t* __restrict__ last, int64_t nodes_per_block, int nslots, int64_t margin,
    int64_t nblk, uint32_t* __restrict__ sinfo, uint32_t* __restrict__ dinfo, uint8_t* __restrict__ served,
    int32_t* __restrict__ peak_out) {
  __shared__ int stack_[PW][PSL];      // free slots (LIFO)
  __shared__ int table_[PW][PSL];      // slot -> source that holds it (-1: free)
  __shared__ int pend_[PW][16];        // slots closed in the previous tile (freed at the start of this one, in closing order)
  __shared__ int res_[PW][16];         // per row of the tile: its source's slot, -2 served without a slot, -1 not served
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t w = (int64_t)blockIdx.x * PW + wv;
  if (w >= nblk) return;
  int* const stk = stack_[wv];
  int* const tab = table_[wv];
  int* const pend = pend_[wv];
  int* const res = res_[wv];
  const int64_t v0 = w * nodes_per_block, v1 = v0 + nodes_per_block < N ? v0 + nodes_per_block : N;
  const int64_t rb = in_ptr[v0], re = in_ptr[v1];
  if (lane < PSL) {
    stk[lane] = nslots - 1 - lane;     // the host pushes nslots-1 .. 0: the first pop is slot 0
    tab[lane] = -1;
  }
  wave_lds_sync();
  int nfree = nslots, npend = 0, live = 0, peak = 0, dslot = 0;      // wave-uniform state
  constexpr uint32_t kOpen = kSweepOpen, kClose = kSweepClose;
  // software pipeline (a tile's work is a chain of three dependent global loads otherwise: rows -> first / last / in_ptr):
  // the rows of tile k+2 and the per-row lookups of tile k+1 are in flight while tile k is planned
  auto load_rows = [&](int64_t r0, int32_t& s_, int32_t& d_) __attribute__((always_inline)) {
    const bool ok = r0 < re && lane < kSweepTileRows && r0 + lane < re;
    s_ = ok ? isrc[r0 + lane] : -1 - lane;                       // distinct dummies for the lanes past the tile
    d_ = ok ? idst[r0 + lane] : -1 - lane;
  };
  auto load_look = [&](int32_t s_, int32_t d_, int32_t& f_, int32_t& l_, int32_t& p0_, int32_t& p1_) __attribute__((always_inline)) {
    const bool ok = s_ >= 0;
    f_ = ok ? first[s_] : 0;
    l_ = ok ? last[s_] : 0;
    p0_ = ok ? in_ptr[d_] : 0;
    p1_ = ok ? in_ptr[d_ + 1] : 0;
  };
  int32_t s, d, s1, d1, s2 = 0, d2 = 0, f, l, p0, p1, f1 = 0, l1 = 0, p01 = 0, p11 = 0;
  load_rows(rb, s, d);
  load_rows(rb + kSweepTileRows, s1, d1);
  load_look(s, d, f, l, p0, p1);
  for (int64_t r0 = rb; r0 < re; r0 += kSweepTileRows) {
    const int nv = re - r0 < kSweepTileRows ? (int)(re - r0) : kSweepTileRows;
    const bool valid = lane < nv;
    load_rows(r0 + 2 * kSweepTileRows, s2, d2);
    load_look(s1, d1, f1, l1, p01, p11);
    // ---- slots closed in the previous tile are free again ----
    for (int q = 0; q < npend; ++q) {
      const int sl = pend[q];
      if (lane == 0) {
        stk[nfree + q] = sl;
        tab[sl] = -1;
      }
    }
    nfree += npend;
    live -= npend;
    npend = 0;
    wave_lds_sync();
    uint32_t ms = 0, md = 0;                                     // rows of the tile with my source / my destination
#pragma unroll
    for (int q = 0; q < kSweepTileRows; ++q) {
      ms |= (__shfl(s, q, 64) == s) ? (1u << q) : 0u;
      md |= (__shfl(d, q, 64) == d) ? (1u << q) : 0u;
    }
    const uint32_t below = (1u << (lane & 31)) - 1u;
    const bool lead_s = valid && (ms & below) == 0;
    const bool lead_d = valid && (md & below) == 0;
    // ---- sources ----
    const bool opens = lead_s && f >= r0;
    const bool closes = lead_s && l < r0 + nv;
    const bool inside = opens && f >= rb && l < re && s >= v0 - margin && s < v1 + margin;
    if (lane < kSweepTileRows) res[lane] = -1;
    wave_lds_sync();
    unsigned long long todo = __ballot(lead_s);
    while (todo) {                                               // the tile's leaders in row order (wave-uniform loop)
      const int r = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int32_t sr = __shfl(s, r, 64);
      const bool op = __shfl((int)opens, r, 64), cl = __shfl((int)closes, r, 64), in_ = __shfl((int)inside, r, 64);
      int slot = -1;                                             // -1: not served
      if (op) {
        if (in_) {
          if (cl) {
            slot = -2;                                           // opens and closes in this tile: no slot
          } else if (nfree > 0) {
            slot = stk[nfree - 1];
            --nfree;
            ++live;
            peak = live > peak ? live : peak;
            if (lane == 0) tab[slot] = sr;
          }
        }
      } else {                                                   // opened in an earlier tile: served iff it holds a slot
        const unsigned long long hit = __ballot(lane < PSL && tab[lane] == sr);
        if (hit) slot = __ffsll((long long)hit) - 1;
      }
      if (cl && slot >= 0) {
        if (lane == 0) pend[npend] = slot;
        ++npend;
      }
      if (lane == 0) res[r] = slot;
      wave_lds_sync();                                           // tab / pend / res as lane 0 left them, before the next leader looks
    }
    if (lead_s) {
      const int sl = res[lane];
      if (sl != -1) {
        uint32_t wd = ms;                                        // a leader: every row with its source is at or below it
        if (sl >= 0) wd |= (uint32_t)sl << 16;
        if (opens) {
          wd |= kOpen;
          served[s] = 1;
        }
        if (closes) wd |= kClose;
        sinfo[r0 + lane] = wd;
      }
    }
    // ---- destinations: contiguous runs; a run that crosses the tile boundary carries its sum in one of two slots ----
    {
      bool dop = false, dcl = false;
      if (lead_d) {
        const int cnt = __builtin_popcount(md);
        dop = r0 + lane == p0;
        dcl = r0 + lane + cnt == p1;
      }
      // only the tile's LAST run can stay open; it toggles the slot when it also opened here
      const unsigned long long stay = __ballot(lead_d && !dcl);
      const unsigned long long tog = __ballot(lead_d && !dcl && dop);
      const int dnew = dslot ^ (tog ? 1 : 0);
      if (lead_d) {
        uint32_t wd = md;
        if (!dop) wd |= (uint32_t)dslot << 16;                   // continues the run of the previous tile: its slot
        if (dop) wd |= kOpen;
        if (dcl) wd |= kClose;
        if (!dcl) wd = (wd & ~(63u << 16)) | ((uint32_t)dnew << 16);
        dinfo[r0 + lane] = wd;
      }
      (void)stay;
      dslot = dnew;
    }
    s = s1; d = d1; s1 = s2; d1 = d2;
    f = f1; l = l1; p0 = p01; p1 = p11;
  }
  if (lane == 0 && peak_out) atomicMax(peak_out, peak);
}

}  // namespace gnm

using namespace gnm;

// sinfo / dinfo [E], served [N] bytes, fix_nodes [N] (optional), first / last [N] int32 scratch, peak [1] int32: all DEVICE memory; everything is
// (re)initialised here.  Returns with the kernels queued on `stream`; no host synchronisation.
namespace gnm {
// fix[v] = v for the nodes the plan does not serve, -1 otherwise (the *_fix kernels skip negative entries)
__global__ void plan_fix_list_k(int64_t N, const uint8_t* __restrict__ served, int32_t* __restrict__ fix) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x)
    fix[v] = served[v] ? -1 : (int32_t)v;
}
}  // namespace gnm

extern "C" int gnm_graph_build_sweep_plan_device(const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, int64_t N,
                                                 int64_t E, int64_t nodes_per_block, int tile_rows, int nslots, int64_t margin,
                                                 uint32_t* sinfo, uint32_t* dinfo, uint8_t* served, int32_t* fix_nodes,
                                                 int32_t* first, int32_t* last, int32_t* peak, void* stream) {
  GNM_CHECK_ARG(N > 0 && E > 0 && N < INT32_MAX && E < INT32_MAX && nodes_per_block > 0 && tile_rows == kSweepTileRows &&
                    nslots > 0 && nslots <= PSL && margin >= 0,
                "graph_build_sweep_plan_device: bad extent (tile_rows = 16, nslots <= 64)");
  GNM_CHECK_ARG(isrc && idst && in_ptr && sinfo && dinfo && served && first && last, "graph_build_sweep_plan_device: null argument");
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(sinfo, 0, sizeof(uint32_t) * (size_t)E, st) != hipSuccess) return -2;
  if (hipMemsetAsync(dinfo, 0, sizeof(uint32_t) * (size_t)E, st) != hipSuccess) return -2;
  if (hipMemsetAsync(served, 0, (size_t)N, st) != hipSuccess) return -2;
  if (hipMemsetAsync(first, 0x7f, sizeof(int32_t) * (size_t)N, st) != hipSuccess) return -2;     // 0x7f7f7f7f > any row
  if (hipMemsetAsync(last, 0xff, sizeof(int32_t) * (size_t)N, st) != hipSuccess) return -2;      // -1
  if (peak && hipMemsetAsync(peak, 0, sizeof(int32_t), st) != hipSuccess) return -2;
  int g = (int)((E + 255) / 256);
  if (g > num_cus() * 8) g = num_cus() * 8;
  hipLaunchKernelGGL(plan_first_last_k, dim3(g), dim3(256), 0, st, E, isrc, first, last);
  GNM_LAUNCH_CHECK("plan_first_last");
  const int64_t nblk = (N + nodes_per_block - 1) / nodes_per_block;
  hipLaunchKernelGGL(plan_build_k, dim3((unsigned)((nblk + PW - 1) / PW)), dim3(64 * PW), 0, st, N, isrc, idst, in_ptr,
                     (const int32_t*)first, (const int32_t*)last, nodes_per_block, nslots, margin, nblk, sinfo, dinfo, served, peak);
  GNM_LAUNCH_CHECK("plan_build");
  if (fix_nodes) {
    int gf = (int)((N + 255) / 256);
    if (gf > num_cus() * 8) gf = num_cus() * 8;
    hipLaunchKernelGGL(plan_fix_list_k, dim3(gf), dim3(256), 0, st, N, (const uint8_t*)served, fix_nodes);
    GNM_LAUNCH_CHECK("plan_fix_list");
  }
  return 0;
}
