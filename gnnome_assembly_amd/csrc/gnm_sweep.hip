// Two-sided forward sweep of one GatedGCN layer (gfx950, H = 128).
//
// The reference aggregates ONE gate twice: by destination on g (gated_gcn_full.py:128-130) and by source on
// dgl.reverse(g) (:133-143).  edge_gate_fwd_k + node_agg_src_fwd_k did that as two passes: the second re-read the whole
// e_out [E,H] only to sum, per source, rows the first pass had on chip.  Here ONE destination-sorted sweep forms both:
//   e_out = relu(bn_e(t)) + e_in;  sigma = sigmoid(e_out)                                             :122-127
//   hf[v] = sum_in  sigma A2h[src] / (sum_in  sigma + 1e-6)   (rows of v are contiguous)              :128-130
//   hb[v] = sum_out sigma A3h[dst] / (sum_out sigma + 1e-6)   (rows of v lie in a few nearby tiles)   :141-143
// through the sweep plan of the graph (gnm_graph_build_sweep_plan): a workgroup owns the rows of a contiguous range of
// destination nodes and walks them in 16-row tiles, thread (row, 4 columns); the per-edge terms of a tile go to three
// [16,128] fp32 images in LDS; then every half-wave serves its own row: if the row LEADS its destination (its source) in
// the tile it adds up the tile's rows of that node, joins the sum carried in the node's accumulator slot and parks it
// there again or -- last tile of the node -- normalises and stores it.  One owner per node and tile, fixed order of the
// additions: deterministic, no atomics.  Nodes the plan does not serve (out-edges in more than one workgroup: chunk
// boundaries, repeat edges; no out-edges) are covered by node_agg_src_fix_k (gathers, a few per cent of the rows);
// z = A1h + hf + hb and the BatchNorm_h column sums by node_z_stats_k.                                 :145-147
#include <type_traits>

#include "gnm_tr.h"
#include "gnm_ln.h"

namespace gnm {

int gate2_wg();          // gnm_fused.hip: gnm_debug_set_variant("gate2_wg", 1 | 2)
constexpr int GT = 512;                     // threads per workgroup (8 waves), two workgroups per CU
constexpr int GR = kSweepTileRows;          // rows per tile
typedef unsigned int u32x4g_ __attribute__((ext_vector_type(4)));

struct Gate2Args {
  int64_t N, E;
  const float* t; const float* e_in; const float* stat; const float* P;
  const int32_t* isrc; const int32_t* idst; const int32_t* in_ptr;
  const uint32_t* sinfo; const uint32_t* dinfo;
  float* e_out; float* hf; float* inv_f; float* hb; float* inv_b;
  int64_t nodes_per_block, margin;
  const float* gamma; const float* beta; int width;      // LN: the edge LayerNorm's affine pair and the layer's real width (stat unused)
};

constexpr int G2_LDS = 3 * GR * SW * 4 + 2 * kSweepSlots * SW * 4 + 2 * 2 * SW * 4 + 2 * SW * 4 + 3 * 4 * GR * 4;

__device__ __forceinline__ u32x4g_ bits4g(const float4& v) {
  const u32x4g_ b = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y),
                     __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
  return b;
}

// INV = false (a forward under no_grad): inv_f / inv_b -- which only the backward reads -- are not stored
// HF = the row pitch of every tensor (floats) = the layer's full width: 128, or 256 with the kernel run once per 128-column
// half (the sweep is column-separable; the caller offsets every pointer by the half's first column)
// LN (round 5, batch_norm = False, HF = 128): e_out = relu(LayerNorm(t)) + e_in with the row statistics taken by the 32 lanes that hold the
// row (row_normalize of gnm_ln.h: the expressions of ln_edge_gate_fwd_k, bit-identical e_out) -- LayerNorm has no global barrier, so the
// sweep needs nothing else changed.
template <bool RES, bool INV, int HF, bool LN = false>
__global__ __launch_bounds__(GT, 4) void edge_gate2_fwd_k(const Gate2Args a) {
  static_assert(!LN || HF == SW, "the LayerNorm sweep is built for 128-wide layers");
  __shared__ __attribute__((aligned(16))) unsigned char lds[G2_LDS];
  float* i1 = reinterpret_cast<float*>(lds);          // sigma * A2h[src]
  float* i2 = i1 + GR * SW;                            // sigma
  float* i3 = i2 + GR * SW;                            // sigma * A3h[dst]
  float* sslots = i3 + GR * SW;                        // [2 sums][kSweepSlots][128]
  float* dslots = sslots + 2 * kSweepSlots * SW;       // [2 sums][2][128]
  float* cs = dslots + 2 * 2 * SW;                     // scale, shift
  int* ring = reinterpret_cast<int*>(cs + 2 * SW);     // 3 tiles x [src | dst | sinfo | dinfo] x 16
  const int tid = threadIdx.x;
  const int row = tid >> 5, c4 = (tid & 31) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * a.nodes_per_block < a.N ? (int64_t)chunk * a.nodes_per_block : a.N;
  const int64_t v1 = v0 + a.nodes_per_block < a.N ? v0 + a.nodes_per_block : a.N;
  const int64_t rb = a.in_ptr[v0], re = a.in_ptr[v1];
  const int64_t ntile = (re - rb + GR - 1) / GR;
  for (int c = tid; c < SW; c += GT) {
    cs[c] = LN ? a.gamma[c] : a.stat[2 * HF + c];
    cs[SW + c] = LN ? a.beta[c] : a.stat[3 * HF + c];
  }
  // outputs as buffers: rows outside the range (a lane that has nothing to store carries offset 0x80000000) are dropped
  const int64_t vbase = v0 - a.margin;
  const int nspan = (int)(v1 - v0 + 2 * a.margin);
  const __amdgpu_buffer_rsrc_t rs_e = __builtin_amdgcn_make_buffer_rsrc(a.e_out + rb * HF, 0, (int)(re - rb) * HF * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hf = __builtin_amdgcn_make_buffer_rsrc(a.hf + v0 * HF, 0, (int)(v1 - v0) * HF * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_if = __builtin_amdgcn_make_buffer_rsrc(INV ? a.inv_f + v0 * HF : a.hf, 0, INV ? (int)(v1 - v0) * HF * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_hb = __builtin_amdgcn_make_buffer_rsrc(a.hb + vbase * HF, 0, nspan * HF * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ib = __builtin_amdgcn_make_buffer_rsrc(INV ? a.inv_b + vbase * HF : a.hb, 0, INV ? nspan * HF * 4 : 0, 0x00020000);
  __syncthreads();
  if (ntile == 0) return;

  float4 ptA, peA = f4(0.f), ptB, peB = f4(0.f), ptC, peC = f4(0.f);   // rows of t, e_in: three tiles in flight
  float4 ga2, ga3;                           // A2h[src], A3h[dst] of this thread's edge
  int fs = 0, fd = 0;                        // indices / plan words of this thread's row two tiles ahead
  unsigned fi = 0, fj = 0;
  const int64_t klast = ntile - 1;
  auto clamp_row = [&](int64_t k) __attribute__((always_inline)) {
    const int64_t left = re - (rb + k * GR);
    const int nv = left < GR ? (int)left : GR;
    return row < nv ? row : nv - 1;
  };
  auto prefetch_idx = [&](int64_t k) __attribute__((always_inline)) {
    const int cr = clamp_row(k);
    const int64_t r = rb + k * GR + cr;
    fs = a.isrc[r];
    fd = a.idst[r];
    const unsigned w1 = a.sinfo[r], w2 = a.dinfo[r];
    fi = cr == row ? w1 : 0u;
    fj = cr == row ? w2 : 0u;
  };
  auto prefetch_rows = [&](int64_t k, float4& pt, float4& pe_) __attribute__((always_inline)) {
    const int64_t o = (rb + k * GR + clamp_row(k)) * HF + c4;
    pt = ld4_nt(a.t + o);
    if constexpr (RES) pe_ = ld4_nt(a.e_in + o);
  };
  auto gather = [&](int64_t s, int64_t d) __attribute__((always_inline)) {
    ga2 = ld4(a.P + s * (5 * HF) + HF + c4);
    ga3 = ld4(a.P + d * (5 * HF) + 2 * HF + c4);
  };
  auto ring_put = [&](int64_t k) __attribute__((always_inline)) {
    if ((tid & 31) == 0) {
      int* rk = ring + (int)(k % 3) * 4 * GR;
      rk[row] = fs;
      rk[GR + row] = fd;
      rk[2 * GR + row] = (int)fi;
      rk[3 * GR + row] = (int)fj;
    }
  };
  prefetch_idx(0);
  ring_put(0);
  const int s0 = fs, d0 = fd;
  prefetch_idx(klast < 1 ? klast : 1);
  prefetch_rows(0, ptA, peA);
  prefetch_rows(klast < 1 ? klast : 1, ptB, peB);
  prefetch_rows(klast < 2 ? klast : 2, ptC, peC);
  gather(s0, d0);
  const float4 sc = ld4(cs + c4), sh = ld4(cs + SW + c4);
  const float4 ln_live = LN ? live_mask(c4, a.width) : f4(1.f);
  const float ln_inv_w = LN ? 1.0f / (float)a.width : 0.f;
  // one tile; its rows are in (pt, pe_), which are refilled with the rows of tile k + 3 behind the first barrier: the row
  // streams are requested THREE tiles ahead (one tile's run sums + barriers are shorter than the HBM latency under load)
  auto tile = [&](int64_t k, float4& pt, float4& pe_) __attribute__((always_inline)) {
    const int64_t r0 = rb + k * GR;
    const int nvalid = re - r0 < GR ? (int)(re - r0) : GR;
    const int* rk = ring + (int)(k % 3) * 4 * GR;
    // ---- per-edge arithmetic of this thread's row: e_out, sigma, the three images ----
    {
      const bool live = row < nvalid;
      float4 x = pt;
      if constexpr (LN) {
        float rstd;
        x = row_normalize<SW>(pt, ln_live, ln_inv_w, rstd);
      }
      const float4 eo = relu4(fma4(x, sc, sh)) + pe_;
      __builtin_amdgcn_raw_buffer_store_b128(bits4g(eo), rs_e, live ? (int)(((k * GR + row) * HF + c4) * 4) : (int)0x80000000, 0, 2);
      const float4 sg = live ? sigmoid4(eo) : f4(0.f);
      st4(i1 + row * SW + c4, sg * ga2);
      st4(i2 + row * SW + c4, sg);
      st4(i3 + row * SW + c4, sg * ga3);
      ring_put(k + 1);                             // indices of tile k+1 (requested a tile ago)
    }
    __syncthreads();
    prefetch_idx(k + 2 < klast ? k + 2 : klast);
    prefetch_rows(k + 3 < klast ? k + 3 : klast, pt, pe_);
    {                                              // the next tile's node rows (its indices went to the ring before the barrier): a whole
      const int* rn = ring + (int)((k + 1) % 3) * 4 * GR;      // run-sum phase ahead of their use
      gather(rn[row], rn[GR + row]);
    }
    // ---- run sums: this half-wave's row as the leader of its destination AND of its source.  Every LDS read both need is
    // issued up front -- the leader's own row (unconditional), the slots (unconditional), the next three rows of the
    // destination run (a destination has ~5 contiguous rows; predicated on the run length) -- so that the phase is two LDS
    // round trips long; loops only for what is left ----
    {
      const unsigned wd = (unsigned)rk[3 * GR + row], ws_ = (unsigned)rk[2 * GR + row];
      const unsigned md = wd & 0xffffu;
      unsigned ms = ws_ & 0xffffu;
      const int nd = __builtin_popcount(md);                   // rows row .. row + nd - 1
      float4 dn_ = ld4(i1 + row * SW + c4), dd = ld4(i2 + row * SW + c4);
      float4 sn_ = ld4(i3 + row * SW + c4), sdn = dd;
      float* pd = dslots + ((wd >> 16) & 1u) * SW + c4;
      float* ps = sslots + ((ws_ >> 16) & (kSweepSlots - 1)) * SW + c4;
      const float4 e1 = ld4(pd), e2 = ld4(pd + 2 * SW);
      const float4 q1 = ld4(ps), q2 = ld4(ps + kSweepSlots * SW);
      float4 x1[3] = {f4(0.f), f4(0.f), f4(0.f)}, x2[3] = {f4(0.f), f4(0.f), f4(0.f)};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j + 1 < nd) {                                        // only the leader of a run reads its rows
          x1[j] = ld4(i1 + (row + 1 + j) * SW + c4);
          x2[j] = ld4(i2 + (row + 1 + j) * SW + c4);
        }
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        dn_ += x1[j];
        dd += x2[j];
      }
      for (int j = 4; j < nd; ++j) {                             // a destination with more than four rows in this tile
        dn_ += ld4(i1 + (row + j) * SW + c4);
        dd += ld4(i2 + (row + j) * SW + c4);
      }
      ms &= ms - 1;
      while (ms) {                                               // a source with several edges into this tile
        const int b = __builtin_ctz(ms);
        ms &= ms - 1;
        sn_ += ld4(i3 + b * SW + c4);
        sdn += ld4(i2 + b * SW + c4);
      }
      const bool ld_ = md != 0, ls_ = (ws_ & 0xffffu) != 0;
      if (ld_ && !(wd & kSweepOpen)) { dn_ += e1; dd += e2; }
      if (ls_ && !(ws_ & kSweepOpen)) { sn_ += q1; sdn += q2; }
      if (ld_ && !(wd & kSweepClose)) { st4(pd, dn_); st4(pd + 2 * SW, dd); }
      if (ls_ && !(ws_ & kSweepClose)) { st4(ps, sn_); st4(ps + kSweepSlots * SW, sdn); }
      const bool outd = ld_ && (wd & kSweepClose), outs = ls_ && (ws_ & kSweepClose);
      if (__builtin_amdgcn_ballot_w64(outd) != 0) {
        const float4 inv = make_float4(1.f / (dd.x + kEpsDen), 1.f / (dd.y + kEpsDen), 1.f / (dd.z + kEpsDen), 1.f / (dd.w + kEpsDen));
        const int o = outd ? ((rk[GR + row] - (int)v0) * HF + c4) * 4 : (int)0x80000000;
        __builtin_amdgcn_raw_buffer_store_b128(bits4g(dn_ * inv), rs_hf, o, 0, 2);
        if constexpr (INV) __builtin_amdgcn_raw_buffer_store_b128(bits4g(inv), rs_if, o, 0, 2);
      }
      if (__builtin_amdgcn_ballot_w64(outs) != 0) {
        const float4 inv = make_float4(1.f / (sdn.x + kEpsDen), 1.f / (sdn.y + kEpsDen), 1.f / (sdn.z + kEpsDen), 1.f / (sdn.w + kEpsDen));
        const int o = outs ? ((rk[row] - (int)vbase) * HF + c4) * 4 : (int)0x80000000;
        __builtin_amdgcn_raw_buffer_store_b128(bits4g(sn_ * inv), rs_hb, o, 0, 2);
        if constexpr (INV) __builtin_amdgcn_raw_buffer_store_b128(bits4g(inv), rs_ib, o, 0, 2);
      }
    }
    __syncthreads();                               // the images are free again
  };
  for (int64_t k = 0; k < ntile; k += 3) {
    tile(k, ptA, peA);
    if (k + 1 < ntile) tile(k + 1, ptB, peB);
    if (k + 2 < ntile) tile(k + 2, ptC, peC);
  }
}

// hf / inv_f of the nodes WITHOUT in-edges (the sweep only ever stores to nodes that own rows): 0 and 1 / 1e-6
template <int HF>
__global__ __launch_bounds__(256) void gate2_empty_segments_k(int64_t N, const int32_t* __restrict__ in_ptr,
                                                              float* __restrict__ hf, float* __restrict__ inv_f) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  const int64_t stride = (int64_t)gridDim.x * 4 * 64;
  for (int64_t base = wave0; base < N; base += stride) {
    const int64_t v = base + lane;
    const bool empty = v < N && in_ptr[v + 1] == in_ptr[v];
    unsigned long long m = __ballot(empty);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int64_t u = base + b;
      const int c4 = (lane & 31) * 4;
      if (lane < 32) st4(hf + u * HF + c4, f4(0.f));
      else if (inv_f) st4(inv_f + u * HF + c4, f4(1.f / kEpsDen));
    }
  }
}

// hb / inv_b of the nodes the sweep plan does not serve: node_agg_src_fwd_k's gathers over a node list
template <int HF>
__global__ __launch_bounds__(kBlock, 8) void node_agg_src_fix_k(int64_t nfix, const int32_t* __restrict__ fix_nodes,
                                                                const float* __restrict__ e_out, const float* __restrict__ P,
                                                                const int32_t* __restrict__ out_ptr,
                                                                const int32_t* __restrict__ out_pos,
                                                                const int32_t* __restrict__ out_dst, float* __restrict__ hb,
                                                                float* __restrict__ inv_b) {
  constexpr int H = SW, G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  for (int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave; i < nfix; i += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t v = fix_nodes[i];
    if (v < 0) continue;                      // a list compacted on the device carries -1 behind its last entry
    const int a = out_ptr[v], b = out_ptr[v + 1];
    float4 num = f4(0.f), den = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = out_pos[m], d = out_dst[m];
      const float4 sg = sigmoid4(ld4_nt(e_out + j * HF + c4));
      num = fma4(sg, ld4(P + d * (5 * HF) + 2 * HF + c4), num);
      den += sg;
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      num += shfl_xor4(num, off);
      den += shfl_xor4(den, off);
    }
    if (sub == 0) {
      const float4 inv = make_float4(1.f / (den.x + kEpsDen), 1.f / (den.y + kEpsDen), 1.f / (den.z + kEpsDen),
                                     1.f / (den.w + kEpsDen));
      st4_nt(hb + v * HF + c4, num * inv);
      if (inv_b) st4_nt(inv_b + v * HF + c4, inv);
    }
  }
}

// z = A1h + hf + hb, partial (sum z, sum z^2)                                     gated_gcn_full.py:145-147
template <int HF>
__global__ __launch_bounds__(kBlock) void node_z_stats_k(int64_t N, const float* __restrict__ P, const float* __restrict__ hf,
                                                         const float* __restrict__ hb, float* __restrict__ z,
                                                         double* __restrict__ partials, int64_t rows_per_block) {
  constexpr int H = SW, G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)chunk * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  Stat4 st;
  st.zero();
  for (int64_t v = r0 + wave * RPW + sub; v < r1; v += kWavesPerBlock * RPW) {
    const float4 zz = ld4_nt(P + v * (5 * HF) + c4) + ld4_nt(hf + v * HF + c4) + ld4_nt(hb + v * HF + c4);
    st4_nt(z + v * HF + c4, zz);
    st.add_prod(zz, zz);
  }
  block_stat_store<H, HF>(st, lds, partials, chunk);
}

}  // namespace gnm

using namespace gnm;

extern "C" int gnm_sweep_partition(int64_t N, int wg_per_cu, int64_t* nodes_per_block, int* grid_out) {
  GNM_CHECK_ARG(N > 0 && nodes_per_block && wg_per_cu >= 1 && wg_per_cu <= 8, "sweep_partition: bad argument");
  const int grid = persistent_grid(N, 64, wg_per_cu, 8);     // call cap 8 >= wg_per_cu: the process-wide occupancy knob does not apply
  *nodes_per_block = (N + grid - 1) / grid;
  if (grid_out) *grid_out = grid;
  return 0;
}

// ln_gamma != NULL: the LayerNorm form (stat_e unused; H = 128)
static int edge_gate2_fwd_impl(int64_t N, int64_t E, int H, const float* t, const float* e_in, const float* stat_e,
                               const float* ln_gamma, const float* ln_beta, int ln_width,
                               const float* P, const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr,
                               const uint32_t* sinfo, const uint32_t* dinfo, int64_t plan_nodes_per_block,
                               int64_t nfix, const int32_t* fix_nodes, const int32_t* out_ptr, const int32_t* out_pos,
                               const int32_t* out_dst, float* e_out, float* hf, float* inv_f, float* hb, float* inv_b,
                               float* z, double* partials, int* nblk_out, void* stream) {
  GNM_CHECK_ARG(H == SW || (H == 2 * SW && !ln_gamma), "edge_gate2_fwd: H=%d (128 and 256 are built; LayerNorm: 128)", H);
  GNM_CHECK_ARG(N > 0 && E > 0 && t && (stat_e || ln_gamma) && P && isrc && idst && in_ptr && sinfo && dinfo && (nfix == 0 || fix_nodes) &&
                    nfix >= 0 && out_ptr && out_pos && out_dst && e_out && hf && hb && z && partials && nblk_out &&
                    (inv_f != nullptr) == (inv_b != nullptr),
                "edge_gate2_fwd: null/neg argument");      // e_in == NULL: no residual; inv_f == inv_b == NULL: not wanted (no backward)
  hipStream_t st = (hipStream_t)stream;
  Gate2Args a{};
  a.N = N; a.E = E; a.t = t; a.e_in = e_in; a.stat = stat_e; a.P = P; a.isrc = isrc; a.idst = idst; a.in_ptr = in_ptr;
  a.sinfo = sinfo; a.dinfo = dinfo; a.e_out = e_out; a.hf = hf; a.inv_f = inv_f; a.hb = hb; a.inv_b = inv_b;
  a.margin = kSweepMargin;
  a.gamma = ln_gamma; a.beta = ln_beta; a.width = ln_width;
  int grid = 0;
  gnm_sweep_partition(N, gate2_wg(), &a.nodes_per_block, &grid);
  GNM_CHECK_ARG(plan_nodes_per_block == a.nodes_per_block, "edge_gate2_fwd: the sweep plan was built for %lld nodes per workgroup, the "
                "kernel uses %lld (gnm_sweep_partition(N, %d))", (long long)plan_nodes_per_block, (long long)a.nodes_per_block, gate2_wg());
  // 32-bit buffer offsets: the rows of one workgroup (x 4 H bytes) and its node range + margins (x 4 H bytes)
  GNM_CHECK_ARG((a.nodes_per_block + 2 * kSweepMargin) * H * 4 < (int64_t)INT32_MAX && (E / grid + 64) * H * 4 < (int64_t)INT32_MAX,
                "edge_gate2_fwd: a workgroup's share exceeds the 32-bit buffer offsets");
  const int gz = H == SW ? persistent_grid(N, 256, occ_blocks<node_z_stats_k<SW>>()) : persistent_grid(N, 256, occ_blocks<node_z_stats_k<2 * SW>>());
  auto run = [&](auto hf_tag, int c0) -> int {
    constexpr int HF = decltype(hf_tag)::value;
    Gate2Args b = a;              // this 128-column problem: every pointer starts at the half's first column
    b.t = t + c0; b.e_in = e_in ? e_in + c0 : nullptr; b.stat = stat_e ? stat_e + c0 : nullptr; b.P = P + c0; b.e_out = e_out + c0; b.hf = hf + c0;
    b.inv_f = inv_f ? inv_f + c0 : nullptr; b.hb = hb + c0; b.inv_b = inv_b ? inv_b + c0 : nullptr;
    hipLaunchKernelGGL(gate2_empty_segments_k<HF>, dim3(num_cus() * 2), dim3(256), 0, st, N, in_ptr, b.hf, b.inv_f);
    auto launch = [&](auto ln_tag) {
      constexpr bool LN_ = decltype(ln_tag)::value;
      if (e_in && inv_f) hipLaunchKernelGGL((edge_gate2_fwd_k<true, true, HF, LN_>), dim3(grid), dim3(GT), 0, st, b);
      else if (e_in) hipLaunchKernelGGL((edge_gate2_fwd_k<true, false, HF, LN_>), dim3(grid), dim3(GT), 0, st, b);
      else if (inv_f) hipLaunchKernelGGL((edge_gate2_fwd_k<false, true, HF, LN_>), dim3(grid), dim3(GT), 0, st, b);
      else hipLaunchKernelGGL((edge_gate2_fwd_k<false, false, HF, LN_>), dim3(grid), dim3(GT), 0, st, b);
    };
    if constexpr (HF == SW) {
      if (ln_gamma) launch(std::true_type{});
      else launch(std::false_type{});
    } else {
      launch(std::false_type{});
    }
    GNM_LAUNCH_CHECK("edge_gate2_fwd");
    if (nfix > 0) {
      int64_t g2 = (nfix + kWavesPerBlock - 1) / kWavesPerBlock;
      const int64_t cap = (int64_t)num_cus() * 8;
      if (g2 > cap) g2 = cap;
      hipLaunchKernelGGL(node_agg_src_fix_k<HF>, dim3((int)g2), dim3(kBlock), 0, st, nfix, fix_nodes, b.e_out, b.P, out_ptr, out_pos,
                         out_dst, b.hb, b.inv_b);
      GNM_LAUNCH_CHECK("edge_gate2_fwd fix");
    }
    hipLaunchKernelGGL(node_z_stats_k<HF>, dim3(gz), dim3(kBlock), 0, st, N, b.P, b.hf, b.hb, z + c0, partials + c0, (N + gz - 1) / gz);
    GNM_LAUNCH_CHECK("edge_gate2_fwd z");
    return 0;
  };
  if (H == SW) {
    if (run(std::integral_constant<int, SW>{}, 0)) return -2;
  } else {
    for (int c0 = 0; c0 < H; c0 += SW)
      if (run(std::integral_constant<int, 2 * SW>{}, c0)) return -2;
  }
  *nblk_out = gz;
  return 0;
}

extern "C" int gnm_edge_gate2_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in, const float* stat_e,
                                  const float* P, const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr,
                                  const uint32_t* sinfo, const uint32_t* dinfo, int64_t plan_nodes_per_block,
                                  int64_t nfix, const int32_t* fix_nodes, const int32_t* out_ptr, const int32_t* out_pos,
                                  const int32_t* out_dst, float* e_out, float* hf, float* inv_f, float* hb, float* inv_b,
                                  float* z, double* partials, int* nblk_out, void* stream) {
  GNM_CHECK_ARG(stat_e, "edge_gate2_fwd: stat_e is null");
  return edge_gate2_fwd_impl(N, E, H, t, e_in, stat_e, nullptr, nullptr, 0, P, isrc, idst, in_ptr, sinfo, dinfo, plan_nodes_per_block, nfix,
                             fix_nodes, out_ptr, out_pos, out_dst, e_out, hf, inv_f, hb, inv_b, z, partials, nblk_out, stream);
}

// The LayerNorm form (batch_norm = False, H = 128; width = the layer's real out_channels <= H): gnm_ln_edge_gate_fwd + gnm_node_agg_src_fwd
// in one sweep.  z = A1h + hf + hb as in the BatchNorm form (the column sums in `partials` are computed and not needed).
extern "C" int gnm_ln_edge_gate2_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in, const float* gamma_e,
                                     const float* beta_e, int width, const float* P, const int32_t* isrc, const int32_t* idst,
                                     const int32_t* in_ptr, const uint32_t* sinfo, const uint32_t* dinfo, int64_t plan_nodes_per_block,
                                     int64_t nfix, const int32_t* fix_nodes, const int32_t* out_ptr, const int32_t* out_pos,
                                     const int32_t* out_dst, float* e_out, float* hf, float* inv_f, float* hb, float* inv_b,
                                     float* z, double* partials, int* nblk_out, void* stream) {
  GNM_CHECK_ARG(gamma_e && beta_e && width >= 1 && width <= H, "ln_edge_gate2_fwd: null argument or width outside [1, H]");
  return edge_gate2_fwd_impl(N, E, H, t, e_in, nullptr, gamma_e, beta_e, width, P, isrc, idst, in_ptr, sinfo, dinfo, plan_nodes_per_block, nfix,
                             fix_nodes, out_ptr, out_pos, out_dst, e_out, hf, inv_f, hb, inv_b, z, partials, nblk_out, stream);
}
