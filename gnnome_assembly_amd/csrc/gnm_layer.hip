// GatedGCN layer kernels (forward + backward), gfx950.
//
// Replaces the implicit DGL/ATen kernels behind layers/gated_gcn_full.py:120-152 and their
// autograd duals.  All kernels here are HBM-bound gather / segmented-sum / normalise passes:
//   * a row of H floats is owned by G = H/4 lanes holding a float4 each (16 B per lane,
//     one global_load_dwordx4), so a 64-lane wave works on 64/G rows at a time;
//   * segmented sums run one destination (or source) node per wave: the node's edge rows are
//     split over the wave's 64/G sub-groups, accumulated in registers and combined with
//     wavefront xor-shuffles -- no atomics, deterministic;
//   * edges are stored sorted by destination ("internal order"), so the by-destination pass
//     streams its [E,H] operands contiguously and the by-source pass gathers 4H-byte rows;
//   * the grid is persistent (<= 8 workgroups per CU) with one contiguous node/edge chunk per
//     workgroup and an XCD-aware chunk map, so the node rows gathered by neighbouring chunks
//     share an L2;
//   * BatchNorm column statistics are accumulated per lane in fp64 and reduced
//     lane -> wave (shuffles) -> workgroup (LDS) -> per-workgroup partial rows.
#include "gnm_common.h"

namespace gnm {

// -------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------

// t[j] += B1h[isrc j] + B2h[idst j]; partial (sum t, sum t^2).   gated_gcn_full.py:120-121
template <int H>
__global__ __launch_bounds__(kBlock) void edge_t_stats_fwd_k(int64_t E, float* __restrict__ t,
                                                             const float* __restrict__ P,
                                                             const int32_t* __restrict__ isrc,
                                                             const int32_t* __restrict__ idst,
                                                             double* __restrict__ partials,
                                                             int64_t rows_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)chunk * rows_per_block;
  const int64_t r1 = min(E, r0 + rows_per_block);
  Stat4 st;
  st.zero();
  for (int64_t j = r0 + wave * RPW + sub; j < r1; j += kWavesPerBlock * RPW) {
    const int64_t s = isrc[j], d = idst[j];
    float4 v = ld4_nt(t + j * H + c4);
    v = v + ld4(P + s * (5 * H) + 3 * H + c4) + ld4(P + d * (5 * H) + 4 * H + c4);
    st4_nt(t + j * H + c4, v);
    st.add_prod(v, v);
  }
  block_stat_store<H>(st, lds, partials, chunk);
}

// e_out = relu(bn(t)) + e_in ; sigma = sigmoid(e_out) ; by-destination gated mean.
// gated_gcn_full.py:122-130
// RES = false: no residual (GatedGCN_1d(residual=False) or in_channels != out_channels, gated_gcn_full.py:41-42,124-125)
template <int H, bool RES = true>
__global__ __launch_bounds__(kBlock, 8) void edge_gate_fwd_k(int64_t N, const float* __restrict__ t,
                                                          const float* __restrict__ e_in,
                                                          const float* __restrict__ stat,
                                                          const float* __restrict__ P,
                                                          const int32_t* __restrict__ isrc,
                                                          const int32_t* __restrict__ in_ptr,
                                                          float* __restrict__ e_out,
                                                          float* __restrict__ hf,
                                                          float* __restrict__ inv_f,
                                                          int64_t nodes_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = in_ptr[v], b = in_ptr[v + 1];
    float4 num = f4(0.f), den = f4(0.f);
    for (int64_t j = a + sub; j < b; j += RPW) {
      const int64_t s = isrc[j];
      const float4 tt = ld4_nt(t + j * H + c4);
      float4 ee = f4(0.f);
      if constexpr (RES) ee = ld4_nt(e_in + j * H + c4);
      const float4 a2 = ld4(P + s * (5 * H) + H + c4);
      const float4 eo = relu4(fma4(tt, sc, sh)) + ee;
      st4_nt(e_out + j * H + c4, eo);
      const float4 sg = sigmoid4(eo);
      num = fma4(sg, a2, num);
      den += sg;
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      num += shfl_xor4(num, off);
      den += shfl_xor4(den, off);
    }
    if (sub == 0) {
      const float4 inv = make_float4(1.f / (den.x + kEpsDen), 1.f / (den.y + kEpsDen),
                                     1.f / (den.z + kEpsDen), 1.f / (den.w + kEpsDen));
      st4_nt(hf + v * H + c4, num * inv);
      st4_nt(inv_f + v * H + c4, inv);
    }
  }
}

// by-source gated mean on the same gate, z = A1h + hf + hb, partial (sum z, sum z^2).
// gated_gcn_full.py:133-145
template <int H>
__global__ __launch_bounds__(kBlock, 8) void node_agg_src_fwd_k(
    int64_t N, const float* __restrict__ e_out, const float* __restrict__ P,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos,
    const int32_t* __restrict__ out_dst, const float* __restrict__ hf, float* __restrict__ hb,
    float* __restrict__ inv_b, float* __restrict__ z, double* __restrict__ partials,
    int64_t nodes_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  Stat4 st;
  st.zero();
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = out_ptr[v], b = out_ptr[v + 1];
    float4 num = f4(0.f), den = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = out_pos[m], d = out_dst[m];
      const float4 sg = sigmoid4(ld4_nt(e_out + j * H + c4));
      const float4 a3 = ld4(P + d * (5 * H) + 2 * H + c4);
      num = fma4(sg, a3, num);
      den += sg;
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      num += shfl_xor4(num, off);
      den += shfl_xor4(den, off);
    }
    if (sub == 0) {
      const float4 inv = make_float4(1.f / (den.x + kEpsDen), 1.f / (den.y + kEpsDen),
                                     1.f / (den.z + kEpsDen), 1.f / (den.w + kEpsDen));
      const float4 b_ = num * inv;
      const float4 zz = ld4_nt(P + v * (5 * H) + c4) + ld4_nt(hf + v * H + c4) + b_;
      st4_nt(hb + v * H + c4, b_);
      st4_nt(inv_b + v * H + c4, inv);
      st4_nt(z + v * H + c4, zz);
      st.add_prod(zz, zz);
    }
  }
  block_stat_store<H>(st, lds, partials, chunk);
}

// h_out = relu(bn(z)) + h_in.   gated_gcn_full.py:147-152
template <int H, bool RES = true>
__global__ __launch_bounds__(kBlock) void node_update_fwd_k(int64_t N, const float* __restrict__ z,
                                                            const float* __restrict__ stat,
                                                            const float* __restrict__ h_in,
                                                            float* __restrict__ h_out) {
  constexpr int G = H / 4;
  const int64_t total = N * G;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c4 = (int)(i % G) * 4;
    const int64_t o = (i / G) * H + c4;
    const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
    float4 hr = f4(0.f);
    if constexpr (RES) hr = ld4(h_in + o);
    st4(h_out + o, relu4(fma4(ld4(z + o), sc, sh)) + hr);
  }
}


// -------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------

// gw = gh_out * [bn(z) > 0]; partial (sum gw, sum gw*zhat).
template <int H>
__global__ __launch_bounds__(kBlock) void node_bwd_stats_k(int64_t N, const float* __restrict__ z,
                                                           const float* __restrict__ stat,
                                                           const float* __restrict__ gh_out,
                                                           double* __restrict__ partials,
                                                           int64_t rows_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)chunk * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  const float4 mu = ld4(stat + c4), rs = ld4(stat + H + c4);
  const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
  Stat4 st;
  st.zero();
  for (int64_t v = r0 + wave * RPW + sub; v < r1; v += kWavesPerBlock * RPW) {
    const float4 zz = ld4(z + v * H + c4);
    const float4 gw = gate4(fma4(zz, sc, sh), ld4(gh_out + v * H + c4));
    st.add_prod(gw, (zz - mu) * rs);
  }
  block_stat_store<H>(st, lds, partials, chunk);
}

// gz = gamma*rstd*(gw - m1 - zhat*m2) -> gP[:,0:H]; Q[N,2H] = [qf = gz*inv_f | qb = gz*inv_b]
// (the by-destination / by-source passes form rf = qf*hf and rb = qb*hb themselves from the saved hf / hb rows:
//  same number of rows read there, two [N,H] writes and two [N,H] reads fewer here)
template <int H>
__global__ __launch_bounds__(kBlock) void node_bwd_apply_k(
    int64_t N, const float* __restrict__ z, const float* __restrict__ stat,
    const float* __restrict__ bstat, const float* __restrict__ gamma,
    const float* __restrict__ gh_out, const float* __restrict__ inv_f,
    const float* __restrict__ inv_b, float* __restrict__ gP,
    float* __restrict__ Q) {
  constexpr int G = H / 4;
  const int64_t total = N * G;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c4 = (int)(i % G) * 4;
    const int64_t v = i / G;
    const int64_t o = v * H + c4;
    const float4 mu = ld4(stat + c4), rs = ld4(stat + H + c4);
    const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
    const float4 m1 = ld4(bstat + c4), m2 = ld4(bstat + H + c4);
    const float4 c = ld4(gamma + c4) * rs;
    const float4 zz = ld4_nt(z + o);
    const float4 gw = gate4(fma4(zz, sc, sh), ld4_nt(gh_out + o));
    const float4 gz = c * (gw - m1 - ((zz - mu) * rs) * m2);
    st4_nt(gP + v * (5 * H) + c4, gz);
    float* q = Q + v * (2 * H) + c4;
    st4_nt(q, gz * ld4_nt(inv_f + o));
    st4_nt(q + H, gz * ld4_nt(inv_b + o));
  }
}

// by-destination backward pass (internal order).  See gnm.h for the arithmetic.
template <int H>
__global__ __launch_bounds__(kBlock) void edge_bwd_dst_k(
    int64_t N, const float* __restrict__ e_out, const float* __restrict__ t,
    const float* __restrict__ stat, float* __restrict__ ge, const float* __restrict__ P,
    const float* __restrict__ Q, const float* __restrict__ hf, const float* __restrict__ hb,
    const int32_t* __restrict__ isrc,
    const int32_t* __restrict__ in_ptr, float* __restrict__ gP, float* __restrict__ Ud,
    float* __restrict__ Td, double* __restrict__ partials, int64_t nodes_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  const float4 mu = ld4(stat + c4), rs = ld4(stat + H + c4);
  const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
  Stat4 st;
  st.zero();
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = in_ptr[v], b = in_ptr[v + 1];
    float4 a3acc = f4(0.f), ud = f4(0.f), td = f4(0.f);
    if (a < b) {
      const float4 qf_d = ld4_nt(Q + v * (2 * H) + c4);       // this node's own rows: read once here
      const float4 rf_d = qf_d * ld4_nt(hf + v * H + c4);
      const float4 a3_d = ld4_nt(P + v * (5 * H) + 2 * H + c4);
      for (int64_t j = a + sub; j < b; j += RPW) {
        const int64_t s = isrc[j];
        float4 sg, dsg;
        sigmoid_grad4(ld4_nt(e_out + j * H + c4), sg, dsg);
        const float4 a2_s = ld4(P + s * (5 * H) + H + c4);
        const float4 qb_s = ld4(Q + s * (2 * H) + H + c4);
        const float4 rb_s = qb_s * ld4(hb + s * H + c4);
        const float4 gsig = fma4(qf_d, a2_s, fma4(qb_s, a3_d, f4(0.f) - rf_d - rb_s));
        const float4 g = fma4(gsig, dsg, ld4_nt(ge + j * H + c4));
        st4_nt(ge + j * H + c4, g);
        const float4 tt = ld4_nt(t + j * H + c4);
        const float4 gu = gate4(fma4(tt, sc, sh), g);
        const float4 th = (tt - mu) * rs;
        st.add_prod(gu, th);
        a3acc = fma4(sg, qb_s, a3acc);
        ud += gu;
        td += th;
      }
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      a3acc += shfl_xor4(a3acc, off);
      ud += shfl_xor4(ud, off);
      td += shfl_xor4(td, off);
    }
    if (sub == 0) {
      st4_nt(gP + v * (5 * H) + 2 * H + c4, a3acc);
      st4_nt(Ud + v * H + c4, ud);
      st4_nt(Td + v * H + c4, td);
    }
  }
  block_stat_store<H>(st, lds, partials, chunk);
}

// by-source backward pass.  See gnm.h for the arithmetic.
template <int H>
__global__ __launch_bounds__(kBlock, 7) void edge_bwd_src_k(
    int64_t N, const float* __restrict__ e_out, const float* __restrict__ t,
    const float* __restrict__ stat, const float* __restrict__ bstat,
    const float* __restrict__ gamma, const float* __restrict__ ge, const float* __restrict__ Q,
    const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr,
    const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst,
    const float* __restrict__ Ud, const float* __restrict__ Td, float* __restrict__ gP,
    int64_t nodes_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  const float4 mu = ld4(stat + c4), rs = ld4(stat + H + c4);
  const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
  // m1, m2, gamma are only needed once per node: they are re-read there (L1 hits) instead of living in 12 registers
  // across the gather loop -- the loop's occupancy (loads in flight) is what this latency-bound kernel runs on
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = out_ptr[v], b = out_ptr[v + 1];
    float4 a2acc = f4(0.f), us = f4(0.f), ts = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = out_pos[m], d = out_dst[m];
      const float4 sg = sigmoid4(ld4_nt(e_out + j * H + c4));
      const float4 qf_d = ld4(Q + d * (2 * H) + c4);
      const float4 tt = ld4_nt(t + j * H + c4);
      const float4 gu = gate4(fma4(tt, sc, sh), ld4_nt(ge + j * H + c4));
      a2acc = fma4(sg, qf_d, a2acc);
      us += gu;
      ts += (tt - mu) * rs;
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      a2acc += shfl_xor4(a2acc, off);
      us += shfl_xor4(us, off);
      ts += shfl_xor4(ts, off);
    }
    if (sub == 0) {
      const float* bs = bstat;
      const float* gm = gamma;
      asm volatile("" : "+s"(bs), "+s"(gm));     // opaque to the loop-invariant hoisting that would keep them live
      const float4 m1 = ld4(bs + c4), m2 = ld4(bs + H + c4);
      const float4 c = ld4(gm + c4) * rs;
      const float outdeg = (float)(b - a);
      const float indeg = (float)(in_ptr[v + 1] - in_ptr[v]);
      float* g = gP + v * (5 * H) + c4;
      st4_nt(g + H, a2acc);
      st4_nt(g + 3 * H, c * (us - m1 * outdeg - m2 * ts));
      st4_nt(g + 4 * H, c * (ld4_nt(Ud + v * H + c4) - m1 * indeg - m2 * ld4_nt(Td + v * H + c4)));
    }
  }
}

// Two-sided sweep (gnm_edge_bwd_chain_src): the by-source sums of the nodes the sweep plan does NOT serve (sources
// with out-edges in more than one workgroup of the sweep -- chunk boundaries, repeat edges --, and the nodes without
// out-edges), one node of the list per wave, by gathers like edge_bwd_src_k:  RAW sums
//   gP[v][H:2H] = sum_out sigma * Qf[dst],   UT[v] = [ sum_out gu | sum_out that ]
template <int H>
__global__ __launch_bounds__(kBlock, 7) void edge_bwd_src_fix_k(
    int64_t nfix, const int32_t* __restrict__ fix_nodes, const float* __restrict__ e_out, const float* __restrict__ t,
    const float* __restrict__ stat, const float* __restrict__ ge, const float* __restrict__ Q,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos, const int32_t* __restrict__ out_dst,
    float* __restrict__ gP, float* __restrict__ UT) {
  constexpr int G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const float4 mu = ld4(stat + c4), rs = ld4(stat + H + c4);
  const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
  for (int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave; i < nfix; i += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t v = fix_nodes[i];
    if (v < 0) continue;                      // a list compacted on the device carries -1 behind its last entry
    const int a = out_ptr[v], b = out_ptr[v + 1];
    float4 a2acc = f4(0.f), us = f4(0.f), ts = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = out_pos[m], d = out_dst[m];
      const float4 sg = sigmoid4(ld4_nt(e_out + j * H + c4));
      const float4 qf_d = ld4(Q + d * (2 * H) + c4);
      const float4 tt = ld4_nt(t + j * H + c4);
      const float4 gu = gate4(fma4(tt, sc, sh), ld4_nt(ge + j * H + c4));
      a2acc = fma4(sg, qf_d, a2acc);
      us += gu;
      ts += (tt - mu) * rs;
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      a2acc += shfl_xor4(a2acc, off);
      us += shfl_xor4(us, off);
      ts += shfl_xor4(ts, off);
    }
    if (sub == 0) {
      st4_nt(gP + v * (5 * H) + H + c4, a2acc);
      st4_nt(UT + v * (2 * H) + c4, us);
      st4_nt(UT + v * (2 * H) + H + c4, ts);
    }
  }
}

// ... and what edge_bwd_src_k did once the BatchNorm-backward means m1, m2 of the layer are known (linearity):
//   gP[v][3H:4H] = gB1h = c (Us - outdeg m1 - m2 Ts),   gP[v][4H:5H] = gB2h = c (Ud - indeg m1 - m2 Td),  c = gamma rstd
template <int H>
__global__ __launch_bounds__(kBlock) void node_bgrad_k(int64_t N, const float* __restrict__ stat,
                                                       const float* __restrict__ bstat, const float* __restrict__ gamma,
                                                       const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr,
                                                       const float* __restrict__ UT, const float* __restrict__ Ud,
                                                       const float* __restrict__ Td, int64_t ud_pitch, float* __restrict__ gP) {
  constexpr int G = H / 4;
  const int64_t total = N * G;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c4 = (int)(i % G) * 4;
    const int64_t v = i / G;
    const float4 m1 = ld4(bstat + c4), m2 = ld4(bstat + H + c4);
    const float4 c = ld4(gamma + c4) * ld4(stat + H + c4);
    const float outdeg = (float)(out_ptr[v + 1] - out_ptr[v]);
    const float indeg = (float)(in_ptr[v + 1] - in_ptr[v]);
    const float4 us = ld4_nt(UT + v * (2 * H) + c4), ts = ld4_nt(UT + v * (2 * H) + H + c4);
    const float4 ud = ld4_nt(Ud + v * ud_pitch + c4), td = ld4_nt(Td + v * ud_pitch + c4);
    float* g = gP + v * (5 * H) + c4;
    st4_nt(g + 3 * H, c * (us - m1 * outdeg - m2 * ts));
    st4_nt(g + 4 * H, c * (ud - m1 * indeg - m2 * td));
  }
}

// gt = gamma*rstd*(gu - m1 - that*m2)
template <int H>
__global__ __launch_bounds__(kBlock) void edge_bwd_gt_k(int64_t E, const float* __restrict__ ge,
                                                        const float* __restrict__ t,
                                                        const float* __restrict__ stat,
                                                        const float* __restrict__ bstat,
                                                        const float* __restrict__ gamma,
                                                        float* __restrict__ gt) {
  constexpr int G = H / 4;
  const int64_t total = E * G;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c4 = (int)(i % G) * 4;
    const int64_t o = (i / G) * H + c4;
    const float4 mu = ld4(stat + c4), rs = ld4(stat + H + c4);
    const float4 sc = ld4(stat + 2 * H + c4), sh = ld4(stat + 3 * H + c4);
    const float4 m1 = ld4(bstat + c4), m2 = ld4(bstat + H + c4);
    const float4 c = ld4(gamma + c4) * rs;
    const float4 tt = ld4_nt(t + o);
    const float4 gu = gate4(fma4(tt, sc, sh), ld4_nt(ge + o));
    st4_nt(gt + o, c * (gu - m1 - ((tt - mu) * rs) * m2));
  }
}

// -------------------------------------------------------------------------------------------
// BatchNorm statistic finalisation: parallel fp64 reduction of the per-workgroup partial rows
// (fixed order -> deterministic) and the per-channel arithmetic, one launch.
// -------------------------------------------------------------------------------------------
// Reduction AND finalisation in one launch (round 5: two launches, 32 per step, on the critical path of the kernel that needs the
// statistics): a workgroup owns 8 channels = the 16 columns {c, H + c} of the partial rows, sums them in a fixed order (16 row groups x four
// independent chains -- a single dependent chain made this 30 us --, joined through LDS: deterministic; the sums are also left in `sums`) and finalises its channels.
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_finalize_k(const double* __restrict__ partials, int nblk, int H, double inv_count,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, double eps,
                                                             double* __restrict__ sums, float* __restrict__ stat,
                                                             float* __restrict__ ggamma, float* __restrict__ gbeta) {
  __shared__ double red[16][17];
  __shared__ double fin[16];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int ch = blockIdx.x * 8 + (c & 7);
  const int col = (c >> 3) * H + ch, total = 2 * H;
  double acc = 0.0;
  if (ch < H) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double* p = partials + col;
    int b = r;
    for (; b + 48 < nblk; b += 64) {
      a0 += p[(size_t)b * total];
      a1 += p[(size_t)(b + 16) * total];
      a2 += p[(size_t)(b + 32) * total];
      a3 += p[(size_t)(b + 48) * total];
    }
    for (; b < nblk; b += 16) a0 += p[(size_t)b * total];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[r][c] = acc;
  __syncthreads();
  if (r == 0) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][c];
    fin[c] = s;
    if (ch < H) sums[col] = s;
  }
  __syncthreads();
  if (threadIdx.x < 8 && ch < H) {
    const double s1 = fin[threadIdx.x], s2 = fin[8 + threadIdx.x];
    if constexpr (BWD) {
      stat[ch] = (float)(s1 * inv_count);           // bstat
      stat[H + ch] = (float)(s2 * inv_count);
      gbeta[ch] = (float)s1;
      ggamma[ch] = (float)s2;
    } else {
      const double mean = s1 * inv_count;
      double var = s2 * inv_count - mean * mean;   // biased variance, exact sums in fp64
      if (var < 0.0) var = 0.0;
      const float rstd = (float)(1.0 / sqrt(var + eps));
      const float scale = gamma[ch] * rstd;
      stat[ch] = (float)mean;
      stat[H + ch] = rstd;
      stat[2 * H + ch] = scale;
      stat[3 * H + ch] = beta[ch] - (float)mean * scale;
    }
  }
}

}  // namespace gnm

// -------------------------------------------------------------------------------------------
// C ABI
// -------------------------------------------------------------------------------------------
using namespace gnm;

#define GNM_DISPATCH_H(H, ...)                                       \
  switch (H) {                                                       \
    case 32: { constexpr int HH = 32; __VA_ARGS__; } break;          \
    case 64: { constexpr int HH = 64; __VA_ARGS__; } break;          \
    case 128: { constexpr int HH = 128; __VA_ARGS__; } break;        \
    case 256: { constexpr int HH = 256; __VA_ARGS__; } break;        \
    default: ::gnm::set_error("H=%d unsupported (32, 64, 128, 256)", (int)(H)); return -1; \
  }

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ew_grid(int64_t items) {
  int64_t g = ceil_div64(items, kBlock);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int gnm_edge_t_stats_fwd(int64_t E, int H, float* t, const float* P, const int32_t* isrc,
                                    const int32_t* idst, double* partials, int* nblk_out,
                                    void* stream) {
  GNM_CHECK_ARG(E >= 0 && t && P && isrc && idst && partials && nblk_out, "edge_t_stats_fwd: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(E, 256, occ_blocks<edge_t_stats_fwd_k<HH>>());
    const int64_t rpb = ceil_div64(E, grid);
    hipLaunchKernelGGL(edge_t_stats_fwd_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, E, t, P, isrc, idst, partials, rpb);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("edge_t_stats_fwd");
  return 0;
}

extern "C" int gnm_edge_gate_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in,
                                 const float* stat_e, const float* P, const int32_t* isrc,
                                 const int32_t* in_ptr, float* e_out, float* hf, float* inv_f,
                                 void* stream) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && t && stat_e && P && isrc && in_ptr && e_out && hf && inv_f,
                "edge_gate_fwd: null/neg argument");      // e_in == NULL: no residual (residual=False / in != out)
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<edge_gate_fwd_k<HH>>());
    const int64_t npb = ceil_div64(N, grid);
    if (e_in)
      hipLaunchKernelGGL((edge_gate_fwd_k<HH, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, t, e_in, stat_e, P, isrc, in_ptr, e_out, hf, inv_f, npb);
    else
      hipLaunchKernelGGL((edge_gate_fwd_k<HH, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, t, e_in, stat_e, P, isrc, in_ptr, e_out, hf, inv_f, npb);
  });
  GNM_LAUNCH_CHECK("edge_gate_fwd");
  return 0;
}

extern "C" int gnm_node_agg_src_fwd(int64_t N, int64_t E, int H, const float* e_out, const float* P,
                                    const int32_t* out_ptr, const int32_t* out_pos,
                                    const int32_t* out_dst, const float* hf, float* hb, float* inv_b,
                                    float* z, double* partials, int* nblk_out, void* stream) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && e_out && P && out_ptr && out_pos && out_dst && hf && hb && inv_b && z &&
                    partials && nblk_out, "node_agg_src_fwd: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<node_agg_src_fwd_k<HH>>());
    const int64_t npb = ceil_div64(N, grid);
    hipLaunchKernelGGL(node_agg_src_fwd_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, e_out, P, out_ptr, out_pos, out_dst, hf, hb, inv_b, z, partials, npb);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("node_agg_src_fwd");
  return 0;
}

extern "C" int gnm_node_update_fwd(int64_t N, int H, const float* z, const float* stat_h,
                                   const float* h_in, float* h_out, void* stream) {
  GNM_CHECK_ARG(N >= 0 && z && stat_h && h_out, "node_update_fwd: null/neg argument");   // h_in == NULL: no residual
  GNM_DISPATCH_H(H, {
    if (h_in)
      hipLaunchKernelGGL((node_update_fwd_k<HH, true>), dim3(ew_grid(N * (HH / 4))), dim3(kBlock), 0, (hipStream_t)stream, N, z, stat_h, h_in, h_out);
    else
      hipLaunchKernelGGL((node_update_fwd_k<HH, false>), dim3(ew_grid(N * (HH / 4))), dim3(kBlock), 0, (hipStream_t)stream, N, z, stat_h, h_in, h_out);
  });
  GNM_LAUNCH_CHECK("node_update_fwd");
  return 0;
}

extern "C" int gnm_node_bwd_stats(int64_t N, int H, const float* z, const float* stat_h,
                                  const float* gh_out, double* partials, int* nblk_out, void* stream) {
  GNM_CHECK_ARG(N >= 0 && z && stat_h && gh_out && partials && nblk_out, "node_bwd_stats: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 256, occ_blocks<node_bwd_stats_k<HH>>());
    const int64_t rpb = ceil_div64(N, grid);
    hipLaunchKernelGGL(node_bwd_stats_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, z, stat_h, gh_out, partials, rpb);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("node_bwd_stats");
  return 0;
}

extern "C" int gnm_node_bwd_apply(int64_t N, int H, const float* z, const float* stat_h,
                                  const float* bstat_h, const float* gamma_h, const float* gh_out,
                                  const float* inv_f, const float* inv_b, float* gP, float* Q, void* stream) {
  GNM_CHECK_ARG(N >= 0 && z && stat_h && bstat_h && gamma_h && gh_out && inv_f && inv_b && gP && Q,
                "node_bwd_apply: null/neg argument");
  GNM_DISPATCH_H(H, hipLaunchKernelGGL(node_bwd_apply_k<HH>, dim3(ew_grid(N * (HH / 4))), dim3(kBlock),
                                       0, (hipStream_t)stream, N, z, stat_h, bstat_h, gamma_h, gh_out,
                                       inv_f, inv_b, gP, Q));
  GNM_LAUNCH_CHECK("node_bwd_apply");
  return 0;
}

extern "C" int gnm_edge_bwd_dst(int64_t N, int64_t E, int H, const float* e_out, const float* t,
                                const float* stat_e, float* ge, const float* P, const float* Q,
                                const float* hf, const float* hb,
                                const int32_t* isrc, const int32_t* in_ptr, float* gP, float* Ud,
                                float* Td, double* partials, int* nblk_out, void* stream) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && e_out && t && stat_e && ge && P && Q && hf && hb && isrc && in_ptr && gP && Ud && Td &&
                    partials && nblk_out, "edge_bwd_dst: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<edge_bwd_dst_k<HH>>());
    const int64_t npb = ceil_div64(N, grid);
    hipLaunchKernelGGL(edge_bwd_dst_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, e_out, t, stat_e, ge, P, Q, hf, hb, isrc, in_ptr, gP, Ud, Td, partials, npb);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("edge_bwd_dst");
  return 0;
}

extern "C" int gnm_edge_bwd_src(int64_t N, int64_t E, int H, const float* e_out, const float* t,
                                const float* stat_e, const float* bstat_e, const float* gamma_e,
                                const float* ge, const float* Q, const int32_t* in_ptr,
                                const int32_t* out_ptr, const int32_t* out_pos, const int32_t* out_dst,
                                const float* Ud, const float* Td, float* gP, int max_blocks_per_cu, void* stream) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && e_out && t && stat_e && bstat_e && gamma_e && ge && Q && in_ptr && out_ptr &&
                    out_pos && out_dst && Ud && Td && gP && max_blocks_per_cu >= 0, "edge_bwd_src: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<edge_bwd_src_k<HH>>(), max_blocks_per_cu);
    const int64_t npb = ceil_div64(N, grid);
    hipLaunchKernelGGL(edge_bwd_src_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, e_out, t, stat_e, bstat_e, gamma_e, ge, Q, in_ptr, out_ptr, out_pos, out_dst, Ud, Td, gP, npb);
  });
  GNM_LAUNCH_CHECK("edge_bwd_src");
  return 0;
}

extern "C" int gnm_edge_bwd_src_fix(int64_t nfix, const int32_t* fix_nodes, int64_t N, int64_t E, int H,
                                    const float* e_out, const float* t, const float* stat_e, const float* ge,
                                    const float* Q, const int32_t* out_ptr, const int32_t* out_pos,
                                    const int32_t* out_dst, float* gP, float* UT, void* stream) {
  GNM_CHECK_ARG(nfix >= 0 && N >= 0 && E >= 0 && (nfix == 0 || fix_nodes) && e_out && t && stat_e && ge && Q && out_ptr &&
                    out_pos && out_dst && gP && UT, "edge_bwd_src_fix: null/neg argument");
  if (nfix == 0) return 0;
  GNM_DISPATCH_H(H, {
    int64_t grid = ceil_div64(nfix, kWavesPerBlock);
    const int64_t cap = (int64_t)num_cus() * occ_blocks<edge_bwd_src_fix_k<HH>>();
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(edge_bwd_src_fix_k<HH>, dim3((int)grid), dim3(kBlock), 0, (hipStream_t)stream, nfix, fix_nodes,
                       e_out, t, stat_e, ge, Q, out_ptr, out_pos, out_dst, gP, UT);
  });
  GNM_LAUNCH_CHECK("edge_bwd_src_fix");
  return 0;
}

extern "C" int gnm_node_bgrad(int64_t N, int H, const float* stat_e, const float* bstat_e, const float* gamma_e,
                              const int32_t* in_ptr, const int32_t* out_ptr, const float* UT, const float* Ud,
                              const float* Td, int64_t ud_pitch, float* gP, void* stream) {
  GNM_CHECK_ARG(N >= 0 && stat_e && bstat_e && gamma_e && in_ptr && out_ptr && UT && Ud && Td && gP,
                "node_bgrad: null/neg argument");
  GNM_CHECK_ARG(ud_pitch == H || ud_pitch == 2 * H, "node_bgrad: ud_pitch must be H (two [N,H] arrays) or 2H (one [N,2H] array)");
  GNM_DISPATCH_H(H, hipLaunchKernelGGL(node_bgrad_k<HH>, dim3(ew_grid(N * (HH / 4))), dim3(kBlock), 0,
                                       (hipStream_t)stream, N, stat_e, bstat_e, gamma_e, in_ptr, out_ptr, UT, Ud, Td, ud_pitch, gP));
  GNM_LAUNCH_CHECK("node_bgrad");
  return 0;
}

extern "C" int gnm_edge_bwd_gt(int64_t E, int H, const float* ge, const float* t, const float* stat_e,
                               const float* bstat_e, const float* gamma_e, float* gt, void* stream) {
  GNM_CHECK_ARG(E >= 0 && ge && t && stat_e && bstat_e && gamma_e && gt, "edge_bwd_gt: null/neg argument");
  GNM_DISPATCH_H(H, hipLaunchKernelGGL(edge_bwd_gt_k<HH>, dim3(ew_grid(E * (HH / 4))), dim3(kBlock), 0,
                                       (hipStream_t)stream, E, ge, t, stat_e, bstat_e, gamma_e, gt));
  GNM_LAUNCH_CHECK("edge_bwd_gt");
  return 0;
}

// The reduced sums live in the row just past the partial rows: the caller's partials buffer
// holds (gnm_max_partial_blocks() + 1) * 2 * 256 doubles (gnm.h).
extern "C" int gnm_bn_finalize(const double* partials, int nblk, int64_t count, int H,
                               const float* gamma, const float* beta, float eps, float* stat,
                               void* stream) {
  GNM_CHECK_ARG(partials && nblk > 0 && nblk <= kMaxPartialBlocks && count > 0 && H > 0 && H <= 256 && gamma &&
                    beta && stat, "bn_finalize: bad argument");
  double* sums = const_cast<double*>(partials) + (size_t)kMaxPartialBlocks * 2 * 256;
  hipLaunchKernelGGL(bn_reduce_finalize_k<false>, dim3((H + 7) / 8), dim3(256), 0, (hipStream_t)stream, partials, nblk, H,
                     1.0 / (double)count, gamma, beta, (double)eps, sums, stat, (float*)nullptr, (float*)nullptr);
  GNM_LAUNCH_CHECK("bn_finalize");
  return 0;
}

extern "C" int gnm_bn_bwd_finalize(const double* partials, int nblk, int64_t count, int H,
                                   float* bstat, float* ggamma, float* gbeta, void* stream) {
  GNM_CHECK_ARG(partials && nblk > 0 && nblk <= kMaxPartialBlocks && count > 0 && H > 0 && H <= 256 && bstat &&
                    ggamma && gbeta, "bn_bwd_finalize: bad argument");
  double* sums = const_cast<double*>(partials) + (size_t)kMaxPartialBlocks * 2 * 256;
  hipLaunchKernelGGL(bn_reduce_finalize_k<true>, dim3((H + 7) / 8), dim3(256), 0, (hipStream_t)stream, partials, nblk, H,
                     1.0 / (double)count, (const float*)nullptr, (const float*)nullptr, 0.0, sums, bstat, ggamma, gbeta);
  GNM_LAUNCH_CHECK("bn_bwd_finalize");
  return 0;
}
