// LayerNorm mode of the GatedGCN layer (batch_norm=False: gated_gcn_full.py:57-59 picks
// nn.LayerNorm(H) for bn_h / bn_e).  Normalisation is per row over the H channels, so there is
// no global barrier: the row statistics are reduced over the G = H/4 lanes that own the row with
// xor-shuffles, and each backward pass can finish its BatchNorm-free arithmetic in place:
//   * the by-destination pass computes gt = LNbwd(gu) directly, sums it per destination (gB2h)
//     and stores it for the by-source pass and the weight-gradient GEMMs;
//   * node backward is one pass (gz, Q, and the column partials of ggamma / gbeta).
// Same data layout, grid policy and determinism rules as gnm_layer.hip.
#include "gnm_common.h"
#include "gnm_ln.h"

namespace gnm {

// e_out = relu(LN(t)) + e_in ; sigma = sigmoid(e_out) ; by-destination gated mean   (:122-130)
template <int H, bool RES = true>
__global__ __launch_bounds__(kBlock) void ln_edge_gate_fwd_k(
    int64_t N, const float* __restrict__ t, const float* __restrict__ e_in, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ P, const int32_t* __restrict__ isrc,
    const int32_t* __restrict__ in_ptr, float* __restrict__ e_out, float* __restrict__ hf,
    float* __restrict__ inv_f, int64_t nodes_per_block, int width) {
  constexpr int G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const float4 live = live_mask(c4, width);
  const float inv_w = 1.0f / (float)width;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  const float4 ga = ld4(gamma + c4), be = ld4(beta + c4);
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = in_ptr[v], b = in_ptr[v + 1];
    float4 num = f4(0.f), den = f4(0.f);
    for (int64_t j = a + sub; j < b; j += RPW) {
      const int64_t s = isrc[j];
      float rstd;
      const float4 th = row_normalize<H>(ld4_nt(t + j * H + c4), live, inv_w, rstd);
      float4 er_ = f4(0.f);
      if constexpr (RES) er_ = ld4_nt(e_in + j * H + c4);
      const float4 eo = relu4(fma4(th, ga, be)) + er_;
      st4_nt(e_out + j * H + c4, eo);
      const float4 sg = sigmoid4(eo);
      num = fma4(sg, ld4(P + s * (5 * H) + H + c4), num);
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

// h_out = relu(LN(z)) + h_in                                                        (:147-152)
template <int H, bool RES = true>
__global__ __launch_bounds__(kBlock) void ln_node_update_fwd_k(int64_t N, const float* __restrict__ z,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ h_in,
                                                               float* __restrict__ h_out, int width) {
  constexpr int G = H / 4;
  const int64_t total = N * G;     // a multiple of G: the lanes of a row are in range together
  const float inv_w = 1.0f / (float)width;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int c4 = (int)(i % G) * 4;
    const int64_t o = (i / G) * H + c4;
    float rstd;
    const float4 zh = row_normalize<H>(ld4(z + o), live_mask(c4, width), inv_w, rstd);
    float4 hr_ = f4(0.f);
    if constexpr (RES) hr_ = ld4(h_in + o);
    st4(h_out + o, relu4(fma4(zh, ld4(gamma + c4), ld4(beta + c4))) + hr_);
  }
}

// node backward in one pass: gw = gh_out*[LN(z)*g+b > 0]; gz = LNbwd(gw) -> gP[:,0:H];
// Q = [gz*inv_f | gz*inv_f*hf | gz*inv_b | gz*inv_b*hb]; partials (sum gw, sum gw*zhat)
template <int H>
__global__ __launch_bounds__(kBlock) void ln_node_bwd_k(
    int64_t N, const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ gh_out, const float* __restrict__ hf, const float* __restrict__ inv_f,
    const float* __restrict__ hb, const float* __restrict__ inv_b, float* __restrict__ gP,
    float* __restrict__ Q, double* __restrict__ partials, int64_t rows_per_block, int width) {
  constexpr int G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, c4 = (lane % G) * 4;
  const float4 live = live_mask(c4, width);
  const float inv_w = 1.0f / (float)width;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)chunk * rows_per_block;
  const int64_t r1 = min(N, r0 + rows_per_block);
  const float4 ga = ld4(gamma + c4), be = ld4(beta + c4);
  Stat4 st;
  st.zero();
  for (int64_t v = r0 + wave * RPW + sub; v < r1; v += kWavesPerBlock * RPW) {
    const int64_t o = v * H + c4;
    float rstd;
    const float4 zh = row_normalize<H>(ld4(z + o), live, inv_w, rstd);
    const float4 gw = gate4(fma4(zh, ga, be), ld4(gh_out + o));
    st.add_prod(gw, zh);
    const float4 a = ga * gw;
    const float m1 = row_sum<G>(hsum4(a)) * inv_w;
    const float m2 = row_sum<G>(hsum4(a * zh)) * inv_w;
    const float4 gz = (a - f4(m1) * live - zh * m2) * rstd;
    st4(gP + v * (5 * H) + c4, gz);
    const float4 qf = gz * ld4(inv_f + o);
    const float4 qb = gz * ld4(inv_b + o);
    float* q = Q + v * (4 * H) + c4;
    st4(q, qf);
    st4(q + H, qf * ld4(hf + o));
    st4(q + 2 * H, qb);
    st4(q + 3 * H, qb * ld4(hb + o));
  }
  block_stat_store<H>(st, lds, partials, chunk);
}

// by-destination backward: ge <- ge + gsigma*sigma'; gu = ge*[u>0]; gt = LNbwd(gu) -> gt[];
// gP[:,2H:3H][d] = sum sigma*Qb[s]; gP[:,4H:5H][d] = sum gt; partials (sum gu, sum gu*that)
template <int H>
__global__ __launch_bounds__(kBlock) void ln_edge_bwd_dst_k(
    int64_t N, const float* __restrict__ e_out, const float* __restrict__ t, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ ge, const float* __restrict__ P,
    const float* __restrict__ Q, const int32_t* __restrict__ isrc, const int32_t* __restrict__ in_ptr,
    float* __restrict__ gP, float* __restrict__ gt, double* __restrict__ partials, int64_t nodes_per_block, int width) {
  constexpr int G = H / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * H];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const float4 live = live_mask(c4, width);
  const float inv_w = 1.0f / (float)width;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  const float4 ga = ld4(gamma + c4), be = ld4(beta + c4);
  Stat4 st;
  st.zero();
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = in_ptr[v], b = in_ptr[v + 1];
    float4 a3acc = f4(0.f), gtsum = f4(0.f);
    if (a < b) {
      const float4 qf_d = ld4_nt(Q + v * (4 * H) + c4);
      const float4 rf_d = ld4_nt(Q + v * (4 * H) + H + c4);
      const float4 a3_d = ld4_nt(P + v * (5 * H) + 2 * H + c4);
      for (int64_t j = a + sub; j < b; j += RPW) {
        const int64_t s = isrc[j];
        float4 sg, dsg;
        sigmoid_grad4(ld4_nt(e_out + j * H + c4), sg, dsg);
        const float4 a2_s = ld4(P + s * (5 * H) + H + c4);
        const float4 qb_s = ld4(Q + s * (4 * H) + 2 * H + c4);
        const float4 rb_s = ld4(Q + s * (4 * H) + 3 * H + c4);
        const float4 gsig = fma4(qf_d, a2_s, fma4(qb_s, a3_d, f4(0.f) - rf_d - rb_s));
        const float4 g = fma4(gsig, dsg, ld4_nt(ge + j * H + c4));
        st4_nt(ge + j * H + c4, g);
        float rstd;
        const float4 th = row_normalize<H>(ld4_nt(t + j * H + c4), live, inv_w, rstd);
        const float4 gu = gate4(fma4(th, ga, be), g);
        st.add_prod(gu, th);
        const float4 ag = ga * gu;
        const float m1 = row_sum<G>(hsum4(ag)) * inv_w;
        const float m2 = row_sum<G>(hsum4(ag * th)) * inv_w;
        const float4 gtv = (ag - f4(m1) * live - th * m2) * rstd;
        st4_nt(gt + j * H + c4, gtv);
        a3acc = fma4(sg, qb_s, a3acc);
        gtsum += gtv;
      }
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      a3acc += shfl_xor4(a3acc, off);
      gtsum += shfl_xor4(gtsum, off);
    }
    if (sub == 0) {
      st4_nt(gP + v * (5 * H) + 2 * H + c4, a3acc);
      st4_nt(gP + v * (5 * H) + 4 * H + c4, gtsum);
    }
  }
  block_stat_store<H>(st, lds, partials, chunk);
}

// by-source backward: gP[:,H:2H][v] = sum_{out(v)} sigma*Qf[dst]; gP[:,3H:4H][v] = sum_{out(v)} gt
template <int H>
__global__ __launch_bounds__(kBlock) void ln_edge_bwd_src_k(
    int64_t N, const float* __restrict__ e_out, const float* __restrict__ gt, const float* __restrict__ Q,
    const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos,
    const int32_t* __restrict__ out_dst, float* __restrict__ gP, int64_t nodes_per_block) {
  constexpr int G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = out_ptr[v], b = out_ptr[v + 1];
    float4 a2acc = f4(0.f), gts = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = out_pos[m], d = out_dst[m];
      const float4 sg = sigmoid4(ld4_nt(e_out + j * H + c4));
      a2acc = fma4(sg, ld4(Q + d * (4 * H) + c4), a2acc);
      gts += ld4_nt(gt + j * H + c4);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      a2acc += shfl_xor4(a2acc, off);
      gts += shfl_xor4(gts, off);
    }
    if (sub == 0) {
      st4_nt(gP + v * (5 * H) + H + c4, a2acc);
      st4_nt(gP + v * (5 * H) + 3 * H + c4, gts);
    }
  }
}

// the by-source sums of the nodes the sweep plan does not serve (gnm_ln_edge_bwd_top): ln_edge_bwd_src_k over a node list
template <int H>
__global__ __launch_bounds__(kBlock) void ln_edge_bwd_src_fix_k(
    int64_t nfix, const int32_t* __restrict__ fix_nodes, const float* __restrict__ e_out, const float* __restrict__ gt,
    const float* __restrict__ Q, const int32_t* __restrict__ out_ptr, const int32_t* __restrict__ out_pos,
    const int32_t* __restrict__ out_dst, float* __restrict__ gP) {
  constexpr int G = H / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  for (int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave; i < nfix; i += (int64_t)gridDim.x * kWavesPerBlock) {
    const int64_t v = fix_nodes[i];
    if (v < 0) continue;                      // a list compacted on the device carries -1 behind its last entry
    const int a = out_ptr[v], b = out_ptr[v + 1];
    float4 a2acc = f4(0.f), gts = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = out_pos[m], d = out_dst[m];
      const float4 sg = sigmoid4(ld4_nt(e_out + j * H + c4));
      a2acc = fma4(sg, ld4(Q + d * (4 * H) + c4), a2acc);
      gts += ld4_nt(gt + j * H + c4);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      a2acc += shfl_xor4(a2acc, off);
      gts += shfl_xor4(gts, off);
    }
    if (sub == 0) {
      st4_nt(gP + v * (5 * H) + H + c4, a2acc);
      st4_nt(gP + v * (5 * H) + 3 * H + c4, gts);
    }
  }
}

}  // namespace gnm

using namespace gnm;

#define GNM_DISPATCH_H(H, ...)                                       \
  switch (H) {                                                       \
    case 32: { constexpr int HH = 32; __VA_ARGS__; } break;          \
    case 64: { constexpr int HH = 64; __VA_ARGS__; } break;          \
    case 128: { constexpr int HH = 128; __VA_ARGS__; } break;        \
    case 256: { constexpr int HH = 256; __VA_ARGS__; } break;        \
    default: ::gnm::set_error("H=%d unsupported (32, 64, 128, 256)", (int)(H)); return -1; \
  }

static inline int64_t cdivl(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ewgrid(int64_t items) {
  int64_t g = cdivl(items, kBlock);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int gnm_ln_edge_gate_fwd(int64_t N, int64_t E, int H, const float* t, const float* e_in,
                                    const float* gamma, const float* beta, const float* P,
                                    const int32_t* isrc, const int32_t* in_ptr, float* e_out, float* hf,
                                    float* inv_f, int width, void* stream) {
  GNM_CHECK_ARG(width >= 1 && width <= H, "ln_edge_gate_fwd: width must be in [1, H]");
  GNM_CHECK_ARG(N >= 0 && E >= 0 && t && gamma && beta && P && isrc && in_ptr && e_out && hf && inv_f,
                "ln_edge_gate_fwd: null/neg argument");      // e_in == NULL: no residual
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<ln_edge_gate_fwd_k<HH>>());
    if (e_in)
      hipLaunchKernelGGL((ln_edge_gate_fwd_k<HH, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, t, e_in, gamma,
                         beta, P, isrc, in_ptr, e_out, hf, inv_f, cdivl(N, grid), width);
    else
      hipLaunchKernelGGL((ln_edge_gate_fwd_k<HH, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, t, e_in, gamma,
                         beta, P, isrc, in_ptr, e_out, hf, inv_f, cdivl(N, grid), width);
  });
  GNM_LAUNCH_CHECK("ln_edge_gate_fwd");
  return 0;
}

extern "C" int gnm_ln_node_update_fwd(int64_t N, int H, const float* z, const float* gamma, const float* beta,
                                      const float* h_in, float* h_out, int width, void* stream) {
  GNM_CHECK_ARG(N >= 0 && z && gamma && beta && h_out && width >= 1 && width <= H, "ln_node_update_fwd: null/neg argument or width outside [1, H]");   // h_in == NULL: no residual
  GNM_DISPATCH_H(H, {
    if (h_in)
      hipLaunchKernelGGL((ln_node_update_fwd_k<HH, true>), dim3(ewgrid(N * (HH / 4))), dim3(kBlock), 0, (hipStream_t)stream, N, z, gamma, beta, h_in, h_out, width);
    else
      hipLaunchKernelGGL((ln_node_update_fwd_k<HH, false>), dim3(ewgrid(N * (HH / 4))), dim3(kBlock), 0, (hipStream_t)stream, N, z, gamma, beta, h_in, h_out, width);
  });
  GNM_LAUNCH_CHECK("ln_node_update_fwd");
  return 0;
}

extern "C" int gnm_ln_node_bwd(int64_t N, int H, const float* z, const float* gamma, const float* beta,
                               const float* gh_out, const float* hf, const float* inv_f, const float* hb,
                               const float* inv_b, float* gP, float* Q, double* partials, int* nblk_out,
                               int width, void* stream) {
  GNM_CHECK_ARG(width >= 1 && width <= H, "ln_node_bwd: width must be in [1, H]");
  GNM_CHECK_ARG(N >= 0 && z && gamma && beta && gh_out && hf && inv_f && hb && inv_b && gP && Q && partials && nblk_out,
                "ln_node_bwd: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 256, occ_blocks<ln_node_bwd_k<HH>>());
    hipLaunchKernelGGL(ln_node_bwd_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, z, gamma, beta,
                       gh_out, hf, inv_f, hb, inv_b, gP, Q, partials, cdivl(N, grid), width);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("ln_node_bwd");
  return 0;
}

extern "C" int gnm_ln_edge_bwd_dst(int64_t N, int64_t E, int H, const float* e_out, const float* t,
                                   const float* gamma, const float* beta, float* ge, const float* P,
                                   const float* Q, const int32_t* isrc, const int32_t* in_ptr, float* gP,
                                   float* gt, double* partials, int* nblk_out, int width, void* stream) {
  GNM_CHECK_ARG(width >= 1 && width <= H, "ln_edge_bwd_dst: width must be in [1, H]");
  GNM_CHECK_ARG(N >= 0 && E >= 0 && e_out && t && gamma && beta && ge && P && Q && isrc && in_ptr && gP && gt &&
                    partials && nblk_out, "ln_edge_bwd_dst: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<ln_edge_bwd_dst_k<HH>>());
    hipLaunchKernelGGL(ln_edge_bwd_dst_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, e_out, t, gamma,
                       beta, ge, P, Q, isrc, in_ptr, gP, gt, partials, cdivl(N, grid), width);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("ln_edge_bwd_dst");
  return 0;
}

extern "C" int gnm_ln_edge_bwd_src(int64_t N, int64_t E, int H, const float* e_out, const float* gt,
                                   const float* Q, const int32_t* out_ptr, const int32_t* out_pos,
                                   const int32_t* out_dst, float* gP, void* stream) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && e_out && gt && Q && out_ptr && out_pos && out_dst && gP,
                "ln_edge_bwd_src: null/neg argument");
  GNM_DISPATCH_H(H, {
    const int grid = persistent_grid(N, 64, occ_blocks<ln_edge_bwd_src_k<HH>>());
    hipLaunchKernelGGL(ln_edge_bwd_src_k<HH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, e_out, gt, Q,
                       out_ptr, out_pos, out_dst, gP, cdivl(N, grid));
  });
  GNM_LAUNCH_CHECK("ln_edge_bwd_src");
  return 0;
}

extern "C" int gnm_ln_edge_bwd_src_fix(int64_t nfix, const int32_t* fix_nodes, int64_t N, int64_t E, int H, const float* e_out,
                                       const float* gt, const float* Q, const int32_t* out_ptr, const int32_t* out_pos,
                                       const int32_t* out_dst, float* gP, void* stream) {
  GNM_CHECK_ARG(nfix >= 0 && (nfix == 0 || fix_nodes) && N >= 0 && E >= 0 && e_out && gt && Q && out_ptr && out_pos && out_dst && gP,
                "ln_edge_bwd_src_fix: null/neg argument");
  if (nfix == 0) return 0;
  GNM_DISPATCH_H(H, {
    int64_t grid = cdivl(nfix, kWavesPerBlock);
    const int64_t cap = (int64_t)num_cus() * 8;
    if (grid > cap) grid = cap;
    hipLaunchKernelGGL(ln_edge_bwd_src_fix_k<HH>, dim3((int)grid), dim3(kBlock), 0, (hipStream_t)stream, nfix, fix_nodes, e_out, gt, Q,
                       out_ptr, out_pos, out_dst, gP);
  });
  GNM_LAUNCH_CHECK("ln_edge_bwd_src_fix");
  return 0;
}
