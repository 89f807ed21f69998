// Edge-feature encoder (full_graph.py:24-26): e0 = linear2_edge(relu(linear1_edge(e_raw))),
// forward and backward, for the reference's sizes edge_features = 2, hidden_edge_features = 16
// (hyperparameters.py:9-10) and H = 128.  Both are single HBM passes over the [E,H] tensor:
//   forward  writes e0 once (the 2 -> 16 -> 128 arithmetic is 24 FMA per output element, VALU);
//   backward reads  ge0 once and produces every encoder gradient: the 16-wide hidden gradient
//            needs a sum over the 128 channels of a row, done as a reduce-scatter over the 32
//            lanes that own the row (16 xor-shuffles instead of 80 for an all-reduce); lane l
//            ends up owning hidden unit q = (l>>1)&15.
// Rows are in internal (destination-sorted) order; e_raw is read through `perm`.
#include "gnm_common.h"

namespace gnm {

constexpr int EH = 128;      // hidden width
constexpr int EQ = 16;       // hidden_edge_features
constexpr int EG = EH / 4;   // lanes per row
constexpr int ENP = EH * EQ + EH + 2 * EQ + EQ;   // gW2 | gb2 | gW1 | gb1 = 2224 partial sums

__global__ __launch_bounds__(kBlock) void edge_encoder_fwd_k(int64_t E, const float* __restrict__ e_raw,
                                                             const int32_t* __restrict__ perm,
                                                             const float* __restrict__ W1,
                                                             const float* __restrict__ b1,
                                                             const float* __restrict__ W2,
                                                             const float* __restrict__ b2,
                                                             float* __restrict__ e0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / EG, c4 = (lane % EG) * 4;
  float w2[4][EQ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < EQ; ++q) w2[i][q] = W2[(c4 + i) * EQ + q];
  const float4 bb = ld4(b2 + c4);
  float w1a[EQ], w1b[EQ], bq[EQ];
#pragma unroll
  for (int q = 0; q < EQ; ++q) { w1a[q] = W1[2 * q]; w1b[q] = W1[2 * q + 1]; bq[q] = b1[q]; }
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * 2;
  // The row's two features come through a dependent pair of loads (perm[j], then e_raw[perm[j]]): ~2 memory
  // latencies per iteration with nothing to overlap them was what bounded this kernel (1.5 ms for one [E,H]
  // write).  Software pipeline: the index two iterations ahead and the features one iteration ahead are in flight
  // under the arithmetic (clamped, branch-free).
  const int64_t Elast = E - 1;
  const int64_t j0 = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * 2 + sub;
  auto clampj = [&](int64_t j) __attribute__((always_inline)) { return j < Elast ? j : Elast; };
  int64_t kn = perm[clampj(j0)];
  float2 xn = *reinterpret_cast<const float2*>(e_raw + 2 * kn);
  kn = perm[clampj(j0 + stride)];
  for (int64_t j = j0; j < E; j += stride) {
    const float x0 = xn.x, x1 = xn.y;
    xn = *reinterpret_cast<const float2*>(e_raw + 2 * kn);
    kn = perm[clampj(j + 2 * stride)];
    float o0 = bb.x, o1 = bb.y, o2 = bb.z, o3 = bb.w;
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
      const float a = fmaxf(fmaf(w1a[q], x0, fmaf(w1b[q], x1, bq[q])), 0.f);
      o0 = fmaf(a, w2[0][q], o0);
      o1 = fmaf(a, w2[1][q], o1);
      o2 = fmaf(a, w2[2][q], o2);
      o3 = fmaf(a, w2[3][q], o3);
    }
    st4(e0 + j * EH + c4, make_float4(o0, o1, o2, o3));
  }
}

// partials[chunk][ENP] (fp64): per-workgroup sums of gW2[c][q], gb2[c], gW1[q][0..1], gb1[q]
__global__ __launch_bounds__(kBlock) void edge_encoder_bwd_k(int64_t E, const float* __restrict__ ge0,
                                                             const float* __restrict__ e_raw,
                                                             const int32_t* __restrict__ perm,
                                                             const float* __restrict__ W1,
                                                             const float* __restrict__ b1,
                                                             const float* __restrict__ W2,
                                                             double* __restrict__ partials,
                                                             int64_t rows_per_block) {
  __shared__ float red[kWavesPerBlock][ENP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / EG, lr = lane % EG, c4 = lr * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)chunk * rows_per_block;
  const int64_t r1 = min(E, r0 + rows_per_block);
  float w2[4][EQ], gw2[4][EQ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < EQ; ++q) { w2[i][q] = W2[(c4 + i) * EQ + q]; gw2[i][q] = 0.f; }
  float w1a[EQ], w1b[EQ], bq[EQ];
#pragma unroll
  for (int q = 0; q < EQ; ++q) { w1a[q] = W1[2 * q]; w1b[q] = W1[2 * q + 1]; bq[q] = b1[q]; }
  float gb2_0 = 0.f, gb2_1 = 0.f, gb2_2 = 0.f, gb2_3 = 0.f, gw1_0 = 0.f, gw1_1 = 0.f, gb1_ = 0.f;
  const int q0 = (lr >> 1) & 15;     // the hidden unit this lane owns after the reduce-scatter
  const bool bit4 = lr & 16, bit3 = lr & 8, bit2 = lr & 4, bit1 = lr & 2;
  // software pipeline as in the forward kernel: index two iterations ahead, features and the gradient row one ahead
  const int64_t Elast = E - 1;
  const int64_t j0 = r0 + wave * 2 + sub;
  auto clampj = [&](int64_t j) __attribute__((always_inline)) { return j < Elast ? j : Elast; };
  int64_t kn = perm[clampj(j0)];
  float2 xn = *reinterpret_cast<const float2*>(e_raw + 2 * kn);
  float4 gn = ld4(ge0 + clampj(j0) * EH + c4);
  kn = perm[clampj(j0 + kWavesPerBlock * 2)];
  for (int64_t j = j0; j < r1; j += kWavesPerBlock * 2) {
    const float x0 = xn.x, x1 = xn.y;
    const float4 g = gn;
    xn = *reinterpret_cast<const float2*>(e_raw + 2 * kn);
    gn = ld4(ge0 + clampj(j + kWavesPerBlock * 2) * EH + c4);
    kn = perm[clampj(j + 2 * kWavesPerBlock * 2)];
    float p[EQ];
    float apre_q0 = 0.f;
#pragma unroll
    for (int q = 0; q < EQ; ++q) {
      const float ap = fmaf(w1a[q], x0, fmaf(w1b[q], x1, bq[q]));
      const float a = fmaxf(ap, 0.f);
      apre_q0 = (q == q0) ? ap : apre_q0;
      gw2[0][q] = fmaf(g.x, a, gw2[0][q]);
      gw2[1][q] = fmaf(g.y, a, gw2[1][q]);
      gw2[2][q] = fmaf(g.z, a, gw2[2][q]);
      gw2[3][q] = fmaf(g.w, a, gw2[3][q]);
      p[q] = fmaf(g.x, w2[0][q], fmaf(g.y, w2[1][q], fmaf(g.z, w2[2][q], g.w * w2[3][q])));
    }
    gb2_0 += g.x; gb2_1 += g.y; gb2_2 += g.z; gb2_3 += g.w;
    // reduce-scatter of p[0..15] over the 32 lanes of the row: 8 + 4 + 2 + 1 + 1 shuffles
    float s8[8], s4[4], s2[2], s1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float send = bit4 ? p[i] : p[i + 8];
      s8[i] = (bit4 ? p[i + 8] : p[i]) + __shfl_xor(send, 16, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float send = bit3 ? s8[i] : s8[i + 4];
      s4[i] = (bit3 ? s8[i + 4] : s8[i]) + __shfl_xor(send, 8, 64);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float send = bit2 ? s4[i] : s4[i + 2];
      s2[i] = (bit2 ? s4[i + 2] : s4[i]) + __shfl_xor(send, 4, 64);
    }
    {
      const float send = bit1 ? s2[0] : s2[1];
      s1 = (bit1 ? s2[1] : s2[0]) + __shfl_xor(send, 2, 64);
    }
    s1 += __shfl_xor(s1, 1, 64);                       // full sum over the 32 lanes, for unit q0
    const float ga = apre_q0 > 0.f ? s1 : 0.f;         // relu backward of linear1_edge
    gw1_0 = fmaf(ga, x0, gw1_0);
    gw1_1 = fmaf(ga, x1, gw1_1);
    gb1_ += ga;
  }
  // combine the two row slots of the wave, then the 4 waves through LDS
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < EQ; ++q) gw2[i][q] += __shfl_xor(gw2[i][q], 32, 64);
  gb2_0 += __shfl_xor(gb2_0, 32, 64); gb2_1 += __shfl_xor(gb2_1, 32, 64);
  gb2_2 += __shfl_xor(gb2_2, 32, 64); gb2_3 += __shfl_xor(gb2_3, 32, 64);
  gw1_0 += __shfl_xor(gw1_0, 32, 64); gw1_1 += __shfl_xor(gw1_1, 32, 64); gb1_ += __shfl_xor(gb1_, 32, 64);
  if (sub == 0) {
    float* r = red[wave];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < EQ; ++q) r[(c4 + i) * EQ + q] = gw2[i][q];
    r[EH * EQ + c4 + 0] = gb2_0; r[EH * EQ + c4 + 1] = gb2_1;
    r[EH * EQ + c4 + 2] = gb2_2; r[EH * EQ + c4 + 3] = gb2_3;
    if ((lr & 1) == 0) {       // lanes l and l^1 hold the same unit: one of them reports it
      r[EH * EQ + EH + 2 * q0 + 0] = gw1_0;
      r[EH * EQ + EH + 2 * q0 + 1] = gw1_1;
      r[EH * EQ + EH + 2 * EQ + q0] = gb1_;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ENP; i += kBlock) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) acc += (double)red[w][i];
    partials[(size_t)chunk * ENP + i] = acc;
  }
}

}  // namespace gnm

using namespace gnm;

extern "C" int gnm_edge_encoder_fwd(int64_t E, int H, int F, int Q, const float* e_raw, const int32_t* perm,
                                    const float* W1, const float* b1, const float* W2, const float* b2,
                                    float* e0, void* stream) {
  GNM_CHECK_ARG(H == EH && F == 2 && Q == EQ, "edge_encoder_fwd: built for H=128, edge_features=2, hidden=16 (got %d,%d,%d)", H, F, Q);
  GNM_CHECK_ARG(E >= 0 && e_raw && perm && W1 && b1 && W2 && b2 && e0, "edge_encoder_fwd: null/neg argument");
  int64_t g = (E + 7) / 8;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(edge_encoder_fwd_k, dim3((unsigned)g), dim3(kBlock), 0, (hipStream_t)stream, E, e_raw, perm,
                     W1, b1, W2, b2, e0);
  GNM_LAUNCH_CHECK("edge_encoder_fwd");
  return 0;
}

// partials must hold gnm_max_partial_blocks() * 2224 doubles
extern "C" size_t gnm_edge_encoder_bwd_workspace_bytes(void) { return (size_t)kMaxPartialBlocks * ENP * sizeof(double); }

extern "C" int gnm_edge_encoder_bwd(int64_t E, int H, int F, int Q, const float* ge0, const float* e_raw,
                                    const int32_t* perm, const float* W1, const float* b1, const float* W2,
                                    float* gW1, float* gb1, float* gW2, float* gb2, void* ws, size_t ws_bytes,
                                    void* stream) {
  GNM_CHECK_ARG(H == EH && F == 2 && Q == EQ, "edge_encoder_bwd: built for H=128, edge_features=2, hidden=16 (got %d,%d,%d)", H, F, Q);
  GNM_CHECK_ARG(E >= 0 && ge0 && e_raw && perm && W1 && b1 && W2 && gW1 && gb1 && gW2 && gb2, "edge_encoder_bwd: null/neg argument");
  const int grid = persistent_grid(E, 256, occ_blocks<edge_encoder_bwd_k>());
  GNM_CHECK_ARG(ws && ws_bytes >= (size_t)grid * ENP * sizeof(double), "edge_encoder_bwd: workspace too small");
  const int64_t rpb = (E + grid - 1) / grid;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(edge_encoder_bwd_k, dim3(grid), dim3(kBlock), 0, st, E, ge0, e_raw, perm, W1, b1, W2,
                     (double*)ws, rpb);
  GNM_LAUNCH_CHECK("edge_encoder_bwd");
  // gW2 | gb2 | gW1 | gb1 are contiguous in the partial rows; reduce each piece into its tensor
  const double* p = (const double*)ws;
  struct Piece { int off, n; float* out; } pieces[4] = {
      {0, EH * EQ, gW2}, {EH * EQ, EH, gb2}, {EH * EQ + EH, 2 * EQ, gW1}, {EH * EQ + EH + 2 * EQ, EQ, gb1}};
  for (const Piece& pc : pieces) {
    // strided view: row stride ENP, `n` columns starting at `off`
    if (reduce_partials_strided(p, grid, ENP, pc.off, pc.n, pc.out, stream)) return -3;
  }
  return 0;
}
