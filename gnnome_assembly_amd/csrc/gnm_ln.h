// Row statistics of the LayerNorm mode (gated_gcn_full.py:57-59: nn.LayerNorm(H) for bn_h / bn_e), shared by the LayerNorm
// kernels (gnm_layernorm.hip) and the LayerNorm form of the two-sided forward sweep (gnm_sweep.hip): the same expressions,
// so the edge outputs of the two routes are bit-identical.
#pragma once
#include "gnm_common.h"

namespace gnm {

constexpr float kEpsLN = 1e-5f;   // nn.LayerNorm default

template <int G>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float hsum4(float4 a) { return (a.x + a.y) + (a.z + a.w); }

// live[j] = 1 for the channels c4 + j < width, 0 for the dead (zero-padded) channels of a layer that runs on the next
// kernel width up (layers.padded_width): nn.LayerNorm(out_channels) normalises over the layer's REAL width
__device__ __forceinline__ float4 live_mask(int c4, int width) {
  return make_float4(c4 < width ? 1.f : 0.f, c4 + 1 < width ? 1.f : 0.f, c4 + 2 < width ? 1.f : 0.f, c4 + 3 < width ? 1.f : 0.f);
}

// xhat = (x - mean_row) * rstd_row over the `width` live channels of the row held by G lanes (dead channels hold 0 on
// entry and get xhat = 0); inv_w = 1 / width
template <int H>
__device__ __forceinline__ float4 row_normalize(float4 x, const float4& live, float inv_w, float& rstd) {
  constexpr int G = H / 4;
  const float mu = row_sum<G>(hsum4(x)) * inv_w;
  const float4 d = (x - f4(mu)) * live;
  const float var = row_sum<G>(hsum4(d * d)) * inv_w;
  rstd = 1.0f / sqrtf(var + kEpsLN);
  return d * rstd;
}

}  // namespace gnm
