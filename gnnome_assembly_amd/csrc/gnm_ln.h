// Row statistics of the LayerNorm mode (gated_gcn_full.py:57-59: nn.LayerNorm(H) for bn_h / bn_e), shared by the LayerNorm
// kernels (gnm_layernorm.hip) and the LayerNorm form of the two-sided forward sweep (gnm_sweep.hip): the same expressions,
// so the edge outputs of the two routes are bit-identical.
#pragma once
#include "gnm_common.h"

namespace gnm {

constexpr float kEpsLN = 1e-5f;   // nn.LayerNorm default

// Sum over the G consecutive lanes that hold one row (G = H / 4), the total in every one of them.
// Round 6: the butterfly v += xor-partner(v) was five ds_bpermute_b32 round trips through the LDS crossbar per sum (hipcc's __shfl_xor), four
// sums per row in the backward -- the reason the LayerNorm sweeps ran 15-25 % behind their BatchNorm twins.  The same butterfly on the
// data-parallel primitives: quad_perm / row_half_mirror / row_mirror (a DPP operand of the add) inside a 16-lane row, gfx950's
// v_permlane16_swap across the two rows of a 32-lane group.  Every step adds the same two partial sums as the xor butterfly did (after a
// step all lanes of a group hold the group's sum, so "the mirrored lane" and "the xor partner" carry the same value) and fp32 addition is
// commutative: bit-identical results.
template <int CTRL>
__device__ __forceinline__ float dpp_add_f32(float v) {
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
  return v + __builtin_bit_cast(float, o);
}
template <int G>
__device__ __forceinline__ float row_sum(float v) {
  static_assert(G == 8 || G == 16 || G == 32 || G == 64, "a row is held by 8, 16, 32 or 64 lanes");
  v = dpp_add_f32<0xB1>(v);                    // quad_perm [1,0,3,2]
  v = dpp_add_f32<0x4E>(v);                    // quad_perm [2,3,0,1]
  v = dpp_add_f32<0x141>(v);                   // row_half_mirror: the other quad of the 8
  if constexpr (G >= 16) v = dpp_add_f32<0x140>(v);      // row_mirror: the other 8 of the 16
  if constexpr (G >= 32) {
    typedef unsigned u32x2_ln_ __attribute__((ext_vector_type(2)));
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const u32x2_ln_ sw = __builtin_amdgcn_permlane16_swap(b, b, false, false);     // (rows 0 0 2 2, rows 1 1 3 3)
    // (the elements go through scalars first: __builtin_bit_cast applied to sw.x / sw.y directly makes hipcc 7.2 read element 0 twice)
    const unsigned lo_ = sw.x, hi_ = sw.y;
    v = __uint_as_float(lo_) + __uint_as_float(hi_);
  }
  if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
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
