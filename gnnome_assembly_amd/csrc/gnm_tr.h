// Swizzled split-bf16 tile images + hardware transpose reads (gfx950 ds_read_b64_tr_b16).
//
// The bf16x3 matmul mode (gnm_fused.hip, MmB3) needs every fp32 tile as three bf16 images (hi / mid / lo).
// A weight-gradient contraction  C[m][n] += sum_rows X[row][m] * Y[row][n]  ("TN") contracts over the tile's
// ROWS, and a lane of v_mfma_f32_32x32x16_bf16 must hold 8 CONSECUTIVE contraction indices of one column --
// a column of the row-major tile.  Round 1 built a second, column-major image set for that through register
// transposition (stage_cols: 110 KB of LDS in the fused edge backward, one workgroup per CU).  gfx950 can
// transpose on the way OUT of LDS instead: ds_read_b64_tr_b16 lets every lane of a 16-lane group name its own
// 8-byte piece (4 bf16 of one row) and returns to lane i the i-th COLUMN of the 4 x 16 block the group named
// (lane i, element j  <-  piece 4j + i/4, element i%4).  So ONE row-major image set serves both contractions:
//   NN / NT  (y = x W):       A fragment = 8 consecutive k of row i          -> ds_read_b128
//   TN       (gW = x^T y):    A / B fragment = 8 consecutive rows of column i -> 2 x ds_read_b64_tr_b16
//
// Image layout: [rows][128] bf16, row pitch 256 B, no padding; the 16-byte slot s (8 columns) of row r lives at
// slot s ^ swz(r) with swz(r) = ((r & 3) << 2) | f((r >> 2) & 3).  Under the bank rules of MI355X_MICROARCH.md
// (LDS section) every access pattern used here is conflict-free:
//   * staging writes (ds_write_b64, 16-lane groups): 16 lanes cover half a row = 8 slots x 2 halves, distinct;
//   * NN reads (ds_read_b128, groups {0-3,12-15,20-27} ...): 16 rows with distinct r mod 16 at one logical
//     slot -> swz is a bijection of r mod 16 -> 16 distinct physical slots = all 64 banks once;
//   * transpose reads (2 x 32 lanes): 4 rows (r & 3 = 0..3, same r >> 2) x 4 logical slots x 2 halves: the row's
//     low bits go to the HIGH slot bits, the slot's low bits stay -> 16 distinct slots x 2 halves = 64 banks once.
#pragma once
#include "gnm_common.h"

namespace gnm {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 tr_bf16x4;   // operand type of the tr builtin

constexpr int SW = 128;                 // tile width (hidden size these kernels are built for)
constexpr int SPITCH = SW * 2;          // bytes per image row

// f = (0, 2, 3, 1): any bijection of (r >> 2) & 3 keeps the 32-row patterns conflict-free; this one also keeps the
// 16-row A-fragment reads of v_mfma_f32_16x16x32_bf16 (lane (i = l & 15, g = l >> 4) reads slot 4 kc + g of row i)
// conflict-free, whose ds_read_b128 lane groups mix g = 0 rows {0-3, 12-15} with g = 1 rows {4-11}.
__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((0x78 >> (((r >> 2) & 3) * 2)) & 3); }
// byte offset, inside one image, of the 4 bf16 at (row r, columns c .. c+3), c % 4 == 0
__device__ __forceinline__ int simg_off(int r, int c) {
  return r * SPITCH + ((((c >> 3) ^ swz(r)) << 4) | ((c & 4) << 1));
}

// x = hi + mid + lo exactly (3 x 8 significand bits); see MmB3 in gnm_fused.hip
__device__ __forceinline__ void split3f(const float4& v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r1 = x[j] - (float)h;      // exact
    const __bf16 m = (__bf16)r1;
    hi[j] = h;
    mid[j] = m;
    lo[j] = (__bf16)(r1 - (float)m);       // exact, fits 8 bits
  }
}

// ---- f16x2: fp32 operands as TWO fp16 terms of a power-of-two multiple (round 5) ----------------------------------
// The package power cap, not HBM or the matrix peak, sets the duration of every kernel that feeds the matrix cores
// (profiles/r05_power.txt): the energy of a contraction is what has to shrink.  x s = h1 + h2 + r with h1 = fp16(x s),
// h2 = fp16(x s - h1), |r| <= 2^-22 |x s| (2 x 11 significand bits), and a product needs THREE MFMAs (h1 w1, h1 w2,
// h2 w1; the dropped h2 w2 <= 2^-22 |x w|) instead of the six of bf16x3.  s is a power of two that puts the largest
// magnitude of the operand's ROW (or of a weight's output column) into [2^14, 2^15): the fp16 exponent range then holds
// both terms of every element down to 2^-17 of that maximum at full width (smaller elements keep an absolute error of
// 2^-25 / s, i.e. 2^-39 of the maximum), nothing overflows, and multiplying the fp32 accumulator by 1 / s is exact.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// max over the 32 consecutive lanes that hold one 128-float row, on the BIT PATTERNS of non-negative floats (ordered as
// unsigned integers; no NaN canonicalisation between the steps): four v_max_u32 with a DPP operand, then the other 16-lane
// row through gfx950's v_permlane16_swap
template <int CTRL>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, true);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned row32_max_bits(unsigned m) {
  m = dpp_max_u32<0xB1>(m);      // quad_perm [1,0,3,2]
  m = dpp_max_u32<0x4E>(m);      // quad_perm [2,3,0,1]
  m = dpp_max_u32<0x141>(m);     // row_half_mirror: the other quad of the 8
  m = dpp_max_u32<0x140>(m);     // row_mirror: the other 8 of the 16
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ sw = __builtin_amdgcn_permlane16_swap(m, m, false, false);   // (rows 0 0 2 2, rows 1 1 3 3)
  return sw.x > sw.y ? sw.x : sw.y;
}
__device__ __forceinline__ unsigned max_abs4_bits(const float4& v) {
  return __float_as_uint(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
}
// max of a SIGNED int over the 16 lanes of a DPP row / over 32 consecutive lanes (the tile-wide largest EA + EB of the f16x2 TN scheme: every
// 16- or 32-lane group holds all the tile's rows, so every wave ends with the same number).  Round 6: these were __shfl_xor butterflies =
// four or five dependent ds_bpermute_b32 round trips per tile and wave.
template <int CTRL>
__device__ __forceinline__ int dpp_max_i32(int v) {
  const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, true);
  return o > v ? o : v;
}
__device__ __forceinline__ int row16_max_i32(int m) {
  m = dpp_max_i32<0xB1>(m);
  m = dpp_max_i32<0x4E>(m);
  m = dpp_max_i32<0x141>(m);
  return dpp_max_i32<0x140>(m);
}
__device__ __forceinline__ int row32_max_i32(int m) {
  m = row16_max_i32(m);
  typedef unsigned u32x2i_ __attribute__((ext_vector_type(2)));
  const u32x2i_ sw = __builtin_amdgcn_permlane16_swap((unsigned)m, (unsigned)m, false, false);
  const int a = (int)sw.x, b = (int)sw.y;
  return a > b ? a : b;
}
// s = 2^(14 - floor(log2 m)), inv = 1 / s; magnitudes below 2^-111 (and zero rows) share s = 2^125
__device__ __forceinline__ void h2_scale(unsigned mbits, float& s, float& inv) {
  int e = (int)(mbits >> 23) & 0xff;
  e = e < 16 ? 16 : e;
  s = __int_as_float((268 - e) << 23);
  inv = __int_as_float((e - 14) << 23);
}
__device__ __forceinline__ void split2(const float4& v, float s, h16x4& hi, h16x4& lo) {
  const float x[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const _Float16 h = (_Float16)x[j];
    hi[j] = h;
    lo[j] = (_Float16)(x[j] - (float)h);     // the difference is exact in fp32
  }
}
__device__ __forceinline__ void mfh(floatx16& acc, const h16x8& a, const h16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

// biased exponent of the largest magnitude of a row (clamped like h2_scale: zero rows count as 2^-111)
__device__ __forceinline__ int h2_row_exp(const float4& v) {
  const int e = (int)(row32_max_bits(max_abs4_bits(v)) >> 23) & 0xff;
  return e < 16 ? 16 : e;
}
__device__ __forceinline__ float pow2_biased(int be) { return __int_as_float((be < 0 ? 0 : be > 254 ? 254 : be) << 23); }   // 2^(be - 127); 0 below the range

// one float4 of row r, columns c .. c+3 -> the three images of a tile (image s at img + s * img_bytes)
__device__ __forceinline__ void simg_stage(unsigned char* img, int img_bytes, int r, int c, const float4& v) {
  bf16x4 hi, mid, lo;
  split3f(v, hi, mid, lo);
  unsigned char* p = img + simg_off(r, c);
  *reinterpret_cast<bf16x4*>(p) = hi;
  *reinterpret_cast<bf16x4*>(p + img_bytes) = mid;
  *reinterpret_cast<bf16x4*>(p + 2 * img_bytes) = lo;
}

// f16x2: one float4 times the power of two sc -> the two images of a tile
__device__ __forceinline__ void simg_stage_h2(unsigned char* img, int img_bytes, int r, int c, const float4& v, float sc) {
  h16x4 hi, lo;
  split2(v, sc, hi, lo);
  unsigned char* p = img + simg_off(r, c);
  *reinterpret_cast<h16x4*>(p) = hi;
  *reinterpret_cast<h16x4*>(p + img_bytes) = lo;
}

// the exact fp32 values back out of the three images (hi + mid is exact in fp32, + lo gives x)
__device__ __forceinline__ float4 simg_load_f32(const unsigned char* img, int img_bytes, int r, int c) {
  const unsigned char* p = img + simg_off(r, c);
  const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p);
  const bf16x4 mid = *reinterpret_cast<const bf16x4*>(p + img_bytes);
  const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p + 2 * img_bytes);
  return make_float4(((float)hi[0] + (float)mid[0]) + (float)lo[0], ((float)hi[1] + (float)mid[1]) + (float)lo[1],
                     ((float)hi[2] + (float)mid[2]) + (float)lo[2], ((float)hi[3] + (float)mid[3]) + (float)lo[3]);
}

// NN / NT A fragment: element j of lane (i = l & 31, g = l >> 5) = X[r0 + i][16 kc + 8 g + j]   (ds_read_b128)
__device__ __forceinline__ bf16x8 simg_row_frag(const unsigned char* img, int r0, int kc, int lane) {
  const int r = r0 + (lane & 31);
  return *reinterpret_cast<const bf16x8*>(img + r * SPITCH + ((((2 * kc + (lane >> 5)) ^ swz(r))) << 4));
}

// TN A / B fragment: element j of lane (i = l & 31, g = l >> 5) = X[k0 + 8 g + j][c0 + i]
// (two transpose reads: lane p = l & 15 of each 16-lane group names the piece (row k0 + 8g + 4q + p/4,
//  columns c0 + 16 ((l >> 4) & 1) + 4 (p & 3) .. +3) and receives column p of the 4 x 16 block)
__device__ __forceinline__ bf16x8 simg_col_frag(const unsigned char* img, int k0, int c0, int lane) {
  const int p = lane & 15;
  const int col = c0 + 16 * ((lane >> 4) & 1) + 4 * (p & 3);
  const int r = k0 + 8 * (lane >> 5) + (p >> 2);
  typedef __attribute__((address_space(3))) tr_bf16x4* lds_ptr;
  const tr_bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(img + simg_off(r, col)));
  const tr_bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(img + simg_off(r + 4, col)));
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = a[j];
    f[4 + j] = b[j];
  }
  return f;
}

// The same reads with the address arithmetic factored so that ONE lane-constant base register serves a whole
// family of fragments (hipcc otherwise keeps one address register per fragment read live across the tile loop).
//   byte offset of the q-th transpose read (rows k0 + 8g + 4q ..) of column block mb (32 columns), chunk kc:
//     trbase[q] ^ (mb << 6)   +   kc * 16 * SPITCH
//   with trbase[q] = (8g + p/4 + 4q) * SPITCH + ((((p/4) << 2) | lowq) << 4) + ((p & 1) << 3),
//        lowq = (2 ih + ((p & 3) >> 1)) ^ f((2g + q) & 3),   p = l & 15, ih = (l >> 4) & 1, g = l >> 5
// (swz of the row is ((p/4) << 2) | f((2g + q) & 3): the row's bits 2-3 are 2g + q modulo 4 for every kc.)
__device__ __forceinline__ int simg_tr_base(int lane, int q) {
  const int p = lane & 15, ih = (lane >> 4) & 1, g = lane >> 5, pr = p >> 2;
  const int lowq = (2 * ih + ((p & 3) >> 1)) ^ ((0x78 >> (((2 * g + q) & 3) * 2)) & 3);
  return (8 * g + pr + 4 * q) * SPITCH + ((((pr << 2) | lowq)) << 4) + ((p & 1) << 3);
}
// fragment of column block mb from precomputed bases; `img` already includes the image (hi/mid/lo) and kc offsets
__device__ __forceinline__ bf16x8 simg_col_frag2(const unsigned char* img, int base0_x_mb, int base1_x_mb) {
  typedef __attribute__((address_space(3))) tr_bf16x4* lds_ptr;
  const tr_bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(img + base0_x_mb));
  const tr_bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(img + base1_x_mb));
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[j] = a[j];
    f[4 + j] = b[j];
  }
  return f;
}

__device__ __forceinline__ h16x8 simg_col_frag_h(const unsigned char* img, int k0, int c0, int lane) {
  return __builtin_bit_cast(h16x8, simg_col_frag(img, k0, c0, lane));     // the transpose read moves 16-bit elements, whatever they hold
}

typedef float floatx4_acc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfb16s(floatx4_acc& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfh16s(floatx4_acc& acc, const h16x8& a, const h16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mfb16(floatx16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

// the six partial products of the bf16x3 scheme, smallest first: (A part, B part) =
// lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi      (parts: 0 = hi, 1 = mid, 2 = lo)
__device__ __forceinline__ constexpr int b3_pa(int t) { return t == 0 ? 2 : (t == 2 || t == 3) ? 1 : 0; }
__device__ __forceinline__ constexpr int b3_pb(int t) { return t == 1 ? 2 : (t == 2 || t == 4) ? 1 : 0; }

// arguments of edge_bwd_chain_k (gnm_tr.hip)
struct ChainArgs {
  int64_t E, N;
  // layer i: gt from (ge, t_hi); gW3(i) against e_mid = e_in(i) = e_out(i-1)
  const float* ge; float* ge_out; const float* t_hi; const float* e_mid;
  const float* stat_hi; const float* bstat_hi; const float* gamma_hi; const bf16x8* Wp;
  float* slab; double* partials;                     // [grid][128][128] partial gW3(i), [grid][128] column sums of gt
  // layer i-1: by-destination backward
  const float* t_lo; const float* stat_lo; const float* P_lo; const float* Q_lo; const float* hf_lo; const float* hb_lo;
  const int32_t* isrc; const int32_t* idst; const int32_t* in_ptr;
  float* gP_lo; float* Ud_lo; float* Td_lo; double* partials_lo;   // gP[:,2H:3H], [N,H], [N,H], [grid][2][128]
  int64_t nodes_per_block;
  // two-sided sweep (edge_bwd_chain_k<., true>): the by-SOURCE sums of layer i-1 through the sweep plan
  // (gnm_graph_build_sweep_plan over THIS partition): gA2h -> gP_lo[:,H:2H], Us | Ts -> UT_lo [N,2H]
  // Ud_lo / Td_lo may be the two halves of ONE [N,2H] array (ud_pitch = 2H) or two [N,H] arrays (ud_pitch = H)
  const uint32_t* sinfo; float* UT_lo; int64_t margin; int ud_pitch;
  int hfull;       // row pitch of the layer-(i-1) tensors (0 / 128: 128; 256: the top sweep of a 256-wide layer, one half per launch)
  // round 6, the LayerNorm form of the sweep without a layer above (edge_bwd_chain_k<..., LN>; gnm_ln_edge_bwd_top): row statistics
  // instead of stat_lo, gt = LNbwd(gu) formed per row and WRITTEN (gt_out), the sums by destination / by source are gB2h / gB1h
  // themselves (no BatchNorm-backward means to wait for)
  const float* ln_gamma; const float* ln_beta; int ln_width; float* gt_out;
  int ut_pitch;    // row pitch of UT_lo (0: 2 H = [Us | Ts]); the LayerNorm form sends its by-source sum of gt straight into gP[:, 3H:4H]
  int q_pitch, qb_off;   // Q_lo row pitch and the offset of Qb in a row (0, 0: 2 H and H = [Qf | Qb]; gnm_ln_node_bwd writes Qf | Rf | Qb | Rb: 4 H, 2 H)
};

// tn_tr_k<., CONV> (gnm_tr.hip): the two column groups of the A operand are formed from raw sums
struct TnConv {
  const float* U[2]; const float* T[2]; int64_t pitch[2]; const int32_t* ptr[2];     // per column group: sums, pitches, CSR pointers
  const float* stat_e; const float* bstat_e; const float* gamma_e;
  float* Xw; int64_t ldxw;                                                            // the formed groups go to Xw[r][cg*128 + c]
};


constexpr int kSweepTileRows = 16;      // rows per tile of the sweep kernels (= ER of gnm_tr.hip)
constexpr int kSweepSlots = 32;         // accumulator slots per workgroup (<= 28 are ever live on the chr19-scale graph)
constexpr int64_t kSweepMargin = 1 << 16;   // a served source lies within this many ids of its workgroup's node range
constexpr unsigned kSweepOpen = 1u << 22, kSweepClose = 1u << 23;

}  // namespace gnm
