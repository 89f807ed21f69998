// W-stationary fused MFMA kernels for H = 128 (gfx950).
//
// The [E,128] x [128,128] contractions of the layer have intensity 32 FLOP/B -- right at the
// fp32-MFMA / HBM balance point of the chip -- so a separate GEMM that writes its result and a
// separate elementwise pass that re-reads it pay the [E,H] stream twice.  These kernels keep
// the 64 KB weight matrix in VGPRs (each of the 4 waves owns 32 output columns = 64 VGPRs of
// B fragments), stream 64-row tiles through LDS once, and do the surrounding elementwise /
// gather / statistics work around the MFMAs:
//
//   rowtile_nt_k<EDGE>   t = e W3^T + b3 + B1h[src] + B2h[dst], BatchNorm column sums
//                        (gated_gcn_full.py:113,120-122) -- replaces gemm + edge_t_stats;
//                        node mode: P = h W5^T + b5 over the five 128-column groups (:107-112)
//   edge_bwd_fused32_k   gt = gamma*rstd*(gu - m1 - that*m2); ge_in = ge + gt W3;
//                        gW3 += gt^T e_in; gb3 += sum gt   (autograd of :113,:122)
//                        -- replaces edge_bwd_gt + two GEMMs + a column sum (9 -> 4 streams)
//   rowtile_nn_acc_k / tn_colgroup_k   autograd of the 5-way node projection (:107-112)
//
// MFMA: v_mfma_f32_32x32x2_f32 (exact fp32).  Lane (i = l&31, g = l>>5) supplies A[i][k'],
// B[k'][i] with k' = g; contraction indices are permuted as k = 8q + 4g + r so that one
// ds_read_b128 feeds four MFMAs; C/D: col = l&31, row = (e&3) + 8*(e>>2) + 4*g.
// LDS row pitch 132 floats: ds_read_b128 fragment reads are bank-conflict-free.  Accumulators
// are transposed through LDS so that every global access is a whole 512-byte row (one float4
// per lane).
//
// Pipelining: every HBM row a tile needs is prefetched one tile ahead into registers, under the
// MFMA phases.  The steady-state loop only runs FULL tiles and is free of divergent branches
// (addresses are clamped instead of predicated), so hipcc can count the vector-memory queue:
// the wait for the prefetched rows is vmcnt(#stores issued after them), not vmcnt(0) -- with a
// predicated epilogue it waited for the previous tile's stores to drain on every iteration.
// The (at most one) ragged tile of a workgroup runs through a predicated copy of the body.
#include <string.h>
#include <type_traits>

#include "gnm_common.h"
#include "gnm_tr.h"

namespace gnm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int FH = 128;          // hidden width these kernels are built for
constexpr int FTR = 64;          // rows per tile
constexpr int FP = FH + 4;       // LDS row pitch (floats)
constexpr int FKQ = FH / 8;      // 16 k-quads

using full_t = std::true_type;
using ragged_t = std::false_type;

// Pack a [rows,128]-shaped weight into MFMA B-fragment order:
//   NT (y = x W^T):  Wp[cb][q][lane][r] = W[(cb*32 + (lane&31)) * ld + 8q + 4(lane>>5) + r]
//   NN (y = x W):    Wp[cb][q][lane][r] = W[(8q + 4(lane>>5) + r) * ld + cb*32 + (lane&31)]
// cb = 32-column block of the output; one float4 per (cb, q, lane).
__global__ void pack_w_k(const float* __restrict__ W, int64_t ld, int ncb, int nn, float* __restrict__ Wp) {
  const int total = ncb * FKQ * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, q = (idx >> 6) % FKQ, cb = idx / (64 * FKQ);
    const int i = lane & 31, g = lane >> 5;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 8 * q + 4 * g + r;
      v[r] = nn ? W[(int64_t)k * ld + cb * 32 + i] : W[(int64_t)(cb * 32 + i) * ld + k];
    }
    reinterpret_cast<float4*>(Wp)[idx] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__device__ __forceinline__ void mfma4(floatx16& acc, const float4& a, const float4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

// acc0/acc1 += (rows 0-31 / 32-63 of the LDS tile) x (this wave's 32 weight columns), K = 128.
// Fragment reads are software-pipelined one k-quad ahead; the sched_barriers keep hipcc from
// hoisting all 32 fragment reads (128 VGPRs) or sinking them right in front of their MFMAs.
__device__ __forceinline__ void mma_tile64(const float* __restrict__ lds, const float4 (&wf)[FKQ],
                                           floatx16& acc0, floatx16& acc1, int li, int lg) {
  const float* p0 = lds + li * FP + 4 * lg;
  const float* p1 = lds + (32 + li) * FP + 4 * lg;
  float4 a0 = ld4(p0), a1 = ld4(p1);
#pragma unroll
  for (int q = 0; q < FKQ; ++q) {
    float4 n0 = a0, n1 = a1;
    if (q + 1 < FKQ) {
      n0 = ld4(p0 + 8 * (q + 1));
      n1 = ld4(p1 + 8 * (q + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    mfma4(acc0, a0, wf[q]);
    mfma4(acc1, a1, wf[q]);
    a0 = n0;
    a1 = n1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// tn[a][b] += gt-tile^T x e-tile over 64 rows: this wave's 64 x 64 block (wn, wc) of the 128 x 128 result
__device__ __forceinline__ void mma_tn64(const float* __restrict__ as, const float* __restrict__ bs,
                                         floatx16 (&tn)[2][2], int wn, int wc, int li, int lg) {
#pragma unroll
  for (int q = 0; q < FTR / 8; ++q) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 8 * q + 4 * lg + r;
      const float a0 = as[row * FP + (2 * wn) * 32 + li];
      const float a1 = as[row * FP + (2 * wn + 1) * 32 + li];
      const float b0 = bs[row * FP + (2 * wc) * 32 + li];
      const float b1 = bs[row * FP + (2 * wc + 1) * 32 + li];
      tn[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, tn[0][0], 0, 0, 0);
      tn[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, tn[0][1], 0, 0, 0);
      tn[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, tn[1][0], 0, 0, 0);
      tn[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, tn[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// MFMA accumulator layout -> row image in LDS
__device__ __forceinline__ void acc_to_lds(float* __restrict__ o, const floatx16& acc0, const floatx16& acc1,
                                           int wave, int li, int lg) {
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * lg;
    o[row * FP + wave * 32 + li] = acc0[e];
    o[(32 + row) * FP + wave * 32 + li] = acc1[e];
  }
}

__device__ __forceinline__ int64_t clampi(int64_t r, int64_t hi) { return r < hi ? r : hi; }

// ------------------------------------------------------------------------------------------
// Matmul policies of the row-tile kernels (how a 64 x 128 fp32 tile meets a 128 x 32 weight block)
//
//   MmF32   v_mfma_f32_32x32x2_f32 on the fp32 tile image (gnm_set_matmul_mode(0)).
//   MmB3    fp32 x fp32 as SIX bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 8x the fp32 rate):
//           x = x1 + x2 + x3 exactly, with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
//           (3 x 8 significand bits = fp32's 24), every product x_i * w_j is exact in the fp32
//           accumulator, and the three products below 2^-24 |x w| (x2 w3, x3 w2, x3 w3) are
//           dropped -- the same order as ONE fp32 rounding of the product.  Accumulation stays
//           fp32.  The default (gnm_set_matmul_mode(0) selects MmF32); inf inputs give NaN (inf - inf in the split).
// A policy stages tile rows into its LDS image(s), keeps this wave's weight block as fragments in
// VGPRs, and accumulates rows 0-31 / 32-63 of the tile into acc0 / acc1.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BP = FH + 8;          // bf16 image row pitch (272 B): ds_read_b128 fragment reads conflict-free
constexpr int BIMG = FTR * BP;      // elements per bf16 image
constexpr int BKC = FH / 16;        // 8 k-chunks of 16

struct MmF32 {
  static constexpr int kImgBytes = FTR * FP * 4;
  static constexpr bool kSplit = false;
  static constexpr bool kScaled = false;
  static constexpr size_t kPackBytes = (size_t)FKQ * 64 * 16;        // per 32-column weight block
  static constexpr int kTnBytes = 2 * FTR * FP * 4;                  // TN operands: two fp32 row images
  // tile row of the it-th float4 of thread-row lrow in the coalesced image
  static __device__ __forceinline__ int row(int lrow, int it) { return lrow + 8 * it; }
  struct Frag { float4 w[FKQ]; };
  static __device__ __forceinline__ void load_w(Frag& f, const void* Wp, int blk, int lane) {
    const float4* p = reinterpret_cast<const float4*>(Wp) + ((int64_t)blk * FKQ) * 64 + lane;
#pragma unroll
    for (int q = 0; q < FKQ; ++q) f.w[q] = p[q * 64];
  }
  static __device__ __forceinline__ void stage(void* img, int row, int c4, const float4& v) {
    st4(reinterpret_cast<float*>(img) + row * FP + c4, v);
  }
  static __device__ __forceinline__ void mma(const void* img, const Frag& f, floatx16& acc0, floatx16& acc1,
                                             int li, int lg) {
    mma_tile64(reinterpret_cast<const float*>(img), f.w, acc0, acc1, li, lg);
  }
};

__device__ __forceinline__ void split3(const float4& v, bf16x4& hi, bf16x4& mid, bf16x4& lo) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 h = (__bf16)x[j];
    const float r1 = x[j] - (float)h;      // exact
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;        // exact, fits 8 bits
    hi[j] = h;
    mid[j] = m;
    lo[j] = (__bf16)r2;
  }
}

__device__ __forceinline__ void mfb(floatx16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

struct MmB3 {
  static constexpr int kImgBytes = 3 * BIMG * 2;
  static constexpr bool kSplit = true;
  static constexpr bool kScaled = false;
  static constexpr size_t kPackBytes = (size_t)BKC * 3 * 64 * 16;
  static __device__ __forceinline__ int row(int lrow, int it) { return 8 * lrow + it; }
  struct Frag { bf16x8 w[BKC][3]; };
  static __device__ __forceinline__ void load_w(Frag& f, const void* Wp, int blk, int lane) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(Wp) + ((int64_t)blk * BKC * 3) * 64 + lane;
#pragma unroll
    for (int c = 0; c < BKC; ++c)
#pragma unroll
      for (int s = 0; s < 3; ++s) f.w[c][s] = p[(c * 3 + s) * 64];
  }
  static __device__ __forceinline__ void stage(void* img, int row, int c4, const float4& v) {
    bf16x4 hi, mid, lo;
    split3(v, hi, mid, lo);
    __bf16* b = reinterpret_cast<__bf16*>(img) + row * BP + c4;
    *reinterpret_cast<bf16x4*>(b) = hi;
    *reinterpret_cast<bf16x4*>(b + BIMG) = mid;
    *reinterpret_cast<bf16x4*>(b + 2 * BIMG) = lo;
  }
  // lane (i, g) of chunk c holds k = 16c + 8g .. +7 of row i (A) / of weight column i (B)
  static __device__ __forceinline__ void mma(const void* img, const Frag& f, floatx16& acc0, floatx16& acc1,
                                             int li, int lg) {
    const __bf16* p0 = reinterpret_cast<const __bf16*>(img) + li * BP + 8 * lg;
    const __bf16* p1 = p0 + 32 * BP;
    bf16x8 a0[3], a1[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      a0[s] = *reinterpret_cast<const bf16x8*>(p0 + s * BIMG);
      a1[s] = *reinterpret_cast<const bf16x8*>(p1 + s * BIMG);
    }
#pragma unroll
    for (int c = 0; c < BKC; ++c) {
      bf16x8 n0[3], n1[3];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        n0[s] = a0[s];
        n1[s] = a1[s];
        if (c + 1 < BKC) {
          n0[s] = *reinterpret_cast<const bf16x8*>(p0 + s * BIMG + 16 * (c + 1));
          n1[s] = *reinterpret_cast<const bf16x8*>(p1 + s * BIMG + 16 * (c + 1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // smallest products first: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
      mfb(acc0, a0[2], f.w[c][0]); mfb(acc1, a1[2], f.w[c][0]);
      mfb(acc0, a0[0], f.w[c][2]); mfb(acc1, a1[0], f.w[c][2]);
      mfb(acc0, a0[1], f.w[c][1]); mfb(acc1, a1[1], f.w[c][1]);
      mfb(acc0, a0[1], f.w[c][0]); mfb(acc1, a1[1], f.w[c][0]);
      mfb(acc0, a0[0], f.w[c][1]); mfb(acc1, a1[0], f.w[c][1]);
      mfb(acc0, a0[0], f.w[c][0]); mfb(acc1, a1[0], f.w[c][0]);
#pragma unroll
      for (int s = 0; s < 3; ++s) { a0[s] = n0[s]; a1[s] = n1[s]; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// MmH2 (gnm_set_matmul_mode(2), "f16x2"): two fp16 terms of a power-of-two multiple per operand, THREE MFMAs per product
// (gnm_tr.h).  A tile row is scaled by its own largest magnitude when it is staged -- stage() returns 1 / s, the thread
// that stages a row piece is the thread that stores the same piece of the result, so the factor stays in a register --
// and a weight block carries the factors of its 32 output columns behind its fragments.
struct MmH2 {
  static constexpr int kImgBytes = 2 * BIMG * 2;
  static constexpr bool kSplit = true;
  static constexpr bool kScaled = true;
  static constexpr size_t kFragBytes = (size_t)BKC * 2 * 64 * 16;
  static constexpr size_t kPackBytes = kFragBytes + 32 * sizeof(float);     // + 1 / s of the block's 32 columns
  struct Frag { h16x8 w[BKC][2]; };
  static __device__ __forceinline__ void load_w(Frag& f, const void* Wp, int blk, int lane) {
    const h16x8* p = reinterpret_cast<const h16x8*>(reinterpret_cast<const char*>(Wp) + (size_t)blk * kPackBytes) + lane;
#pragma unroll
    for (int c = 0; c < BKC; ++c)
#pragma unroll
      for (int s = 0; s < 2; ++s) f.w[c][s] = p[(c * 2 + s) * 64];
  }
  // 1 / s of output columns c4 .. c4+3 of the 128-column group whose first block is blk0
  static __device__ __forceinline__ float4 col_inv4(const void* Wp, int blk0, int c4) {
    return ld4(reinterpret_cast<const float*>(reinterpret_cast<const char*>(Wp) + (size_t)(blk0 + (c4 >> 5)) * kPackBytes + kFragBytes) + (c4 & 31));
  }
  // called by all 32 lanes of a row together; returns the row's 1 / s
  static __device__ __forceinline__ float stage(void* img, int row, int c4, const float4& v) {
    float sc, inv;
    h2_scale(row32_max_bits(max_abs4_bits(v)), sc, inv);
    h16x4 hi, lo;
    split2(v, sc, hi, lo);
    _Float16* b = reinterpret_cast<_Float16*>(img) + row * BP + c4;
    *reinterpret_cast<h16x4*>(b) = hi;
    *reinterpret_cast<h16x4*>(b + BIMG) = lo;
    return inv;
  }
  static __device__ __forceinline__ void mma(const void* img, const Frag& f, floatx16& acc0, floatx16& acc1,
                                             int li, int lg) {
    const _Float16* p0 = reinterpret_cast<const _Float16*>(img) + li * BP + 8 * lg;
    const _Float16* p1 = p0 + 32 * BP;
    h16x8 a0[2], a1[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a0[s] = *reinterpret_cast<const h16x8*>(p0 + s * BIMG);
      a1[s] = *reinterpret_cast<const h16x8*>(p1 + s * BIMG);
    }
#pragma unroll
    for (int c = 0; c < BKC; ++c) {
      h16x8 n0[2], n1[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        n0[s] = a0[s];
        n1[s] = a1[s];
        if (c + 1 < BKC) {
          n0[s] = *reinterpret_cast<const h16x8*>(p0 + s * BIMG + 16 * (c + 1));
          n1[s] = *reinterpret_cast<const h16x8*>(p1 + s * BIMG + 16 * (c + 1));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // smallest products first: lo*hi, hi*lo, hi*hi
      mfh(acc0, a0[1], f.w[c][0]); mfh(acc1, a1[1], f.w[c][0]);
      mfh(acc0, a0[0], f.w[c][1]); mfh(acc1, a1[0], f.w[c][1]);
      mfh(acc0, a0[0], f.w[c][0]); mfh(acc1, a1[0], f.w[c][0]);
#pragma unroll
      for (int s = 0; s < 2; ++s) { a0[s] = n0[s]; a1[s] = n1[s]; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// MM::stage for every policy: the row's 1 / s (1 where the policy does not scale)
template <class MM>
__device__ __forceinline__ float stage_row(void* img, int row, int c4, const float4& v) {
  if constexpr (MM::kScaled) return MM::stage(img, row, c4, v);
  else { MM::stage(img, row, c4, v); return 1.f; }
}

// ---- MmH2 in the row kernels whose contraction runs over SEVERAL 128-column groups (y = x W, K = ncg * 128) ----------
// The accumulator of a row stays in registers across the groups, so its unit has to: a row keeps a REFERENCE exponent in
// LDS that only grows (the largest magnitude of the row's groups so far); a group is staged in the unit of the new
// reference, and the waves multiply the accumulator rows by 2^(old - new) <= 1 (exact) before they add the group's
// products.  The weight's column factors are taken over the whole K (pack_w2_gen_k), one per output column.
__device__ __forceinline__ void h2_stage_nn(void* img, int row, int c4, const float4& v, int* eref, float* fs, bool first,
                                            bool writer) {
  const unsigned mb = row32_max_bits(max_abs4_bits(v));
  int e = (int)(mb >> 23) & 0xff;
  e = e < 16 ? 16 : e;
  const int eo = first ? e : *eref;          // every lane of the row reads before the row's writer stores (one wave)
  const int en = e > eo ? e : eo;
  if (writer) {
    *eref = en;
    const int fe = 127 + eo - en;
    *fs = __int_as_float((fe < 0 ? 0 : fe) << 23);
  }
  h16x4 hi, lo;
  split2(v, __int_as_float((268 - en) << 23), hi, lo);
  _Float16* b = reinterpret_cast<_Float16*>(img) + row * BP + c4;
  *reinterpret_cast<h16x4*>(b) = hi;
  *reinterpret_cast<h16x4*>(b + BIMG) = lo;
}
// accumulator element e of lane (li, lg) is row (e & 3) + 8 (e >> 2) + 4 lg of the 32-row tile
__device__ __forceinline__ void acc_rescale_rows(floatx16& acc, const float* fs, int lg) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 f = ld4(fs + 8 * q + 4 * lg);
    acc[4 * q + 0] *= f.x; acc[4 * q + 1] *= f.y; acc[4 * q + 2] *= f.z; acc[4 * q + 3] *= f.w;
  }
}
__device__ __forceinline__ float h2_ref_inv(int eref) { return __int_as_float((eref - 14) << 23); }

__device__ __forceinline__ void mma32_h2(const void* img, int row0, const MmH2::Frag& f, floatx16& acc, int li, int lg) {
  const _Float16* p0 = reinterpret_cast<const _Float16*>(img) + (row0 + li) * BP + 8 * lg;
#pragma unroll
  for (int c = 0; c < BKC; ++c) {
    h16x8 a0[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) a0[s] = *reinterpret_cast<const h16x8*>(p0 + s * BIMG + 16 * c);
    mfh(acc, a0[1], f.w[c][0]);
    mfh(acc, a0[0], f.w[c][1]);
    mfh(acc, a0[0], f.w[c][0]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// MmH2 in the 256-wide fused edge kernels: a row is 256 columns = the 64 lanes of ONE wave (thread = 4 columns; the row's two 128-column
// halves go to the images of the two contraction halves, which meet by addition: ONE factor per row).  The row's 1 / s goes to LDS: the
// epilogue of these kernels runs in another thread layout.
__device__ __forceinline__ void h2_stage_row64(void* img, int row, int c4, const float4& v, float* rinv_slot, bool writer) {
  unsigned m = row32_max_bits(max_abs4_bits(v));
  const unsigned o = (unsigned)__shfl_xor((int)m, 32);
  m = o > m ? o : m;
  float sc, inv;
  h2_scale(m, sc, inv);
  if (writer) *rinv_slot = inv;
  h16x4 hi, lo;
  split2(v, sc, hi, lo);
  _Float16* b = reinterpret_cast<_Float16*>(img) + row * BP + c4;
  *reinterpret_cast<h16x4*>(b) = hi;
  *reinterpret_cast<h16x4*>(b + BIMG) = lo;
}

// this wave's 64 x 64 block of the TN result -> sl[n][c] (128 x 128, row-major)
template <class MM>
__device__ __forceinline__ void tn_store_slab(float* __restrict__ sl, const floatx16 (&tn)[2][2], int wn, int wc,
                                              int li, int lg) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = (e & 3) + 8 * (e >> 2) + 4 * lg;
        const int n = (2 * wn + a) * 32 + i;
        const int c = (2 * wc + b) * 32 + li;
        sl[n * FH + c] = tn[a][b][e];
      }
}

// rows 0-31 of a split row image x this wave's 32 weight columns (32-row tiles of the two-workgroup kernels)
__device__ __forceinline__ void mma32_b3(const void* img, int row0, const MmB3::Frag& f, floatx16& acc, int li, int lg) {
  const __bf16* p0 = reinterpret_cast<const __bf16*>(img) + (row0 + li) * BP + 8 * lg;
#pragma unroll
  for (int c = 0; c < BKC; ++c) {
    bf16x8 a0[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) a0[s] = *reinterpret_cast<const bf16x8*>(p0 + s * BIMG + 16 * c);
    mfb(acc, a0[2], f.w[c][0]);
    mfb(acc, a0[0], f.w[c][2]);
    mfb(acc, a0[1], f.w[c][1]);
    mfb(acc, a0[1], f.w[c][0]);
    mfb(acc, a0[0], f.w[c][1]);
    mfb(acc, a0[0], f.w[c][0]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Split-bf16 fragment pack: Wp3[cb][c][s][lane] (bf16x8), s = hi/mid/lo;
//   NT: element j of lane (i,g) = W[(cb*32 + i) * ld + 16c + 8g + j];  NN: W[(16c + 8g + j) * ld + cb*32 + i]
__global__ void pack_w3_k(const float* __restrict__ W, int64_t ld, int ncb, int nn, bf16x8* __restrict__ Wp) {
  const int total = ncb * BKC * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, c = (idx >> 6) % BKC, cb = idx / (64 * BKC);
    const int i = lane & 31, g = lane >> 5;
    bf16x8 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * c + 8 * g + j;
      const float x = nn ? W[(int64_t)k * ld + cb * 32 + i] : W[(int64_t)(cb * 32 + i) * ld + k];
      const __bf16 h = (__bf16)x;
      const float r1 = x - (float)h;
      const __bf16 m = (__bf16)r1;
      hi[j] = h;
      mid[j] = m;
      lo[j] = (__bf16)(r1 - (float)m);
    }
    bf16x8* o = Wp + ((int64_t)(cb * BKC + c) * 3) * 64 + lane;
    o[0] = hi;
    o[64] = mid;
    o[128] = lo;
  }
}

// f16x2 fragment pack: block = { h16x8 frag[c][s][lane] (s = hi/lo of W s_n), float inv[32] = 1 / s_n of its 32 output columns }, s_n from
// the largest magnitude of column n over its whole contraction (taken by the pack kernel itself); element order as pack_w3_k.  Defined with the
// general row GEMM below (pack_w2_gen_k): a single 128-deep weight is its ncg = 1 case.
static void launch_pack_w2_gen(const float* W, int64_t ld, int ncls, int ncg, int nn, void* ws, hipStream_t st);

// 0: fp32 MFMA; 1: bf16x3 split (the default of rounds 2-4: same parity bars, 2.7x the matrix rate);
// 2: f16x2 (MmH2 here, the H2 forms of gnm_tr.hip) in every split-mode matrix kernel -- the default
//    since round 5: every fused step runs at the package power cap, three MFMAs per product instead of six is energy that comes
//    back as time, and its results are no further from fp64 than those of mode 0 or 1 (profiles/r05_f16x2_accuracy.txt)
static int g_matmul_mode = 2;
constexpr size_t kPackBytesPerBlk = MmB3::kPackBytes;   // workspace sizing: the larger of the two

template <class MM>
static void launch_pack(const float* W, int64_t ld, int ncb, int nn, void* wp, hipStream_t st) {
  if (MM::kScaled) launch_pack_w2_gen(W, ld, ncb / 4, 1, nn, wp, st);      // ncb % 4 == 0: whole 128-column groups
  else if (MM::kSplit) hipLaunchKernelGGL(pack_w3_k, dim3(4 * ncb), dim3(256), 0, st, W, ld, ncb, nn, (bf16x8*)wp);
  else hipLaunchKernelGGL(pack_w_k, dim3(4 * ncb), dim3(256), 0, st, W, ld, ncb, nn, (float*)wp);
}

// ------------------------------------------------------------------------------------------
// Y[:, cg*128 + c] = X W_cg^T + bias   (+ gathers and column statistics when EDGE)
// One workgroup per CU = one wave per SIMD with the whole 512-entry register file.
// ------------------------------------------------------------------------------------------
// NCG (number of 128-column groups) is a template parameter so that the column-group loop unrolls:
// a loop with a run-time trip count around the stores would defeat the vmcnt counting below.
template <class MM, bool EDGE, int NCG>
__global__ __launch_bounds__(kBlock, ((MM::kSplit && !EDGE && NCG == 1) || (EDGE && !MM::kSplit)) ? 2 : 1) void rowtile_nt_k(
    int64_t M, const float* __restrict__ X, const void* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t ldy, const float* __restrict__ P,
    const int32_t* __restrict__ isrc, const int32_t* __restrict__ idst, double* __restrict__ partials,
    int64_t tiles_per_block, int ncgs) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];   // X tile image(s)
  constexpr bool INPLACE = EDGE || NCG == 1;  // the output image may overwrite the X image
  __shared__ float ys[INPLACE ? 4 : FTR * FP];   // node mode: output image (X is reused by 5 column groups)
  __shared__ int sd[2 * FTR];
  float* xs = reinterpret_cast<float*>(xraw);   // edge mode: reused as the fp32 output image
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  // ncgs > 1: the output's 128-column groups are spread over workgroups (weights stay in VGPRs for
  // the whole run); the ncgs workgroups of a row range sit on one XCD and share the X tiles in its L2.
  int chunk = xcd_chunk(blockIdx.x, gridDim.x), cgb = 0;
  if (ncgs > 1) {
    const int xcd = blockIdx.x % kXcds, j = blockIdx.x / kXcds;
    cgb = j % ncgs;
    chunk = xcd * (gridDim.x / ncgs / kXcds) + j / ncgs;
  }
  const int64_t ntiles = (M + FTR - 1) / FTR;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, M / FTR);      // tiles [tb0, nfull) are full
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;   // this thread's slot in the coalesced tile image
  const int64_t Mlast = M - 1;

  typename MM::Frag wf;
  auto load_w = [&](int cg) __attribute__((always_inline)) { MM::load_w(wf, Wp, (cgb + cg) * 4 + wave, lane); };
  constexpr int ncg = NCG;
  if (ncg == 1) load_w(0);

  float4 pre[8];
  int pre_idx = 0;
  // branch-free: rows past the end are clamped to the last row (their results are never stored)
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const float* px = X + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4;
      pre[it] = EDGE ? ld4_nt(px) : ld4(px);      // edge rows stream through once; keep L2 for the gathered node rows
    }
    if (EDGE) {
      const int64_t r = clampi(r0 + (tid & (FTR - 1)), Mlast);
      pre_idx = (tid & FTR) ? idst[r] : isrc[r];     // threads 0-63: src, 64-127: dst (128-255 unused)
    }
  };

  Stat4 st;
  st.zero();
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    float rsc[8];      // scaled policies: 1 / s of the rows this thread stages (and stores)
    __syncthreads();   // everyone is done with the previous tile's LDS images
#pragma unroll
    for (int it = 0; it < 8; ++it) rsc[it] = stage_row<MM>(xraw, lrow + 8 * it, lc4, pre[it]);
    if (EDGE && tid < 2 * FTR) sd[tid] = pre_idx;
    __syncthreads();
    const int64_t r0 = tile * FTR;
    // gathers of this tile's B1h[src] / B2h[dst] rows: issued now, consumed in the epilogue
    // (two workgroups per CU in the fp32 edge kernel: the gathers are issued after the MFMAs instead,
    //  the other workgroup's MFMAs cover their latency, and the kernel fits 256 VGPRs)
    constexpr bool LATE = EDGE && !MM::kSplit;
    float4 g1[8], g2[8];
    if (EDGE && !LATE) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = lrow + 8 * it;
        const int64_t s_ = sd[row], d_ = sd[FTR + row];
        g1[it] = ld4(P + s_ * (5 * FH) + 3 * FH + lc4);
        g2[it] = ld4(P + d_ * (5 * FH) + 4 * FH + lc4);
      }
    }
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);   // next tile's X rows, in flight under the MFMAs
#pragma unroll
    for (int cg = 0; cg < ncg; ++cg) {
      if (ncg > 1) load_w(cg);
      floatx16 acc0, acc1;
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
      MM::mma(xraw, wf, acc0, acc1, li, lg);
      if (LATE) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = lrow + 8 * it;
          const int64_t s_ = sd[row], d_ = sd[FTR + row];
          g1[it] = ld4(P + s_ * (5 * FH) + 3 * FH + lc4);
          g2[it] = ld4(P + d_ * (5 * FH) + 4 * FH + lc4);
        }
      }
      float* os = INPLACE ? xs : ys;
      if (INPLACE) __syncthreads();         // all waves are done reading the X image
      else if (cg > 0) __syncthreads();     // previous column group's epilogue is done with ys
      acc_to_lds(os, acc0, acc1, wave, li, lg);
      __syncthreads();
      const float4 b4 = ld4(bias + (cgb + cg) * FH + lc4);
      float4 ci4 = f4(1.f);
      if constexpr (MM::kScaled) ci4 = MM::col_inv4(Wp, (cgb + cg) * 4, lc4);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = lrow + 8 * it;
        const int64_t grow = r0 + row;
        float4 v = ld4(os + row * FP + lc4);
        if constexpr (MM::kScaled) v = v * (ci4 * rsc[it]);      // exact: both factors are powers of two
        v = v + b4;
        if (EDGE) v = v + g1[it] + g2[it];
        if (FULL || grow < M) {
          if (EDGE) st4_nt(Y + grow * ldy + (cgb + cg) * FH + lc4, v);
          else st4(Y + grow * ldy + (cgb + cg) * FH + lc4, v);
          if (EDGE) st.add_prod(v, v);
        }
      }
    }
  };

  if (tb0 < tb1) prefetch(tb0);
  if (tb0 < nfull) {
    // throw-away stores behind the first prefetch (same addresses the first epilogue rewrites): hipcc merges the vector-memory
    // scoreboard of the loop entry with that of the back edge and keeps the weaker guarantee -- without stores behind the first
    // prefetch it would wait vmcnt(0) (= for the previous tile's stores) before the last prefetched row on EVERY iteration;
    // these make both edges alike
#pragma unroll
    for (int it = 0; it < 8; ++it) st4(Y + (tb0 * FTR + lrow + 8 * it) * ldy + cgb * FH + lc4, f4(0.f));
    for (int64_t tile = tb0; tile < nfull; ++tile) body(full_t{}, tile);
  }
  if (nfull < tb1 && nfull >= tb0) body(ragged_t{}, nfull);
  if (EDGE) {
    __syncthreads();   // the last epilogue is done with the LDS image we reuse for the reduction
    block_stat_store<FH>(st, reinterpret_cast<double*>(xs), partials, chunk);
  }
}

// ------------------------------------------------------------------------------------------
// Split-mode edge t kernel with 32-row tiles: t = e W3^T + b3 + B1h[src] + B2h[dst] + BatchNorm sums.
// 69 KB of LDS and <= 256 VGPRs -> two workgroups per CU (the 64-row version needs 330 registers, and
// with one wave per SIMD its split staging and epilogue leave the matrix pipe 32 % busy).  The two
// 32-row halves of the three bf16 images are two tile buffers, the fp32 output image is separate.
// ------------------------------------------------------------------------------------------
constexpr int ER3 = 32;
// (round 4: the split / staging of tile k+1 is issued inside the matrix phase of tile k, see edge_t32_h256p_k; the kernel that ran
//  the phases one after the other -- 14.9 against 14.3 ms per step, bit-identical -- was removed in round 5, numbers in
//  profiles/r04_ab_kernels.txt)
// MM = MmB3 (six MFMAs per product) or MmH2 (three; the row factors of a tile are made when it is staged, inside the matrix
// phase of the tile before it, and used by the same thread in the tile's epilogue)
template <class MM>
__global__ __launch_bounds__(kBlock, 2) void edge_t32_b3p_k(
    int64_t M, const float* __restrict__ X, const void* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ Y, const float* __restrict__ P, const int32_t* __restrict__ isrc,
    const int32_t* __restrict__ idst, double* __restrict__ partials, int64_t tiles_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];
  __shared__ float os[ER3 * FP];
  __shared__ int sd[2][2 * ER3];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (M + ER3 - 1) / ER3;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, M / ER3);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  const int32_t* const ibase = (lane & 32) ? idst : isrc;

  typename MM::Frag wf;
  MM::load_w(wf, Wp, wave, lane);
  const float4 b4 = ld4(bias + lc4);
  float4 ci4 = f4(1.f);
  if constexpr (MM::kScaled) ci4 = MM::col_inv4(Wp, 0, lc4);
  float rsc[2][4];                   // scaled policies: 1 / s of this thread's rows in tile buffer 0 / 1
  float4 pre[2][4];
  int pidx[2] = {0, 0};
  auto prefetch = [&](float4 (&buf)[4], int& idx, int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = (tile < tb1 ? tile : tb1 - 1) * ER3;
#pragma unroll
    for (int it = 0; it < 4; ++it) buf[it] = ld4_nt(X + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
    idx = ibase[clampi(r0 + (lane & 31), Mlast)];
  };
  Stat4 st;
  st.zero();
  auto body = [&](auto tag, float4 (&nbuf)[4], int& nidx, int64_t tile, int hb) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * ER3;
    float4 g1[4], g2[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = lrow + 8 * it;
      const int64_t s_ = sd[hb][row], d_ = sd[hb][ER3 + row];
      g1[it] = ld4(P + s_ * (5 * FH) + 3 * FH + lc4);
      g2[it] = ld4(P + d_ * (5 * FH) + 4 * FH + lc4);
    }
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if constexpr (MM::kScaled) {
      const _Float16* p0 = reinterpret_cast<const _Float16*>(xraw) + (32 * hb + li) * BP + 8 * lg;
#pragma unroll
      for (int c = 0; c < BKC; ++c) {
        h16x8 a0[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) a0[s_] = *reinterpret_cast<const h16x8*>(p0 + s_ * BIMG + 16 * c);
        mfh(acc, a0[1], wf.w[c][0]);
        if (c < 4) rsc[hb ^ 1][c] = MM::stage(xraw, 32 * (hb ^ 1) + lrow + 8 * c, lc4, nbuf[c]);
        mfh(acc, a0[0], wf.w[c][1]);
        mfh(acc, a0[0], wf.w[c][0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const __bf16* p0 = reinterpret_cast<const __bf16*>(xraw) + (32 * hb + li) * BP + 8 * lg;
#pragma unroll
      for (int c = 0; c < BKC; ++c) {
        bf16x8 a0[3];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) a0[s_] = *reinterpret_cast<const bf16x8*>(p0 + s_ * BIMG + 16 * c);
        mfb(acc, a0[2], wf.w[c][0]);
        mfb(acc, a0[0], wf.w[c][2]);
        mfb(acc, a0[1], wf.w[c][1]);
        if (c < 4) MM::stage(xraw, 32 * (hb ^ 1) + lrow + 8 * c, lc4, nbuf[c]);
        mfb(acc, a0[1], wf.w[c][0]);
        mfb(acc, a0[0], wf.w[c][1]);
        mfb(acc, a0[0], wf.w[c][0]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    sd[hb ^ 1][lane] = nidx;
    prefetch(nbuf, nidx, tile + 3);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) os[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + wave * 32 + li] = acc[e];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = lrow + 8 * it;
      const int64_t grow = r0 + row;
      float4 v = ld4(os + row * FP + lc4);
      if constexpr (MM::kScaled) v = v * (ci4 * rsc[hb][it]);      // exact: powers of two
      v = v + b4 + g1[it] + g2[it];
      if (FULL || grow < M) {
        st4_nt(Y + grow * FH + lc4, v);
        st.add_prod(v, v);
      }
    }
  };
  if (tb0 < tb1) {
    prefetch(pre[0], pidx[0], tb0);
    prefetch(pre[1], pidx[1], tb0 + 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) rsc[0][it] = stage_row<MM>(xraw, lrow + 8 * it, lc4, pre[0][it]);
    sd[0][lane] = pidx[0];
    prefetch(pre[0], pidx[0], tb0 + 2);
    __syncthreads();
  }
  int64_t tile = tb0;
  for (; tile + 2 <= nfull; tile += 2) {
    body(full_t{}, pre[1], pidx[1], tile, 0);
    body(full_t{}, pre[0], pidx[0], tile + 1, 1);
  }
  int hb = 0;
  for (; tile < tb1; ++tile, hb ^= 1) {
    if (hb == 0) body(ragged_t{}, pre[1], pidx[1], tile, 0);
    else body(ragged_t{}, pre[0], pidx[0], tile, 1);
  }
  __syncthreads();
  block_stat_store<FH>(st, reinterpret_cast<double*>(xraw), partials, chunk);
}

// ------------------------------------------------------------------------------------------
// H = 256 (the reference's default dim_latent, hyperparameters.py:8): t = e W3^T + b3 + B1h[src] + B2h[dst] and the
// BatchNorm sums in ONE pass over e (gated_gcn_full.py:113,120-122), split mode.  The three bf16 images of the
// 256 x 256 weight are 384 KB -- no workgroup can keep them -- so a workgroup keeps ONE output half J (128 columns)
// stationary in the registers of EIGHT waves = 4 column blocks of 32 x 2 contraction halves of 128 (96 VGPRs per wave,
// as in edge_t32_b3_k), and the two workgroups of a row chunk (J = 0, 1) sit on one XCD and read the same e rows through
// its L2.  Per 32-row tile: 512 threads stage the 32 x 256 tile into two image sets (one per contraction half), every
// wave runs its 48 MFMAs, the two halves of the contraction meet in two fp32 LDS images that the epilogue adds.
// 130 KB of LDS, one workgroup per CU.  Replaces gemm NT [E,256,256] + edge_t_stats_fwd (t written and re-read once).
// ------------------------------------------------------------------------------------------
constexpr int WH = 2 * FH;          // the wide hidden size
constexpr int kBlockW = 512;
// The split / staging of tile k+1 is issued INSIDE the matrix phase of tile k: a wave's 48 MFMAs are one dependent chain (62 cycles
// apiece on its own), which leaves the vector ALU idle -- the next tile's 4 rows per thread are split and written to the OTHER half
// of the image set between the chunks of the chain, so that per tile only the epilogue stays outside the matrix phase (27.1 -> 24.5
// ms per step against the kernel that ran the phases one after the other, bit-identical; that kernel was removed in round 5).
template <class MM>
__global__ __launch_bounds__(kBlockW, 1) void edge_t32_h256p_k(
    int64_t M, const float* __restrict__ X, const void* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ Y, const float* __restrict__ P, const int32_t* __restrict__ isrc,
    const int32_t* __restrict__ idst, double* __restrict__ partials, int nchunk, int64_t tiles_per_chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[2 * MM::kImgBytes];
  __shared__ float os[2 * ER3 * FP];
  __shared__ int sd[2][2 * ER3];
  __shared__ float rinv[2][ER3];       // MmH2: 1 / s of the rows of tile buffer 0 / 1 (h2_stage_row64)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int cb = wave & 3, kh = wave >> 2;
  const int xcd = blockIdx.x % kXcds, jj = blockIdx.x / kXcds;
  const int J = jj & 1, chunk = xcd * (nchunk / kXcds) + (jj >> 1);
  const int64_t ntiles = (M + ER3 - 1) / ER3;
  const int64_t tb0 = (int64_t)chunk * tiles_per_chunk;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_chunk);
  const int64_t nfull = min(tb1, M / ER3);
  const int srow = tid >> 6, sc = (tid & 63) * 4;
  unsigned char* const simg = xraw + (sc >> 7) * MM::kImgBytes;
  const int slc4 = sc & (FH - 1);
  const int erow = tid >> 5, ec4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  const int32_t* const ibase = (lane & 32) ? idst : isrc;

  typename MM::Frag wf;
  MM::load_w(wf, Wp, (J * 2 + kh) * 4 + cb, lane);
  const float4 b4 = ld4(bias + J * FH + ec4);
  float4 ci4 = f4(1.f);
  if constexpr (MM::kScaled) ci4 = MM::col_inv4(Wp, J * 2 * 4, ec4);      // the column factors span both contraction halves
  auto stage = [&](int buf, int r, const float4& v) __attribute__((always_inline)) {
    if constexpr (MM::kScaled) h2_stage_row64(simg, 32 * buf + r, slc4, v, &rinv[buf][r], (tid & 63) == 0);
    else MM::stage(simg, 32 * buf + r, slc4, v);
  };
  float4 pre[2][4];
  int pidx[2] = {0, 0};
  auto prefetch = [&](float4 (&buf)[4], int& idx, int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = (tile < tb1 ? tile : tb1 - 1) * ER3;
#pragma unroll
    for (int it = 0; it < 4; ++it) buf[it] = ld4(X + clampi(r0 + srow + 8 * it, Mlast) * WH + sc);
    idx = ibase[clampi(r0 + (lane & 31), Mlast)];
  };
  Stat4 st;
  st.zero();
  // tile `tile` is staged in image half hb; (nbuf, nidx) hold tile + 1, which is staged into half hb ^ 1 during the MFMAs
  auto body = [&](auto tag, float4 (&nbuf)[4], int& nidx, int64_t tile, int hb) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * ER3;
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // the factors of THIS tile's rows (staged during the previous body, two barriers ago) before anybody stages into this buffer again
    float ri[2] = {1.f, 1.f};
    if constexpr (MM::kScaled) { ri[0] = rinv[hb][erow]; ri[1] = rinv[hb][erow + 16]; }
    if constexpr (MM::kScaled) {
      const _Float16* p0 = reinterpret_cast<const _Float16*>(xraw + kh * MM::kImgBytes) + (32 * hb + li) * BP + 8 * lg;
      h16x8 a0[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) a0[s_] = *reinterpret_cast<const h16x8*>(p0 + s_ * BIMG);
#pragma unroll
      for (int c = 0; c < BKC; ++c) {
        h16x8 n0[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) n0[s_] = c + 1 < BKC ? *reinterpret_cast<const h16x8*>(p0 + s_ * BIMG + 16 * (c + 1)) : a0[s_];
        __builtin_amdgcn_sched_barrier(0);
        mfh(acc, a0[1], wf.w[c][0]);
        if (c < 4) stage(hb ^ 1, srow + 8 * c, nbuf[c]);     // the next tile, one row per chunk
        mfh(acc, a0[0], wf.w[c][1]);
        mfh(acc, a0[0], wf.w[c][0]);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) a0[s_] = n0[s_];
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const __bf16* p0 = reinterpret_cast<const __bf16*>(xraw + kh * MM::kImgBytes) + (32 * hb + li) * BP + 8 * lg;
      bf16x8 a0[3];
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) a0[s_] = *reinterpret_cast<const bf16x8*>(p0 + s_ * BIMG);
#pragma unroll
      for (int c = 0; c < BKC; ++c) {
        bf16x8 n0[3];               // the next chunk's fragments are requested before this chunk's MFMAs (one workgroup per CU:
#pragma unroll                      // nobody else hides the LDS round trip)
        for (int s_ = 0; s_ < 3; ++s_) n0[s_] = c + 1 < BKC ? *reinterpret_cast<const bf16x8*>(p0 + s_ * BIMG + 16 * (c + 1)) : a0[s_];
        __builtin_amdgcn_sched_barrier(0);
        mfb(acc, a0[2], wf.w[c][0]);
        mfb(acc, a0[0], wf.w[c][2]);
        mfb(acc, a0[1], wf.w[c][1]);
        if (c < 4) stage(hb ^ 1, srow + 8 * c, nbuf[c]);     // the next tile, one row per chunk
        mfb(acc, a0[1], wf.w[c][0]);
        mfb(acc, a0[0], wf.w[c][1]);
        mfb(acc, a0[0], wf.w[c][0]);
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) a0[s_] = n0[s_];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    float4 g1[2], g2[2];          // this tile's B1h[src] / B2h[dst] rows: in flight across the two barriers below
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = erow + 16 * it;
      const int64_t s_ = sd[hb][row], d_ = sd[hb][ER3 + row];
      g1[it] = ld4(P + s_ * (5 * WH) + 3 * WH + J * FH + ec4);
      g2[it] = ld4(P + d_ * (5 * WH) + 4 * WH + J * FH + ec4);
    }
    if (wave == 0) sd[hb ^ 1][lane] = nidx;
    prefetch(nbuf, nidx, tile + 3);
    __syncthreads();          // every wave is past the previous tile's epilogue: the partial-result images are free
    float* const oh = os + kh * (ER3 * FP);
#pragma unroll
    for (int e = 0; e < 16; ++e) oh[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + cb * 32 + li] = acc[e];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = erow + 16 * it;
      const int64_t grow = r0 + row;
      float4 v = ld4(os + row * FP + ec4) + ld4(os + ER3 * FP + row * FP + ec4);
      if constexpr (MM::kScaled) v = v * (ci4 * ri[it]);      // exact: powers of two
      v = v + b4 + g1[it] + g2[it];
      if (FULL || grow < M) {
        st4_nt(Y + grow * WH + J * FH + ec4, v);
        st.add_prod(v, v);
      }
    }
  };
  if (tb0 < tb1) {
    prefetch(pre[0], pidx[0], tb0);
    prefetch(pre[1], pidx[1], tb0 + 1);
#pragma unroll
    for (int it = 0; it < 4; ++it) stage(0, srow + 8 * it, pre[0][it]);
    if (wave == 0) sd[0][lane] = pidx[0];
    prefetch(pre[0], pidx[0], tb0 + 2);
    __syncthreads();
  }
  int64_t tile = tb0;
  for (; tile + 2 <= nfull; tile += 2) {
    body(full_t{}, pre[1], pidx[1], tile, 0);
    body(full_t{}, pre[0], pidx[0], tile + 1, 1);
  }
  int hb = 0;
  for (; tile < tb1; ++tile, hb ^= 1) {
    if (hb == 0) body(ragged_t{}, pre[1], pidx[1], tile, 0);
    else body(ragged_t{}, pre[0], pidx[0], tile, 1);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st.a[i] += __shfl_xor(st.a[i], 32, 64);
    st.b[i] += __shfl_xor(st.b[i], 32, 64);
  }
  double* red = reinterpret_cast<double*>(xraw);
  if (lane < 32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[(wave * 2 + 0) * FH + lane * 4 + i] = st.a[i];
      red[(wave * 2 + 1) * FH + lane * 4 + i] = st.b[i];
    }
  }
  __syncthreads();
  if (tid < 2 * FH) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < kBlockW / 64; ++w) acc += red[w * 2 * FH + tid];
    partials[((size_t)chunk * 2 + (tid >> 7)) * WH + J * FH + (tid & (FH - 1))] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// H = 256 backward twin: gt = gamma*rstd*(gu - m1 - that*m2), gu = ge*[t*scale+shift > 0] (autograd of
// gated_gcn_full.py:122) formed while the tile is staged, written ONCE for the weight-gradient GEMM, and
// ge_out = ge + gt W3 (autograd of :113) in the same pass; organisation of edge_t32_h256_k (class J = output half of
// ge_out, eight waves = 4 column blocks x 2 contraction halves of gt).  Both classes read all 256 columns of ge / t
// (the second through L2), each writes its own half of gt and of ge_out: ge_out must NOT alias ge.
// Replaces edge_bwd_gt + gemm NN [E,256,256] (gt re-read, ge read twice).  (Staging the next tile inside the matrix phase as
// edge_t32_h256p_k does was measured here too: 31.2 vs 30.8 ms per step -- this kernel moves twice the bytes and waits on HBM.)
// ------------------------------------------------------------------------------------------
template <class MM>
__global__ __launch_bounds__(kBlockW, 1) void edge_gt_nn_h256_k(
    int64_t M, const float* __restrict__ ge, const float* __restrict__ t, const float* __restrict__ stat,
    const float* __restrict__ bstat, const float* __restrict__ gamma, const void* __restrict__ Wp,
    float* __restrict__ gt, float* __restrict__ ge_out, int nchunk, int64_t tiles_per_chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[2 * MM::kImgBytes];   // [contraction half][split images] (rows 0-31 used)
  __shared__ float os[2 * ER3 * FP];
  __shared__ float rinv[ER3];          // MmH2: 1 / s of the tile's rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int cb = wave & 3, kh = wave >> 2;
  const int xcd = blockIdx.x % kXcds, jj = blockIdx.x / kXcds;
  const int J = jj & 1, chunk = xcd * (nchunk / kXcds) + (jj >> 1);
  const int64_t ntiles = (M + ER3 - 1) / ER3;
  const int64_t tb0 = (int64_t)chunk * tiles_per_chunk;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_chunk);
  const int64_t nfull = min(tb1, M / ER3);
  const int srow = tid >> 6, sc = (tid & 63) * 4;
  unsigned char* const simg = xraw + (sc >> 7) * MM::kImgBytes;
  const int slc4 = sc & (FH - 1);
  const bool mine = (sc >> 7) == J;                          // this thread's gt columns belong to this class's half
  const int erow = tid >> 5, ec4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;

  const float4 mu = ld4(stat + sc), rs = ld4(stat + WH + sc), scl = ld4(stat + 2 * WH + sc), sh = ld4(stat + 3 * WH + sc);
  const float4 m1 = ld4(bstat + sc), m2 = ld4(bstat + WH + sc), cc = ld4(gamma + sc) * rs;
  typename MM::Frag wf;
  MM::load_w(wf, Wp, (J * 2 + kh) * 4 + cb, lane);
  float4 ci4 = f4(1.f);
  if constexpr (MM::kScaled) ci4 = MM::col_inv4(Wp, J * 2 * 4, ec4);
  float4 pg[4], pt[4];
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = (tile < tb1 ? tile : tb1 - 1) * ER3;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t o = clampi(r0 + srow + 8 * it, Mlast) * WH + sc;
      pg[it] = ld4(ge + o);
      pt[it] = ld4(t + o);
    }
  };
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * ER3;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t grow = r0 + srow + 8 * it;
      const float4 gu = gate4(fma4(pt[it], scl, sh), pg[it]);
      float4 g = cc * (gu - m1 - ((pt[it] - mu) * rs) * m2);
      if (!FULL && grow >= M) g = f4(0.f);
      if (mine && (FULL || grow < M)) st4_nt(gt + grow * WH + sc, g);
      if constexpr (MM::kScaled) h2_stage_row64(simg, srow + 8 * it, slc4, g, &rinv[srow + 8 * it], (tid & 63) == 0);
      else MM::stage(simg, srow + 8 * it, slc4, g);
    }
    __syncthreads();
    float ri[2] = {1.f, 1.f};        // before the next tile's staging (no barrier at the top of a tile) can rewrite them
    if constexpr (MM::kScaled) { ri[0] = rinv[erow]; ri[1] = rinv[erow + 16]; }
    float4 rr[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) rr[it] = ld4(ge + clampi(r0 + erow + 16 * it, Mlast) * WH + J * FH + ec4);   // L2: staged a moment ago
    prefetch(tile + 1);
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    if constexpr (MM::kScaled) mma32_h2(xraw + kh * MM::kImgBytes, 0, wf, acc, li, lg);
    else mma32_b3(xraw + kh * MM::kImgBytes, 0, wf, acc, li, lg);
    float* const oh = os + kh * (ER3 * FP);
#pragma unroll
    for (int e = 0; e < 16; ++e) oh[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + cb * 32 + li] = acc[e];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = erow + 16 * it;
      const int64_t grow = r0 + row;
      float4 v = ld4(os + row * FP + ec4) + ld4(os + ER3 * FP + row * FP + ec4);
      if constexpr (MM::kScaled) v = v * (ci4 * ri[it]);
      v = v + rr[it];
      if (FULL || grow < M) st4_nt(ge_out + grow * WH + J * FH + ec4, v);
    }
  };
  if (tb0 < tb1) prefetch(tb0);
  int64_t tile = tb0;
  for (; tile < nfull; ++tile) body(full_t{}, tile);
  for (; tile < tb1; ++tile) body(ragged_t{}, tile);
}

// ------------------------------------------------------------------------------------------
// fused edge backward, fp32-MFMA mode: gt prologue + NN (ge_in) + TN (gW3 slab) + column sum of gt
// (the split mode runs edge_bwd_tr_k, gnm_tr.hip; the 64-row split kernel of round 1 -- 159 KB of LDS, one workgroup per CU,
//  4.0-4.3 against 3.5 ms -- was removed in round 5)
// ------------------------------------------------------------------------------------------
// 32-row tiles: 236 VGPRs and 37 KB of LDS, so TWO workgroups share a CU and one's gt prologue / epilogue runs under the
// other's MFMAs (a 64-row kernel holds 416 registers and leaves the matrix pipe idle during those phases: 68 % busy).
constexpr int FTR2 = 32;

__global__ __launch_bounds__(kBlock, 2) void edge_bwd_fused32_k(
    int64_t E, const float* ge, float* ge_out, const float* __restrict__ t, const float* __restrict__ e_in,
    const float* __restrict__ stat, const float* __restrict__ bstat, const float* __restrict__ gamma,
    const void* __restrict__ Wp, float* __restrict__ slab, double* __restrict__ partials, int64_t tiles_per_block) {
  __shared__ float gs[FTR2 * FP];      // gt tile, later the transposed output image
  __shared__ float es[FTR2 * FP];      // e_in tile
  __shared__ float cs[7 * FH];         // mu, rstd, scale, shift, m1, m2, c = gamma*rstd
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + FTR2 - 1) / FTR2;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, E / FTR2);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Elast = E - 1;
  for (int c = tid; c < FH; c += kBlock) {
    cs[c] = stat[c];
    cs[FH + c] = stat[FH + c];
    cs[2 * FH + c] = stat[2 * FH + c];
    cs[3 * FH + c] = stat[3 * FH + c];
    cs[4 * FH + c] = bstat[c];
    cs[5 * FH + c] = bstat[FH + c];
    cs[6 * FH + c] = gamma[c] * stat[FH + c];
  }
  MmF32::Frag wf;
  MmF32::load_w(wf, Wp, wave, lane);
  floatx16 tn[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) tn[a][b][e] = 0.f;
  double cg0 = 0.0, cg1 = 0.0, cg2 = 0.0, cg3 = 0.0;
  __syncthreads();

  float4 pg[4], pt[4], pe_[4];
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR2;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int64_t o = clampi(r0 + lrow + 8 * it, Elast) * FH + lc4;
      pg[it] = ld4(ge + o);
      pt[it] = ld4(t + o);
      pe_[it] = ld4(e_in + o);
    }
  };
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * FTR2;
    float4 gk[4];
    {
      const float4 mu = ld4(cs + lc4), rs = ld4(cs + FH + lc4), sc = ld4(cs + 2 * FH + lc4),
                   sh = ld4(cs + 3 * FH + lc4), m1 = ld4(cs + 4 * FH + lc4), m2 = ld4(cs + 5 * FH + lc4),
                   cc = ld4(cs + 6 * FH + lc4);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = lrow + 8 * it;
        const bool ok = FULL || (r0 + row < E);
        gk[it] = pg[it];
        const float4 gu = gate4(fma4(pt[it], sc, sh), pg[it]);
        float4 gt = cc * (gu - m1 - ((pt[it] - mu) * rs) * m2);
        float4 ev = pe_[it];
        if (!ok) { gt = f4(0.f); ev = f4(0.f); }
        cg0 += (double)gt.x; cg1 += (double)gt.y; cg2 += (double)gt.z; cg3 += (double)gt.w;
        st4(gs + row * FP + lc4, gt);
        st4(es + row * FP + lc4, ev);
      }
    }
    __syncthreads();
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);
    // ---- acc = gt W3 (32 rows x this wave's 32 columns) ----
    floatx16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    {
      const float* p0 = gs + li * FP + 4 * lg;
      float4 a0 = ld4(p0);
#pragma unroll
      for (int q = 0; q < FKQ; ++q) {
        float4 n0 = a0;
        if (q + 1 < FKQ) n0 = ld4(p0 + 8 * (q + 1));
        __builtin_amdgcn_sched_barrier(0);
        mfma4(acc, a0, wf.w[q]);
        a0 = n0;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- gW3[n][c] += sum_rows gt[row][n] e_in[row][c] over the tile's 32 rows ----
    {
      const float* ga = gs + 4 * lg * FP + li;
      const float* eb = es + 4 * lg * FP + li;
#pragma unroll
      for (int q = 0; q < FTR2 / 8; ++q) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = (8 * q + r) * FP;
          const float a0 = ga[o + (2 * wn) * 32], a1 = ga[o + (2 * wn + 1) * 32];
          const float b0 = eb[o + (2 * wc) * 32], b1 = eb[o + (2 * wc + 1) * 32];
          tn[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, tn[0][0], 0, 0, 0);
          tn[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, tn[0][1], 0, 0, 0);
          tn[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, tn[1][0], 0, 0, 0);
          tn[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, tn[1][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();   // gt / e_in images are dead: reuse gs as the transposed output image
#pragma unroll
    for (int e = 0; e < 16; ++e) gs[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + wave * 32 + li] = acc[e];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = lrow + 8 * it;
      const int64_t grow = r0 + row;
      if (FULL || grow < E) st4(ge_out + grow * FH + lc4, ld4(gs + row * FP + lc4) + gk[it]);
    }
    __syncthreads();
  };
  if (tb0 < tb1) prefetch(tb0);
#pragma unroll
  for (int it = 0; it < 4; ++it) st4(slab + (size_t)chunk * FH * FH + (lrow + 8 * it) * FH + lc4, f4(0.f));   // scoreboard equalisation, see rowtile_nt_k
  for (int64_t tile = tb0; tile < nfull; ++tile) body(full_t{}, tile);
  if (nfull < tb1 && nfull >= tb0) body(ragged_t{}, nfull);

  tn_store_slab<MmF32>(slab + (size_t)chunk * FH * FH, tn, wn, wc, li, lg);
  double* red = reinterpret_cast<double*>(gs);     // 8 row-slots x 128 doubles = 8 KB
  red[lrow * FH + lc4 + 0] = cg0;
  red[lrow * FH + lc4 + 1] = cg1;
  red[lrow * FH + lc4 + 2] = cg2;
  red[lrow * FH + lc4 + 3] = cg3;
  __syncthreads();
  if (tid < FH) {
    double s_ = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s_ += red[k * FH + tid];
    partials[(size_t)chunk * FH + tid] = s_;
  }
}

// ------------------------------------------------------------------------------------------
// node-level backward of the 5-way projection (autograd of gated_gcn_full.py:107-112):
//   rowtile_nn_acc_k   gh_in = gh_out + gP W5            (K = 5*128, accumulated over 5 groups)
//   tn_colgroup_k      gW5[cg] = gP[:,cg]^T h_in, gb5[cg] = sum gP[:,cg]   (5 workgroup classes)
// ------------------------------------------------------------------------------------------
template <class MM>
__global__ __launch_bounds__(kBlock, MM::kSplit ? 1 : 2) void rowtile_nn_acc_k(
    int64_t M, const float* __restrict__ X, int64_t ldx, int ncg, const void* __restrict__ Wp,
    const float* __restrict__ R, float* __restrict__ Y, int64_t tiles_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];
  float* xs = reinterpret_cast<float*>(xraw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (M + FTR - 1) / FTR;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, M / FTR);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  float4 pre[8];
  auto prefetch = [&](int64_t tile, int cg) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) pre[it] = ld4(X + clampi(r0 + lrow + 8 * it, Mlast) * ldx + cg * FH + lc4);
  };
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * FTR;
    float4 rr[8];   // residual rows, consumed in the epilogue
#pragma unroll
    for (int it = 0; it < 8; ++it) rr[it] = ld4(R + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
    floatx16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (int cg = 0; cg < ncg; ++cg) {
      typename MM::Frag wf;
      MM::load_w(wf, Wp, cg * 4 + wave, lane);
      __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
      for (int it = 0; it < 8; ++it) MM::stage(xraw, lrow + 8 * it, lc4, pre[it]);
      __syncthreads();
      if (cg + 1 < ncg) prefetch(tile, cg + 1);
      else prefetch(tile + 1 < tb1 ? tile + 1 : tile, 0);
      MM::mma(xraw, wf, acc0, acc1, li, lg);
    }
    __syncthreads();
    acc_to_lds(xs, acc0, acc1, wave, li, lg);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = lrow + 8 * it;
      const int64_t grow = r0 + row;
      if (FULL || grow < M) st4(Y + grow * FH + lc4, ld4(xs + row * FP + lc4) + rr[it]);
    }
  };
  if (tb0 < tb1) prefetch(tb0, 0);
  if (tb0 < nfull) {
#pragma unroll
    for (int it = 0; it < 8; ++it) st4(Y + (tb0 * FTR + lrow + 8 * it) * FH + lc4, f4(0.f));   // scoreboard equalisation, see rowtile_nt_k
    for (int64_t tile = tb0; tile < nfull; ++tile) body(full_t{}, tile);
  }
  if (nfull < tb1 && nfull >= tb0) body(ragged_t{}, nfull);
}

// Split-mode variant: the weight fragments of a column group (96 VGPRs) are loaded once per GROUP of
// T row tiles whose accumulators stay in registers -- in split mode the MFMAs are cheap enough that
// re-reading 5 x 96 KB of fragments from L2 for every 64-row tile was the bound.
template <class MM, int T>
__global__ __launch_bounds__(kBlock, 1) void rowtile_nn_group_k(
    int64_t M, const float* __restrict__ X, int64_t ldx, int ncg, const void* __restrict__ Wp,
    const float* __restrict__ R, float* __restrict__ Y, int64_t groups_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];
  float* xs = reinterpret_cast<float*>(xraw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ngroups = (M + FTR * T - 1) / (FTR * T);
  const int64_t g0 = (int64_t)chunk * groups_per_block;
  const int64_t g1 = min(ngroups, g0 + groups_per_block);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  // Two steps (tile, column group) of rows in flight: with cheap MFMAs one step is shorter than the
  // HBM latency.  T is even, so the buffer parity of a step is its tile index within the group.
  static_assert(T % 2 == 0, "two-deep prefetch assumes an even group size");
  float4 pre[2][8];
  auto prefetch = [&](float4 (&buf)[8], int64_t g, int cg, int tl) __attribute__((always_inline)) {
    if (tl >= T) { tl -= T; ++cg; }
    if (cg >= ncg) { cg = 0; g = g + 1 < g1 ? g + 1 : g; }
    const int64_t r0 = (g * T + tl) * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) buf[it] = ld4(X + clampi(r0 + lrow + 8 * it, Mlast) * ldx + cg * FH + lc4);
  };
  if (g0 < g1) {
    prefetch(pre[0], g0, 0, 0);
    prefetch(pre[1], g0, 0, 1);
  }
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t t0 = g * T;
    floatx16 acc[T][2];
#pragma unroll
    for (int tl = 0; tl < T; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[tl][0][e] = 0.f; acc[tl][1][e] = 0.f; }
    for (int cg = 0; cg < ncg; ++cg) {
      typename MM::Frag wf;
      MM::load_w(wf, Wp, cg * 4 + wave, lane);
#pragma unroll
      for (int tl = 0; tl < T; ++tl) {
        __syncthreads();   // previous fragment reads are done
#pragma unroll
        for (int it = 0; it < 8; ++it) MM::stage(xraw, lrow + 8 * it, lc4, pre[tl & 1][it]);
        __syncthreads();
        prefetch(pre[tl & 1], g, cg, tl + 2);
        MM::mma(xraw, wf, acc[tl][0], acc[tl][1], li, lg);
      }
    }
#pragma unroll
    for (int tl = 0; tl < T; ++tl) {
      const int64_t r0 = (t0 + tl) * FTR;
      float4 rr[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) rr[it] = ld4(R + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
      __syncthreads();
      acc_to_lds(xs, acc[tl][0], acc[tl][1], wave, li, lg);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = lrow + 8 * it;
        const int64_t grow = r0 + row;
        if (grow < M) st4(Y + grow * FH + lc4, ld4(xs + row * FP + lc4) + rr[it]);
      }
    }
  }
}

// The same with 32-row tiles: 26 KB of LDS and <= 256 VGPRs -> two workgroups per CU, so that the split
// staging of one overlaps the MFMAs of the other.
constexpr int NR3 = 32;
template <class MM, int T>
__global__ __launch_bounds__(kBlock, 2) void rowtile_nn_group32_b3_k(
    int64_t M, const float* __restrict__ X, int64_t ldx, int ncg, const void* __restrict__ Wp,
    const float* __restrict__ R, float* __restrict__ Y, int64_t groups_per_block) {
  // the split images keep the 64-row layout of MM::stage; rows 0-31 / 32-63 are two tile buffers,
  // so a step needs ONE barrier (the next step stages into the half the slower waves are not reading)
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];
  __shared__ int eref[MM::kScaled ? T * NR3 : 1];                // MmH2: the rows' reference exponents (h2_stage_nn)
  __shared__ __attribute__((aligned(16))) float fsc[MM::kScaled ? 2 * NR3 : 4];   // ... and the accumulator factors, per tile buffer
  float* xs = reinterpret_cast<float*>(xraw);                    // later: 32 x 132 fp32 output image
  static_assert(T % 2 == 0, "two-deep prefetch assumes an even group size");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ngroups = (M + NR3 * T - 1) / (NR3 * T);
  const int64_t g0 = (int64_t)chunk * groups_per_block;
  const int64_t g1 = min(ngroups, g0 + groups_per_block);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  float4 ci4 = f4(1.f);
  if constexpr (MM::kScaled) ci4 = MM::col_inv4(Wp, 0, lc4);      // the column factors span the whole K: block 0's serve every group
  // T steps (one whole column group of the row group) of rows in flight: a 32-row step of cheap MFMAs is
  // far shorter than the HBM latency.  pre[tl] always holds tile tl of the NEXT column group.
  float4 pre[T][4];
  auto prefetch = [&](float4 (&buf)[4], int64_t g, int cg, int tl) __attribute__((always_inline)) {
    if (cg >= ncg) { cg = 0; g = g + 1 < g1 ? g + 1 : g; }
    const int64_t r0 = (g * T + tl) * NR3;
#pragma unroll
    for (int it = 0; it < 4; ++it) buf[it] = ld4_nt(X + clampi(r0 + lrow + 8 * it, Mlast) * ldx + cg * FH + lc4);
  };
  if (g0 < g1) {
#pragma unroll
    for (int tl = 0; tl < T; ++tl) prefetch(pre[tl], g0, 0, tl);
  }
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t t0 = g * T;
    floatx16 acc[T];
#pragma unroll
    for (int tl = 0; tl < T; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tl][e] = 0.f;
    for (int cg = 0; cg < ncg; ++cg) {
      typename MM::Frag wf;
      MM::load_w(wf, Wp, cg * 4 + wave, lane);
#pragma unroll
      for (int tl = 0; tl < T; ++tl) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = lrow + 8 * it;
          if constexpr (MM::kScaled) h2_stage_nn(xraw, 32 * (tl & 1) + row, lc4, pre[tl][it], eref + tl * NR3 + row, fsc + (tl & 1) * NR3 + row,
                                                 cg == 0, (tid & 31) == 0);
          else MM::stage(xraw, 32 * (tl & 1) + row, lc4, pre[tl][it]);
        }
        __syncthreads();
        prefetch(pre[tl], g, cg + 1, tl);
        if constexpr (MM::kScaled) {
          if (cg > 0) acc_rescale_rows(acc[tl], fsc + (tl & 1) * NR3, lg);
          mma32_h2(xraw, 32 * (tl & 1), wf, acc[tl], li, lg);
        } else {
          mma32_b3(xraw, 32 * (tl & 1), wf, acc[tl], li, lg);
        }
      }
    }
#pragma unroll
    for (int tl = 0; tl < T; ++tl) {
      const int64_t r0 = (t0 + tl) * NR3;
      float4 rr[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) rr[it] = ld4(R + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
      __syncthreads();   // MFMAs (first pass) / the previous tile's row reads are done with the image memory
#pragma unroll
      for (int e = 0; e < 16; ++e) xs[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + wave * 32 + li] = acc[tl][e];
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = lrow + 8 * it;
        const int64_t grow = r0 + row;
        float4 v = ld4(xs + row * FP + lc4);
        if constexpr (MM::kScaled) v = v * (ci4 * h2_ref_inv(eref[tl * NR3 + row]));
        if (grow < M) st4(Y + grow * FH + lc4, v + rr[it]);
      }
    }
    __syncthreads();   // the output image overlays both tile buffers: done before the next group stages
  }
}

// ------------------------------------------------------------------------------------------
// rowtile_nn2_k (round 5): rowtile_nn_group32_b3_k with the BatchNorm_h backward sums of the layer BELOW in its epilogue.
// The epilogue holds the rows of gh_in = the gh_out of the layer below, whose BatchNorm_h backward begins with
// (sum gw, sum gw zhat), gw = gh_out [bn_h(z) > 0] (node_bwd_stats_k: z and gh_out read once more, a launch of its own):
// the sums are taken here from the rows on chip and the z rows, in fp64, parked per thread in LDS between tiles (no
// registers live across the matrix phase: the kernel sits at 256 with the weight fragments, sixteen accumulators per tile
// and four tiles of prefetched rows).
// (The conversion of the raw by-source / by-destination sums into gB1h / gB2h -- node_bgrad_k -- was built into this
//  kernel's operand load first: two rows of loads per row of operand, +36 live registers, 250-580 spilled in every arrangement
//  tried.  It went to the weight-gradient kernel instead, which has the registers: tn_tr_k<., CONV>.)
// ------------------------------------------------------------------------------------------
struct Nn2Args {
  int64_t M; const float* X; int64_t ldx; int ncg; const void* Wp; const float* R; float* Y; int64_t groups_per_block;
  const float* z_lo; const float* stat_lo; double* partials;
};

template <class MM, int T>
__global__ __launch_bounds__(kBlock, 2) void rowtile_nn2_k(const Nn2Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];
  __shared__ double sacc[8 * 2 * FH];      // [row slot 0..7][sum a | sum b][128 columns], one owner thread per entry
  __shared__ int eref[MM::kScaled ? T * NR3 : 1];                // MmH2: as in rowtile_nn_group32_b3_k
  __shared__ __attribute__((aligned(16))) float fsc[MM::kScaled ? 2 * NR3 : 4];
  float* xs = reinterpret_cast<float*>(xraw);
  static_assert(T % 2 == 0, "two tile buffers");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t M = a.M;
  const int ncg = a.ncg;
  const int64_t ngroups = (M + NR3 * T - 1) / (NR3 * T);
  const int64_t g0 = (int64_t)chunk * a.groups_per_block;
  const int64_t g1 = min(ngroups, g0 + a.groups_per_block);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  float4 ci4 = f4(1.f);
  if constexpr (MM::kScaled) ci4 = MM::col_inv4(a.Wp, 0, lc4);
  const float* __restrict__ X = a.X;
  const int64_t ldx = a.ldx;
#pragma unroll
  for (int j = 0; j < 4; ++j) { sacc[(lrow * 2 + 0) * FH + lc4 + j] = 0.0; sacc[(lrow * 2 + 1) * FH + lc4 + j] = 0.0; }
  float4 pre[T][4];
  auto prefetch = [&](float4 (&buf)[4], int64_t g, int cg, int tl) __attribute__((always_inline)) {
    if (cg >= ncg) { cg = 0; g = g + 1 < g1 ? g + 1 : g; }
    const int64_t r0 = (g * T + tl) * NR3;
#pragma unroll
    for (int it = 0; it < 4; ++it) buf[it] = ld4_nt(X + clampi(r0 + lrow + 8 * it, Mlast) * ldx + cg * FH + lc4);
  };
  if (g0 < g1) {
#pragma unroll
    for (int tl = 0; tl < T; ++tl) prefetch(pre[tl], g0, 0, tl);
  }
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t t0 = g * T;
    floatx16 acc[T];
#pragma unroll
    for (int tl = 0; tl < T; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tl][e] = 0.f;
    for (int cg = 0; cg < ncg; ++cg) {
      typename MM::Frag wf;
      MM::load_w(wf, a.Wp, cg * 4 + wave, lane);
#pragma unroll
      for (int tl = 0; tl < T; ++tl) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = lrow + 8 * it;
          if constexpr (MM::kScaled) h2_stage_nn(xraw, 32 * (tl & 1) + row, lc4, pre[tl][it], eref + tl * NR3 + row, fsc + (tl & 1) * NR3 + row,
                                                 cg == 0, (tid & 31) == 0);
          else MM::stage(xraw, 32 * (tl & 1) + row, lc4, pre[tl][it]);
        }
        __syncthreads();
        prefetch(pre[tl], g, cg + 1, tl);
        if constexpr (MM::kScaled) {
          if (cg > 0) acc_rescale_rows(acc[tl], fsc + (tl & 1) * NR3, lg);
          mma32_h2(xraw, 32 * (tl & 1), wf, acc[tl], li, lg);
        } else {
          mma32_b3(xraw, 32 * (tl & 1), wf, acc[tl], li, lg);
        }
      }
    }
#pragma unroll
    for (int tl = 0; tl < T; ++tl) {
      const int64_t r0 = (t0 + tl) * NR3;
      float4 rr[4], zz[4];
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        rr[it] = ld4(a.R + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
        zz[it] = ld4(a.z_lo + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
      }
      __syncthreads();   // MFMAs (first pass) / the previous tile's row reads are done with the image memory
#pragma unroll
      for (int e = 0; e < 16; ++e) xs[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + wave * 32 + li] = acc[tl][e];
      __syncthreads();
      Stat4 st;
      st.zero();
      const float4 mu = ld4(a.stat_lo + lc4), rs = ld4(a.stat_lo + FH + lc4);
      const float4 sc = ld4(a.stat_lo + 2 * FH + lc4), sh = ld4(a.stat_lo + 3 * FH + lc4);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = lrow + 8 * it;
        const int64_t grow = r0 + row;
        float4 v = ld4(xs + row * FP + lc4);
        if constexpr (MM::kScaled) v = v * (ci4 * h2_ref_inv(eref[tl * NR3 + row]));
        v = v + rr[it];
        if (grow < M) {
          st4(a.Y + grow * FH + lc4, v);
          const float4 gw = gate4(fma4(zz[it], sc, sh), v);      // node_bwd_stats_k's expressions
          st.add_prod(gw, (zz[it] - mu) * rs);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        sacc[(lrow * 2 + 0) * FH + lc4 + j] += st.a[j];
        sacc[(lrow * 2 + 1) * FH + lc4 + j] += st.b[j];
      }
    }
    __syncthreads();   // the output image overlays both tile buffers: done before the next group stages
  }
  __syncthreads();
  // partials[chunk][2][128], as block_stat_store lays them out; the 8 row slots added in a fixed order
  for (int idx = tid; idx < 2 * FH; idx += kBlock) {
    double s_ = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s_ += sacc[(k * 2 + idx / FH) * FH + idx % FH];
    a.partials[(size_t)chunk * 2 * FH + idx] = s_;
  }
}

// ------------------------------------------------------------------------------------------
// General row GEMM in split mode (the hidden sizes the fused 128-wide kernels are not built for, e.g. the reference's
// own default 256):  Y[r][cls*128 + c] = sum_cg X[r][cg*128 + k] Wblk[cls][cg][k][c]  (+ bias, + R, relu)
// with K = ncg * 128 and N = ncls * 128.  The kernel above with an output-column class per workgroup: the ncls
// workgroups of a row chunk sit on one XCD (workgroup b runs on XCD b % 8) and read the same X rows through its L2.
// R may be Y (ge = ge + gt W3 in place): a thread reads its R elements before it stores the same Y elements.
// ------------------------------------------------------------------------------------------
template <class MM, int T>
__global__ __launch_bounds__(kBlock, 2) void gemm_rows_b3_k(
    int64_t M, const float* __restrict__ X, int64_t ldx, int ncg, int ncls, const void* __restrict__ Wp,
    const float* __restrict__ bias, const float* R, int64_t ldr, int relu, float* Y, int64_t ldy, int nchunk,
    int64_t groups_per_chunk) {
  __shared__ __attribute__((aligned(16))) unsigned char xraw[MM::kImgBytes];
  __shared__ int eref[MM::kScaled ? T * NR3 : 1];                // MmH2: as in rowtile_nn_group32_b3_k
  __shared__ __attribute__((aligned(16))) float fsc[MM::kScaled ? 2 * NR3 : 4];
  float* xs = reinterpret_cast<float*>(xraw);
  static_assert(T % 2 == 0, "two-deep prefetch assumes an even group size");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int xcd = blockIdx.x % kXcds, jj = blockIdx.x / kXcds;
  const int cls = jj % ncls, chunk = xcd * (nchunk / kXcds) + jj / ncls;
  const int64_t ngroups = (M + NR3 * T - 1) / (NR3 * T);
  const int64_t g0 = (int64_t)chunk * groups_per_chunk;
  const int64_t g1 = min(ngroups, g0 + groups_per_chunk);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  float4 ci4 = f4(1.f);
  if constexpr (MM::kScaled) ci4 = MM::col_inv4(Wp, cls * ncg * 4, lc4);    // the class's column factors span its whole K
  float4 pre[T][4];
  auto prefetch = [&](float4 (&buf)[4], int64_t g, int cg, int tl) __attribute__((always_inline)) {
    if (cg >= ncg) { cg = 0; g = g + 1 < g1 ? g + 1 : g; }
    const int64_t r0 = (g * T + tl) * NR3;
#pragma unroll
    for (int it = 0; it < 4; ++it) buf[it] = ld4_nt(X + clampi(r0 + lrow + 8 * it, Mlast) * ldx + cg * FH + lc4);
  };
  if (g0 < g1) {
#pragma unroll
    for (int tl = 0; tl < T; ++tl) prefetch(pre[tl], g0, 0, tl);
  }
  for (int64_t g = g0; g < g1; ++g) {
    const int64_t t0 = g * T;
    floatx16 acc[T];
#pragma unroll
    for (int tl = 0; tl < T; ++tl)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tl][e] = 0.f;
    for (int cg = 0; cg < ncg; ++cg) {
      typename MM::Frag wf;
      MM::load_w(wf, Wp, (cls * ncg + cg) * 4 + wave, lane);
#pragma unroll
      for (int tl = 0; tl < T; ++tl) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int row = lrow + 8 * it;
          if constexpr (MM::kScaled) h2_stage_nn(xraw, 32 * (tl & 1) + row, lc4, pre[tl][it], eref + tl * NR3 + row, fsc + (tl & 1) * NR3 + row,
                                                 cg == 0, (tid & 31) == 0);
          else MM::stage(xraw, 32 * (tl & 1) + row, lc4, pre[tl][it]);
        }
        __syncthreads();
        prefetch(pre[tl], g, cg + 1, tl);
        if constexpr (MM::kScaled) {
          if (cg > 0) acc_rescale_rows(acc[tl], fsc + (tl & 1) * NR3, lg);
          mma32_h2(xraw, 32 * (tl & 1), wf, acc[tl], li, lg);
        } else {
          mma32_b3(xraw, 32 * (tl & 1), wf, acc[tl], li, lg);
        }
      }
    }
#pragma unroll
    for (int tl = 0; tl < T; ++tl) {
      const int64_t r0 = (t0 + tl) * NR3;
      const float4 bv = bias ? ld4(bias + cls * FH + lc4) : f4(0.f);
      float4 rr[4];
#pragma unroll
      for (int it = 0; it < 4; ++it)
        rr[it] = R ? ld4(R + clampi(r0 + lrow + 8 * it, Mlast) * ldr + cls * FH + lc4) + bv : bv;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 16; ++e) xs[((e & 3) + 8 * (e >> 2) + 4 * lg) * FP + wave * 32 + li] = acc[tl][e];
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = lrow + 8 * it;
        const int64_t grow = r0 + row;
        float4 v = ld4(xs + row * FP + lc4);
        if constexpr (MM::kScaled) v = v * (ci4 * h2_ref_inv(eref[tl * NR3 + row]));
        v = v + rr[it];
        if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if (grow < M) st4(Y + grow * ldy + cls * FH + lc4, v);
      }
    }
    __syncthreads();
  }
}

// fragment blocks of the [K = ncg*128] x [N = ncls*128] weight for gemm_rows_b3_k: block ((cls*ncg + cg)*4 + wv) in the
// layout of pack_w3_k;  NT: W is [N,K] row-major (y = x W^T),  NN: W is [K,N] row-major (y = x W)
__global__ void pack_w3_gen_k(const float* __restrict__ W, int64_t ld, int ncls, int ncg, int nn, bf16x8* __restrict__ Wp) {
  const int total = ncls * ncg * 4 * BKC * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, c = (idx >> 6) % BKC, blk = idx / (64 * BKC);
    const int wv = blk & 3, cg = (blk >> 2) % ncg, cls = (blk >> 2) / ncg;
    const int i = lane & 31, g = lane >> 5;
    const int64_t n = (int64_t)cls * FH + wv * 32 + i;
    bf16x8 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t k = (int64_t)cg * FH + 16 * c + 8 * g + j;
      const float x = nn ? W[k * ld + n] : W[n * ld + k];
      const __bf16 h = (__bf16)x;
      const float r1 = x - (float)h;
      const __bf16 m = (__bf16)r1;
      hi[j] = h;
      mid[j] = m;
      lo[j] = (__bf16)(r1 - (float)m);
    }
    bf16x8* o = Wp + ((int64_t)(blk * BKC + c) * 3) * 64 + lane;
    o[0] = hi;
    o[64] = mid;
    o[128] = lo;
  }
}

// the same for MmH2 (block layout of pack_w2_k); a column's factor is taken over its WHOLE contraction (all ncg groups), so that
// an accumulator that runs over the groups keeps one unit per column.  One workgroup per fragment block (32 output columns x one
// 128-deep group): it first takes the largest magnitude of its 32 columns over the whole K = ncg x 128 (32 columns x 8 contraction
// slices, joined through LDS -- the weight is L2-resident, the ncg-fold re-read is a few hundred KB), then packs.  (Until the end of
// round 5 the maxima came from a launch of their own, col_amax_k: 26 more launches per step on the critical path of the kernel behind.)
__global__ __launch_bounds__(256) void pack_w2_gen_k(const float* __restrict__ W, int64_t ld, int ncg, int nn, unsigned char* __restrict__ Wp) {
  __shared__ float part[8][32];
  const int blk = blockIdx.x;
  const int wv = blk & 3, cg = (blk >> 2) % ncg, cls = (blk >> 2) / ncg;
  {
    const int c = threadIdx.x & 31, ks = threadIdx.x >> 5;
    const int64_t n = (int64_t)cls * FH + wv * 32 + c, K = (int64_t)ncg * FH;
    float m = 0.f;
    if (nn) {
      for (int64_t k = ks; k < K; k += 8) m = fmaxf(m, fabsf(W[k * ld + n]));
    } else {
      const float* row = W + n * ld;
      for (int64_t k = ks; k < K; k += 8) m = fmaxf(m, fabsf(row[k]));
    }
    part[ks][c] = m;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < BKC * 64; idx += 256) {
    const int lane = idx & 63, c = idx >> 6;
    const int i = lane & 31, g = lane >> 5;
    const int64_t n = (int64_t)cls * FH + wv * 32 + i;
    float m = part[0][i];
#pragma unroll
    for (int q = 1; q < 8; ++q) m = fmaxf(m, part[q][i]);
    float sc, inv;
    h2_scale(__float_as_uint(m), sc, inv);
    h16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t k = (int64_t)cg * FH + 16 * c + 8 * g + j;
      const float x = (nn ? W[k * ld + n] : W[n * ld + k]) * sc;
      const _Float16 h = (_Float16)x;
      hi[j] = h;
      lo[j] = (_Float16)(x - (float)h);
    }
    unsigned char* b = Wp + (size_t)blk * MmH2::kPackBytes;
    h16x8* o = reinterpret_cast<h16x8*>(b) + (c * 2) * 64 + lane;
    o[0] = hi;
    o[64] = lo;
    if (c == 0 && g == 0) reinterpret_cast<float*>(b + MmH2::kFragBytes)[i] = inv;
  }
}

static void launch_pack_w2_gen(const float* W, int64_t ld, int ncls, int ncg, int nn, void* ws, hipStream_t st) {
  hipLaunchKernelGGL(pack_w2_gen_k, dim3(4 * ncls * ncg), dim3(256), 0, st, W, ld, ncg, nn, (unsigned char*)ws);
}

// slab[(cg*nslot + slot)][n][c] = sum over the slot's rows of A[row][cg*128+n] * B[row][c];
// partials[(cg*nslot + slot)][128] = column sums of A[:, cg*128 ..]
// (fp32-MFMA mode; the split mode runs tn_tr_k, gnm_tr.hip)
template <class MM>
__global__ __launch_bounds__(kBlock, 2) void tn_colgroup_k(
    int64_t M, const float* __restrict__ A, int64_t lda, int ncg, const float* __restrict__ B,
    float* __restrict__ slab, double* __restrict__ partials, int nslot, int64_t tiles_per_slot) {
  static_assert(!MM::kSplit, "the split mode has its own weight-gradient kernel (tn_tr_k)");
  __shared__ __attribute__((aligned(16))) unsigned char raw[MM::kTnBytes];
  float* as = reinterpret_cast<float*>(raw);             // two row images
  float* bs = as + FTR * FP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  // The ncg workgroups of a slot read the same B rows: keep them on one XCD (workgroup b runs on XCD
  // b % 8) so that the tile comes out of that XCD's L2 instead of HBM ncg times.  nslot % 8 == 0.
  const int xcd = blockIdx.x % kXcds, j = blockIdx.x / kXcds;
  const int cg = j % ncg, slot = xcd * (nslot / kXcds) + j / ncg;
  const int64_t ntiles = (M + FTR - 1) / FTR;
  const int64_t tb0 = (int64_t)slot * tiles_per_slot;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_slot);
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  floatx16 tn[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) tn[a][b][e] = 0.f;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
  // loads only (no stores in the loop): clamped and branch-free, rows past the end are zeroed below.
  constexpr int DEPTH = 1;
  float4 pa[DEPTH][8], pb[DEPTH][8];
  auto prefetch = [&](float4 (&qa)[8], float4 (&qb)[8], int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = (tile < tb1 ? tile : tb1 - 1) * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t r = clampi(r0 + MM::row(lrow, it), Mlast);
      qa[it] = ld4(A + r * lda + cg * FH + lc4);
      qb[it] = ld4(B + r * FH + lc4);
    }
  };
  auto step = [&](float4 (&qa)[8], float4 (&qb)[8], int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const bool ok = r0 + MM::row(lrow, it) < M;
      if (!ok) qa[it] = f4(0.f);
      const float4 av = qa[it];
      c0 += (double)av.x; c1 += (double)av.y; c2 += (double)av.z; c3 += (double)av.w;
      st4(as + MM::row(lrow, it) * FP + lc4, av);
      st4(bs + MM::row(lrow, it) * FP + lc4, qb[it]);
    }
    __syncthreads();
    prefetch(qa, qb, tile + DEPTH);
    mma_tn64(as, bs, tn, wn, wc, li, lg);
  };
  if (tb0 < tb1) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) prefetch(pa[d], pb[d], tb0 + d);
    int64_t tile = tb0;
    for (; tile + DEPTH <= tb1; tile += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) step(pa[d], pb[d], tile + d);
    }
    if (DEPTH == 2 && tile < tb1) step(pa[0], pb[0], tile);
  }
  float* sl = slab + (size_t)(cg * nslot + slot) * FH * FH;
  tn_store_slab<MM>(sl, tn, wn, wc, li, lg);
  __syncthreads();
  double* red = reinterpret_cast<double*>(raw);
  red[lrow * FH + lc4 + 0] = c0;
  red[lrow * FH + lc4 + 1] = c1;
  red[lrow * FH + lc4 + 2] = c2;
  red[lrow * FH + lc4 + 3] = c3;
  __syncthreads();
  if (tid < FH) {
    double s_ = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s_ += red[k * FH + tid];
    partials[(size_t)(cg * nslot + slot) * FH + tid] = s_;
  }
}

// out[i] = sum_b slab[b][i], fixed order -> deterministic (gnm_common.h slab_reduce_128); grid (total / 128, batch) x 256 threads
__global__ __launch_bounds__(256) void slab_reduce_k(const float* __restrict__ slab, int nslab, int total, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float red[8 * 128];
  slab += (size_t)blockIdx.y * nslab * total;       // blockIdx.y = batch: its own nslab slabs -> its own `total` outputs
  out += (size_t)blockIdx.y * total;
  const int i0 = blockIdx.x * 128;
  const float4 s_ = slab_reduce_128(slab, nslab, total, i0, red);
  if (threadIdx.x < 32) st4(out + i0 + threadIdx.x * 4, s_);
}

// out[m * ldc + n] = sum_b slab[b][m][n] for the 128 x 128 block blockIdx.y = cls of a [ncga x ncgb] grid of blocks (a workgroup
// = one row m of one block; class cls owns slabs cls * nslab .. and the block (cls / ncgb, cls % ncgb) of C)
__global__ __launch_bounds__(256) void slab_reduce_ld_k(const float* __restrict__ slab, int nslab, int ncgb, float* __restrict__ out, int64_t ldc) {
  __shared__ __attribute__((aligned(16))) float red[8 * 128];
  const int m = blockIdx.x, cls = blockIdx.y;
  slab += (size_t)cls * nslab * FH * FH;
  out += (int64_t)(cls / ncgb) * FH * ldc + (cls % ncgb) * FH;
  const float4 s_ = slab_reduce_128(slab, nslab, FH * FH, m * FH, red);
  if (threadIdx.x < 32) st4(out + (int64_t)m * ldc + threadIdx.x * 4, s_);
}

}  // namespace gnm

using namespace gnm;

static inline int64_t cdiv_(int64_t a, int64_t b) { return (a + b - 1) / b; }

namespace gnm {   // gnm_tr.hip: the split-mode TN kernel on swizzled row-major images + transpose reads
int tn_tr_rows_per_tile();
int tn_tr_occupancy(bool conv = false, bool h2 = false);
void tn_tr_launch(int64_t M, const float* A, int64_t lda, int ncg, const void* B, int64_t ldb, int ncgb, float* slab,
                  double* partials, int nslot, int64_t tiles_per_slot, hipStream_t st, const TnConv* cv = nullptr, bool h2 = false);
size_t edge_bwd_tr_pack_bytes();
int edge_bwd_tr_launch(int64_t E, const float* ge, float* ge_out, const float* t, const float* e_in, const float* stat_e,
                       const float* bstat_e, const float* gamma_e, const float* W3, void* wpack, float* slab,
                       double* partials, hipStream_t st, bool h2 = false, bool given = false);
}
// kernel-generation switches for same-process A/B runs (gnm_debug_set_variant, GNM_VARIANTS).  Round 5 removed the generations
// that had lost their A/B (the round-1 split-mode edge backward / weight gradient / VALU encoders, the unpipelined t kernels,
// the role-split chained kernel): their numbers are in profiles/r02_ab_kernels.txt, r03_chain_phases.txt, r04_ab_*.txt.
static int g_eb_variant = 1;     // edge_bwd_tr_k: 1 = LDS stash + pinned prefetch (default), 2 = registers, unpinned (within 1 %)
// round 6: workgroups per CU of the two-sided forward sweep (2 = default).  1 = the occupancy a sweep that ALSO carried the B_3 product
// (W3 stationary: +32 registers per thread, over the 128 of four waves per SIMD) would be confined to: the measured price of
// "t never materialised" in the forward sweep (profiles/r06_ab_gate2_occupancy.txt).  The caller passes the plan built for that partition.
static int g_gate2_wg = 2;
namespace gnm {
int eb_variant() { return g_eb_variant; }
int gate2_wg() { return g_gate2_wg; }
}
extern "C" int gnm_debug_set_variant(const char* what, int v) {
  if (what && !strcmp(what, "edge_bwd")) { g_eb_variant = v; return 0; }
  if (what && !strcmp(what, "gate2_wg") && (v == 1 || v == 2)) { g_gate2_wg = v; return 0; }
  ::gnm::set_error("debug_set_variant: unknown switch");
  return -1;
}


// workspace: packed weights (ncb * 16 * 64 float4)
extern "C" size_t gnm_rowtile_workspace_bytes(int ncols) { return (size_t)(ncols / 32) * kPackBytesPerBlk; }

extern "C" int gnm_set_matmul_mode(int mode) {
  GNM_CHECK_ARG(mode >= 0 && mode <= 2, "set_matmul_mode: mode %d (0 = fp32 MFMA, 1 = bf16x3 split, 2 = f16x2 split)", mode);
  g_matmul_mode = mode;
  return 0;
}
extern "C" int gnm_get_matmul_mode(void) { return g_matmul_mode; }

template <class MM>
static int edge_t_fused_impl(int64_t E, const float* e_in, const float* W3, const float* b3, const float* P,
                             const int32_t* isrc, const int32_t* idst, float* t, double* partials, int* nblk_out,
                             void* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  launch_pack<MM>(W3, FH, FH / 32, 0, ws, st);
  GNM_LAUNCH_CHECK("pack_w (NT)");
  if constexpr (MM::kSplit) {
    const int64_t ntiles = cdiv_(E, ER3);
    const int grid = persistent_grid(ntiles, 8, occ_blocks<edge_t32_b3p_k<MM>>());
    hipLaunchKernelGGL(edge_t32_b3p_k<MM>, dim3(grid), dim3(kBlock), 0, st, E, e_in, (const void*)ws, b3, t, P, isrc, idst,
                       partials, cdiv_(ntiles, grid));
    GNM_LAUNCH_CHECK("edge_t_fused_fwd");
    *nblk_out = grid;
    return 0;
  }
  const int64_t ntiles = cdiv_(E, FTR);
  const int grid = persistent_grid(ntiles, 4, occ_blocks<rowtile_nt_k<MM, true, 1>>());
  hipLaunchKernelGGL((rowtile_nt_k<MM, true, 1>), dim3(grid), dim3(kBlock), 0, st, E, e_in, (const void*)ws, b3, t,
                     (int64_t)FH, P, isrc, idst, partials, cdiv_(ntiles, grid), 1);
  GNM_LAUNCH_CHECK("edge_t_fused_fwd");
  *nblk_out = grid;
  return 0;
}

extern "C" int gnm_edge_t_fused_fwd(int64_t E, int H, const float* e_in, const float* W3, const float* b3,
                                    const float* P, const int32_t* isrc, const int32_t* idst, float* t,
                                    double* partials, int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH || (H == WH && g_matmul_mode), "edge_t_fused_fwd: H=%d (128, and 256 in the split matmul modes, are built)", H);
  GNM_CHECK_ARG(E > 0 && e_in && W3 && b3 && P && isrc && idst && t && partials && nblk_out, "edge_t_fused_fwd: null/neg argument");
  if (H == WH) {
    GNM_CHECK_ARG(ws && ws_bytes >= (size_t)16 * MmB3::kPackBytes, "edge_t_fused_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool h2 = g_matmul_mode == 2;
    if (h2) launch_pack_w2_gen(W3, (int64_t)WH, 2, 2, 0, ws, st);
    else hipLaunchKernelGGL(pack_w3_gen_k, dim3(64), dim3(256), 0, st, W3, (int64_t)WH, 2, 2, 0, (bf16x8*)ws);
    GNM_LAUNCH_CHECK("pack_w3_gen (NT 256)");
    const int64_t ntiles = cdiv_(E, ER3);
    int nchunk = num_cus() / 2 / kXcds * kXcds;            // one 8-wave workgroup per CU, two classes per chunk
    if (nchunk < kXcds) nchunk = kXcds;
    if (nchunk > kMaxPartialBlocks) nchunk = kMaxPartialBlocks / kXcds * kXcds;
    if (h2) hipLaunchKernelGGL(edge_t32_h256p_k<MmH2>, dim3(nchunk * 2), dim3(kBlockW), 0, st, E, e_in, (const void*)ws, b3, t, P, isrc, idst,
                               partials, nchunk, cdiv_(ntiles, nchunk));
    else hipLaunchKernelGGL(edge_t32_h256p_k<MmB3>, dim3(nchunk * 2), dim3(kBlockW), 0, st, E, e_in, (const void*)ws, b3, t, P, isrc, idst,
                            partials, nchunk, cdiv_(ntiles, nchunk));
    GNM_LAUNCH_CHECK("edge_t_fused_fwd (256)");
    *nblk_out = nchunk;
    return 0;
  }
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_rowtile_workspace_bytes(FH), "edge_t_fused_fwd: workspace too small");
  return g_matmul_mode == 2 ? edge_t_fused_impl<MmH2>(E, e_in, W3, b3, P, isrc, idst, t, partials, nblk_out, ws, stream)
         : g_matmul_mode  ? edge_t_fused_impl<MmB3>(E, e_in, W3, b3, P, isrc, idst, t, partials, nblk_out, ws, stream)
                          : edge_t_fused_impl<MmF32>(E, e_in, W3, b3, P, isrc, idst, t, partials, nblk_out, ws, stream);
}

// edge_bwd_gt + ge_out = ge + gt W3 at H = 256 (bf16x3 matmul mode), see edge_gt_nn_h256_k; ge_out != ge
extern "C" int gnm_edge_bwd_gt_nn(int64_t E, int H, const float* ge, const float* t, const float* stat_e, const float* bstat_e,
                                  const float* gamma_e, const float* W3, float* gt, float* ge_out, void* ws, size_t ws_bytes,
                                  void* stream) {
  GNM_CHECK_ARG(H == WH && g_matmul_mode, "edge_bwd_gt_nn: H=%d (256 in the split matmul modes is what is built)", H);
  GNM_CHECK_ARG(E > 0 && ge && t && stat_e && bstat_e && gamma_e && W3 && gt && ge_out && ge_out != ge && gt != ge,
                "edge_bwd_gt_nn: null / aliased argument (ge_out and gt must not be ge)");
  GNM_CHECK_ARG(ws && ws_bytes >= (size_t)16 * MmB3::kPackBytes, "edge_bwd_gt_nn: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const bool h2 = g_matmul_mode == 2;
  if (h2) launch_pack_w2_gen(W3, (int64_t)WH, 2, 2, 1, ws, st);
  else hipLaunchKernelGGL(pack_w3_gen_k, dim3(64), dim3(256), 0, st, W3, (int64_t)WH, 2, 2, 1, (bf16x8*)ws);
  GNM_LAUNCH_CHECK("pack_w3_gen (NN 256)");
  const int64_t ntiles = cdiv_(E, ER3);
  int nchunk = num_cus() / 2 / kXcds * kXcds;
  if (nchunk < kXcds) nchunk = kXcds;
  if (h2) hipLaunchKernelGGL(edge_gt_nn_h256_k<MmH2>, dim3(nchunk * 2), dim3(kBlockW), 0, st, E, ge, t, stat_e, bstat_e, gamma_e, (const void*)ws,
                             gt, ge_out, nchunk, cdiv_(ntiles, nchunk));
  else hipLaunchKernelGGL(edge_gt_nn_h256_k<MmB3>, dim3(nchunk * 2), dim3(kBlockW), 0, st, E, ge, t, stat_e, bstat_e, gamma_e, (const void*)ws,
                          gt, ge_out, nchunk, cdiv_(ntiles, nchunk));
  GNM_LAUNCH_CHECK("edge_bwd_gt_nn");
  return 0;
}

template <class MM>
static int node_proj_fwd_impl(int64_t N, int ncols, const float* h, const float* W, const float* b, float* Pout,
                              void* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  launch_pack<MM>(W, FH, ncols / 32, 0, ws, st);
  GNM_LAUNCH_CHECK("pack_w (NT, node)");
  const int64_t ntiles = cdiv_(N, FTR);
  if constexpr (MM::kSplit) {
    // HBM-bound in split mode: column groups over workgroups, weights stationary (no reload per tile)
    const int ncg = ncols / FH;
    int nslot = (num_cus() * occ_blocks<rowtile_nt_k<MM, false, 1>>()) / ncg / kXcds * kXcds;
    if (nslot > (int)((ntiles + kXcds - 1) / kXcds * kXcds)) nslot = (int)((ntiles + kXcds - 1) / kXcds * kXcds);
    if (nslot < kXcds) nslot = kXcds;
    hipLaunchKernelGGL((rowtile_nt_k<MM, false, 1>), dim3(nslot * ncg), dim3(kBlock), 0, st, N, h, (const void*)ws, b,
                       Pout, (int64_t)ncols, (const float*)nullptr, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (double*)nullptr, cdiv_(ntiles, nslot), ncg);
  } else {
    const int grid = persistent_grid(ntiles, 2, occ_blocks<rowtile_nt_k<MM, false, 5>>());
    hipLaunchKernelGGL((rowtile_nt_k<MM, false, 5>), dim3(grid), dim3(kBlock), 0, st, N, h, (const void*)ws, b, Pout,
                       (int64_t)ncols, (const float*)nullptr, (const int32_t*)nullptr,
                       (const int32_t*)nullptr, (double*)nullptr, cdiv_(ntiles, grid), 1);
  }
  GNM_LAUNCH_CHECK("node_proj_fwd");
  return 0;
}

extern "C" int gnm_node_proj_fwd(int64_t N, int H, int ncols, const float* h, const float* W, const float* b,
                                 float* Pout, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "node_proj_fwd: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(N > 0 && ncols == 5 * FH && h && W && b && Pout, "node_proj_fwd: bad argument (ncols must be 5*128)");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_rowtile_workspace_bytes(ncols), "node_proj_fwd: workspace too small");
  return g_matmul_mode == 2 ? node_proj_fwd_impl<MmH2>(N, ncols, h, W, b, Pout, ws, stream)
         : g_matmul_mode  ? node_proj_fwd_impl<MmB3>(N, ncols, h, W, b, Pout, ws, stream)
                          : node_proj_fwd_impl<MmF32>(N, ncols, h, W, b, Pout, ws, stream);
}

namespace gnm {
int edge_bwd_chain_launch(const ChainArgs& in, const float* W3, void* wpack, hipStream_t st, bool h2 = false);   // gnm_tr.hip
}
extern "C" size_t gnm_edge_bwd_fused_workspace_bytes(void);

// The fused edge backward of layer i chained with the by-destination backward pass of layer i-1 (gnm.h).
static int edge_bwd_chain_impl(int64_t N, int64_t E, int H, const float* ge, float* ge_out, const float* t_hi,
                               const float* e_mid, const float* stat_hi, const float* bstat_hi, const float* gamma_hi,
                               const float* W3_hi, float* gW3_hi, float* gb3_hi, double* partials_hi,
                               const float* t_lo, const float* stat_lo, const float* P_lo, const float* Q_lo,
                               const float* hf_lo, const float* hb_lo, const int32_t* isrc, const int32_t* idst,
                               const int32_t* in_ptr, float* gP_lo, float* Ud_lo, float* Td_lo, double* partials_lo,
                               const uint32_t* sinfo, int64_t plan_nodes_per_block, float* UT_lo,
                               int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "edge_bwd_chain: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(g_matmul_mode >= 1, "edge_bwd_chain: only built for the split matmul modes");
  GNM_CHECK_ARG(N > 0 && E > 0 && ge && ge_out && t_hi && e_mid && stat_hi && bstat_hi && gamma_hi && W3_hi && gW3_hi &&
                    gb3_hi && partials_hi && t_lo && stat_lo && P_lo && Q_lo && hf_lo && hb_lo && isrc && idst && in_ptr &&
                    gP_lo && Ud_lo && Td_lo && partials_lo && nblk_out && partials_hi != partials_lo,
                "edge_bwd_chain: null / aliased argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_edge_bwd_fused_workspace_bytes(), "edge_bwd_chain: workspace %zu < %zu", ws_bytes,
                gnm_edge_bwd_fused_workspace_bytes());
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
  ChainArgs a{};
  a.E = E; a.N = N;
  a.ge = ge; a.ge_out = ge_out; a.t_hi = t_hi; a.e_mid = e_mid;
  a.stat_hi = stat_hi; a.bstat_hi = bstat_hi; a.gamma_hi = gamma_hi;
  a.slab = slab; a.partials = partials_hi;
  a.t_lo = t_lo; a.stat_lo = stat_lo; a.P_lo = P_lo; a.Q_lo = Q_lo; a.hf_lo = hf_lo; a.hb_lo = hb_lo;
  a.isrc = isrc; a.idst = idst; a.in_ptr = in_ptr;
  a.gP_lo = gP_lo; a.Ud_lo = Ud_lo; a.Td_lo = Td_lo; a.partials_lo = partials_lo;
  a.sinfo = sinfo; a.UT_lo = UT_lo; a.margin = kSweepMargin;
  a.ud_pitch = Td_lo == Ud_lo + FH ? 2 * FH : FH;       // [Ud | Td] as one [N,2H] array, or two [N,H] arrays
  int64_t npb = 0;
  gnm_sweep_partition(N, 1, &npb, nullptr);      // one 512-thread workgroup per CU
  // the walkers' / run sums' rows are addressed through 32-bit buffer offsets over the workgroup's node range
  GNM_CHECK_ARG((npb + 2 * kSweepMargin) * 5 * FH * 4 < (int64_t)INT32_MAX, "edge_bwd_chain: %lld nodes per workgroup exceed the 32-bit buffer offsets",
                (long long)npb);
  if (sinfo) {
    GNM_CHECK_ARG(UT_lo, "edge_bwd_chain_src: UT_lo is null");
    GNM_CHECK_ARG(plan_nodes_per_block == npb, "edge_bwd_chain_src: the sweep plan was built for %lld nodes per workgroup, the kernel uses %lld "
                  "(gnm_sweep_partition(N, 1))", (long long)plan_nodes_per_block, (long long)npb);
  }
  const int grid = edge_bwd_chain_launch(a, W3_hi, ws, st, g_matmul_mode == 2);
  GNM_LAUNCH_CHECK("edge_bwd_chain");
  hipLaunchKernelGGL(slab_reduce_k, dim3(FH * FH / 128), dim3(256), 0, st, (const float*)slab, grid, FH * FH, gW3_hi);
  GNM_LAUNCH_CHECK("edge_bwd_chain slab reduce");
  *nblk_out = grid;
  return gnm_reduce_partials(partials_hi, grid, 1, FH, gb3_hi, stream) ? -3 : 0;
}

// the top of the stack: gnm_edge_bwd_dst (+ the by-source sums through the sweep plan) on the chained kernel's sweep
extern "C" int gnm_edge_bwd_top(int64_t N, int64_t E, int H, float* ge, const float* e_out, const float* t,
                                const float* stat_e, const float* P, const float* Q, const float* hf, const float* hb,
                                const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, float* gP, float* Ud,
                                float* Td, double* partials, const uint32_t* sinfo, int64_t plan_nodes_per_block,
                                float* UT, int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH || (H == 2 * FH && sinfo), "edge_bwd_top: H=%d (128; 256 with a sweep plan, one sweep per 128-column half)", H);
  GNM_CHECK_ARG(N > 0 && E > 0 && ge && e_out && t && stat_e && P && Q && hf && hb && isrc && idst && in_ptr && gP && Ud &&
                    Td && partials && nblk_out && (!sinfo || UT), "edge_bwd_top: null argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_edge_bwd_fused_workspace_bytes(), "edge_bwd_top: workspace %zu < %zu", ws_bytes,
                gnm_edge_bwd_fused_workspace_bytes());
  int64_t npb = 0;
  gnm_sweep_partition(N, 1, &npb, nullptr);
  GNM_CHECK_ARG((npb + 2 * kSweepMargin) * 5 * H * 4 < (int64_t)INT32_MAX, "edge_bwd_top: %lld nodes per workgroup exceed the 32-bit buffer offsets",
                (long long)npb);
  GNM_CHECK_ARG(!sinfo || plan_nodes_per_block == npb, "edge_bwd_top: the sweep plan was built for %lld nodes per workgroup, the kernel uses %lld",
                (long long)plan_nodes_per_block, (long long)npb);
  for (int c0 = 0; c0 < H; c0 += FH) {                             // a 256-wide layer = two 128-column problems with row pitch 256
    ChainArgs a{};
    a.E = E; a.N = N; a.hfull = H;
    a.ge = ge + c0; a.ge_out = ge + c0; a.e_mid = e_out + c0;      // t_hi == NULL selects the sweep without a layer above
    a.slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
    a.t_lo = t + c0; a.stat_lo = stat_e + c0; a.P_lo = P + c0; a.Q_lo = Q + c0; a.hf_lo = hf + c0; a.hb_lo = hb + c0;
    a.isrc = isrc; a.idst = idst; a.in_ptr = in_ptr;
    a.gP_lo = gP + c0; a.Ud_lo = Ud + c0; a.Td_lo = Td + c0; a.partials_lo = partials + c0;
    a.sinfo = sinfo; a.UT_lo = UT ? UT + c0 : nullptr; a.margin = kSweepMargin;
    a.ud_pitch = Td == Ud + H ? 2 * H : H;
    const int g = edge_bwd_chain_launch(a, nullptr, nullptr, (hipStream_t)stream);
    GNM_CHECK_ARG(g > 0, "edge_bwd_top: no kernel for this configuration");
    *nblk_out = g;
    GNM_LAUNCH_CHECK("edge_bwd_top");
  }
  return 0;
}

// The LayerNorm form of gnm_edge_bwd_top (round 6; H = 128, with a sweep plan): gnm_ln_edge_bwd_dst + gnm_ln_edge_bwd_src as ONE two-sided sweep.
// ge <- ge + gsigma sigma' in place; gt [E,H] = LNbwd(gu) written for gnm_edge_bwd_fused_gt; gP[:, H:5H] = gA2h | gA3h | gB1h | gB2h for the
// nodes the plan serves (the rest: gnm_ln_edge_bwd_src_fix); partials: (sum gu, sum gu that) = the LayerNorm weight / bias gradient sums.
// Q is gnm_ln_node_bwd's [N,4H] = Qf | Rf | Qb | Rb.                                   autograd of gated_gcn_full.py:120-143 under nn.LayerNorm
extern "C" int gnm_ln_edge_bwd_top(int64_t N, int64_t E, int H, float* ge, const float* e_out, const float* t, const float* gamma_e,
                                   const float* beta_e, int width, const float* P, const float* Q, const float* hf, const float* hb,
                                   const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, float* gP, float* gt,
                                   double* partials, const uint32_t* sinfo, int64_t plan_nodes_per_block, int* nblk_out, void* ws,
                                   size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "ln_edge_bwd_top: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(width >= 1 && width <= H, "ln_edge_bwd_top: width must be in [1, H]");
  GNM_CHECK_ARG(N > 0 && E > 0 && ge && e_out && t && gamma_e && beta_e && P && Q && hf && hb && isrc && idst && in_ptr && gP && gt &&
                    partials && sinfo && nblk_out, "ln_edge_bwd_top: null argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_edge_bwd_fused_workspace_bytes(), "ln_edge_bwd_top: workspace %zu < %zu", ws_bytes,
                gnm_edge_bwd_fused_workspace_bytes());
  int64_t npb = 0;
  gnm_sweep_partition(N, 1, &npb, nullptr);
  GNM_CHECK_ARG((npb + 2 * kSweepMargin) * 5 * H * 4 < (int64_t)INT32_MAX, "ln_edge_bwd_top: %lld nodes per workgroup exceed the 32-bit buffer offsets",
                (long long)npb);
  GNM_CHECK_ARG(plan_nodes_per_block == npb, "ln_edge_bwd_top: the sweep plan was built for %lld nodes per workgroup, the kernel uses %lld",
                (long long)plan_nodes_per_block, (long long)npb);
  ChainArgs a{};
  a.E = E; a.N = N; a.hfull = H;
  a.ge = ge; a.ge_out = ge; a.e_mid = e_out;
  a.slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
  a.t_lo = t; a.P_lo = P; a.Q_lo = Q; a.q_pitch = 4 * H; a.qb_off = 2 * H; a.hf_lo = hf; a.hb_lo = hb;
  a.isrc = isrc; a.idst = idst; a.in_ptr = in_ptr;
  a.gP_lo = gP; a.Ud_lo = gP + 4 * H; a.Td_lo = gP + 4 * H; a.ud_pitch = 5 * H;     // sum_dst gt = gB2h, straight into its column group
  a.UT_lo = gP + 3 * H; a.ut_pitch = 5 * H;                                            // sum_src gt = gB1h
  a.partials_lo = partials;
  a.sinfo = sinfo; a.margin = kSweepMargin;
  a.ln_gamma = gamma_e; a.ln_beta = beta_e; a.ln_width = width; a.gt_out = gt;
  const int g = edge_bwd_chain_launch(a, nullptr, nullptr, (hipStream_t)stream);
  GNM_CHECK_ARG(g > 0, "ln_edge_bwd_top: no kernel for this configuration");
  *nblk_out = g;
  GNM_LAUNCH_CHECK("ln_edge_bwd_top");
  return 0;
}

// The CHAINED LayerNorm backward (round 6; H = 128, split matmul modes, with a sweep plan): gnm_edge_bwd_fused_gt of layer i (its gt GIVEN:
// gt_hi, written by layer i's own sweep a launch earlier) and gnm_ln_edge_bwd_top of layer i-1 in one sweep -- read ge'(i), gt(i),
// e_mid = e_in(i) = e_out(i-1), t(i-1); write ge'(i-1) (in place) and gt(i-1): 6 [E,H] streams instead of 4 + 5.
extern "C" int gnm_ln_edge_bwd_chain(int64_t N, int64_t E, int H, float* ge, const float* gt_hi, const float* e_mid, const float* W3_hi,
                                     float* gW3_hi, float* gb3_hi, double* partials_hi, const float* t_lo, const float* gamma_lo,
                                     const float* beta_lo, int width, const float* P_lo, const float* Q_lo, const float* hf_lo,
                                     const float* hb_lo, const int32_t* isrc, const int32_t* idst, const int32_t* in_ptr, float* gP_lo,
                                     float* gt_lo, double* partials_lo, const uint32_t* sinfo, int64_t plan_nodes_per_block,
                                     int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "ln_edge_bwd_chain: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(g_matmul_mode >= 1, "ln_edge_bwd_chain: only built for the split matmul modes");
  GNM_CHECK_ARG(width >= 1 && width <= H, "ln_edge_bwd_chain: width must be in [1, H]");
  GNM_CHECK_ARG(N > 0 && E > 0 && ge && gt_hi && e_mid && W3_hi && gW3_hi && gb3_hi && partials_hi && t_lo && gamma_lo && beta_lo && P_lo &&
                    Q_lo && hf_lo && hb_lo && isrc && idst && in_ptr && gP_lo && gt_lo && partials_lo && sinfo && nblk_out &&
                    partials_hi != partials_lo && gt_hi != gt_lo, "ln_edge_bwd_chain: null / aliased argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_edge_bwd_fused_workspace_bytes(), "ln_edge_bwd_chain: workspace %zu < %zu", ws_bytes,
                gnm_edge_bwd_fused_workspace_bytes());
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
  int64_t npb = 0;
  gnm_sweep_partition(N, 1, &npb, nullptr);
  GNM_CHECK_ARG((npb + 2 * kSweepMargin) * 5 * FH * 4 < (int64_t)INT32_MAX, "ln_edge_bwd_chain: %lld nodes per workgroup exceed the 32-bit buffer offsets",
                (long long)npb);
  GNM_CHECK_ARG(plan_nodes_per_block == npb, "ln_edge_bwd_chain: the sweep plan was built for %lld nodes per workgroup, the kernel uses %lld",
                (long long)plan_nodes_per_block, (long long)npb);
  ChainArgs a{};
  a.E = E; a.N = N; a.hfull = H;
  a.ge = ge; a.ge_out = ge; a.t_hi = gt_hi; a.e_mid = e_mid;
  a.slab = slab; a.partials = partials_hi;
  a.t_lo = t_lo; a.P_lo = P_lo; a.Q_lo = Q_lo; a.q_pitch = 4 * H; a.qb_off = 2 * H; a.hf_lo = hf_lo; a.hb_lo = hb_lo;
  a.isrc = isrc; a.idst = idst; a.in_ptr = in_ptr;
  a.gP_lo = gP_lo; a.Ud_lo = gP_lo + 4 * H; a.Td_lo = gP_lo + 4 * H; a.ud_pitch = 5 * H;
  a.UT_lo = gP_lo + 3 * H; a.ut_pitch = 5 * H;
  a.partials_lo = partials_lo;
  a.sinfo = sinfo; a.margin = kSweepMargin;
  a.ln_gamma = gamma_lo; a.ln_beta = beta_lo; a.ln_width = width; a.gt_out = gt_lo;
  const int grid = edge_bwd_chain_launch(a, W3_hi, ws, st, g_matmul_mode == 2);
  GNM_CHECK_ARG(grid > 0, "ln_edge_bwd_chain: no kernel for this configuration");
  GNM_LAUNCH_CHECK("ln_edge_bwd_chain");
  hipLaunchKernelGGL(slab_reduce_k, dim3(FH * FH / 128), dim3(256), 0, st, (const float*)slab, grid, FH * FH, gW3_hi);
  GNM_LAUNCH_CHECK("ln_edge_bwd_chain slab reduce");
  *nblk_out = grid;
  return gnm_reduce_partials(partials_hi, grid, 1, FH, gb3_hi, stream) ? -3 : 0;
}

extern "C" int gnm_edge_bwd_chain(int64_t N, int64_t E, int H, const float* ge, float* ge_out, const float* t_hi,
                                  const float* e_mid, const float* stat_hi, const float* bstat_hi, const float* gamma_hi,
                                  const float* W3_hi, float* gW3_hi, float* gb3_hi, double* partials_hi,
                                  const float* t_lo, const float* stat_lo, const float* P_lo, const float* Q_lo,
                                  const float* hf_lo, const float* hb_lo, const int32_t* isrc, const int32_t* idst,
                                  const int32_t* in_ptr, float* gP_lo, float* Ud_lo, float* Td_lo, double* partials_lo,
                                  int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  return edge_bwd_chain_impl(N, E, H, ge, ge_out, t_hi, e_mid, stat_hi, bstat_hi, gamma_hi, W3_hi, gW3_hi, gb3_hi, partials_hi,
                             t_lo, stat_lo, P_lo, Q_lo, hf_lo, hb_lo, isrc, idst, in_ptr, gP_lo, Ud_lo, Td_lo, partials_lo,
                             nullptr, 0, nullptr, nblk_out, ws, ws_bytes, stream);
}

extern "C" int gnm_edge_bwd_chain_src(int64_t N, int64_t E, int H, const float* ge, float* ge_out, const float* t_hi,
                                      const float* e_mid, const float* stat_hi, const float* bstat_hi, const float* gamma_hi,
                                      const float* W3_hi, float* gW3_hi, float* gb3_hi, double* partials_hi,
                                      const float* t_lo, const float* stat_lo, const float* P_lo, const float* Q_lo,
                                      const float* hf_lo, const float* hb_lo, const int32_t* isrc, const int32_t* idst,
                                      const int32_t* in_ptr, float* gP_lo, float* Ud_lo, float* Td_lo, double* partials_lo,
                                      const uint32_t* sinfo, int64_t plan_nodes_per_block,
                                      float* UT_lo, int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(sinfo, "edge_bwd_chain_src: sinfo is null");
  return edge_bwd_chain_impl(N, E, H, ge, ge_out, t_hi, e_mid, stat_hi, bstat_hi, gamma_hi, W3_hi, gW3_hi, gb3_hi, partials_hi,
                             t_lo, stat_lo, P_lo, Q_lo, hf_lo, hb_lo, isrc, idst, in_ptr, gP_lo, Ud_lo, Td_lo, partials_lo,
                             sinfo, plan_nodes_per_block, UT_lo, nblk_out, ws, ws_bytes, stream);
}

extern "C" size_t gnm_edge_bwd_fused_workspace_bytes(void) {
  // packed W3 + one 128x128 slab per possible workgroup
  return gnm_rowtile_workspace_bytes(FH) + (size_t)kMaxPartialBlocks * FH * FH * sizeof(float);
}

template <class MM>
static int edge_bwd_fused_impl(int64_t E, const float* ge, float* ge_out, const float* t, const float* e_in,
                               const float* stat_e, const float* bstat_e, const float* gamma_e, const float* W3,
                               float* gW3, float* gb3, double* partials, void* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
  int grid;
  if constexpr (MM::kSplit) {
    grid = edge_bwd_tr_launch(E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e, W3, ws, slab, partials, st, g_matmul_mode == 2);
    GNM_LAUNCH_CHECK("edge_bwd_fused (tr)");
    hipLaunchKernelGGL(slab_reduce_k, dim3(FH * FH / 128), dim3(256), 0, st, (const float*)slab, grid, FH * FH, gW3);
    GNM_LAUNCH_CHECK("edge_bwd_fused slab reduce");
    return gnm_reduce_partials(partials, grid, 1, FH, gb3, stream) ? -3 : 0;
  }
  launch_pack<MM>(W3, FH, FH / 32, 1, ws, st);
  GNM_LAUNCH_CHECK("pack_w (NN)");
  const int64_t ntiles = cdiv_(E, FTR2);
  grid = persistent_grid(ntiles, 8, occ_blocks<edge_bwd_fused32_k>());
  hipLaunchKernelGGL(edge_bwd_fused32_k, dim3(grid), dim3(kBlock), 0, st, E, ge, ge_out, t, e_in, stat_e, bstat_e,
                     gamma_e, (const void*)ws, slab, partials, cdiv_(ntiles, grid));
  GNM_LAUNCH_CHECK("edge_bwd_fused");
  hipLaunchKernelGGL(slab_reduce_k, dim3(FH * FH / 128), dim3(256), 0, st, (const float*)slab, grid, FH * FH, gW3);
  GNM_LAUNCH_CHECK("edge_bwd_fused slab reduce");
  return gnm_reduce_partials(partials, grid, 1, FH, gb3, stream) ? -3 : 0;
}

extern "C" int gnm_edge_bwd_fused(int64_t E, int H, const float* ge, float* ge_out, const float* t, const float* e_in,
                                  const float* stat_e, const float* bstat_e, const float* gamma_e,
                                  const float* W3, float* gW3, float* gb3, double* partials, void* ws,
                                  size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "edge_bwd_fused: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(E > 0 && ge && ge_out && t && e_in && stat_e && bstat_e && gamma_e && W3 && gW3 && gb3 && partials,
                "edge_bwd_fused: null/neg argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_edge_bwd_fused_workspace_bytes(), "edge_bwd_fused: workspace %zu < %zu", ws_bytes,
                gnm_edge_bwd_fused_workspace_bytes());
  return g_matmul_mode ? edge_bwd_fused_impl<MmB3>(E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e, W3, gW3, gb3, partials, ws, stream)
                       : edge_bwd_fused_impl<MmF32>(E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e, W3, gW3, gb3, partials, ws, stream);
}

// gt GIVEN (the LayerNorm backward, whose gt comes out of its by-destination pass): gW3 = gt^T e_in, gb3 = sum gt, ge_out = ge + gt W3 in one
// pass over gt, ge, e_in -- the two generic GEMMs + column sum of the LayerNorm mode at H = 128.  Split matmul modes only (-4 otherwise:
// the caller keeps the generic route).
extern "C" int gnm_edge_bwd_fused_gt(int64_t E, int H, const float* ge, float* ge_out, const float* gt, const float* e_in,
                                     const float* W3, float* gW3, float* gb3, double* partials, void* ws, size_t ws_bytes,
                                     void* stream) {
  GNM_CHECK_ARG(H == FH, "edge_bwd_fused_gt: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(E > 0 && ge && ge_out && gt && e_in && W3 && gW3 && gb3 && partials, "edge_bwd_fused_gt: null/neg argument");
  GNM_CHECK_ARG(gt != ge_out, "edge_bwd_fused_gt: gt and ge_out must not alias");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_edge_bwd_fused_workspace_bytes(), "edge_bwd_fused_gt: workspace %zu < %zu", ws_bytes,
                gnm_edge_bwd_fused_workspace_bytes());
  if (!g_matmul_mode) { ::gnm::set_error("edge_bwd_fused_gt: built for the split matmul modes (1, 2) only"); return -4; }
  hipStream_t st = (hipStream_t)stream;
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
  const int grid = edge_bwd_tr_launch(E, ge, ge_out, gt, e_in, nullptr, nullptr, nullptr, W3, ws, slab, partials, st, g_matmul_mode == 2, true);
  GNM_LAUNCH_CHECK("edge_bwd_fused_gt (tr)");
  hipLaunchKernelGGL(slab_reduce_k, dim3(FH * FH / 128), dim3(256), 0, st, (const float*)slab, grid, FH * FH, gW3);
  GNM_LAUNCH_CHECK("edge_bwd_fused_gt slab reduce");
  return gnm_reduce_partials(partials, grid, 1, FH, gb3, stream) ? -3 : 0;
}

static int tn_colgroups(int64_t N, const float* A, int64_t lda, int ncg, const float* B, float* gW, float* gb,
                        double* partials, float* slab, void* stream, int max_blocks_per_cu = 0, const TnConv* cv = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  const bool tr = g_matmul_mode || cv;                   // split mode: tn_tr_k (transpose reads); fp32-MFMA mode: tn_colgroup_k
  const int64_t ntiles = cdiv_(N, tr ? tn_tr_rows_per_tile() : FTR);
  const bool h2 = g_matmul_mode == 2;                    // f16x2 (tn_tr_k<., ., true>)
  int occ = tr ? tn_tr_occupancy(cv != nullptr, h2) : occ_blocks<tn_colgroup_k<MmF32>>();
  if (max_blocks_per_cu > 0 && occ > max_blocks_per_cu) occ = max_blocks_per_cu;    // the caller shares the CUs with another stream
  int nslot = (num_cus() * occ) / ncg;
  if (nslot > kMaxPartialBlocks / ncg) nslot = kMaxPartialBlocks / ncg;
  if ((int64_t)nslot > ntiles) nslot = (int)ntiles;
  nslot = nslot / kXcds * kXcds;             // whole slots per XCD (see tn_colgroup_k)
  if (nslot < kXcds) nslot = kXcds;          // empty slots write zero slabs
  if (tr)
    tn_tr_launch(N, A, lda, ncg, B, FH, 1, slab, partials, nslot, cdiv_(ntiles, nslot), st, cv, h2);
  else
    hipLaunchKernelGGL(tn_colgroup_k<MmF32>, dim3(nslot * ncg), dim3(kBlock), 0, st, N, A, lda, ncg, B, slab, partials,
                       nslot, cdiv_(ntiles, nslot));
  GNM_LAUNCH_CHECK("tn_colgroup");
  // one launch each for all column groups (same per-element summation order as one launch per group)
  hipLaunchKernelGGL(slab_reduce_k, dim3(FH * FH / 128, ncg), dim3(256), 0, st, (const float*)slab, nslot, FH * FH, gW);
  if (reduce_partials_batched(partials, ncg, nslot, FH, gb, stream)) return -3;
  GNM_LAUNCH_CHECK("tn_colgroup reduce");
  return 0;
}

// ------------------------------------------------------------------------------------------
// Split-mode route of gnm_gemm_f32 (gnm_gemm.hip) for the big-M shapes whose other two dimensions are multiples of
// 128: NT / NN through gemm_rows_b3_k, TN through tn_tr_k with (A group, B group) classes.  Returns 1 = done,
// 0 = not eligible (the caller runs the fp32-MFMA kernel), < 0 = error.
// ------------------------------------------------------------------------------------------
namespace gnm {
static bool gemm_b3_shape_ok(int mode, int64_t M, int64_t N, int64_t K) {
  if (!g_matmul_mode) return false;
  if (mode == GNM_GEMM_TN) return K >= 4096 && M > 0 && N > 0 && M % FH == 0 && N % FH == 0 && (M / FH) * (N / FH) <= 64;
  return M >= 2048 && N > 0 && K > 0 && N % FH == 0 && K % FH == 0 && (N / FH) * (K / FH) <= 256;
}
static int gemm_b3_tn_slots(int64_t rows, int ncls) {
  const int64_t ntiles = cdiv_(rows, tn_tr_rows_per_tile());
  int nslot = (num_cus() * tn_tr_occupancy(false, g_matmul_mode == 2)) / ncls;
  if (nslot > kMaxPartialBlocks / ncls) nslot = kMaxPartialBlocks / ncls;
  if ((int64_t)nslot > ntiles) nslot = (int)ntiles;
  nslot = nslot / kXcds * kXcds;
  if (nslot < kXcds) nslot = kXcds;
  return nslot;
}
size_t gemm_b3_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
  if (!gemm_b3_shape_ok(mode, M, N, K)) return 0;
  if (mode == GNM_GEMM_TN) {
    const int ncls = (int)((M / FH) * (N / FH));
    const size_t blocks = (size_t)ncls * gemm_b3_tn_slots(K, ncls);
    return blocks * FH * FH * sizeof(float) + blocks * FH * sizeof(double);
  }
  return (size_t)(N / FH) * (K / FH) * 4 * MmB3::kPackBytes;
}
int gemm_b3_try(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, const float* bias, const float* resid, int64_t ldr, int relu, void* ws,
                size_t ws_bytes, hipStream_t st) {
  if (!gemm_b3_shape_ok(mode, M, N, K)) return 0;
  auto al = [](const void* p, int64_t ld) { return (uintptr_t)p % 16 == 0 && ld % 4 == 0; };
  if (!al(A, lda) || !ws || ws_bytes < gemm_b3_workspace_bytes(mode, M, N, K)) return 0;
  if (mode == GNM_GEMM_TN) {            // C[M,N] = A[K,M]^T B[K,N]
    if (bias || resid || relu || !al(B, ldb)) return 0;
    const int ncga = (int)(M / FH), ncgb = (int)(N / FH), ncls = ncga * ncgb;
    const int nslot = gemm_b3_tn_slots(K, ncls);
    const int64_t ntiles = cdiv_(K, tn_tr_rows_per_tile());
    float* slab = (float*)ws;
    double* partials = (double*)((char*)ws + (size_t)ncls * nslot * FH * FH * sizeof(float));
    tn_tr_launch(K, A, lda, ncga, B, ldb, ncgb, slab, partials, nslot, cdiv_(ntiles, nslot), st, nullptr, g_matmul_mode == 2);
    hipLaunchKernelGGL(slab_reduce_ld_k, dim3(FH, ncls), dim3(256), 0, st, (const float*)slab, nslot, ncgb, C, ldc);   // one launch for all blocks
    return hipGetLastError() == hipSuccess ? 1 : -2;
  }
  if (!al(C, ldc) || (resid && !al(resid, ldr)) || (bias && (uintptr_t)bias % 16 != 0)) return 0;
  const int ncls = (int)(N / FH), ncg = (int)(K / FH);
  const bool h2 = g_matmul_mode == 2;
  if (h2) launch_pack_w2_gen(B, ldb, ncls, ncg, mode == GNM_GEMM_NN ? 1 : 0, ws, st);
  else hipLaunchKernelGGL(pack_w3_gen_k, dim3(4 * ncls * ncg), dim3(256), 0, st, B, ldb, ncls, ncg, mode == GNM_GEMM_NN ? 1 : 0,
                          (bf16x8*)ws);
  constexpr int T = 4;
  const int64_t ngroups = cdiv_(M, NR3 * T);
  int nchunk = (num_cus() * (h2 ? occ_blocks<gemm_rows_b3_k<MmH2, T>>() : occ_blocks<gemm_rows_b3_k<MmB3, T>>())) / ncls;
  if ((int64_t)nchunk > ngroups) nchunk = (int)ngroups;
  nchunk = nchunk / kXcds * kXcds;
  if (nchunk < kXcds) nchunk = kXcds;
  if (h2) hipLaunchKernelGGL((gemm_rows_b3_k<MmH2, T>), dim3(nchunk * ncls), dim3(kBlock), 0, st, M, A, lda, ncg, ncls, (const void*)ws,
                             bias, resid, ldr, relu, C, ldc, nchunk, cdiv_(ngroups, nchunk));
  else hipLaunchKernelGGL((gemm_rows_b3_k<MmB3, T>), dim3(nchunk * ncls), dim3(kBlock), 0, st, M, A, lda, ncg, ncls, (const void*)ws,
                          bias, resid, ldr, relu, C, ldc, nchunk, cdiv_(ngroups, nchunk));
  return hipGetLastError() == hipSuccess ? 1 : -2;
}
// the same TN product plus the column sums of A (a Linear's bias gradient beside its weight gradient): tn_tr_k keeps them
// in fp64 per workgroup anyway
int gemm_b3_tn_colsum_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                          int64_t ldc, float* colsum, void* ws, size_t ws_bytes, hipStream_t st) {
  const int rc = gemm_b3_try(GNM_GEMM_TN, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, 0, ws, ws_bytes, st);
  if (rc != 1) return rc;
  const int ncga = (int)(M / FH), ncgb = (int)(N / FH), ncls = ncga * ncgb;
  const int nslot = gemm_b3_tn_slots(K, ncls);
  const double* partials = (const double*)((const char*)ws + (size_t)ncls * nslot * FH * FH * sizeof(float));
  for (int cga = 0; cga < ncga; ++cga)
    if (gnm_reduce_partials(partials + (size_t)(cga * ncgb) * nslot * FH, nslot, 1, FH, colsum + (size_t)cga * FH, (void*)st)) return -3;
  return 1;
}
}  // namespace gnm

// gh_in = gh_out + gP W  (W [ncols,128] row-major, ncols % 128 == 0);  gW = gP^T h_in;  gb = sum gP.
// ws: packed W (ncols/32 fragment blocks) + slabs [ncg][nslot][128][128]; partials double[ncg*nslot][128]
extern "C" size_t gnm_node_proj_bwd_workspace_bytes(int ncols) {
  return gnm_rowtile_workspace_bytes(ncols) + (size_t)kMaxPartialBlocks * FH * FH * sizeof(float);
}

template <class MM>
static int node_proj_bwd_nn(int64_t N, int ncols, const float* gP, const float* W, const float* gh_out, float* gh_in,
                            void* ws, hipStream_t st) {
  const int ncg = ncols / FH;
  if constexpr (MM::kSplit) {        // W rows cg*128.. form the [k=128, c=128] block of group cg: ONE launch for all groups
    // (pack_w3_gen_k with one output class: k runs over the whole stacked weight, block (cg*4 + wv) as pack_w3_k lays it out)
    if constexpr (MM::kScaled) launch_pack_w2_gen(W, (int64_t)FH, 1, ncg, 1, ws, st);
    else hipLaunchKernelGGL(pack_w3_gen_k, dim3(4 * ncg), dim3(256), 0, st, W, (int64_t)FH, 1, ncg, 1, (bf16x8*)ws);
  } else {
    for (int cg = 0; cg < ncg; ++cg)
      launch_pack<MM>(W + (size_t)cg * FH * FH, FH, FH / 32, 1, (char*)ws + (size_t)cg * 4 * MM::kPackBytes, st);
  }
  GNM_LAUNCH_CHECK("pack_w (NN, node)");
  if constexpr (MM::kSplit) {
    constexpr int T = 4;
    const int64_t ngroups = cdiv_(N, NR3 * T);
    const int grid = persistent_grid(ngroups, 1, occ_blocks<rowtile_nn_group32_b3_k<MM, T>>());
    hipLaunchKernelGGL((rowtile_nn_group32_b3_k<MM, T>), dim3(grid), dim3(kBlock), 0, st, N, gP, (int64_t)ncols, ncg,
                       (const void*)ws, gh_out, gh_in, cdiv_(ngroups, grid));
  } else {
    const int64_t ntiles = cdiv_(N, FTR);
    const int grid = persistent_grid(ntiles, 2, occ_blocks<rowtile_nn_acc_k<MM>>());
    hipLaunchKernelGGL(rowtile_nn_acc_k<MM>, dim3(grid), dim3(kBlock), 0, st, N, gP, (int64_t)ncols, ncg,
                       (const void*)ws, gh_out, gh_in, cdiv_(ntiles, grid));
  }
  GNM_LAUNCH_CHECK("node_proj_bwd (NN)");
  return 0;
}

// The two halves of gnm_node_proj_bwd as separate entry points (same workspace layout), so that a caller can
// time / schedule the NN and the TN kernel on their own.
extern "C" int gnm_node_proj_bwd_nn(int64_t N, int H, int ncols, const float* gP, const float* W, const float* gh_out,
                                    float* gh_in, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "node_proj_bwd_nn: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(N > 0 && ncols > 0 && ncols % FH == 0 && gP && W && gh_out && gh_in, "node_proj_bwd_nn: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_rowtile_workspace_bytes(ncols), "node_proj_bwd_nn: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  return (g_matmul_mode == 2 ? node_proj_bwd_nn<MmH2>(N, ncols, gP, W, gh_out, gh_in, ws, st)
          : g_matmul_mode  ? node_proj_bwd_nn<MmB3>(N, ncols, gP, W, gh_out, gh_in, ws, st)
                           : node_proj_bwd_nn<MmF32>(N, ncols, gP, W, gh_out, gh_in, ws, st)) ? -2 : 0;
}

// gnm_node_proj_bwd_nn with the BatchNorm_h backward sums of the layer below (gnm_node_bwd_stats over gh_in and z_lo) in its
// epilogue; see rowtile_nn2_k.  bf16x3 mode, H = 128.  partials / *nblk_out: as gnm_node_bwd_stats leaves them.
extern "C" int gnm_node_proj_bwd_nn_stats(int64_t N, int H, int ncols, const float* gP, const float* W, const float* gh_out,
                                          float* gh_in, const float* z_lo, const float* stat_h_lo, double* partials,
                                          int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "node_proj_bwd_nn_stats: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(g_matmul_mode >= 1, "node_proj_bwd_nn_stats: split matmul modes only");
  GNM_CHECK_ARG(N > 0 && ncols > 0 && ncols % FH == 0 && gP && W && gh_out && gh_in && z_lo && stat_h_lo && partials && nblk_out,
                "node_proj_bwd_nn_stats: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_rowtile_workspace_bytes(ncols), "node_proj_bwd_nn_stats: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int ncg = ncols / FH;
  const bool h2 = g_matmul_mode == 2;
  if (h2) launch_pack_w2_gen(W, (int64_t)FH, 1, ncg, 1, ws, st);
  else hipLaunchKernelGGL(pack_w3_gen_k, dim3(4 * ncg), dim3(256), 0, st, W, (int64_t)FH, 1, ncg, 1, (bf16x8*)ws);
  GNM_LAUNCH_CHECK("pack_w (NN, node, stats)");
  constexpr int T = 4;
  const int64_t ngroups = cdiv_(N, NR3 * T);
  const int grid = persistent_grid(ngroups, 1, h2 ? occ_blocks<rowtile_nn2_k<MmH2, T>>() : occ_blocks<rowtile_nn2_k<MmB3, T>>());
  const Nn2Args a{N, gP, (int64_t)ncols, ncg, (const void*)ws, gh_out, gh_in, cdiv_(ngroups, grid), z_lo, stat_h_lo, partials};
  if (h2) hipLaunchKernelGGL((rowtile_nn2_k<MmH2, T>), dim3(grid), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL((rowtile_nn2_k<MmB3, T>), dim3(grid), dim3(kBlock), 0, st, a);
  *nblk_out = grid;
  GNM_LAUNCH_CHECK("node_proj_bwd_nn_stats");
  return 0;
}

extern "C" int gnm_node_proj_bwd_tn(int64_t N, int H, int ncols, const float* gP, const float* h_in, float* gW, float* gb,
                                    double* partials, void* ws, size_t ws_bytes, int max_blocks_per_cu, void* stream) {
  GNM_CHECK_ARG(H == FH, "node_proj_bwd_tn: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(N > 0 && ncols > 0 && ncols % FH == 0 && gP && h_in && gW && gb && partials && max_blocks_per_cu >= 0,
                "node_proj_bwd_tn: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_node_proj_bwd_workspace_bytes(ncols), "node_proj_bwd_tn: workspace too small");
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(ncols));
  return tn_colgroups(N, gP, ncols, ncols / FH, h_in, gW, gb, partials, slab, stream, max_blocks_per_cu);
}

extern "C" int gnm_node_proj_bwd(int64_t N, int H, int ncols, const float* gP, const float* h_in, const float* W,
                                 const float* gh_out, float* gh_in, float* gW, float* gb, double* partials,
                                 void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_node_proj_bwd_workspace_bytes(ncols), "node_proj_bwd: workspace too small");
  const int rc = gnm_node_proj_bwd_nn(N, H, ncols, gP, W, gh_out, gh_in, ws, ws_bytes, stream);
  return rc ? rc : gnm_node_proj_bwd_tn(N, H, ncols, gP, h_in, gW, gb, partials, ws, ws_bytes, 0, stream);
}

// out[cg*128 + n][c] = sum_rows A[row][cg*128 + n] * B[row][c];  colsum[cg*128 + n] = sum_rows A[row][cg*128 + n]
// (A [M, lda >= ncg*128], B [M,128]): the weight-gradient shape of every Linear whose input is 128 wide.
extern "C" size_t gnm_tn128_workspace_bytes(void) { return (size_t)kMaxPartialBlocks * FH * FH * sizeof(float); }

extern "C" int gnm_tn128(int64_t M, const float* A, int64_t lda, int ncg, const float* B, float* out, float* colsum,
                         double* partials, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(M > 0 && A && B && out && colsum && partials && ncg > 0 && lda >= (int64_t)ncg * FH && ncg <= 16,
                "tn128: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_tn128_workspace_bytes(), "tn128: workspace too small");
  return tn_colgroups(M, A, lda, ncg, B, out, colsum, partials, (float*)ws, stream);
}

// gnm_tn128 over the column groups gB1h | gB2h of gP (ncg = 2: out = gW5[3H:5H], colsum = gb5[3H:5H]) with gnm_node_bgrad in
// its operand load: the groups are formed from the raw sums UT = [Us | Ts] (pitch 2H), Ud, Td (pitch ud_pitch = H or 2H) and
// the BatchNorm_e backward means, and WRITTEN to gP[:, 3H:5H] (row pitch 5H) for gnm_node_proj_bwd_nn*, which runs behind
// this call.  B: the layer's h_in [M,128].  Split matmul modes, H = 128.
extern "C" int gnm_tn128_bgrad(int64_t M, int H, const float* UT, const float* Ud, const float* Td, int64_t ud_pitch,
                               const float* stat_e, const float* bstat_e, const float* gamma_e, const int32_t* in_ptr,
                               const int32_t* out_ptr, float* gP, const float* B, float* out, float* colsum,
                               double* partials, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "tn128_bgrad: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(g_matmul_mode >= 1, "tn128_bgrad: split matmul modes only");
  GNM_CHECK_ARG(M > 0 && UT && Ud && Td && (ud_pitch == H || ud_pitch == 2 * H) && stat_e && bstat_e && gamma_e && in_ptr &&
                    out_ptr && gP && B && out && colsum && partials, "tn128_bgrad: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_tn128_workspace_bytes(), "tn128_bgrad: workspace too small");
  const TnConv cv{{UT, Ud}, {UT + H, Td}, {(int64_t)2 * H, ud_pitch}, {out_ptr, in_ptr}, stat_e, bstat_e, gamma_e,
                  gP + 3 * H, (int64_t)5 * H};
  return tn_colgroups(M, nullptr, 0, 2, B, out, colsum, partials, (float*)ws, stream, 0, &cv);
}

