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
//   edge_bwd_fused_k     gt = gamma*rstd*(gu - m1 - that*m2); ge_in = ge + gt W3;
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
#include <type_traits>

#include "gnm_common.h"

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
// Y[:, cg*128 + c] = X W_cg^T + bias   (+ gathers and column statistics when EDGE)
// One workgroup per CU = one wave per SIMD with the whole 512-entry register file.
// ------------------------------------------------------------------------------------------
// NCG (number of 128-column groups) is a template parameter so that the column-group loop unrolls:
// a loop with a run-time trip count around the stores would defeat the vmcnt counting below.
template <bool EDGE, int NCG>
__global__ __launch_bounds__(kBlock, 1) void rowtile_nt_k(
    int64_t M, const float* __restrict__ X, const float* __restrict__ Wp, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t ldy, const float* __restrict__ P,
    const int32_t* __restrict__ isrc, const int32_t* __restrict__ idst, double* __restrict__ partials,
    int64_t tiles_per_block) {
  __shared__ float xs[FTR * FP];
  __shared__ float ys[EDGE ? 4 : FTR * FP];   // node mode: output image (X is reused by 5 column groups)
  __shared__ int sd[2 * FTR];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (M + FTR - 1) / FTR;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, M / FTR);      // tiles [tb0, nfull) are full
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;   // this thread's slot in the coalesced tile image
  const int64_t Mlast = M - 1;

  float4 wf[FKQ];
  auto load_w = [&](int cg) __attribute__((always_inline)) {
    const float4* p = reinterpret_cast<const float4*>(Wp) + ((int64_t)(cg * 4 + wave) * FKQ) * 64 + lane;
#pragma unroll
    for (int q = 0; q < FKQ; ++q) wf[q] = p[q * 64];
  };
  constexpr int ncg = NCG;
  if (ncg == 1) load_w(0);

  float4 pre[8];
  int pre_idx = 0;
  // branch-free: rows past the end are clamped to the last row (their results are never stored)
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) pre[it] = ld4(X + clampi(r0 + lrow + 8 * it, Mlast) * FH + lc4);
    if (EDGE) {
      const int64_t r = clampi(r0 + (tid & (FTR - 1)), Mlast);
      pre_idx = (tid & FTR) ? idst[r] : isrc[r];     // threads 0-63: src, 64-127: dst (128-255 unused)
    }
  };

  Stat4 st;
  st.zero();
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    __syncthreads();   // everyone is done with the previous tile's LDS images
#pragma unroll
    for (int it = 0; it < 8; ++it) st4(xs + (lrow + 8 * it) * FP + lc4, pre[it]);
    if (EDGE && tid < 2 * FTR) sd[tid] = pre_idx;
    __syncthreads();
    const int64_t r0 = tile * FTR;
    // gathers of this tile's B1h[src] / B2h[dst] rows: issued now, consumed in the epilogue
    float4 g1[8], g2[8];
    if (EDGE) {
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
      mma_tile64(xs, wf, acc0, acc1, li, lg);
      float* os = EDGE ? xs : ys;
      if (EDGE) __syncthreads();            // all waves are done reading the X image
      else if (cg > 0) __syncthreads();     // previous column group's epilogue is done with ys
      acc_to_lds(os, acc0, acc1, wave, li, lg);
      __syncthreads();
      const float4 b4 = ld4(bias + cg * FH + lc4);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int row = lrow + 8 * it;
        const int64_t grow = r0 + row;
        float4 v = ld4(os + row * FP + lc4) + b4;
        if (EDGE) v = v + g1[it] + g2[it];
        if (FULL || grow < M) {
          st4(Y + grow * ldy + cg * FH + lc4, v);
          if (EDGE) st.add_prod(v, v);
        }
      }
    }
  };

  if (tb0 < tb1) prefetch(tb0);
  if (tb0 < nfull) {
    // throw-away stores behind the first prefetch (same addresses the first epilogue rewrites):
    // they make the loop-entry scoreboard equal to the back edge's, see edge_bwd_fused_k
#pragma unroll
    for (int it = 0; it < 8; ++it) st4(Y + (tb0 * FTR + lrow + 8 * it) * ldy + lc4, f4(0.f));
    for (int64_t tile = tb0; tile < nfull; ++tile) body(full_t{}, tile);
  }
  if (nfull < tb1 && nfull >= tb0) body(ragged_t{}, nfull);
  if (EDGE) {
    __syncthreads();   // the last epilogue is done with the LDS image we reuse for the reduction
    block_stat_store<FH>(st, reinterpret_cast<double*>(xs), partials, chunk);
  }
}

// ------------------------------------------------------------------------------------------
// fused edge backward: gt prologue + NN (ge_in) + TN (gW3 slab) + column sum of gt
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 1) void edge_bwd_fused_k(
    int64_t E, const float* ge, float* ge_out, const float* __restrict__ t, const float* __restrict__ e_in,
    const float* __restrict__ stat, const float* __restrict__ bstat, const float* __restrict__ gamma,
    const float* __restrict__ Wp,                    // W3 packed NN
    float* __restrict__ slab,                        // [grid][128][128] partial gW3
    double* __restrict__ partials,                   // [grid][128]: per-workgroup column sums of gt
    int64_t tiles_per_block) {
  __shared__ float gs[FTR * FP];       // gt tile, later the transposed output image
  __shared__ float es[FTR * FP];       // e_in tile
  __shared__ float cs[7 * FH];         // mu, rstd, scale, shift, m1, m2, c = gamma*rstd
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + FTR - 1) / FTR;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, E / FTR);
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
  float4 wf[FKQ];
  {
    const float4* p = reinterpret_cast<const float4*>(Wp) + ((int64_t)wave * FKQ) * 64 + lane;
#pragma unroll
    for (int q = 0; q < FKQ; ++q) wf[q] = p[q * 64];
  }
  floatx16 tn[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) tn[a][b][e] = 0.f;
  double cg0 = 0.0, cg1 = 0.0, cg2 = 0.0, cg3 = 0.0;   // column sums of gt for columns lc4..lc4+3
  __syncthreads();

  // ge / t / e_in rows of a tile are prefetched one tile ahead (96 VGPRs), under the MFMA phases
  float4 pg[8], pt[8], pe_[8];
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t o = clampi(r0 + lrow + 8 * it, Elast) * FH + lc4;
      pg[it] = ld4(ge + o);
      pt[it] = ld4(t + o);
      pe_[it] = ld4(e_in + o);
    }
  };

  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * FTR;
    // ---- phase 0: gt tile and e_in tile into LDS (coalesced float4 image) ----
    float4 gk[8];   // this tile's ge rows, kept for the residual add in the epilogue
    {
      const float4 mu = ld4(cs + lc4), rs = ld4(cs + FH + lc4), sc = ld4(cs + 2 * FH + lc4),
                   sh = ld4(cs + 3 * FH + lc4), m1 = ld4(cs + 4 * FH + lc4), m2 = ld4(cs + 5 * FH + lc4),
                   cc = ld4(cs + 6 * FH + lc4);
#pragma unroll
      for (int it = 0; it < 8; ++it) {
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
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);   // in flight under the MFMAs below
    // ---- phase 1: acc = gt W3   (this wave: output columns wave*32 .. +31) ----
    floatx16 acc0, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    mma_tile64(gs, wf, acc0, acc1, li, lg);
    // ---- phase 2: gW3[n][c] += sum_rows gt[row][n] e_in[row][c]  (this wave: 64 x 64 block) ----
    mma_tn64(gs, es, tn, wn, wc, li, lg);
    __syncthreads();   // gt / e_in images are dead: reuse gs as the transposed output image
    acc_to_lds(gs, acc0, acc1, wave, li, lg);
    __syncthreads();
    // ---- ge_in = ge + gt W3, whole 512-byte rows, one float4 per lane ----
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = lrow + 8 * it;
      const int64_t grow = r0 + row;
      if (FULL || grow < E) st4(ge_out + grow * FH + lc4, ld4(gs + row * FP + lc4) + gk[it]);
    }
    __syncthreads();   // gs is rewritten by the next tile's phase 0
  };

  if (tb0 < tb1) prefetch(tb0);
  // hipcc merges the vector-memory scoreboard of the loop entry with that of the back edge and
  // keeps the weaker guarantee: without stores behind the first prefetch it would wait vmcnt(0)
  // (= for the previous tile's stores) before the last prefetched row on EVERY iteration.  Eight
  // throw-away stores into this workgroup's slab (rewritten at the end) make both edges alike.
#pragma unroll
  for (int it = 0; it < 8; ++it) st4(slab + (size_t)chunk * FH * FH + (lrow + 8 * it) * FH + lc4, f4(0.f));
  for (int64_t tile = tb0; tile < nfull; ++tile) body(full_t{}, tile);
  if (nfull < tb1 && nfull >= tb0) body(ragged_t{}, nfull);

  // ---- write the partial gW3 slab and the column sums ----
  float* sl = slab + (size_t)chunk * FH * FH;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = (2 * wn + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
        const int c = (2 * wc + b) * 32 + li;
        sl[n * FH + c] = tn[a][b][e];
      }
  // 8 row-slots (lrow) x 128 columns -> 128 column sums (the gt tile's LDS is free now)
  double* red = reinterpret_cast<double*>(gs);
  red[lrow * FH + lc4 + 0] = cg0;
  red[lrow * FH + lc4 + 1] = cg1;
  red[lrow * FH + lc4 + 2] = cg2;
  red[lrow * FH + lc4 + 3] = cg3;
  __syncthreads();
  if (tid < FH) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k * FH + tid];
    partials[(size_t)chunk * FH + tid] = s;
  }
}

// ------------------------------------------------------------------------------------------
// node-level backward of the 5-way projection (autograd of gated_gcn_full.py:107-112):
//   rowtile_nn_acc_k   gh_in = gh_out + gP W5            (K = 5*128, accumulated over 5 groups)
//   tn_colgroup_k      gW5[cg] = gP[:,cg]^T h_in, gb5[cg] = sum gP[:,cg]   (5 workgroup classes)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock, 2) void rowtile_nn_acc_k(
    int64_t M, const float* __restrict__ X, int64_t ldx, int ncg, const float* __restrict__ Wp,
    const float* __restrict__ R, float* __restrict__ Y, int64_t tiles_per_block) {
  __shared__ float xs[FTR * FP];
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
      float4 wf[FKQ];
      {
        const float4* p = reinterpret_cast<const float4*>(Wp) + ((int64_t)(cg * 4 + wave) * FKQ) * 64 + lane;
#pragma unroll
        for (int q = 0; q < FKQ; ++q) wf[q] = p[q * 64];
      }
      __syncthreads();   // previous chunk's fragment reads are done
#pragma unroll
      for (int it = 0; it < 8; ++it) st4(xs + (lrow + 8 * it) * FP + lc4, pre[it]);
      __syncthreads();
      if (cg + 1 < ncg) prefetch(tile, cg + 1);
      else prefetch(tile + 1 < tb1 ? tile + 1 : tile, 0);
      mma_tile64(xs, wf, acc0, acc1, li, lg);
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
    for (int it = 0; it < 8; ++it) st4(Y + (tb0 * FTR + lrow + 8 * it) * FH + lc4, f4(0.f));   // see edge_bwd_fused_k
    for (int64_t tile = tb0; tile < nfull; ++tile) body(full_t{}, tile);
  }
  if (nfull < tb1 && nfull >= tb0) body(ragged_t{}, nfull);
}

// slab[(cg*nslot + slot)][n][c] = sum over the slot's rows of A[row][cg*128+n] * B[row][c];
// partials[(cg*nslot + slot)][128] = column sums of A[:, cg*128 ..]
__global__ __launch_bounds__(kBlock, 2) void tn_colgroup_k(
    int64_t M, const float* __restrict__ A, int64_t lda, int ncg, const float* __restrict__ B,
    float* __restrict__ slab, double* __restrict__ partials, int nslot, int64_t tiles_per_slot) {
  __shared__ float as[FTR * FP];
  __shared__ float bs[FTR * FP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  const int cg = blockIdx.x % ncg, slot = blockIdx.x / ncg;
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
  float4 pa[8], pb[8];
  // loads only (no stores in the loop): clamped and branch-free, rows past the end are zeroed below
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * FTR;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int64_t r = clampi(r0 + lrow + 8 * it, Mlast);
      pa[it] = ld4(A + r * lda + cg * FH + lc4);
      pb[it] = ld4(B + r * FH + lc4);
    }
  };
  if (tb0 < tb1) prefetch(tb0);
  for (int64_t tile = tb0; tile < tb1; ++tile) {
    const int64_t r0 = tile * FTR;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const bool ok = r0 + lrow + 8 * it < M;
      const float4 av = ok ? pa[it] : f4(0.f);
      st4(as + (lrow + 8 * it) * FP + lc4, av);
      st4(bs + (lrow + 8 * it) * FP + lc4, pb[it]);
      c0 += (double)av.x; c1 += (double)av.y; c2 += (double)av.z; c3 += (double)av.w;
    }
    __syncthreads();
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);
    mma_tn64(as, bs, tn, wn, wc, li, lg);
  }
  float* sl = slab + (size_t)(cg * nslot + slot) * FH * FH;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = (2 * wn + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
        const int c = (2 * wc + b) * 32 + li;
        sl[n * FH + c] = tn[a][b][e];
      }
  __syncthreads();
  double* red = reinterpret_cast<double*>(as);
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

// out[i] = sum_b slab[b][i], fixed order -> deterministic
__global__ void slab_reduce_k(const float* __restrict__ slab, int nslab, int total, float* __restrict__ out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int b = 0; b < nslab; ++b) acc += slab[(size_t)b * total + i];
    out[i] = acc;
  }
}

}  // namespace gnm

using namespace gnm;

static inline int64_t cdiv_(int64_t a, int64_t b) { return (a + b - 1) / b; }

// workspace: packed weights (ncb * 16 * 64 float4)
extern "C" size_t gnm_rowtile_workspace_bytes(int ncols) { return (size_t)(ncols / 32) * FKQ * 64 * 16; }

extern "C" int gnm_edge_t_fused_fwd(int64_t E, int H, const float* e_in, const float* W3, const float* b3,
                                    const float* P, const int32_t* isrc, const int32_t* idst, float* t,
                                    double* partials, int* nblk_out, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "edge_t_fused_fwd: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(E > 0 && e_in && W3 && b3 && P && isrc && idst && t && partials && nblk_out, "edge_t_fused_fwd: null/neg argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_rowtile_workspace_bytes(FH), "edge_t_fused_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_w_k, dim3(16), dim3(256), 0, st, W3, (int64_t)FH, FH / 32, 0, (float*)ws);
  GNM_LAUNCH_CHECK("pack_w (NT)");
  const int64_t ntiles = cdiv_(E, FTR);
  const int grid = persistent_grid(ntiles, 4, occ_blocks<rowtile_nt_k<true, 1>>());
  hipLaunchKernelGGL((rowtile_nt_k<true, 1>), dim3(grid), dim3(kBlock), 0, st, E, e_in, (const float*)ws, b3, t,
                     (int64_t)FH, P, isrc, idst, partials, cdiv_(ntiles, grid));
  GNM_LAUNCH_CHECK("edge_t_fused_fwd");
  *nblk_out = grid;
  return 0;
}

extern "C" int gnm_node_proj_fwd(int64_t N, int H, int ncols, const float* h, const float* W, const float* b,
                                 float* Pout, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "node_proj_fwd: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(N > 0 && ncols == 5 * FH && h && W && b && Pout, "node_proj_fwd: bad argument (ncols must be 5*128)");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_rowtile_workspace_bytes(ncols), "node_proj_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_w_k, dim3(16 * ncols / FH), dim3(256), 0, st, W, (int64_t)FH, ncols / 32, 0, (float*)ws);
  GNM_LAUNCH_CHECK("pack_w (NT, node)");
  const int64_t ntiles = cdiv_(N, FTR);
  const int grid = persistent_grid(ntiles, 2, occ_blocks<rowtile_nt_k<false, 5>>());
  hipLaunchKernelGGL((rowtile_nt_k<false, 5>), dim3(grid), dim3(kBlock), 0, st, N, h, (const float*)ws, b, Pout,
                     (int64_t)ncols, (const float*)nullptr, (const int32_t*)nullptr,
                     (const int32_t*)nullptr, (double*)nullptr, cdiv_(ntiles, grid));
  GNM_LAUNCH_CHECK("node_proj_fwd");
  return 0;
}

extern "C" size_t gnm_edge_bwd_fused_workspace_bytes(void) {
  // packed W3 + one 128x128 slab per possible workgroup
  return gnm_rowtile_workspace_bytes(FH) + (size_t)kMaxPartialBlocks * FH * FH * sizeof(float);
}

extern "C" int gnm_edge_bwd_fused(int64_t E, int H, const float* ge, float* ge_out, const float* t, const float* e_in,
                                  const float* stat_e, const float* bstat_e, const float* gamma_e,
                                  const float* W3, float* gW3, float* gb3, double* partials, void* ws,
                                  size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "edge_bwd_fused: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(E > 0 && ge && ge_out && t && e_in && stat_e && bstat_e && gamma_e && W3 && gW3 && gb3 && partials,
                "edge_bwd_fused: null/neg argument");
  const int64_t ntiles = cdiv_(E, FTR);
  const int grid = persistent_grid(ntiles, 4, occ_blocks<edge_bwd_fused_k>());
  const size_t need = gnm_rowtile_workspace_bytes(FH) + (size_t)grid * FH * FH * sizeof(float);
  GNM_CHECK_ARG(ws && ws_bytes >= need, "edge_bwd_fused: workspace %zu < %zu", ws_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  float* wp = (float*)ws;
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(FH));
  hipLaunchKernelGGL(pack_w_k, dim3(16), dim3(256), 0, st, W3, (int64_t)FH, FH / 32, 1, wp);
  GNM_LAUNCH_CHECK("pack_w (NN)");
  hipLaunchKernelGGL(edge_bwd_fused_k, dim3(grid), dim3(kBlock), 0, st, E, ge, ge_out, t, e_in, stat_e, bstat_e,
                     gamma_e, (const float*)wp, slab, partials, cdiv_(ntiles, grid));
  GNM_LAUNCH_CHECK("edge_bwd_fused");
  hipLaunchKernelGGL(slab_reduce_k, dim3(64), dim3(256), 0, st, (const float*)slab, grid, FH * FH, gW3);
  GNM_LAUNCH_CHECK("edge_bwd_fused slab reduce");
  return gnm_reduce_partials(partials, grid, 1, FH, gb3, stream) ? -3 : 0;
}

// gh_in = gh_out + gP W  (W [ncols,128] row-major, ncols % 128 == 0);  gW = gP^T h_in;  gb = sum gP.
// ws: packed W (ncols/32 fragment blocks) + slabs [ncg][nslot][128][128]; partials double[ncg*nslot][128]
extern "C" size_t gnm_node_proj_bwd_workspace_bytes(int ncols) {
  return gnm_rowtile_workspace_bytes(ncols) + (size_t)kMaxPartialBlocks * FH * FH * sizeof(float);
}

extern "C" int gnm_node_proj_bwd(int64_t N, int H, int ncols, const float* gP, const float* h_in, const float* W,
                                 const float* gh_out, float* gh_in, float* gW, float* gb, double* partials,
                                 void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(H == FH, "node_proj_bwd: H=%d (only 128 is built)", H);
  GNM_CHECK_ARG(N > 0 && ncols > 0 && ncols % FH == 0 && gP && h_in && W && gh_out && gh_in && gW && gb && partials,
                "node_proj_bwd: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_node_proj_bwd_workspace_bytes(ncols), "node_proj_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int ncg = ncols / FH;
  float* wp = (float*)ws;
  float* slab = (float*)((char*)ws + gnm_rowtile_workspace_bytes(ncols));
  for (int cg = 0; cg < ncg; ++cg) {   // W rows cg*128.. form the [k=128, c=128] block of group cg
    hipLaunchKernelGGL(pack_w_k, dim3(16), dim3(256), 0, st, W + (size_t)cg * FH * FH, (int64_t)FH, FH / 32, 1,
                       wp + (size_t)cg * 4 * FKQ * 64 * 4);
  }
  GNM_LAUNCH_CHECK("pack_w (NN, node)");
  const int64_t ntiles = cdiv_(N, FTR);
  {
    const int grid = persistent_grid(ntiles, 2, occ_blocks<rowtile_nn_acc_k>());
    hipLaunchKernelGGL(rowtile_nn_acc_k, dim3(grid), dim3(kBlock), 0, st, N, gP, (int64_t)ncols, ncg,
                       (const float*)wp, gh_out, gh_in, cdiv_(ntiles, grid));
    GNM_LAUNCH_CHECK("node_proj_bwd (NN)");
  }
  {
    int nslot = (num_cus() * occ_blocks<tn_colgroup_k>()) / ncg;
    if (nslot > kMaxPartialBlocks / ncg) nslot = kMaxPartialBlocks / ncg;
    if ((int64_t)nslot > ntiles) nslot = (int)ntiles;
    if (nslot < 1) nslot = 1;
    hipLaunchKernelGGL(tn_colgroup_k, dim3(nslot * ncg), dim3(kBlock), 0, st, N, gP, (int64_t)ncols, ncg, h_in,
                       slab, partials, nslot, cdiv_(ntiles, nslot));
    GNM_LAUNCH_CHECK("node_proj_bwd (TN)");
    for (int cg = 0; cg < ncg; ++cg) {
      hipLaunchKernelGGL(slab_reduce_k, dim3(64), dim3(256), 0, st, (const float*)slab + (size_t)cg * nslot * FH * FH,
                         nslot, FH * FH, gW + (size_t)cg * FH * FH);
      if (gnm_reduce_partials(partials + (size_t)cg * nslot * FH, nslot, 1, FH, gb + cg * FH, stream)) return -3;
    }
    GNM_LAUNCH_CHECK("node_proj_bwd reduce");
  }
  return 0;
}
