// bf16x3 kernels on swizzled row-major split images + hardware transpose reads (see gnm_tr.h), gfx950.
//
//   tn_tr_k     weight gradient of a Linear whose input is 128 wide:  gW[cg] = A[:, cg]^T B,  gb[cg] = sum A[:, cg]
//               (autograd of gated_gcn_full.py:107-112 and of the predictor's node halves, score_predictor.py:13-17)
#include <stdlib.h>

#include "gnm_tr.h"
#include "gnm_ln.h"

namespace gnm {

constexpr int TRR = 32;                     // rows per tile
constexpr int TRIMG = TRR * SPITCH;         // bytes per image (8 KB)

// slab[(cg*nslot + slot)][m][n] = sum over the slot's rows of A[row][cg*128 + m] * B[row][n];
// partials[(cg*nslot + slot)][128] = column sums of A[:, cg*128 ..].
// 32-row tiles, 48 KB of LDS, coalesced float4 loads one tile ahead; each of the 4 waves owns a 64 x 64 block of
// the 128 x 128 result (64 accumulator registers).  Two or three workgroups share a CU, so one's split / staging
// VALU work and HBM waits run under the others' MFMAs.
// (Round 5 also built a variant that copied a PRE-SPLIT image of B instead of splitting the tile in every workgroup class:
//  -0.9 ms per step here, +1.7 in the kernel that wrote the image; removed)
// CONV (round 5): the A operand does not exist yet -- the two column groups of this launch are gB1h, gB2h, which the chained
// edge kernel left as RAW sums (Us | Ts by source, Ud | Td by destination); with the BatchNorm_e backward means m1, m2
//     gB1h = c (Us - outdeg m1 - m2 Ts),   gB2h = c (Ud - indeg m1 - m2 Td),   c = gamma_e rstd_e      (node_bgrad_k)
// A tile is FORMED from the sums on its way into the images (two rows of loads per operand row) and written once to
// gP[:, 3H:5H] for the kernel that multiplies gP by W5 right behind this one: the elementwise launch in between (6 [N,H]
// streams, 0.87 ms per layer) is gone.
// H2: the binades a row that lies d = R - P below the reference gives up on its A side (the B side gives up d - da)
__device__ __forceinline__ int h2_tn_da(int R, int P) { const int d = R - P; return d > 0 ? (d > 200 ? 100 : d >> 1) : 0; }
// H2 (round 5, gnm_set_matmul_mode(2)): the f16x2 form of gnm_tr.h -- three MFMAs per product, two images per operand.  A TN
// contraction runs over the tile's ROWS, so the unit of the products must be ONE for all rows that meet in an accumulator:
//   A row r is staged as A_r 2^(4 - EA_r - da), B row r as B_r 2^(4 - R + EA_r + da),
// R = the REFERENCE exponent of this workgroup: the largest EA + EB of any row so far; d = R - (EA_r + EB_r) >= 0 is what the row
// lies below it and da = d / 2 -- every product carries 2^(8 - R), a row AT the reference has both operands' largest
// magnitudes at 2^4 (22 bits each), a row below it gives up d / 2 binades on either side: its operands are held to 2^-25
// absolute = 2^-(29 - d/2) of their largest magnitude, its products to 2^-(28 + d/2) of the largest row product met -- always
// below the 2^-24 the fp32 accumulator resolves -- and it is gone at d = 56.  A row ABOVE R (da = 0) has room for 2^10 before
// the fp16 range ends.  R is known without a barrier: the row owners leave EA + EB in LDS when they stage, after the
// staging barrier every wave takes the tile's maximum of them -- the same number in every wave -- and the accumulators
// move to the new unit (one exact multiplication) after the tile's MFMAs.  Only a tile with a row more than 2^10 above R
// (the first tile; a jump of three decades between neighbouring tiles) is staged again with its own maximum as R.
template <bool CONV, int OCC, bool H2 = false>
__global__ __launch_bounds__(kBlock, OCC) void tn_tr_k(int64_t M, const float* __restrict__ A, int64_t lda, int ncg,
                                                     const void* __restrict__ Bv, int64_t ldb, int ncgb,
                                                     float* __restrict__ slab, double* __restrict__ partials, int nslot,
                                                     int64_t tiles_per_slot, const TnConv cv) {
  const float* __restrict__ B = reinterpret_cast<const float*>(Bv);
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * TRIMG];
  unsigned char* ia = lds;
  unsigned char* ib = lds + 3 * TRIMG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  // The ncg workgroups of a slot read the same B rows: keep them on one XCD (workgroup b runs on XCD b % 8) so
  // that the tile comes out of that XCD's L2 instead of HBM ncg times.  nslot % 8 == 0.
  const int xcd = blockIdx.x % kXcds, jj = blockIdx.x / kXcds;
  // class = (column group of A, column group of B); ncgb = 1, ldb = 128 is the original one-group-of-B case
  const int ncls = ncg * ncgb;
  const int cls = jj % ncls, slot = xcd * (nslot / kXcds) + jj / ncls;
  const int cg = cls / ncgb, cgb = cls - cg * ncgb;
  const int64_t ntiles = (M + TRR - 1) / TRR;
  const int64_t tb0 = (int64_t)slot * tiles_per_slot;
  const int64_t tb1 = tb0 + tiles_per_slot < ntiles ? tb0 + tiles_per_slot : ntiles;
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;
  const int64_t Mlast = M - 1;
  floatx16 tn[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) tn[a][b][e] = 0.f;
  double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;     // column sums of A for columns lc4 .. lc4+3
  float4 pa[4], pb[4];
  float4 pt[CONV ? 4 : 1];                        // CONV: the T rows and the CSR pointers the degrees come from
  int dg[CONV ? 4 : 1][2];
  float4 cc = f4(0.f), m1 = f4(0.f), m2 = f4(0.f);
  if constexpr (CONV) {
    cc = ld4(cv.gamma_e + lc4) * ld4(cv.stat_e + SW + lc4);
    m1 = ld4(cv.bstat_e + lc4);
    m2 = ld4(cv.bstat_e + SW + lc4);
  }
  const float* const Ucg = CONV ? cv.U[cg] : nullptr;
  const float* const Tcg = CONV ? cv.T[cg] : nullptr;
  const int64_t pcg = CONV ? cv.pitch[cg] : 0;
  const int32_t* const dptr = CONV ? cv.ptr[cg] : nullptr;
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = (tile < tb1 ? tile : tb1 - 1) * TRR;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int64_t r = r0 + lrow + 8 * it;
      r = r < Mlast ? r : Mlast;
      if constexpr (CONV) {
        pa[it] = ld4_nt(Ucg + r * pcg + lc4);
        pt[it] = ld4_nt(Tcg + r * pcg + lc4);
        dg[it][0] = dptr[r];
        dg[it][1] = dptr[r + 1];
      } else {
        pa[it] = ld4_nt(A + r * lda + cg * SW + lc4);
      }
      pb[it] = ld4(B + r * ldb + cgb * SW + lc4);        // shared by the workgroups of the slot through L2
    }
  };
  constexpr int kNoRef = -100000;            // H2: no row met yet (every row of the first tile lies above it)
  int Rref = kNoRef;                         // H2: reference exponent (unbiased EA + EB), the same in every thread
  int* const rexp = reinterpret_cast<int*>(lds + 2 * TRIMG);   // H2: [32] EA + EB of the tile's rows, [32] = "a row lies above R + 10"
                                                               //     (the third A image of the bf16x3 layout, unused here)
  if constexpr (H2) { if (tid == 0) rexp[TRR] = 0; }
  if (tb0 < tb1) prefetch(tb0);
  for (int64_t tile = tb0; tile < tb1; ++tile) {
    const int64_t r0 = tile * TRR;
    __syncthreads();                          // the previous tile's fragment reads are done
    int rowa[H2 ? 4 : 1];                     // H2: EA of this thread's rows (a tile staged again needs them)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if constexpr (CONV) {
        const float deg = (float)(dg[it][1] - dg[it][0]);
        pa[it] = cc * (pa[it] - m1 * deg - m2 * pt[it]);                      // node_bgrad_k's expression
        int64_t r = r0 + lrow + 8 * it;                                      // rows past the end were loaded from the last row:
        r = r < Mlast ? r : Mlast;                                           // they rewrite its value
        st4(cv.Xw + r * cv.ldxw + cg * SW + lc4, pa[it]);
      }
      if (r0 + lrow + 8 * it >= M) pa[it] = f4(0.f);      // rows past the end contribute nothing
      c0 += (double)pa[it].x; c1 += (double)pa[it].y; c2 += (double)pa[it].z; c3 += (double)pa[it].w;
      if constexpr (H2) {
        const int ea = h2_row_exp(pa[it]) - 127, eb = h2_row_exp(pb[it]) - 127;
        rowa[it] = ea;
        if ((tid & 31) == 0) {
          rexp[lrow + 8 * it] = ea + eb;
          if (ea + eb > Rref + 10) rexp[TRR] = 1;
        }
        const int da = h2_tn_da(Rref, ea + eb);
        simg_stage_h2(ia, TRIMG, lrow + 8 * it, lc4, pa[it], pow2_biased(127 + 4 - ea - da));
        simg_stage_h2(ib, TRIMG, lrow + 8 * it, lc4, pb[it], pow2_biased(127 + 4 - Rref + ea + da));
      } else {
        simg_stage(ia, TRIMG, lrow + 8 * it, lc4, pa[it]);
        simg_stage(ib, TRIMG, lrow + 8 * it, lc4, pb[it]);
      }
    }
    __syncthreads();
    int Rnext = Rref;
    if constexpr (H2) {
      // the tile's largest EA + EB: lane l holds row l & 31's, the maximum over the wave is the same number in every wave
      const int m = row32_max_i32(rexp[lane & 31]);
      const bool redo = rexp[TRR] != 0;       // (uniform: written before the barrier)
      if (redo) {                             // rare: the tile is staged again in the unit of its own maximum
        const float f = pow2_biased(127 + (Rref == kNoRef ? -127 : Rref - m));      // the accumulators so far move to it (<= 1, exact)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) tn[a][b][e] *= f;
        Rref = m;
        __syncthreads();                      // every wave has read the flag and the exponents
        if (tid == 0) rexp[TRR] = 0;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int da = h2_tn_da(Rref, rexp[lrow + 8 * it]);
          simg_stage_h2(ia, TRIMG, lrow + 8 * it, lc4, pa[it], pow2_biased(127 + 4 - rowa[it] - da));
          simg_stage_h2(ib, TRIMG, lrow + 8 * it, lc4, pb[it], pow2_biased(127 + 4 - Rref + rowa[it] + da));
        }
        __syncthreads();
      }
      Rnext = m > Rref ? m : Rref;
    }
    prefetch(tile + 1);                       // in flight under the MFMAs
    if constexpr (H2) {
#pragma unroll
      for (int kc = 0; kc < TRR / 16; ++kc) {
        h16x8 a[2][2], b[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            a[x][s] = simg_col_frag_h(ia + s * TRIMG, 16 * kc, (2 * wn + x) * 32, lane);
            b[x][s] = simg_col_frag_h(ib + s * TRIMG, 16 * kc, (2 * wc + x) * 32, lane);
          }
#pragma unroll
        for (int t = 0; t < 3; ++t) {         // lo*hi, hi*lo, hi*hi
          const int sa = t == 0 ? 1 : 0, sb = t == 1 ? 1 : 0;
          mfh(tn[0][0], a[0][sa], b[0][sb]);
          mfh(tn[0][1], a[0][sa], b[1][sb]);
          mfh(tn[1][0], a[1][sa], b[0][sb]);
          mfh(tn[1][1], a[1][sa], b[1][sb]);
        }
      }
      if (Rnext != Rref) {                    // the next tile is staged in the unit of the new reference
        const float f = pow2_biased(127 + Rref - Rnext);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) tn[a][b][e] *= f;
        Rref = Rnext;
      }
    } else {
#pragma unroll
    for (int kc = 0; kc < TRR / 16; ++kc) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          a[x][s] = simg_col_frag(ia + s * TRIMG, 16 * kc, (2 * wn + x) * 32, lane);
          b[x][s] = simg_col_frag(ib + s * TRIMG, 16 * kc, (2 * wc + x) * 32, lane);
        }
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        mfb16(tn[0][0], a[0][b3_pa(t)], b[0][b3_pb(t)]);
        mfb16(tn[0][1], a[0][b3_pa(t)], b[1][b3_pb(t)]);
        mfb16(tn[1][0], a[1][b3_pa(t)], b[0][b3_pb(t)]);
        mfb16(tn[1][1], a[1][b3_pa(t)], b[1][b3_pb(t)]);
      }
    }
    }
  }
  if constexpr (H2) {                         // out of the products' unit 2^(8 - R)
    const float f = Rref == kNoRef ? 0.f : pow2_biased(127 + Rref - 8);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) tn[a][b][e] *= f;
  }
  // C / D layout of the 32 x 32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float* sl = slab + (size_t)(cls * nslot + slot) * SW * SW;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = (2 * wn + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
        sl[m * SW + (2 * wc + b) * 32 + li] = tn[a][b][e];
      }
  __syncthreads();
  double* red = reinterpret_cast<double*>(lds);          // 8 row slots x 128 columns
  red[lrow * SW + lc4 + 0] = c0;
  red[lrow * SW + lc4 + 1] = c1;
  red[lrow * SW + lc4 + 2] = c2;
  red[lrow * SW + lc4 + 3] = c3;
  __syncthreads();
  if (tid < SW) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k * SW + tid];
    partials[(size_t)(cls * nslot + slot) * SW + tid] = s;
  }
}

// ------------------------------------------------------------------------------------------
// fused edge backward, split mode (autograd of gated_gcn_full.py:113,:122), second generation:
//   gt = gamma*rstd*(gu - m1 - that*m2), gu = ge*[t*scale+shift > 0];   ge_out = ge + gt W3;
//   gW3 += gt^T e_in;   gb3 += sum gt
// ONE swizzled row-major image set per operand feeds both contractions: TN (this wave's 64 x 64 block of gW3) by
// transpose reads into v_mfma_f32_32x32x16_bf16 (the tile's 16 rows are its contraction chunk), NN (16 rows x this
// wave's 32 columns of gt W3) by ds_read_b128 into v_mfma_f32_16x16x32_bf16, W3 stationary in 96 VGPRs.
// 16-ROW tiles: with the weight fragments (96) and the TN accumulators (64) pinned, a 32-row tile's prefetch
// (48 registers) no longer fits 256 registers next to the phase-0 temporaries; 16 rows need 24, and the kernel
// runs TWO workgroups per CU (45 KB of LDS each) -- one's gt prologue / split staging / stores and HBM waits
// under the other's MFMAs.  Round 1's split kernel kept a second, register-transposed image set (159 KB, one
// workgroup per CU, matrix pipe 39 % busy).
// ------------------------------------------------------------------------------------------
constexpr int ER = 16;                      // rows per tile
constexpr int EIMG = ER * SPITCH;           // bytes per image (4 KB)
constexpr int EOP = SW + 4;                 // fp32 output image pitch (floats)

// this wave's 32 output columns of W3 as B fragments of v_mfma_f32_16x16x32_bf16: [nb][kc][hi/mid/lo] = 96 VGPRs
struct W3Frag { bf16x8 w[2][SW / 32][3]; };

// Wp[cb (16-column block)][kc][s][lane] (bf16x8): element j of lane (n = l & 15, g = l >> 4) =
// part s of W[(32 kc + 8 g + j) * ld + 16 cb + n]      (y = x W: contraction index first)
__global__ void pack_w3_nn16_k(const float* __restrict__ W, int64_t ld, bf16x8* __restrict__ Wp) {
  const int total = (SW / 16) * (SW / 32) * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, kc = (idx >> 6) % (SW / 32), cb = idx / (64 * (SW / 32));
    const int n = lane & 15, g = lane >> 4;
    bf16x8 hi, mid, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = W[(int64_t)(32 * kc + 8 * g + j) * ld + 16 * cb + n];
      const __bf16 h = (__bf16)x;
      const float r1 = x - (float)h;
      const __bf16 m = (__bf16)r1;
      hi[j] = h;
      mid[j] = m;
      lo[j] = (__bf16)(r1 - (float)m);
    }
    bf16x8* o = Wp + ((int64_t)(cb * (SW / 32) + kc) * 3) * 64 + lane;
    o[0] = hi;
    o[64] = mid;
    o[128] = lo;
  }
}

// f16x2 form (edge_bwd_chain_k<..., H2>): Wp2[cb][kc][s = hi/lo][lane] (h16x8) of W s_n, s_n = the power of two that puts the largest
// magnitude of output column n at 2^14 (h2_scale); 1 / s_n of the 128 columns as floats behind the 64 KB of fragments.  One workgroup per
// 16-column block cb (grid = SW / 16): it takes its columns' largest magnitudes itself (16 columns x 16 contraction slices through LDS;
// a launch of its own until the end of round 5) and packs.
constexpr size_t kW2Nn16FragBytes = (size_t)(SW / 16) * (SW / 32) * 2 * 64 * 16;
__global__ __launch_bounds__(256) void pack_w2_nn16_k(const float* __restrict__ W, int64_t ld, unsigned char* __restrict__ Wp) {
  __shared__ float part[16][16];
  const int cb = blockIdx.x;
  {
    const int c = threadIdx.x & 15, ks = threadIdx.x >> 4;
    float m = 0.f;
    for (int k = ks; k < SW; k += 16) m = fmaxf(m, fabsf(W[(int64_t)k * ld + 16 * cb + c]));
    part[ks][c] = m;
  }
  __syncthreads();
  static_assert((SW / 32) * 64 == 256, "one pass of 256 threads packs a 16-column block");
  const int lane = threadIdx.x & 63, kc = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4;
  float m = part[0][n];
#pragma unroll
  for (int q = 1; q < 16; ++q) m = fmaxf(m, part[q][n]);
  float sc, inv;
  h2_scale(__float_as_uint(m), sc, inv);
  h16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = W[(int64_t)(32 * kc + 8 * g + j) * ld + 16 * cb + n] * sc;
    const _Float16 h = (_Float16)x;
    hi[j] = h;
    lo[j] = (_Float16)(x - (float)h);
  }
  h16x8* o = reinterpret_cast<h16x8*>(Wp) + ((int64_t)(cb * (SW / 32) + kc) * 2) * 64 + lane;
  o[0] = hi;
  o[64] = lo;
  if (kc == 0 && g == 0) reinterpret_cast<float*>(Wp + kW2Nn16FragBytes)[16 * cb + n] = inv;
}

template <bool FULL>
struct tile_tag { static constexpr bool full = FULL; };

// VAR bit 0: residual ge rows and the fp64 column sums wait in LDS (1) or in registers (0);
//     bit 1: the prefetch is pinned right behind the first barrier (1) or left to hipcc's scheduler (0)
//     bit 2 (round 6): gt is GIVEN -- `t` points at the [E,128] gt rows another kernel formed (the LayerNorm backward: its gt comes out
//           of the by-destination pass, row statistics and all), stat / bstat / gamma are not read; everything else -- gW3 += gt^T e_in,
//           gb3 += sum gt, ge_out = ge + gt W3 -- as before: the LayerNorm mode's two generic GEMMs + column sum in one pass
// H2 (round 5): the f16x2 form, as in edge_bwd_chain_k<..., H2> (the NN product per row of gt and column of W3, the TN product through the
// workgroup's reference exponent; a thread owns two rows of a tile here)
template <int VAR, bool H2 = false>
__global__ __launch_bounds__(kBlock, 2) void edge_bwd_tr_k(
    int64_t E, const float* ge, float* ge_out, const float* __restrict__ t, const float* __restrict__ e_in,
    const float* __restrict__ stat, const float* __restrict__ bstat, const float* __restrict__ gamma,
    const bf16x8* __restrict__ Wp,                   // W3 packed by pack_w3_nn16_k
    float* __restrict__ slab,                        // [grid][128][128] partial gW3
    double* __restrict__ partials,                   // [grid][128]: per-workgroup column sums of gt
    int64_t tiles_per_block) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * EIMG + ER * EOP * 4 + 7 * SW * 4 + kBlock * 32];
  unsigned char* ig = lds;                                               // gt images
  unsigned char* ie = lds + 3 * EIMG;                                    // e_in images
  float* og = reinterpret_cast<float*>(lds + 6 * EIMG);                  // residual ge rows, then ge + gt W3 (row layout)
  float* cs = og + ER * EOP;                                             // mu, rstd, scale, shift, m1, m2, c = gamma*rstd
  // this thread's four fp64 column sums of gt live in LDS, like the residual rows: the weight fragments (96
  // registers) and the TN accumulators (64) leave no room for them beside the prefetched rows
  double* cgs = reinterpret_cast<double*>(cs + 7 * SW) + 4 * threadIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 1, wc = wave & 1;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + ER - 1) / ER;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = tb0 + tiles_per_block < ntiles ? tb0 + tiles_per_block : ntiles;
  const int64_t nfull = tb1 < E / ER ? tb1 : E / ER;        // tiles [tb0, nfull) are full
  const int lrow = tid >> 5, lc4 = (tid & 31) * 4;           // rows lrow and lrow + 8 of a tile
  const int64_t Elast = E - 1;
  constexpr bool GIVEN = (VAR & 4) != 0;
  if constexpr (!GIVEN) {
    for (int c = tid; c < SW; c += kBlock) {
      cs[c] = stat[c];
      cs[SW + c] = stat[SW + c];
      cs[2 * SW + c] = stat[2 * SW + c];
      cs[3 * SW + c] = stat[3 * SW + c];
      cs[4 * SW + c] = bstat[c];
      cs[5 * SW + c] = bstat[SW + c];
      cs[6 * SW + c] = gamma[c] * stat[SW + c];
    }
  }
  W3Frag wf;
  h16x8 wfh[H2 ? 2 : 1][H2 ? SW / 32 : 1][2];       // H2: [nb][kc][hi/lo] = 64 VGPRs
  float cinv[2] = {1.f, 1.f};                       //     1 / s_n of this lane's two output columns
  if constexpr (!H2) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const bf16x8* p = Wp + ((int64_t)(2 * wave + nb) * (SW / 32) * 3) * 64 + lane;
#pragma unroll
      for (int kc = 0; kc < SW / 32; ++kc)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) wf.w[nb][kc][s_] = p[(kc * 3 + s_) * 64];
    }
  } else {
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(Wp);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const h16x8* p = reinterpret_cast<const h16x8*>(wp) + ((int64_t)(2 * wave + nb) * (SW / 32) * 2) * 64 + lane;
#pragma unroll
      for (int kc = 0; kc < SW / 32; ++kc)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) wfh[nb][kc][s_] = p[(kc * 2 + s_) * 64];
      cinv[nb] = reinterpret_cast<const float*>(wp + kW2Nn16FragBytes)[(2 * wave + nb) * 16 + (lane & 15)];
    }
  }
  int* const rexp = reinterpret_cast<int*>(lds + 2 * EIMG);          // H2: as in edge_bwd_chain_k (the unused third gt image)
  float* const rinv = reinterpret_cast<float*>(lds + 2 * EIMG) + ER;
  constexpr int kNoRef = -100000;
  int Rref = kNoRef;
  if constexpr (H2) { if (threadIdx.x == 0) rexp[2 * ER] = 0; }
  floatx16 tn[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) tn[a][b][e] = 0.f;
  constexpr bool STASH = VAR & 1, PIN = VAR & 2;
  double cg0 = 0.0, cg1 = 0.0, cg2 = 0.0, cg3 = 0.0;        // column sums of gt for columns lc4 .. lc4+3 (!STASH)
  cgs[0] = 0.0; cgs[1] = 0.0; cgs[2] = 0.0; cgs[3] = 0.0;   // ... (STASH)
  const int trq0 = simg_tr_base(lane, 0), trq1 = simg_tr_base(lane, 1);   // transpose-read bases (gnm_tr.h)
  // NN A fragment (16x16x32): lane (i = l & 15, g = l >> 4) reads slot 4 kc + g of row i:
  //   i * SPITCH + (((kc ^ (i & 3)) << 2 | (g ^ f(i >> 2))) << 4)  =  nnb ^ (kc << 6)
  const int ni = lane & 15, ng = lane >> 4;
  const int nnb = ni * SPITCH + ((((ni & 3) << 2) | (ng ^ (swz(ni) & 3))) << 4);
  __syncthreads();

  float4 pg[2], pt[2], pe_[2];
  // wave-uniform tile base (scalar registers) + a 32-bit lane offset: no 64-bit address registers are kept;
  // rows past the end are clamped to the last valid row (branch-free, never stored)
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * ER;
    const int64_t left = E - r0;                                   // >= 1
    const int last = left < ER ? (int)left - 1 : ER - 1;
    const float* bg = ge + r0 * SW;
    const float* bt = t + r0 * SW;
    const float* be = e_in + r0 * SW;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rl = lrow + 8 * it;
      const int o = (rl < last ? rl : last) * SW + lc4;
      pg[it] = ld4(bg + o);
      pt[it] = ld4(bt + o);
      pe_[it] = ld4(be + o);
    }
  };
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::full;
    const int64_t r0 = tile * ER;
    // ---- phase 0: gt tile and e_in tile -> split images; residual ge rows -> og ----
    float4 gk[2];
    float4 gt_keep[H2 ? 2 : 1], ev_keep[H2 ? 2 : 1];      // H2: this thread's rows until the tile is known not to be staged again
    int row_ea[H2 ? 2 : 1];
    {
      double c0 = cg0, c1 = cg1, c2 = cg2, c3 = cg3;
      if (STASH) { c0 = cgs[0]; c1 = cgs[1]; c2 = cgs[2]; c3 = cgs[3]; }
      float4 mu = f4(0.f), rs = f4(0.f), sc = f4(0.f), sh = f4(0.f), m1 = f4(0.f), m2 = f4(0.f), cc = f4(0.f);
      if constexpr (!GIVEN) {
        mu = ld4(cs + lc4); rs = ld4(cs + SW + lc4); sc = ld4(cs + 2 * SW + lc4);
        sh = ld4(cs + 3 * SW + lc4); m1 = ld4(cs + 4 * SW + lc4); m2 = ld4(cs + 5 * SW + lc4);
        cc = ld4(cs + 6 * SW + lc4);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = lrow + 8 * it;
        const bool ok = FULL || (r0 + row < E);
        if (STASH) st4(og + row * EOP + lc4, pg[it]);   // this thread read exactly these og elements in the previous epilogue
        else gk[it] = pg[it];
        float4 gt;
        if constexpr (GIVEN) {
          gt = pt[it];
        } else {
          const float4 gu = gate4(fma4(pt[it], sc, sh), pg[it]);
          gt = cc * (gu - m1 - ((pt[it] - mu) * rs) * m2);
        }
        float4 ev = pe_[it];
        if (!ok) { gt = f4(0.f); ev = f4(0.f); }
        c0 += (double)gt.x; c1 += (double)gt.y; c2 += (double)gt.z; c3 += (double)gt.w;
        if constexpr (H2) {
          gt_keep[it] = gt;
          ev_keep[it] = ev;
          const int ea = h2_row_exp(gt) - 127, eb = h2_row_exp(ev) - 127;
          const int da = h2_tn_da(Rref, ea + eb);
          row_ea[it] = ea;
          if ((tid & 31) == 0) {
            rexp[row] = ea + eb;
            rinv[row] = pow2_biased(127 + ea + da - 4);
            if (ea + eb > Rref + 10) rexp[2 * ER] = 1;
          }
          simg_stage_h2(ig, EIMG, row, lc4, gt, pow2_biased(127 + 4 - ea - da));
          simg_stage_h2(ie, EIMG, row, lc4, ev, pow2_biased(127 + 4 - Rref + ea + da));
        } else {
          simg_stage(ig, EIMG, row, lc4, gt);
          simg_stage(ie, EIMG, row, lc4, ev);
        }
      }
      if (STASH) { cgs[0] = c0; cgs[1] = c1; cgs[2] = c2; cgs[3] = c3; }
      else { cg0 = c0; cg1 = c1; cg2 = c2; cg3 = c3; }
    }
    __syncthreads();   // images and residual rows ready
    int Rnext = Rref;
    if constexpr (H2) {
      const int m = row16_max_i32(rexp[lane & (ER - 1)]);
      if (rexp[2 * ER] != 0) {                    // rare: the tile is staged again in the unit of its own maximum
        const float f = pow2_biased(127 + (Rref == kNoRef ? -127 : Rref - m));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) tn[a][b][e] *= f;
        Rref = m;
        __syncthreads();
        if (tid == 0) rexp[2 * ER] = 0;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = lrow + 8 * it;
          const int da = h2_tn_da(Rref, rexp[row]);
          if ((tid & 31) == 0) rinv[row] = pow2_biased(127 + row_ea[it] + da - 4);
          simg_stage_h2(ig, EIMG, row, lc4, gt_keep[it], pow2_biased(127 + 4 - row_ea[it] - da));
          simg_stage_h2(ie, EIMG, row, lc4, ev_keep[it], pow2_biased(127 + 4 - Rref + row_ea[it] + da));
        }
        __syncthreads();
      }
      Rnext = m > Rref ? m : Rref;
    }
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);   // in flight under the MFMAs, the epilogue and the partner workgroup
    if (PIN) __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE (hipcc otherwise sinks them behind the MFMAs)
    // ---- TN: gW3[n][c] += sum_rows gt[row][n] e_in[row][c], this wave's 64 x 64 block (transpose reads) ----
    if constexpr (H2) {
      int tr0 = trq0, tr1 = trq1;
      asm volatile("" : "+v"(tr0), "+v"(tr1));
      h16x8 a[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
          a[x][s_] = __builtin_bit_cast(h16x8, simg_col_frag2(ig + s_ * EIMG, tr0 ^ ((2 * wn + x) << 6), tr1 ^ ((2 * wn + x) << 6)));
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const h16x8 b0 = __builtin_bit_cast(h16x8, simg_col_frag2(ie + sb * EIMG, tr0 ^ ((2 * wc) << 6), tr1 ^ ((2 * wc) << 6)));
        const h16x8 b1 = __builtin_bit_cast(h16x8, simg_col_frag2(ie + sb * EIMG, tr0 ^ ((2 * wc + 1) << 6), tr1 ^ ((2 * wc + 1) << 6)));
#pragma unroll
        for (int sa = 0; sa < 2; ++sa) {
          if (sa + sb > 1) continue;               // lo*lo is dropped
          mfh(tn[0][0], a[0][sa], b0);
          mfh(tn[0][1], a[0][sa], b1);
          mfh(tn[1][0], a[1][sa], b0);
          mfh(tn[1][1], a[1][sa], b1);
        }
      }
    } else {
      // two lane-constant bases; the column-block term is a wave-uniform XOR applied at the read.  The empty asm
      // keeps hipcc from hoisting the eight XORed addresses into registers that live across the whole loop.
      int tr0 = trq0, tr1 = trq1;
      asm volatile("" : "+v"(tr0), "+v"(tr1));
      bf16x8 a[2][3];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
          a[x][s_] = simg_col_frag2(ig + s_ * EIMG, tr0 ^ ((2 * wn + x) << 6), tr1 ^ ((2 * wn + x) << 6));
#pragma unroll
      for (int sb = 0; sb < 3; ++sb) {             // B part by B part: only two B fragments live at a time
        const bf16x8 b0 = simg_col_frag2(ie + sb * EIMG, tr0 ^ ((2 * wc) << 6), tr1 ^ ((2 * wc) << 6));
        const bf16x8 b1 = simg_col_frag2(ie + sb * EIMG, tr0 ^ ((2 * wc + 1) << 6), tr1 ^ ((2 * wc + 1) << 6));
#pragma unroll
        for (int sa = 0; sa < 3; ++sa) {
          if (sa + sb > 2) continue;               // the three products below 2^-24 are dropped
          mfb16(tn[0][0], a[0][sa], b0);
          mfb16(tn[0][1], a[0][sa], b1);
          mfb16(tn[1][0], a[1][sa], b0);
          mfb16(tn[1][1], a[1][sa], b1);
        }
      }
    }
    // ---- NN: acc = gt W3 (16 rows x this wave's 2 x 16 columns) ----
    floatx4_acc acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[nb][e] = 0.f;
    if constexpr (H2) {
#pragma unroll
      for (int kc = 0; kc < SW / 32; ++kc) {
        h16x8 a[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) a[s_] = *reinterpret_cast<const h16x8*>(ig + s_ * EIMG + (nnb ^ (kc << 6)));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          mfh16s(acc[nb], a[1], wfh[nb][kc][0]);
          mfh16s(acc[nb], a[0], wfh[nb][kc][1]);
          mfh16s(acc[nb], a[0], wfh[nb][kc][0]);
        }
      }
      const float4 ri = ld4(rinv + 4 * ng);
      const float rf[4] = {ri.x, ri.y, ri.z, ri.w};
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[nb][e] *= rf[e] * cinv[nb];          // exact factors
      if (Rnext != Rref) {
        const float f = pow2_biased(127 + Rref - Rnext);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) tn[a][b][e] *= f;
        Rref = Rnext;
      }
    } else {
#pragma unroll
    for (int kc = 0; kc < SW / 32; ++kc) {
      bf16x8 a[3];
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) a[s_] = *reinterpret_cast<const bf16x8*>(ig + s_ * EIMG + (nnb ^ (kc << 6)));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        mfb16s(acc[nb], a[2], wf.w[nb][kc][0]);
        mfb16s(acc[nb], a[0], wf.w[nb][kc][2]);
        mfb16s(acc[nb], a[1], wf.w[nb][kc][1]);
        mfb16s(acc[nb], a[1], wf.w[nb][kc][0]);
        mfb16s(acc[nb], a[0], wf.w[nb][kc][1]);
        mfb16s(acc[nb], a[0], wf.w[nb][kc][0]);
      }
    }
    }
    // C / D of the 16 x 16 MFMA: column = lane & 15, row = 4 (lane >> 4) + e; every og element is touched by
    // exactly one lane: ge_in = ge + gt W3 is formed in place
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* o = og + (4 * ng + e) * EOP + wave * 32 + nb * 16 + ni;
        *o = STASH ? *o + acc[nb][e] : acc[nb][e];
      }
    __syncthreads();   // og complete; every wave is done with the images (the next phase 0 overwrites them)
    // ---- ge_in = ge + gt W3, whole 512-byte rows, one float4 per lane ----
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = lrow + 8 * it;
      const int64_t grow = r0 + row;
      if (FULL || grow < E) st4(ge_out + grow * SW + lc4, STASH ? ld4(og + row * EOP + lc4) : ld4(og + row * EOP + lc4) + gk[it]);
    }
  };

  if (tb0 < tb1) prefetch(tb0);
  // throw-away stores behind the first prefetch make the loop-entry scoreboard equal to the back edge's (counted
  // vmcnt instead of vmcnt(0): see rowtile_nt_k in gnm_fused.hip), into a slab behind the gridDim.x result slabs
#pragma unroll
  for (int it = 0; it < 2; ++it) st4(slab + (size_t)(gridDim.x + chunk) * SW * SW + (lrow + 8 * it) * SW + lc4, f4(0.f));
  for (int64_t tile = tb0; tile < nfull; ++tile) body(tile_tag<true>{}, tile);
  if (nfull < tb1 && nfull >= tb0) body(tile_tag<false>{}, nfull);

  float* sl = slab + (size_t)chunk * SW * SW;
  const float unit = !H2 ? 1.f : Rref == kNoRef ? 0.f : pow2_biased(127 + Rref - 8);       // H2: out of the products' unit 2^(8 - R)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = (2 * wn + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
        sl[m * SW + (2 * wc + b) * 32 + li] = H2 ? tn[a][b][e] * unit : tn[a][b][e];
      }
  __syncthreads();
  if (!STASH) { cgs[0] = cg0; cgs[1] = cg1; cgs[2] = cg2; cgs[3] = cg3; }
  __syncthreads();
  if (tid < SW) {       // column c is held by the threads 32 k + c/4 (k = 0..7), entry c % 4
    const double* all = reinterpret_cast<const double*>(cs + 7 * SW);
    double s_ = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s_ += all[4 * (32 * k + (tid >> 2)) + (tid & 3)];
    partials[(size_t)chunk * SW + tid] = s_;
  }
}

int eb_variant();   // gnm_fused.hip
size_t edge_bwd_tr_pack_bytes() { return (size_t)(SW / 16) * (SW / 32) * 3 * 64 * sizeof(bf16x8); }
// returns the grid size (= number of slabs / partial rows written)
int edge_bwd_tr_launch(int64_t E, const float* ge, float* ge_out, const float* t, const float* e_in, const float* stat_e,
                       const float* bstat_e, const float* gamma_e, const float* W3, void* wpack, float* slab,
                       double* partials, hipStream_t st, bool h2, bool given) {
  if (h2) {
    hipLaunchKernelGGL(pack_w2_nn16_k, dim3(SW / 16), dim3(256), 0, st, W3, (int64_t)SW, (unsigned char*)wpack);
    const int64_t ntiles = (E + ER - 1) / ER;
    const int grid = persistent_grid(ntiles, 16, occ_blocks<edge_bwd_tr_k<3, true>>());
    if (given)
      hipLaunchKernelGGL((edge_bwd_tr_k<7, true>), dim3(grid), dim3(kBlock), 0, st, E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e,
                         (const bf16x8*)wpack, slab, partials, (ntiles + grid - 1) / grid);
    else
      hipLaunchKernelGGL((edge_bwd_tr_k<3, true>), dim3(grid), dim3(kBlock), 0, st, E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e,
                         (const bf16x8*)wpack, slab, partials, (ntiles + grid - 1) / grid);
    return grid;
  }
  hipLaunchKernelGGL(pack_w3_nn16_k, dim3(8), dim3(256), 0, st, W3, (int64_t)SW, (bf16x8*)wpack);
  const int64_t ntiles = (E + ER - 1) / ER;
  if (given) {
    const int grid = persistent_grid(ntiles, 16, occ_blocks<edge_bwd_tr_k<3>>());
    hipLaunchKernelGGL(edge_bwd_tr_k<7>, dim3(grid), dim3(kBlock), 0, st, E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e,
                       (const bf16x8*)wpack, slab, partials, (ntiles + grid - 1) / grid);
    return grid;
  }
  const int var = eb_variant();      // 1: LDS stash + pinned prefetch (default); 2: registers, unpinned (A/B: within 1 % of each other)
  const int grid = persistent_grid(ntiles, 16, occ_blocks<edge_bwd_tr_k<3>>());
#define GNM_EB_LAUNCH(V)                                                                                               \
  hipLaunchKernelGGL(edge_bwd_tr_k<V>, dim3(grid), dim3(kBlock), 0, st, E, ge, ge_out, t, e_in, stat_e, bstat_e, gamma_e, \
                     (const bf16x8*)wpack, slab, partials, (ntiles + grid - 1) / grid)
  if (var == 2) GNM_EB_LAUNCH(0);
  else GNM_EB_LAUNCH(3);
#undef GNM_EB_LAUNCH
  return grid;
}

// ------------------------------------------------------------------------------------------
// CHAINED edge backward: the fused edge backward of layer i and the by-destination backward pass of layer i-1 in
// one sweep over the destination-sorted edge rows.
//
// Layer i's fused kernel ends by writing ge_in(i) = d loss / d e_out(i-1); layer i-1's by-destination pass begins
// by reading exactly those rows, in the same order, together with e_out(i-1) -- which the fused kernel has just
// read as e_in(i).  Run back to back they move 4 + 4 [E,H] streams; chained, the rows stay on chip:
//     read ge'(i), t(i), e_out(i-1), t(i-1);  write ge'(i-1)            (5 streams, 3 fewer)
// and the matrix-core work (TN + NN, what bounds the fused kernel) runs under the gather / normalise work of the
// by-destination pass (what that pass waits on) instead of beside a half idle memory system.
// Per 16-row tile: phases 0 / TN / NN of edge_bwd_tr_k; then each thread takes its two rows of ge(i-1) = ge'(i) +
// gt W3 from the row image and does edge_bwd_dst_k's per-edge arithmetic (gated_gcn_full.py:122-130 backward)
// with the node rows gathered through isrc / idst; the per-destination sums (gA3h, Ud, Td) and the BatchNorm
// column sums of layer i-1 are taken by 256 column walkers over three [16,128] fp32 images of the per-edge
// terms, sequentially in edge order -- no atomics, deterministic.  A workgroup owns a contiguous range of
// destination NODES (its rows are in_ptr[v0] .. in_ptr[v1]), so every segment sum is complete inside it.
// ------------------------------------------------------------------------------------------

// One 512-thread workgroup per CU (8 waves, two per SIMD): thread (row = tid >> 5, 4 columns) owns ONE row of the
// tile in phase 0 and in the by-destination arithmetic; wave w owns output columns 16w .. 16w+15 of gt W3 (48
// weight-fragment registers) and the 64 x 32 block (w >> 2, w & 3) of gW3 (32 accumulator registers) -- half the
// pinned registers of the 4-wave kernel, which leaves room for the software pipeline this kernel lives on: the
// four row streams AND the six gathered node rows of the NEXT tile are requested a tile ahead, so that no phase
// waits on memory (the eight waves share every barrier, there is no second workgroup to hide a wait).
constexpr int CT = 512;                    // threads per workgroup
constexpr int WG_ = 4;                     // rows per LDS read group of the column walk (8 and 16 measured the same)
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
constexpr int CH_LDS = 6 * EIMG + ER * EOP * 4 + 7 * SW * 4 + 4 * SW * 4 + 5 * ER * SW * 4 + 3 * 2 * ER * 4;
// two-sided sweep: + ring of the plan words, the sigma * Qf[dst] image, three accumulator slot arrays (gA2h, Us, Ts)
constexpr int CH_LDS_SRC = CH_LDS + 3 * ER * 4 + ER * SW * 4 + 3 * kSweepSlots * SW * 4;
static_assert(ER == kSweepTileRows, "the sweep plan is built for the chained kernel's tile");
static_assert(CH_LDS % 16 == 0 && CH_LDS_SRC <= 160 * 1024, "LDS layout");

struct W3Frag16 { bf16x8 w[SW / 32][3]; };     // 16 output columns of W3: [kc][hi/mid/lo] = 48 VGPRs

// ABL (timing experiments only, results wrong when non-zero): bit 0 no column walk, bit 1 no gather arithmetic, bit 2 no MFMAs
// SRC: the two-sided sweep.  The by-SOURCE sums of layer i-1 (gA2h = sum_out sigma*Qf[dst], Us = sum_out gu, Ts = sum_out
// that: what edge_bwd_src_k re-read three [E,H] streams for) are formed here too, from the per-edge terms that are in LDS
// anyway, by the waves the column walk leaves idle (3-7): row q of the tile is served by one half-wave, which -- if the
// row LEADS its source in this tile (sweep plan, gnm_graph_build_sweep_plan) -- adds up the tile's rows of that source, joins
// the sum carried in the source's accumulator slot and either parks it there again or, in the source's last tile, stores
// it.  One owner per source and tile, fixed order: deterministic, no atomics.  The stores are unconditional buffer stores
// (offset out of range unless the leader closes its source), like the walkers'.
// (Taking the by-DESTINATION sums the same way -- run sums on all sixteen half-waves instead of the column walk on three waves --
// was built and measured twice this round, with data-dependent loops and with up-front predicated reads: 7.58 / 7.55 ms per
// launch against 7.15 with the walk.  The walk's three waves overlap the other five's next phase 0; run sums on every wave do
// not.  Not shipped.)
// HI = false: no layer i above (the TOP layer of the stack: ge is d loss / d e_out of layer i-1 as the predictor's backward
// left it): phase 0 only parks the rows, no gt, no MFMAs, no gW3 -- the sweep is edge_bwd_dst_k (+ the by-source sums).
// HF = the row pitch (floats) of every layer-(i-1) tensor = that layer's full width: 128, or 256 with the HI = false sweep run
// once per 128-column half (column-separable; the caller offsets every pointer by the half's first column)
// H2 (round 5): the matrix half in the f16x2 form -- the NN product scales gt per row and W3 per output column, the TN product
// keeps one unit across rows through this workgroup's reference exponent, exactly as tn_tr_k<., ., true> (see there): rows staged
// with the reference of the tiles before, the tile's own maximum taken by every wave after the staging barrier, a tile with a row
// 2^10 above the reference staged again (gt stays in registers until then, the e row is still in its prefetch registers).
// LN (round 6; HI = false, SRC, HF = 128): the LayerNorm form of the sweep without a layer above.  The row statistics are taken by the 32 lanes
// that hold the row (gnm_ln.h, the expressions of ln_edge_bwd_dst_k); gt = LNbwd(gu) is complete inside the row, so it is formed here, written
// once for the fused edge pass (gnm_edge_bwd_fused_gt) and summed by destination (the "Ud" walker -> gB2h) and by source (the "Us" run sums
// -> gB1h) directly; the "Td" / "Ts" sums do not exist; the column sums (sum gu, sum gu that) are the LayerNorm weight / bias gradients.
template <int ABL, bool SRC, bool WSKIP = true, bool HI = true, int HF = SW, bool H2 = false, bool LN = false>
__global__ __launch_bounds__(CT, 2) void edge_bwd_chain_k(const ChainArgs a) {
  static_assert(HF == SW || !HI, "the chained (matrix) half of the kernel is built for 128-wide layers only");
  static_assert(HI || !H2, "f16x2 belongs to the matrix half");
  static_assert(!LN || (SRC && HF == SW), "the LayerNorm sweep: two-sided, 128 wide");
  // LN && HI (the chained LayerNorm backward): layer i's gt is GIVEN -- t_hi points at the gt rows layer i's own sweep wrote a launch
  // earlier (LayerNorm's gt needs no global statistics), stat_hi / bstat_hi / gamma_hi are not read -- and layer i-1's gt is written
  __shared__ __attribute__((aligned(16))) unsigned char lds[(SRC ? CH_LDS_SRC : CH_LDS) + (LN ? ER * SW * 4 : 0)];
  unsigned char* ig = lds;                                               // gt images
  unsigned char* ie = lds + 3 * EIMG;                                    // e_in(i) = e_out(i-1) images
  float* og = reinterpret_cast<float*>(lds + 6 * EIMG);                  // residual ge rows, then ge + gt W3 (row layout)
  float* cs = og + ER * EOP;                                             // layer i:   mu, rstd, scale, shift, m1, m2, c
  float* cl = cs + 7 * SW;                                               // layer i-1: mu, rstd, scale, shift
  float* v1 = cl + 4 * SW;                                               // sigma * Qb[src]
  float* v2 = v1 + ER * SW;                                              // gu
  float* v3 = v2 + ER * SW;                                              // that
  float* tl = v3 + ER * SW;                                              // t(i-1) rows (written and read by the same thread)
  float* ef = tl + ER * SW;                                              // e_out(i-1) rows in fp32 (same thread writes and reads)
  int* sd = reinterpret_cast<int*>(ef + ER * SW);                        // ring of 3 tiles x [src 16 | dst 16]
  unsigned* si = reinterpret_cast<unsigned*>(sd + 3 * 2 * ER);           // SRC: ring of 3 tiles x [plan word 16]
  float* v5 = reinterpret_cast<float*>(si + 3 * ER);                     // SRC: sigma * Qf[dst]
  float* slots = v5 + ER * SW;                                           // SRC: [3 sums][kSweepSlots][128]
  float* v6 = slots + 3 * kSweepSlots * SW;                              // LN: gu (v2 holds gt there)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int wn = wave >> 2, wc = wave & 3;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * a.nodes_per_block < a.N ? (int64_t)chunk * a.nodes_per_block : a.N;
  const int64_t v1n = v0 + a.nodes_per_block < a.N ? v0 + a.nodes_per_block : a.N;
  const int64_t rb = a.in_ptr[v0], re = a.in_ptr[v1n];                   // this workgroup's rows
  const int64_t ntile = (re - rb + ER - 1) / ER;
  const int row = tid >> 5, lc4 = (tid & 31) * 4;                        // this thread's row of every tile
  for (int c = tid; c < SW; c += CT) {
    if constexpr (HI && !LN) {
      cs[c] = a.stat_hi[c];
      cs[SW + c] = a.stat_hi[SW + c];
      cs[2 * SW + c] = a.stat_hi[2 * SW + c];
      cs[3 * SW + c] = a.stat_hi[3 * SW + c];
      cs[4 * SW + c] = a.bstat_hi[c];
      cs[5 * SW + c] = a.bstat_hi[SW + c];
      cs[6 * SW + c] = a.gamma_hi[c] * a.stat_hi[SW + c];
    }
    if constexpr (LN) {
      cl[2 * SW + c] = a.ln_gamma[c];
      cl[3 * SW + c] = a.ln_beta[c];
    } else {
      cl[c] = a.stat_lo[c];
      cl[SW + c] = a.stat_lo[HF + c];
      cl[2 * SW + c] = a.stat_lo[2 * HF + c];
      cl[3 * SW + c] = a.stat_lo[3 * HF + c];
    }
  }
  W3Frag16 wf;
  h16x8 wfh[H2 ? SW / 32 : 1][2];           // H2: 16 output columns of W3 s_n: [kc][hi/lo] = 32 VGPRs
  float cinv = 1.f;                          //     1 / s_n of this lane's output column (wave * 16 + (lane & 15))
  if constexpr (HI && !H2) {
    const bf16x8* p = a.Wp + ((int64_t)wave * (SW / 32) * 3) * 64 + lane;
#pragma unroll
    for (int kc = 0; kc < SW / 32; ++kc)
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) wf.w[kc][s_] = p[(kc * 3 + s_) * 64];
  }
  if constexpr (H2) {
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.Wp);
    const h16x8* p = reinterpret_cast<const h16x8*>(wp) + ((int64_t)wave * (SW / 32) * 2) * 64 + lane;
#pragma unroll
    for (int kc = 0; kc < SW / 32; ++kc)
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) wfh[kc][s_] = p[(kc * 2 + s_) * 64];
    cinv = reinterpret_cast<const float*>(wp + kW2Nn16FragBytes)[wave * 16 + (lane & 15)];
  }
  // H2: per-tile scratch in the third gt image of the bf16x3 layout (unused here): [16] EA + EB of the rows, [16] 1 / (gt row factor),
  //     [32] = "a row lies more than 2^10 above the reference"
  int* const rexp = reinterpret_cast<int*>(lds + 2 * EIMG);
  float* const rinv = reinterpret_cast<float*>(lds + 2 * EIMG) + ER;
  constexpr int kNoRef = -100000;
  int Rref = kNoRef;
  if constexpr (H2) { if (tid == 0) rexp[2 * ER] = 0; }
  floatx16 tn[2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int e = 0; e < 16; ++e) tn[x][e] = 0.f;
  double cg0 = 0.0, cg1 = 0.0, cg2 = 0.0, cg3 = 0.0;       // column sums of gt for columns lc4 .. lc4+3
  const int trq0 = simg_tr_base(lane, 0), trq1 = simg_tr_base(lane, 1);   // transpose-read bases (gnm_tr.h)
  const int ni = lane & 15, ng = lane >> 4;
  const int nnb = ni * SPITCH + ((((ni & 3) << 2) | (ng ^ (swz(ni) & 3))) << 4);
  // column walkers: column wcol; role 0 sums sigma*Qb (-> gA3h), 1 that (-> Td), 2 gu (-> Ud), 3 the BatchNorm column
  // sums of layer i-1 (sum gu, sum gu*that, fp64)
  // column walkers: threads 0-95 (waves 0 and 1), one float4 of columns each; role = tid >> 5: 0 sums sigma*Qb
  // (-> gA3h), 1 that (-> Td), 2 gu (-> Ud).  The BatchNorm column sums of layer i-1 (sum gu, sum gu*that, fp64:
  // the expensive part) are taken beside them by threads 128-255 (waves 2 and 3), four rows of a tile each.
  // (round 3: one role per WAVE -- lanes 0-31 of waves 0, 1, 2 -- so that the walkers' output is a wave-uniform buffer
  //  resource; the BatchNorm sums moved to waves 4-7)
  const bool walker = wave < 3 && lane < 32 && !(LN && wave == 1), bnsum = tid >= 256;   // BatchNorm sums: waves 4-7, two rows of a tile each (LN: no Td)
  const int role = wave, wc4 = (tid & 31) * 4;
  const int brow = 2 * ((tid - 256) >> 5);                         // first of this thread's two rows (0, 2, .. 14)
  int cur = -1;                             // node whose segment is being summed (wave-uniform)
  float4 acc0 = f4(0.f);
  double s_gu[4] = {0.0, 0.0, 0.0, 0.0}, s_gut[4] = {0.0, 0.0, 0.0, 0.0};
  // Every walker step STORES the running sum to its node's output row (the last store of a segment holds the
  // whole sum; a node's ~5 stores meet in L2): no data-dependent branch around a memory operation, so hipcc keeps
  // COUNTED vmcnt waits for the prefetched rows (a store under such a branch costs vmcnt(0) = a full drain of the
  // software pipeline on every tile).  Nodes without in-edges are zeroed by zero_empty_segments_k beforehand.
  float* const wout = role == 0 ? a.gP_lo + 2 * HF : role == 1 ? a.Td_lo : a.Ud_lo;      // (wave-uniform)
  const int wpitch32 = role == 0 ? 5 * HF : a.ud_pitch;
  // the walkers' output rows as a buffer over THIS workgroup's node range (32-bit offsets whatever N; rows outside are dropped)
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(wout + v0 * wpitch32, 0, (int)(v1n - v0) * wpitch32 * 4, 0x00020000);
  // target of the throw-away stores (rows past the chunk, scoreboard equalisation): a slab of its own BEHIND the
  // gridDim.x slabs that carry results -- a late throw-away store must never meet this workgroup's final slab store
  float* const dummy_row = a.slab + (size_t)(gridDim.x + chunk) * SW * SW + lc4;
  // SRC: the by-source sums' output rows as buffers over [v0 - margin, v1 + margin) (the plan serves no source outside)
  const int64_t vbase = v0 - a.margin;
  const int nspan = (int)(v1n - v0 + 2 * a.margin);
  const __amdgpu_buffer_rsrc_t srs_g = __builtin_amdgcn_make_buffer_rsrc(
      SRC ? a.gP_lo + vbase * (5 * HF) + HF : a.gP_lo, 0, SRC ? nspan * 5 * HF * 4 : 0, 0x00020000);
  const int utp = a.ut_pitch ? a.ut_pitch : 2 * HF;                  // row pitch of the by-source sums' array
  const __amdgpu_buffer_rsrc_t srs_u = __builtin_amdgcn_make_buffer_rsrc(
      SRC ? a.UT_lo + vbase * utp : a.gP_lo, 0, SRC ? nspan * utp * 4 : 0, 0x00020000);
  const int qp = a.q_pitch ? a.q_pitch : 2 * HF, qbo = a.q_pitch ? a.qb_off : HF;      // Q_lo row layout (see ChainArgs)
  const float4 ln_live = LN ? live_mask((tid & 31) * 4, a.ln_width) : f4(1.f);
  const float ln_inv_w = LN ? 1.0f / (float)a.ln_width : 0.f;
  __syncthreads();

  float4 pg, pt, pe_, pl;                    // the next tile's row of ge'(i), t(i), e_out(i-1), t(i-1)
  float4 gt_keep = f4(0.f);                  // H2: this thread's gt row piece and its exponent, until the tile is known not to be staged again
  int row_ea = 0;
  float4 ga2, gqb, ghb, gqf, ghf, ga3;       // the node rows of this thread's edge: A2h[s] Qb[s] hb[s] | Qf[d] hf[d] A3h[d]
  int fs = 0, fd = 0;                        // source / destination node of this thread's row TWO tiles ahead (in flight)
  unsigned fi = 0;                           // SRC: its plan word (0 for the clamped rows past the chunk)
  // Software pipeline (every request is issued a full tile before its first use; sd is a ring of three tiles):
  //   after the first barrier of tile k:   indices of tile k+2 (registers), row streams of tile k+1 (registers)
  //   phase 0 of tile k+1:                 indices of tile k+2 -> sd ring;  rows of tile k+1 -> images
  //   after the gather arithmetic of k+1:  node rows of tile k+2 through sd (registers, used a tile later)
  // wave-uniform tile base + 32-bit lane offset; rows past the end of the chunk are clamped (never stored)
  const int64_t klast = ntile - 1;
  auto clamp_row = [&](int64_t k) __attribute__((always_inline)) {
    const int64_t left = re - (rb + k * ER);                             // >= 1
    const int nv = left < ER ? (int)left : ER;
    return row < nv ? row : nv - 1;
  };
  auto prefetch_idx = [&](int64_t k) __attribute__((always_inline)) {
    const int cr = clamp_row(k);
    const int64_t r = rb + k * ER + cr;
    fs = a.isrc[r];
    fd = a.idst[r];
    if constexpr (SRC) {
      const unsigned w = a.sinfo[r];
      fi = cr == row ? w : 0u;
    }
  };
  auto prefetch_rows = [&](int64_t k) __attribute__((always_inline)) {
    const int64_t r0 = rb + k * ER;
    const int o = clamp_row(k) * HF + lc4;
    pg = ld4(a.ge + r0 * HF + o);
    if constexpr (HI) pt = ld4(a.t_hi + r0 * HF + o);
    pe_ = ld4(a.e_mid + r0 * HF + o);
    pl = ld4_nt(a.t_lo + r0 * HF + o);
  };
  auto gather = [&](int64_t s, int64_t d) __attribute__((always_inline)) {   // node rows of the edge s -> d
    ga2 = ld4(a.P_lo + s * (5 * HF) + HF + lc4);
    gqb = ld4(a.Q_lo + s * qp + qbo + lc4);
    ghb = ld4(a.hb_lo + s * HF + lc4);
    gqf = ld4(a.Q_lo + d * qp + lc4);
    ghf = ld4(a.hf_lo + d * HF + lc4);
    ga3 = ld4(a.P_lo + d * (5 * HF) + 2 * HF + lc4);
  };

  if (ntile > 0) {
    prefetch_idx(0);
    const int s0 = fs, d0 = fd;
    if ((tid & 31) == 0) {
      sd[row] = s0;
      sd[ER + row] = d0;
      if constexpr (SRC) si[row] = fi;
    }
    prefetch_idx(klast < 1 ? klast : 1);            // written to the ring in phase 0 of tile 0
    prefetch_rows(0);
    // hipcc merges the vector-memory scoreboard of the loop entry with the back edge's and keeps the weaker
    // guarantee.  Throw-away stores (into this workgroup's slab, rewritten at the end) give the entry the queue a
    // steady-state iteration leaves behind -- row loads | 1 store | 16 walker stores (threads 0-95) | 6 gathers -- so that phase 0
    // waits with a COUNTED vmcnt for the row loads only and the gathers / stores stay in flight.
    st4(dummy_row, f4(0.f));
    if constexpr (LN) st4(dummy_row + SW * 17, f4(0.f));          // the LayerNorm form stores two rows per tile (ge_tot, gt)
    if (walker) {
#pragma unroll
      for (int r = 0; r < ER; ++r) st4(dummy_row + SW * (1 + r), f4(0.f));
    }
    gather(s0, d0);
  }
  for (int64_t k = 0; k < ntile; ++k) {
    const int64_t r0 = rb + k * ER;
    const int nvalid = re - r0 < ER ? (int)(re - r0) : ER;
    const int* sdk = sd + (int)(k % 3) * 2 * ER;
    // ---- phase 0: gt row and e row -> split images; residual ge row -> og; t(i-1) row -> tl; indices -> sd ----
    {
      const float4 mu = ld4(cs + lc4), rs = ld4(cs + SW + lc4), sc = ld4(cs + 2 * SW + lc4),
                   sh = ld4(cs + 3 * SW + lc4), m1 = ld4(cs + 4 * SW + lc4), m2 = ld4(cs + 5 * SW + lc4),
                   cc = ld4(cs + 6 * SW + lc4);
      st4(og + row * EOP + lc4, pg);
      st4(tl + row * SW + lc4, pl);
      st4(ef + row * SW + lc4, pe_);                  // for the sigmoid of the by-destination pass (cheaper than re-joining the split images)
      if constexpr (HI) {
        float4 gt;
        if constexpr (LN) {
          gt = pt;                                       // given (see the static_assert above)
        } else {
          const float4 gu = gate4(fma4(pt, sc, sh), pg);
          gt = cc * (gu - m1 - ((pt - mu) * rs) * m2);
        }
        if (row >= nvalid) gt = f4(0.f);               // rows past the chunk contribute nothing to gW3 / gb3
        cg0 += (double)gt.x; cg1 += (double)gt.y; cg2 += (double)gt.z; cg3 += (double)gt.w;
        if constexpr (H2) {
          gt_keep = gt;
          const int ea = h2_row_exp(gt) - 127, eb = h2_row_exp(pe_) - 127;
          const int da = h2_tn_da(Rref, ea + eb);
          row_ea = ea;
          if ((tid & 31) == 0) {
            rexp[row] = ea + eb;
            rinv[row] = pow2_biased(127 + ea + da - 4);
            if (ea + eb > Rref + 10) rexp[2 * ER] = 1;
          }
          simg_stage_h2(ig, EIMG, row, lc4, gt, pow2_biased(127 + 4 - ea - da));
          simg_stage_h2(ie, EIMG, row, lc4, pe_, pow2_biased(127 + 4 - Rref + ea + da));
        } else {
          simg_stage(ig, EIMG, row, lc4, gt);
          simg_stage(ie, EIMG, row, lc4, pe_);
        }
      }
      if ((tid & 31) == 0) {                        // indices of tile k+1 (requested a tile ago) -> ring
        int* sdn = sd + (int)((k + 1) % 3) * 2 * ER;
        sdn[row] = fs;
        sdn[ER + row] = fd;
        if constexpr (SRC) si[(int)((k + 1) % 3) * ER + row] = fi;
      }
    }
    __syncthreads();   // images, residual rows, the next tile's indices ready
    int Rnext = Rref;
    if constexpr (H2) {
      const int m = row16_max_i32(rexp[lane & (ER - 1)]);     // the tile's largest EA + EB: the same number in every wave
      if (rexp[2 * ER] != 0) {                    // rare (the first tile; a three-decade jump): staged again in the unit of its own maximum
        const float f = pow2_biased(127 + (Rref == kNoRef ? -127 : Rref - m));
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int e = 0; e < 16; ++e) tn[x][e] *= f;
        Rref = m;
        __syncthreads();                          // every wave has read the flag and the exponents
        if (tid == 0) rexp[2 * ER] = 0;
        const int da = h2_tn_da(Rref, rexp[row]);
        if ((tid & 31) == 0) rinv[row] = pow2_biased(127 + row_ea + da - 4);
        simg_stage_h2(ig, EIMG, row, lc4, gt_keep, pow2_biased(127 + 4 - row_ea - da));
        simg_stage_h2(ie, EIMG, row, lc4, pe_, pow2_biased(127 + 4 - Rref + row_ea + da));
        __syncthreads();
      }
      Rnext = m > Rref ? m : Rref;
    }
    // a tile (two for the indices) ahead, in flight under the MFMAs and the gather arithmetic; past the end the
    // last tile is requested again instead of branching around the loads
    prefetch_idx(k + 2 < klast ? k + 2 : klast);
    prefetch_rows(k + 1 < klast ? k + 1 : klast);
    __builtin_amdgcn_sched_barrier(0);
    // ---- TN: gW3[n][c] += sum_rows gt[row][n] e[row][c], this wave's 64 x 32 block (transpose reads) ----
    if constexpr (HI && H2 && !(ABL & 4)) {
      int tr0 = trq0, tr1 = trq1;
      asm volatile("" : "+v"(tr0), "+v"(tr1));
      h16x8 fa[2][2];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
          fa[x][s_] = __builtin_bit_cast(h16x8, simg_col_frag2(ig + s_ * EIMG, tr0 ^ ((2 * wn + x) << 6), tr1 ^ ((2 * wn + x) << 6)));
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const h16x8 fb = __builtin_bit_cast(h16x8, simg_col_frag2(ie + sb * EIMG, tr0 ^ (wc << 6), tr1 ^ (wc << 6)));
#pragma unroll
        for (int sa = 0; sa < 2; ++sa) {
          if (sa + sb > 1) continue;               // lo*lo is dropped
          mfh(tn[0], fa[0][sa], fb);
          mfh(tn[1], fa[1][sa], fb);
        }
      }
      floatx4_acc acc;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kc = 0; kc < SW / 32; ++kc) {
        h16x8 fn[2];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) fn[s_] = *reinterpret_cast<const h16x8*>(ig + s_ * EIMG + (nnb ^ (kc << 6)));
        mfh16s(acc, fn[1], wfh[kc][0]);
        mfh16s(acc, fn[0], wfh[kc][1]);
        mfh16s(acc, fn[0], wfh[kc][0]);
      }
      const float4 ri = ld4(rinv + 4 * ng);        // 1 / (row factor) of rows 4 ng .. 4 ng + 3
      const float rf[4] = {ri.x, ri.y, ri.z, ri.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) og[(4 * ng + e) * EOP + wave * 16 + ni] += acc[e] * (rf[e] * cinv);     // exact factors
      if (Rnext != Rref) {                         // the next tile is staged in the unit of the new reference
        const float f = pow2_biased(127 + Rref - Rnext);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int e = 0; e < 16; ++e) tn[x][e] *= f;
        Rref = Rnext;
      }
    }
    if (HI && !H2 && !(ABL & 4)) {
      int tr0 = trq0, tr1 = trq1;
      asm volatile("" : "+v"(tr0), "+v"(tr1));
      bf16x8 fa[2][3];
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
          fa[x][s_] = simg_col_frag2(ig + s_ * EIMG, tr0 ^ ((2 * wn + x) << 6), tr1 ^ ((2 * wn + x) << 6));
#pragma unroll
      for (int sb = 0; sb < 3; ++sb) {
        const bf16x8 fb = simg_col_frag2(ie + sb * EIMG, tr0 ^ (wc << 6), tr1 ^ (wc << 6));
#pragma unroll
        for (int sa = 0; sa < 3; ++sa) {
          if (sa + sb > 2) continue;               // the three products below 2^-24 are dropped
          mfb16(tn[0], fa[0][sa], fb);
          mfb16(tn[1], fa[1][sa], fb);
        }
      }
    }
    // ---- NN: acc = gt W3 (16 rows x this wave's 16 columns), joined with the residual rows in og ----
    if (HI && !H2 && !(ABL & 4)) {
      floatx4_acc acc;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kc = 0; kc < SW / 32; ++kc) {
        bf16x8 fa[3];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) fa[s_] = *reinterpret_cast<const bf16x8*>(ig + s_ * EIMG + (nnb ^ (kc << 6)));
        mfb16s(acc, fa[2], wf.w[kc][0]);
        mfb16s(acc, fa[0], wf.w[kc][2]);
        mfb16s(acc, fa[1], wf.w[kc][1]);
        mfb16s(acc, fa[1], wf.w[kc][0]);
        mfb16s(acc, fa[0], wf.w[kc][1]);
        mfb16s(acc, fa[0], wf.w[kc][0]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) og[(4 * ng + e) * EOP + wave * 16 + ni] += acc[e];
    }
    __syncthreads();   // og = ge(i-1) rows complete; every wave is done with the gt images
    // ---- by-destination backward of layer i-1 on this thread's row (edge_bwd_dst_k's arithmetic) ----
    {
      const float4 mu = ld4(cl + lc4), rs = ld4(cl + SW + lc4), sc = ld4(cl + 2 * SW + lc4), sh = ld4(cl + 3 * SW + lc4);
      const float4 ge4 = ld4(og + row * EOP + lc4);
      const float4 tt = ld4(tl + row * SW + lc4);
      float4 sg = ge4, dsg = tt;
      if (!(ABL & 2)) sigmoid_grad4(ld4(ef + row * SW + lc4), sg, dsg);
      const float4 gsig = fma4(gqf, ga2, fma4(gqb, ga3, f4(0.f) - gqf * ghf - gqb * ghb));
      const float4 g = fma4(gsig, dsg, ge4);
      const bool live = row < nvalid;        // rows past the chunk repeat its last row's indices: their terms are zeroed HERE
      st4_nt(live ? a.ge_out + (r0 + row) * HF + lc4 : dummy_row, g);
      st4(v1 + row * SW + lc4, live ? sg * gqb : f4(0.f));
      if constexpr (LN) {
        // ln_edge_bwd_dst_k's arithmetic: that and rstd from the row's own statistics, gt = LNbwd(gu) complete inside the row
        float rstd;
        const float4 th = row_normalize<SW>(tt, ln_live, ln_inv_w, rstd);
        const float4 gu = gate4(fma4(th, sc, sh), g);               // sc = gamma, sh = beta
        const float4 ag = sc * gu;
        const float m1r = row_sum<32>(hsum4(ag)) * ln_inv_w;
        const float m2r = row_sum<32>(hsum4(ag * th)) * ln_inv_w;
        const float4 gtv = (ag - f4(m1r) * ln_live - th * m2r) * rstd;
        st4_nt(live ? a.gt_out + (r0 + row) * HF + lc4 : dummy_row + SW * 17, gtv);
        st4(v2 + row * SW + lc4, live ? gtv : f4(0.f));
        st4(v3 + row * SW + lc4, live ? th : f4(0.f));
        st4(v6 + row * SW + lc4, live ? gu : f4(0.f));
      } else {
        st4(v2 + row * SW + lc4, live ? gate4(fma4(tt, sc, sh), g) : f4(0.f));
        st4(v3 + row * SW + lc4, live ? (tt - mu) * rs : f4(0.f));
      }
      if constexpr (SRC) st4(v5 + row * SW + lc4, live ? sg * gqf : f4(0.f));
    }
    __syncthreads();   // per-edge terms of the tile are in v1 / v2 / v3
    // ---- column walkers (waves 0 and 1) and BatchNorm sums (waves 2 and 3); the rest go on to the next phase 0 ----
    if (!(ABL & 1) && walker) {
      // The 16-step chain is the critical path of this phase (the other waves wait for it at the next barrier), so a
      // step is kept to two packed FMAs and a store: the destination node is wave-uniform (scalar compare, scalar
      // part of the address), a new segment multiplies the running sum by 0 instead of selecting, and rows past
      // the chunk were zeroed when they were written.
      const float* vsrc = (role == 0 ? v1 : role == 1 ? v3 : v2) + wc4;
#pragma unroll
      for (int r4 = 0; r4 < ER; r4 += WG_) {        // WG_ rows' LDS reads up front, then the dependent chain
        int dn[WG_];
        float4 xs[WG_];
#pragma unroll
        for (int q = 0; q < WG_; ++q) {
          dn[q] = __builtin_amdgcn_readfirstlane(sdk[ER + r4 + q]);
          xs[q] = ld4(vsrc + (r4 + q) * SW);
        }
        // node of the row after this group of rows: the next group's first, or (tile end) the next tile's first row, which
        // phase 0 put into the ring before this tile's first barrier; the chunk's last tile closes every segment
        const int dafter = r4 + WG_ < ER ? __builtin_amdgcn_readfirstlane(sdk[ER + r4 + WG_])
                                         : (k == klast ? -1 : __builtin_amdgcn_readfirstlane(sd[(int)((k + 1) % 3) * 2 * ER + ER]));
#pragma unroll
        for (int q = 0; q < WG_; ++q) {
          const float keep = dn[q] == cur ? 1.f : 0.f;
          acc0 = fma4(acc0, f4(keep), xs[q]);
          cur = dn[q];
          // Round 3: only the LAST row of a segment reaches memory.  The store stays unconditional (a buffer store: a row that
          // does not end its segment carries an out-of-range offset and is dropped by the hardware), so there is still no
          // data-dependent branch around a memory operation; what goes away is 4/5 of the walk's traffic through the CU's
          // vector-memory path (24 KB of 136 KB per tile).
          const bool ends = (q + 1 < WG_ ? dn[q + 1] : dafter) != cur;
          const u32x4_ bits = {__builtin_bit_cast(unsigned, acc0.x), __builtin_bit_cast(unsigned, acc0.y),
                               __builtin_bit_cast(unsigned, acc0.z), __builtin_bit_cast(unsigned, acc0.w)};
          if (WSKIP) { if (ends) __builtin_amdgcn_raw_buffer_store_b128(bits, wrs, ((cur - (int)v0) * wpitch32 + wc4) * 4, 0, 0); }
          else __builtin_amdgcn_raw_buffer_store_b128(bits, wrs, ends ? ((cur - (int)v0) * wpitch32 + wc4) * 4 : (int)0x80000000, 0, 0);
        }
      }
    }
    if (!(ABL & 1) && bnsum) {                       // LDS reads and fp64 arithmetic only (rows past the chunk hold zeros)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 x = ld4((LN ? v6 : v2) + (brow + q) * SW + wc4);
        const float4 th = ld4(v3 + (brow + q) * SW + wc4);
        s_gu[0] += (double)x.x; s_gu[1] += (double)x.y; s_gu[2] += (double)x.z; s_gu[3] += (double)x.w;
        s_gut[0] += (double)x.x * (double)th.x; s_gut[1] += (double)x.y * (double)th.y;
        s_gut[2] += (double)x.z * (double)th.z; s_gut[3] += (double)x.w * (double)th.w;
      }
    }
    if constexpr (SRC) {
      auto bits4 = [](const float4& v) __attribute__((always_inline)) {
        const u32x4_ b = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y),
                          __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
        return b;
      };
      if (!(ABL & 1) && !(ABL & 8) && wave >= 3) {   // beside the column walk: half-wave q serves the tile rows q and q + 10
        // The two rows of a half-wave are served TOGETHER, every LDS read issued unconditionally and up front -- the plan
        // words, then the leader's own row of the three images and the slot contents for both rows (12 + 12 independent
        // reads), a loop only for the rare further rows of a run -- so that the phase is three LDS round trips long instead
        // of ten: served one after the other through data-dependent loops it was 0.68 ms of the kernel (ablation, r04).
        const int q = (tid - 192) >> 5;                // 0 .. 9
        const unsigned* sik = si + (int)(k % 3) * ER;
        const int rr[2] = {q, (q + 10) & (ER - 1)};
        unsigned w[2] = {sik[rr[0]], q + 10 < ER ? sik[rr[1]] : 0u};
        float4 s1[2], s2[2], s3[2], p1[2], p2[2], p3[2];
        unsigned m[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          m[u] = w[u] & 0xffffu;
          const int b = rr[u];                         // a leader's first row is its own
          s1[u] = ld4(v5 + b * SW + wc4);
          s2[u] = ld4(v2 + b * SW + wc4);
          s3[u] = ld4(v3 + b * SW + wc4);
          const float* p = slots + ((w[u] >> 16) & (kSweepSlots - 1)) * SW + wc4;
          p1[u] = ld4(p);
          p2[u] = ld4(p + kSweepSlots * SW);
          p3[u] = ld4(p + 2 * kSweepSlots * SW);
          m[u] &= m[u] - 1;                            // (0 stays 0)
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          while (m[u]) {                               // further rows of the run (a source with several edges into this tile)
            const int b = __builtin_ctz(m[u]);
            m[u] &= m[u] - 1;
            s1[u] += ld4(v5 + b * SW + wc4);
            s2[u] += ld4(v2 + b * SW + wc4);
            s3[u] += ld4(v3 + b * SW + wc4);
          }
          const bool lead = (w[u] & 0xffffu) != 0;
          if (lead && !(w[u] & kSweepOpen)) {
            s1[u] += p1[u];
            s2[u] += p2[u];
            s3[u] += p3[u];
          }
          float* p = slots + ((w[u] >> 16) & (kSweepSlots - 1)) * SW + wc4;
          if (lead && !(w[u] & kSweepClose)) {
            st4(p, s1[u]);
            st4(p + kSweepSlots * SW, s2[u]);
            st4(p + 2 * kSweepSlots * SW, s3[u]);
          }
          const bool out = lead && (w[u] & kSweepClose);
          const int sn = sdk[rr[u]] - (int)vbase;
          const int og_ = out ? (sn * (5 * HF) + wc4) * 4 : (int)0x80000000;
          const int ou_ = out ? (sn * utp + wc4) * 4 : (int)0x80000000;
          if (!(ABL & 16) && __builtin_amdgcn_ballot_w64(out) != 0) {
            __builtin_amdgcn_raw_buffer_store_b128(bits4(s1[u]), srs_g, og_, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(bits4(s2[u]), srs_u, ou_, 0, 0);
            if constexpr (!LN) __builtin_amdgcn_raw_buffer_store_b128(bits4(s3[u]), srs_u, ou_, HF * 4, 0);
          }
        }
      }
    }
    {                                              // the next tile's node rows, through the ring (written before this tile's first barrier)
      const int* sdn = sd + (int)((k + 1) % 3) * 2 * ER;
      gather(sdn[row], sdn[ER + row]);
    }
    // no barrier: a wave reaches the next tile's post-phase-0 barrier only after its own walk, v1 - v3 are rewritten
    // after that barrier, sd is a ring of three; what the next phase 0 writes before it (images, og, tl) was last
    // read by the SAME thread (same row / column mapping) or before this tile's second barrier (gt images)
  }

  if constexpr (HI) {
    float* sl = a.slab + (size_t)chunk * SW * SW;
    const float unit = !H2 ? 1.f : Rref == kNoRef ? 0.f : pow2_biased(127 + Rref - 8);       // H2: out of the products' unit 2^(8 - R)
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = (2 * wn + x) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lg;
        sl[m * SW + wc * 32 + li] = H2 ? tn[x][e] * unit : tn[x][e];
      }
  }
  __syncthreads();
  {   // BatchNorm column sums of layer i-1: eight row-pair groups (waves 2, 3, 6, 7) -> one row of partials_lo
    double* bnr = reinterpret_cast<double*>(lds);      // [8 groups][2][128] doubles = 16 KB (the images are dead)
    if (bnsum) {
      const int grp = brow >> 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bnr[(grp * 2 + 0) * SW + wc4 + j] = s_gu[j];
        bnr[(grp * 2 + 1) * SW + wc4 + j] = s_gut[j];
      }
    }
    __syncthreads();
    if (tid < 2 * SW) {
      double s_ = 0.0;
#pragma unroll
      for (int g8 = 0; g8 < 8; ++g8) s_ += bnr[g8 * 2 * SW + tid];
      a.partials_lo[(size_t)chunk * 2 * HF + (tid / SW) * HF + (tid % SW)] = s_;
    }
    __syncthreads();
  }
  if constexpr (!HI) return;
  double* red = reinterpret_cast<double*>(lds);          // 16 row slots x 128 columns = 16 KB (the images are dead)
  red[row * SW + lc4 + 0] = cg0;
  red[row * SW + lc4 + 1] = cg1;
  red[row * SW + lc4 + 2] = cg2;
  red[row * SW + lc4 + 3] = cg3;
  __syncthreads();
  if (tid < SW) {
    double s_ = 0.0;
#pragma unroll
    for (int k = 0; k < ER; ++k) s_ += red[k * SW + tid];
    a.partials[(size_t)chunk * SW + tid] = s_;
  }
}


// gA3h / Ud / Td rows of the nodes WITHOUT in-edges (the column walkers only ever store to nodes that own rows)
template <int HF>
__global__ __launch_bounds__(256) void zero_empty_segments_k(int64_t N, const int32_t* __restrict__ in_ptr,
                                                             float* __restrict__ gP, float* __restrict__ Ud,
                                                             float* __restrict__ Td, int ud_pitch) {
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
      if (lane < 32) {
        st4(gP + u * (5 * HF) + 2 * HF + c4, f4(0.f));
        st4(Td + u * ud_pitch + c4, f4(0.f));
      } else {
        st4(Ud + u * ud_pitch + c4, f4(0.f));
      }
    }
  }
}

// returns the grid size (= number of gW3 slabs / partial rows of both kinds)
int edge_bwd_chain_launch(const ChainArgs& in, const float* W3, void* wpack, hipStream_t st, bool h2) {
  if (W3 && h2) {
    hipLaunchKernelGGL(pack_w2_nn16_k, dim3(SW / 16), dim3(256), 0, st, W3, (int64_t)SW, (unsigned char*)wpack);
  }
  else if (W3) hipLaunchKernelGGL(pack_w3_nn16_k, dim3(8), dim3(256), 0, st, W3, (int64_t)SW, (bf16x8*)wpack);
  ChainArgs a = in;
  a.Wp = (const bf16x8*)wpack;
  int grid = 0;
  gnm_sweep_partition(a.N, 1, &a.nodes_per_block, &grid);  // one 512-thread workgroup per CU
  if (a.hfull == 2 * SW) {                           // a 256-wide layer's sweep, one 128-column half: top of the stack only
    hipLaunchKernelGGL(zero_empty_segments_k<2 * SW>, dim3(num_cus() * 2), dim3(256), 0, st, a.N, a.in_ptr, a.gP_lo, a.Ud_lo, a.Td_lo, a.ud_pitch);
    if (a.t_hi || !a.sinfo) return -1;
    hipLaunchKernelGGL((edge_bwd_chain_k<0, true, true, false, 2 * SW>), dim3(grid), dim3(CT), 0, st, a);
    return grid;
  }
  hipLaunchKernelGGL(zero_empty_segments_k<SW>, dim3(num_cus() * 2), dim3(256), 0, st, a.N, a.in_ptr, a.gP_lo, a.Ud_lo, a.Td_lo, a.ud_pitch);
#ifdef GNM_TIMING_ABLATIONS      // builds for timing experiments only: the ablated kernels give wrong results
  static const int abl = getenv("GNM_CHAIN_ABL") ? atoi(getenv("GNM_CHAIN_ABL")) : 0;
  switch (abl) {
    case 1: hipLaunchKernelGGL((edge_bwd_chain_k<1, false>), dim3(grid), dim3(CT), 0, st, a); return grid;
    case 2: hipLaunchKernelGGL((edge_bwd_chain_k<2, false>), dim3(grid), dim3(CT), 0, st, a); return grid;
    case 4: hipLaunchKernelGGL((edge_bwd_chain_k<4, false>), dim3(grid), dim3(CT), 0, st, a); return grid;
    case 7: hipLaunchKernelGGL((edge_bwd_chain_k<7, false>), dim3(grid), dim3(CT), 0, st, a); return grid;
    case 8: if (a.sinfo && a.t_hi) { hipLaunchKernelGGL((edge_bwd_chain_k<8, true>), dim3(grid), dim3(CT), 0, st, a); return grid; } break;
    case 16: if (a.sinfo && a.t_hi) { hipLaunchKernelGGL((edge_bwd_chain_k<16, true>), dim3(grid), dim3(CT), 0, st, a); return grid; } break;
    case 9: if (a.sinfo && a.t_hi) { hipLaunchKernelGGL((edge_bwd_chain_k<9, true>), dim3(grid), dim3(CT), 0, st, a); return grid; } break;
    default: break;
  }
#endif
  if (!a.t_hi) {                                     // top of the stack: the sweep without a layer above
    if (a.ln_gamma) {                                // ... in its LayerNorm form (two-sided only)
      if (!a.sinfo || !a.gt_out || !a.ln_beta) return -1;
      hipLaunchKernelGGL((edge_bwd_chain_k<0, true, true, false, SW, false, true>), dim3(grid), dim3(CT), 0, st, a);
      return grid;
    }
    if (a.sinfo) hipLaunchKernelGGL((edge_bwd_chain_k<0, true, true, false>), dim3(grid), dim3(CT), 0, st, a);
    else hipLaunchKernelGGL((edge_bwd_chain_k<0, false, true, false>), dim3(grid), dim3(CT), 0, st, a);
    return grid;
  }
  if (a.ln_gamma) {                                  // the chained LayerNorm backward (two-sided only): layer i's gt given in t_hi
    if (!a.sinfo || !a.gt_out || !a.ln_beta) return -1;
    if (h2) hipLaunchKernelGGL((edge_bwd_chain_k<0, true, true, true, SW, true, true>), dim3(grid), dim3(CT), 0, st, a);
    else hipLaunchKernelGGL((edge_bwd_chain_k<0, true, true, true, SW, false, true>), dim3(grid), dim3(CT), 0, st, a);
    return grid;
  }
  if (h2) {
    if (a.sinfo) hipLaunchKernelGGL((edge_bwd_chain_k<0, true, true, true, SW, true>), dim3(grid), dim3(CT), 0, st, a);
    else hipLaunchKernelGGL((edge_bwd_chain_k<0, false, true, true, SW, true>), dim3(grid), dim3(CT), 0, st, a);
    return grid;
  }
  if (a.sinfo) hipLaunchKernelGGL((edge_bwd_chain_k<0, true>), dim3(grid), dim3(CT), 0, st, a);     // column walk + by-source run sums
  else hipLaunchKernelGGL((edge_bwd_chain_k<0, false>), dim3(grid), dim3(CT), 0, st, a);
  return grid;
}

int tn_tr_rows_per_tile() { return TRR; }
int tn_tr_occupancy(bool conv, bool h2) {
  if (h2) return conv ? occ_blocks<tn_tr_k<true, 2, true>>() : occ_blocks<tn_tr_k<false, 2, true>>();
  return conv ? occ_blocks<tn_tr_k<true, 2>>() : occ_blocks<tn_tr_k<false, 2>>();
}
// cv: the A operand is formed from raw sums (ncg = 2, ncgb = 1);  h2: the f16x2 form
void tn_tr_launch(int64_t M, const float* A, int64_t lda, int ncg, const void* B, int64_t ldb, int ncgb, float* slab,
                  double* partials, int nslot, int64_t tiles_per_slot, hipStream_t st, const TnConv* cv, bool h2) {
  const TnConv none{};
  const dim3 grid(nslot * ncg * (cv ? 1 : ncgb));
#define GNM_TN_LAUNCH(CONV, H2)                                                                                             \
  hipLaunchKernelGGL((tn_tr_k<CONV, 2, H2>), grid, dim3(kBlock), 0, st, M, A, lda, ncg, B, ldb, cv ? 1 : ncgb, slab, partials, \
                     nslot, tiles_per_slot, cv ? *cv : none)
  if (cv && h2) GNM_TN_LAUNCH(true, true);
  else if (cv) GNM_TN_LAUNCH(true, false);
  else if (h2) GNM_TN_LAUNCH(false, true);
  else GNM_TN_LAUNCH(false, false);
#undef GNM_TN_LAUNCH
}

}  // namespace gnm
