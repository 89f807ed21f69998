// bf16x3 kernels on swizzled row-major split images + hardware transpose reads (see gnm_tr.h), gfx950.
//
//   tn_tr_k     weight gradient of a Linear whose input is 128 wide:  gW[cg] = A[:, cg]^T B,  gb[cg] = sum A[:, cg]
//               (autograd of gated_gcn_full.py:107-112 and of the predictor's node halves, score_predictor.py:13-17)
#include "gnm_tr.h"

namespace gnm {

constexpr int TRR = 32;                     // rows per tile
constexpr int TRIMG = TRR * SPITCH;         // bytes per image (8 KB)

// slab[(cg*nslot + slot)][m][n] = sum over the slot's rows of A[row][cg*128 + m] * B[row][n];
// partials[(cg*nslot + slot)][128] = column sums of A[:, cg*128 ..].
// 32-row tiles, 48 KB of LDS, coalesced float4 loads one tile ahead; each of the 4 waves owns a 64 x 64 block of
// the 128 x 128 result (64 accumulator registers).  Two or three workgroups share a CU, so one's split / staging
// VALU work and HBM waits run under the others' MFMAs.
__global__ __launch_bounds__(kBlock, 2) void tn_tr_k(int64_t M, const float* __restrict__ A, int64_t lda, int ncg,
                                                     const float* __restrict__ B, float* __restrict__ slab,
                                                     double* __restrict__ partials, int nslot, int64_t tiles_per_slot) {
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
  const int cg = jj % ncg, slot = xcd * (nslot / kXcds) + jj / ncg;
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
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = (tile < tb1 ? tile : tb1 - 1) * TRR;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      int64_t r = r0 + lrow + 8 * it;
      r = r < Mlast ? r : Mlast;
      pa[it] = ld4_nt(A + r * lda + cg * SW + lc4);
      pb[it] = ld4(B + r * SW + lc4);        // shared by the ncg workgroups of the slot through L2
    }
  };
  if (tb0 < tb1) prefetch(tb0);
  for (int64_t tile = tb0; tile < tb1; ++tile) {
    const int64_t r0 = tile * TRR;
    __syncthreads();                          // the previous tile's fragment reads are done
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      if (r0 + lrow + 8 * it >= M) pa[it] = f4(0.f);      // rows past the end contribute nothing
      c0 += (double)pa[it].x; c1 += (double)pa[it].y; c2 += (double)pa[it].z; c3 += (double)pa[it].w;
      simg_stage(ia, TRIMG, lrow + 8 * it, lc4, pa[it]);
      simg_stage(ib, TRIMG, lrow + 8 * it, lc4, pb[it]);
    }
    __syncthreads();
    prefetch(tile + 1);                       // in flight under the MFMAs
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
  // C / D layout of the 32 x 32 MFMA: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  float* sl = slab + (size_t)(cg * nslot + slot) * SW * SW;
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
    partials[(size_t)(cg * nslot + slot) * SW + tid] = s;
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

template <bool FULL>
struct tile_tag { static constexpr bool full = FULL; };

// VAR bit 0: residual ge rows and the fp64 column sums wait in LDS (1) or in registers (0);
//     bit 1: the prefetch is pinned right behind the first barrier (1) or left to hipcc's scheduler (0)
template <int VAR>
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
  for (int c = tid; c < SW; c += kBlock) {
    cs[c] = stat[c];
    cs[SW + c] = stat[SW + c];
    cs[2 * SW + c] = stat[2 * SW + c];
    cs[3 * SW + c] = stat[3 * SW + c];
    cs[4 * SW + c] = bstat[c];
    cs[5 * SW + c] = bstat[SW + c];
    cs[6 * SW + c] = gamma[c] * stat[SW + c];
  }
  W3Frag wf;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const bf16x8* p = Wp + ((int64_t)(2 * wave + nb) * (SW / 32) * 3) * 64 + lane;
#pragma unroll
    for (int kc = 0; kc < SW / 32; ++kc)
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) wf.w[nb][kc][s_] = p[(kc * 3 + s_) * 64];
  }
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
    {
      double c0 = cg0, c1 = cg1, c2 = cg2, c3 = cg3;
      if (STASH) { c0 = cgs[0]; c1 = cgs[1]; c2 = cgs[2]; c3 = cgs[3]; }
      const float4 mu = ld4(cs + lc4), rs = ld4(cs + SW + lc4), sc = ld4(cs + 2 * SW + lc4),
                   sh = ld4(cs + 3 * SW + lc4), m1 = ld4(cs + 4 * SW + lc4), m2 = ld4(cs + 5 * SW + lc4),
                   cc = ld4(cs + 6 * SW + lc4);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = lrow + 8 * it;
        const bool ok = FULL || (r0 + row < E);
        if (STASH) st4(og + row * EOP + lc4, pg[it]);   // this thread read exactly these og elements in the previous epilogue
        else gk[it] = pg[it];
        const float4 gu = gate4(fma4(pt[it], sc, sh), pg[it]);
        float4 gt = cc * (gu - m1 - ((pt[it] - mu) * rs) * m2);
        float4 ev = pe_[it];
        if (!ok) { gt = f4(0.f); ev = f4(0.f); }
        c0 += (double)gt.x; c1 += (double)gt.y; c2 += (double)gt.z; c3 += (double)gt.w;
        simg_stage(ig, EIMG, row, lc4, gt);
        simg_stage(ie, EIMG, row, lc4, ev);
      }
      if (STASH) { cgs[0] = c0; cgs[1] = c1; cgs[2] = c2; cgs[3] = c3; }
      else { cg0 = c0; cg1 = c1; cg2 = c2; cg3 = c3; }
    }
    __syncthreads();   // images and residual rows ready
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);   // in flight under the MFMAs, the epilogue and the partner workgroup
    if (PIN) __builtin_amdgcn_sched_barrier(0);   // keep the loads HERE (hipcc otherwise sinks them behind the MFMAs)
    // ---- TN: gW3[n][c] += sum_rows gt[row][n] e_in[row][c], this wave's 64 x 64 block (transpose reads) ----
    {
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
  // vmcnt instead of vmcnt(0): see edge_bwd_fused_k in gnm_fused.hip); the slab is rewritten at the end
#pragma unroll
  for (int it = 0; it < 2; ++it) st4(slab + (size_t)chunk * SW * SW + (lrow + 8 * it) * SW + lc4, f4(0.f));
  for (int64_t tile = tb0; tile < nfull; ++tile) body(tile_tag<true>{}, tile);
  if (nfull < tb1 && nfull >= tb0) body(tile_tag<false>{}, nfull);

  float* sl = slab + (size_t)chunk * SW * SW;
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
                       double* partials, hipStream_t st) {
  hipLaunchKernelGGL(pack_w3_nn16_k, dim3(8), dim3(256), 0, st, W3, (int64_t)SW, (bf16x8*)wpack);
  const int64_t ntiles = (E + ER - 1) / ER;
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

int tn_tr_rows_per_tile() { return TRR; }
int tn_tr_occupancy() { return occ_blocks<tn_tr_k>(); }
void tn_tr_launch(int64_t M, const float* A, int64_t lda, int ncg, const float* B, float* slab, double* partials,
                  int nslot, int64_t tiles_per_slot, hipStream_t st) {
  hipLaunchKernelGGL(tn_tr_k, dim3(nslot * ncg), dim3(kBlock), 0, st, M, A, lda, ncg, B, slab, partials, nslot,
                     tiles_per_slot);
}

}  // namespace gnm
