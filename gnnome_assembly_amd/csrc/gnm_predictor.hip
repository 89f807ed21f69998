// Fused ScorePredictor kernels for H = 128, hidden_edge_scores = 64 (score_predictor.py:12-18 in
// the split-W1 form  score = W2 relu(Ps[src] + Pd[dst] + e W1e^T + b1) + b2).
//
//   pred_fwd_k   one pass over e: 64-row tiles through LDS, hid = e W1e^T on the matrix cores
//                (fp32 MFMA, W1e stationary in VGPRs), + b1 + Ps[src] + Pd[dst], relu, dot with
//                W2, scores scattered to the caller's edge ids.  Replaces gemm NT [E,64,128] +
//                predictor_score_fwd (hid was written and re-read once).
//   pred_bwd_k   one pass: ghid = gscore W2 [hid > 0] (written in place, the by-node sums need
//                it), ge = ghid W1e (NN), gW1e += ghid^T e (TN), column sums for gW2 / gb1 / gb2.
//                Replaces predictor_score_bwd + colsum + gemm TN [64,128,E] + gemm NN [E,128,64].
// Both kernels run two workgroups per CU (<= 256 VGPRs, <= 52 KB LDS), so one workgroup's global
// loads and epilogue overlap the other's MFMAs.
#include <type_traits>

#include "gnm_common.h"

namespace gnm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int PH = 128;          // edge feature width
constexpr int PS = 64;           // hidden_edge_scores
constexpr int PT = 64;           // rows per tile
constexpr int PEP = PH + 4;      // LDS pitch of the e tile / ge output image
constexpr int PGP = PS + 4;      // LDS pitch of the hid / ghid tile

__device__ __forceinline__ int64_t clampr(int64_t r, int64_t hi) { return r < hi ? r : hi; }

// B-fragment pack for v_mfma_f32_32x32x2_f32 with the k = 8q + 4g + r permutation (see gnm_fused.hip):
//   NT (y = x W^T): Wp[cb][q][lane] = W[(cb*32 + i) * ld + 8q + 4g + 0..3]
//   NN (y = x W):   Wp[cb][q][lane] = W[(8q + 4g + 0..3) * ld + cb*32 + i]
__global__ void pack_wk_k(const float* __restrict__ W, int64_t ld, int ncb, int nkq, int nn, float* __restrict__ Wp) {
  const int total = ncb * nkq * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, q = (idx >> 6) % nkq, cb = idx / (64 * nkq);
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

__device__ __forceinline__ void mf4(floatx16& acc, const float4& a, const float4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

// The steady-state loops below run FULL tiles only and are branch-free (clamped addresses, every
// lane loads its index / stores its result even when a neighbour holds the same one): with a
// divergent branch around a load or store hipcc falls back to s_waitcnt vmcnt(0), which drained the
// prefetch in front of the MFMAs (measured: 70 % of wave time parked).  The ragged last tile runs
// through a predicated copy of the body.
// ------------------------------------------------------------------------------------------
// PHT = the edge feature width: 128 (two workgroups per CU), or 256 (the reference's default dim_latent: the e tile image is 66 KB,
// one workgroup per CU, W1e's 64 x 256 block of a wave in 128 registers)
template <int PHT> struct PredDims {
  static constexpr int PEP = PHT + 4;                 // LDS pitch of the e tile / ge output image
  static constexpr int CPR = PHT / 4;                 // threads per row of the coalesced tile image
  static constexpr int RPP = kBlock / CPR;            // rows per pass
  static constexpr int NP = PT / RPP;                 // passes per 64-row tile
  static constexpr int NB = PHT / 128;                // 32-column blocks of the e width per wave (backward)
};

template <bool SAVE, int PHT>
__global__ __launch_bounds__(kBlock, PHT == 128 ? 2 : 1) void pred_fwd_k(
    int64_t E, const float* __restrict__ e, const float* __restrict__ Wp, const float* __restrict__ b1,
    const float* __restrict__ Pn, const int32_t* __restrict__ isrc, const int32_t* __restrict__ idst,
    const int32_t* __restrict__ perm, const float* __restrict__ W2, const float* __restrict__ b2,
    float* __restrict__ hid, float* __restrict__ scores, int64_t tiles_per_block) {
  using D = PredDims<PHT>;
  constexpr int PEP = D::PEP, NP = D::NP, RPP = D::RPP;
  __shared__ float xs[PT * PEP];
  __shared__ float os[PT * PGP];
  __shared__ int sidx[4 * PT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int rb = wave >> 1, cb = wave & 1;     // this wave's 32 x 32 block of the 64 x 64 tile result
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + PT - 1) / PT;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, E / PT);
  const int lrow = tid / D::CPR, lc4 = (tid % D::CPR) * 4;   // e tile image: RPP rows x PHT / 4 float4 per pass
  const int er = tid >> 4, ec4 = (tid & 15) * 4;        // hid tile image: 16 rows x 16 float4 per pass
  const int64_t Elast = E - 1;
  const int32_t* const ibase = wave == 0 ? isrc : wave == 1 ? idst : perm;   // wave 3 re-reads perm (unused)

  float4 wf[PHT / 8];
  {
    const float4* p = reinterpret_cast<const float4*>(Wp) + ((int64_t)cb * (PHT / 8)) * 64 + lane;
#pragma unroll
    for (int q = 0; q < PHT / 8; ++q) wf[q] = p[q * 64];
  }
  const float4 b1v = ld4(b1 + ec4), w2v = ld4(W2 + ec4);
  const float b2v = b2[0];

  float4 pre[NP];
  int pre_idx = 0;
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * PT;
#pragma unroll
    for (int it = 0; it < NP; ++it) pre[it] = ld4_nt(e + clampr(r0 + lrow + RPP * it, Elast) * PHT + lc4);
    pre_idx = ibase[clampr(r0 + lane, Elast)];
  };
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * PT;
    __syncthreads();   // previous tile's epilogue is done with os / sidx, its MFMAs with xs
#pragma unroll
    for (int it = 0; it < NP; ++it) st4(xs + (lrow + RPP * it) * PEP + lc4, pre[it]);
    sidx[tid] = pre_idx;
    __syncthreads();
    // this tile's Ps[src] / Pd[dst] rows: issued now, consumed after the MFMAs
    float4 gsv[PT / 16], gdv[PT / 16];
#pragma unroll
    for (int p = 0; p < PT / 16; ++p) {
      const int row = p * 16 + er;
      gsv[p] = ld4(Pn + (int64_t)sidx[row] * (2 * PS) + ec4);
      gdv[p] = ld4(Pn + (int64_t)sidx[PT + row] * (2 * PS) + PS + ec4);
    }
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);
    floatx16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    {
      const float* p = xs + (rb * 32 + li) * PEP + 4 * lg;
#pragma unroll
      for (int q = 0; q < PHT / 8; ++q) {
        mf4(acc, ld4(p + 8 * q), wf[q]);
        if (PHT > 128 && (q & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep hipcc from hoisting every fragment read
      }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) os[(rb * 32 + (k & 3) + 8 * (k >> 2) + 4 * lg) * PGP + cb * 32 + li] = acc[k];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PT / 16; ++p) {
      const int row = p * 16 + er;
      const int64_t grow = r0 + row;
      float4 v = ld4(os + row * PGP + ec4) + b1v + gsv[p] + gdv[p];
      if (SAVE && (FULL || grow < E)) st4_nt(hid + grow * PS + ec4, v);
      const float4 r = relu4(v);
      float dot = r.x * w2v.x + r.y * w2v.y + r.z * w2v.z + r.w * w2v.w;
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) dot += __shfl_xor(dot, off, 64);
      // all 16 lanes of a row hold the same dot and store it to the same address (one transaction)
      if (FULL || grow < E) scores[sidx[2 * PT + row]] = dot + b2v;
    }
  };
  if (tb0 < tb1) prefetch(tb0);
  for (int64_t tile = tb0; tile < nfull; ++tile) body(std::true_type{}, tile);
  if (nfull < tb1 && nfull >= tb0) body(std::false_type{}, nfull);
}

// ------------------------------------------------------------------------------------------
template <int PHT>
__global__ __launch_bounds__(kBlock, PHT == 128 ? 2 : 1) void pred_bwd_k(
    int64_t E, float* __restrict__ hid, const float* __restrict__ gscore, const int32_t* __restrict__ perm,
    const float* __restrict__ W2, const float* __restrict__ e, const float* __restrict__ Wp,   // W1e packed NN
    float* __restrict__ ge, float* __restrict__ slab,      // [grid][64][PHT] partial gW1e
    double* __restrict__ partials,                         // [grid][3][64]: sum gs*relu(hid) | sum ghid | sum gs
    int64_t tiles_per_block) {
  using D = PredDims<PHT>;
  constexpr int PEP = D::PEP, NP = D::NP, RPP = D::RPP, NB = D::NB;
  __shared__ float gsm[PT * PGP];      // ghid tile
  __shared__ float es[PT * PEP];       // e tile, later the ge output image
  __shared__ float gsc[4 * PT];        // gscore of the tile's rows (one copy per wave)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lg = lane >> 5;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + PT - 1) / PT;
  const int64_t tb0 = (int64_t)chunk * tiles_per_block;
  const int64_t tb1 = min(ntiles, tb0 + tiles_per_block);
  const int64_t nfull = min(tb1, E / PT);
  const int lrow = tid / D::CPR, lc4 = (tid % D::CPR) * 4;
  const int er = tid >> 4, ec4 = (tid & 15) * 4;
  const int64_t Elast = E - 1;
  const int cb0 = wave * NB;           // this wave's NB 32-column blocks of the e width: cb0 .. cb0 + NB - 1

  float4 wf[NB][PS / 8];   // W1e as the NN operand: this wave's output columns, K = 64
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float4* p = reinterpret_cast<const float4*>(Wp) + ((int64_t)(cb0 + b) * (PS / 8)) * 64 + lane;
#pragma unroll
    for (int q = 0; q < PS / 8; ++q) wf[b][q] = p[q * 64];
  }
  const float4 w2v = ld4(W2 + ec4);
  floatx16 tn[NB][2];      // gW1e blocks: rows (ghid columns) 0-31 / 32-63 x this wave's e columns
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int k = 0; k < 16; ++k) { tn[b][0][k] = 0.f; tn[b][1][k] = 0.f; }
  Stat4 st;            // a: sum gs*relu(hid), b: sum ghid  (this thread's 4 columns)
  st.zero();
  double sgs = 0.0;    // sum of gscore over this lane's rows (every wave holds a copy; wave 0's is used)

  float4 pe_[NP], ph[PT / 16];
  int pidx = 0;        // perm of the tile AFTER the prefetched one (row = lane; all four waves alike)
  float pgs = 0.f;     // gscore of the prefetched tile's row
  auto prefetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t r0 = tile * PT;
#pragma unroll
    for (int it = 0; it < NP; ++it) pe_[it] = ld4(e + clampr(r0 + lrow + RPP * it, Elast) * PHT + lc4);
#pragma unroll
    for (int p = 0; p < PT / 16; ++p) ph[p] = ld4(hid + clampr(r0 + p * 16 + er, Elast) * PS + ec4);
    pgs = gscore[pidx];                               // pidx was loaded one prefetch earlier
    pidx = perm[clampr(r0 + PT + lane, Elast)];
  };
  auto body = [&](auto tag, int64_t tile) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(tag)::value;
    const int64_t r0 = tile * PT;
    __syncthreads();   // previous tile's output image / gsc are consumed
    {
      const float g_ = (FULL || r0 + lane < E) ? pgs : 0.f;
      gsc[tid] = g_;
      sgs += (double)g_;
    }
#pragma unroll
    for (int it = 0; it < NP; ++it) st4(es + (lrow + RPP * it) * PEP + lc4, pe_[it]);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PT / 16; ++p) {
      const int row = p * 16 + er;
      const float g_ = gsc[row];           // 0 for rows past the end
      const float4 h_ = ph[p];
      const float4 gh = gate4(h_, w2v * g_);
      if (FULL || r0 + row < E) st4(hid + (r0 + row) * PS + ec4, gh);
      st4(gsm + row * PGP + ec4, gh);
      st.add(relu4(h_) * g_, gh);
    }
    __syncthreads();
    prefetch(tile + 1 < tb1 ? tile + 1 : tile);
    // ---- ge tile = ghid W1e  (K = 64) ----
    floatx16 acc[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < 16; ++k) { acc[b][0][k] = 0.f; acc[b][1][k] = 0.f; }
    {
      const float* p0 = gsm + li * PGP + 4 * lg;
      const float* p1 = gsm + (32 + li) * PGP + 4 * lg;
#pragma unroll
      for (int q = 0; q < PS / 8; ++q) {
        const float4 a0 = ld4(p0 + 8 * q), a1 = ld4(p1 + 8 * q);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          mf4(acc[b][0], a0, wf[b][q]);
          mf4(acc[b][1], a1, wf[b][q]);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep hipcc from hoisting every fragment read (spills)
      }
    }
    // ---- gW1e[n][c] += sum_rows ghid[row][n] e[row][c] ----
    const float* ga = gsm + 4 * lg * PGP + li;               // row 8q + 4lg + r: one lane-dependent base,
    const float* eb = es + 4 * lg * PEP + cb0 * 32 + li;     // compile-time offsets (ds_read immediates)
#pragma unroll
    for (int q = 0; q < PT / 8; ++q) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a0 = ga[(8 * q + r) * PGP], a1 = ga[(8 * q + r) * PGP + 32];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float bv = eb[(8 * q + r) * PEP + 32 * b];
          tn[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, tn[b][0], 0, 0, 0);
          tn[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, tn[b][1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // e tile is dead: reuse it as the ge output image
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = (k & 3) + 8 * (k >> 2) + 4 * lg;
        es[row * PEP + (cb0 + b) * 32 + li] = acc[b][0][k];
        es[(32 + row) * PEP + (cb0 + b) * 32 + li] = acc[b][1][k];
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int row = lrow + RPP * it;
      if (FULL || r0 + row < E) st4(ge + (r0 + row) * PHT + lc4, ld4(es + row * PEP + lc4));
    }
  };
  if (tb0 < tb1) {
    pidx = perm[clampr(tb0 * PT + lane, Elast)];
    prefetch(tb0);
  }
  for (int64_t tile = tb0; tile < nfull; ++tile) body(std::true_type{}, tile);
  if (nfull < tb1 && nfull >= tb0) body(std::false_type{}, nfull);
  // ---- partial gW1e slab, column sums ----
  float* sl = slab + (size_t)chunk * PS * PHT;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int n = (k & 3) + 8 * (k >> 2) + 4 * lg;
      sl[n * PHT + (cb0 + b) * 32 + li] = tn[b][0][k];
      sl[(32 + n) * PHT + (cb0 + b) * 32 + li] = tn[b][1][k];
    }
  __syncthreads();
  double* red = reinterpret_cast<double*>(es);     // [16 row-slots][2][64] doubles = 16 KB
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[(er * 2 + 0) * PS + ec4 + i] = st.a[i];
    red[(er * 2 + 1) * PS + ec4 + i] = st.b[i];
  }
  double* red2 = red + 16 * 2 * PS;
  if (tid < PT) red2[tid] = sgs;
  __syncthreads();
  if (tid < 2 * PS) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k * 2 * PS + tid];
    partials[(size_t)chunk * 3 * PS + tid] = s;
  } else if (tid < 3 * PS) {
    double s = 0.0;
    if (tid == 2 * PS)
      for (int k = 0; k < PT; ++k) s += red2[k];
    partials[(size_t)chunk * 3 * PS + tid] = s;
  }
}

// out[i] = sum_b slab[b][i], fixed order -> deterministic (gnm_common.h slab_reduce_128); grid total / 128 x 256 threads
__global__ __launch_bounds__(256) void pred_slab_reduce_k(const float* __restrict__ slab, int nslab, int total, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float red[8 * 128];
  const int i0 = blockIdx.x * 128;
  const float4 s_ = slab_reduce_128(slab, nslab, total, i0, red);
  if (threadIdx.x < 32) st4(out + i0 + threadIdx.x * 4, s_);
}

}  // namespace gnm

using namespace gnm;

static inline int64_t cdivp(int64_t a, int64_t b) { return (a + b - 1) / b; }
static size_t pack_f_bytes(int H) { return (size_t)2 * (H / 8) * 64 * 16; }     // W1e NT pack: 2 column blocks x H / 8 k-quads
static size_t pack_b_bytes(int H) { return (size_t)(H / 32) * (PS / 8) * 64 * 16; }   // W1e NN pack: H / 32 column blocks x 8 k-quads

// sized for H = 256 (the larger of the two widths that are built)
extern "C" size_t gnm_predictor_fused_workspace_bytes(void) {
  return pack_f_bytes(2 * PH) + pack_b_bytes(2 * PH) + (size_t)kMaxPartialBlocks * PS * 2 * PH * sizeof(float);
}

template <int PHT>
static int predictor_fused_fwd_impl(int64_t E, const float* e, const float* W1e, int64_t ldw, const float* b1, const float* Pn,
                                    const int32_t* isrc, const int32_t* idst, const int32_t* perm, const float* W2,
                                    const float* b2, float* hid, float* scores, void* ws, hipStream_t st) {
  hipLaunchKernelGGL(pack_wk_k, dim3(8), dim3(256), 0, st, W1e, ldw, 2, PHT / 8, 0, (float*)ws);
  GNM_LAUNCH_CHECK("predictor pack (NT)");
  const int64_t ntiles = cdivp(E, PT);
  const int grid = persistent_grid(ntiles, 4, occ_blocks<pred_fwd_k<true, PHT>>());
  if (hid)
    hipLaunchKernelGGL((pred_fwd_k<true, PHT>), dim3(grid), dim3(kBlock), 0, st, E, e, (const float*)ws, b1, Pn, isrc, idst,
                       perm, W2, b2, hid, scores, cdivp(ntiles, grid));
  else
    hipLaunchKernelGGL((pred_fwd_k<false, PHT>), dim3(grid), dim3(kBlock), 0, st, E, e, (const float*)ws, b1, Pn, isrc, idst,
                       perm, W2, b2, hid, scores, cdivp(ntiles, grid));
  GNM_LAUNCH_CHECK("predictor_fused_fwd");
  return 0;
}

extern "C" int gnm_predictor_fused_fwd(int64_t E, int H, int HS, const float* e, const float* W1e, int64_t ldw,
                                       const float* b1, const float* Pn, const int32_t* isrc, const int32_t* idst,
                                       const int32_t* perm, const float* W2, const float* b2, float* hid,
                                       float* scores, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG((H == PH || H == 2 * PH) && HS == PS, "predictor_fused_fwd: H=%d HS=%d (128 or 256 / 64 are built)", H, HS);
  GNM_CHECK_ARG(E > 0 && e && W1e && ldw >= H && b1 && Pn && isrc && idst && perm && W2 && b2 && scores,
                "predictor_fused_fwd: null/neg argument");
  GNM_CHECK_ARG(ws && ws_bytes >= pack_f_bytes(H), "predictor_fused_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  return H == PH ? predictor_fused_fwd_impl<PH>(E, e, W1e, ldw, b1, Pn, isrc, idst, perm, W2, b2, hid, scores, ws, st)
                 : predictor_fused_fwd_impl<2 * PH>(E, e, W1e, ldw, b1, Pn, isrc, idst, perm, W2, b2, hid, scores, ws, st);
}

template <int PHT>
static int predictor_fused_bwd_impl(int64_t E, float* hid, const float* gscore, const int32_t* perm, const float* W2,
                                    const float* e, const float* W1e, int64_t ldw, float* ge, float* gW1e, float* gsums,
                                    double* partials, void* ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  float* wp = (float*)((char*)ws + pack_f_bytes(PHT));
  float* slab = (float*)((char*)ws + pack_f_bytes(PHT) + pack_b_bytes(PHT));
  hipLaunchKernelGGL(pack_wk_k, dim3(8), dim3(256), 0, st, W1e, ldw, PHT / 32, PS / 8, 1, wp);
  GNM_LAUNCH_CHECK("predictor pack (NN)");
  const int64_t ntiles = cdivp(E, PT);
  const int grid = persistent_grid(ntiles, 4, occ_blocks<pred_bwd_k<PHT>>());
  hipLaunchKernelGGL(pred_bwd_k<PHT>, dim3(grid), dim3(kBlock), 0, st, E, hid, gscore, perm, W2, e, (const float*)wp, ge,
                     slab, partials, cdivp(ntiles, grid));
  GNM_LAUNCH_CHECK("predictor_fused_bwd");
  static_assert((PS * PHT) % 128 == 0, "slab_reduce_128 covers 128 elements per workgroup");
  hipLaunchKernelGGL(pred_slab_reduce_k, dim3(PS * PHT / 128), dim3(256), 0, st, (const float*)slab, grid, PS * PHT, gW1e);
  GNM_LAUNCH_CHECK("predictor_fused_bwd slab reduce");
  // gsums[0:64] = gW2, [64:128] = gb1, [128] = gb2 (129..191: zeros)
  return gnm_reduce_partials(partials, grid, 3, PS, gsums, stream) ? -3 : 0;
}

extern "C" int gnm_predictor_fused_bwd(int64_t E, int H, int HS, float* hid, const float* gscore,
                                       const int32_t* perm, const float* W2, const float* e, const float* W1e,
                                       int64_t ldw, float* ge, float* gW1e, float* gsums, double* partials,
                                       void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG((H == PH || H == 2 * PH) && HS == PS, "predictor_fused_bwd: H=%d HS=%d (128 or 256 / 64 are built)", H, HS);
  GNM_CHECK_ARG(E > 0 && hid && gscore && perm && W2 && e && W1e && ldw >= H && ge && gW1e && gsums && partials,
                "predictor_fused_bwd: null/neg argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_predictor_fused_workspace_bytes(), "predictor_fused_bwd: workspace too small");
  return H == PH ? predictor_fused_bwd_impl<PH>(E, hid, gscore, perm, W2, e, W1e, ldw, ge, gW1e, gsums, partials, ws, stream)
                 : predictor_fused_bwd_impl<2 * PH>(E, hid, gscore, perm, W2, e, W1e, ldw, ge, gW1e, gsums, partials, ws, stream);
}
