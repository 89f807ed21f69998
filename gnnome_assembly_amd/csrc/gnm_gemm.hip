// fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered fmaf
// chain per product -- no reduced-precision path exists or is used).
//
// Replaces the nn.Linear call sites of the hot path and their autograd duals:
//   NT  y  = x W^T (+b)(+relu)      gated_gcn_full.py:107-113, full_graph.py:23-26,
//                                   score_predictor.py:15-17
//   NN  gx = gy W (+resid)          input gradients
//   TN  gW = gy^T x                 weight gradients, contraction over the (millions of)
//                                   rows, split-K with deterministic fp32 partial slabs
//
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 =
// 2x2 MFMA 32x32 blocks, 64 accumulator VGPRs), BK = 32.
// An operand is staged in LDS in one of two images:
//   R ("row"): elem(i,k) = X[i*ld + k]  -> LDS [128][36]  (pitch 36 floats: ds_read_b128
//              fragment reads are bank-conflict-free, rows stay 16-B aligned)
//   C ("col"): elem(i,k) = X[k*ld + i]  -> LDS [32][132]  (ds_read_b32, lanes contiguous)
// MFMA 32x32x2 wants, per k-step, A[i][k'] from lane (i = l&31, g = l>>5) with k' = g.
// The contraction index may be permuted freely as long as A and B agree, so lane group g
// takes k = 8q + 4g + r (q = 0..3, r = 0..3) of the tile: one 16-byte LDS read feeds four
// MFMAs for an R operand.
#include "gnm_common.h"

namespace gnm {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH_R = BK + 4;    // 36
constexpr int PITCH_C = BM + 4;    // 132
constexpr int LDS_R = BM * PITCH_R;  // 4608 floats
constexpr int LDS_C = BK * PITCH_C;  // 4224 floats
constexpr int LDS_OP = LDS_R > LDS_C ? LDS_R : LDS_C;

struct GemmArgs {
  int64_t M, N, K;        // logical: C[M,N], contraction K
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  const float* bias;
  const float* resid; int64_t ldr;
  int relu;
  int64_t k_per_split;    // multiple of BK
  float* slab;            // split-K: [splits][M][N] partials (NULL -> direct epilogue)
  bool a_aligned, b_aligned;  // base pointer and leading dimension allow float4 loads
};

// Load one operand tile (128 x 32) from global into registers (4 float4 per thread).
// ROWL: elem(i,k) at X[i*ld+k]; else elem(i,k) at X[k*ld+i].  Interior tiles of a 16-B
// aligned operand take float4 loads; edge tiles fall back to per-element bounds checks
// (the choice is workgroup-uniform).
template <bool ROWL>
__device__ __forceinline__ void load_tile(const float* __restrict__ X, int64_t ld, int64_t i0,
                                          int64_t ilim, int64_t k0, int64_t klim, bool aligned,
                                          float4 (&r)[4]) {
  const int tid = threadIdx.x;
  // VEC: float4 granularity is safe (aligned operand, full k-range, i-limit a multiple of 4 for
  // the column image); rows / column quads beyond the i-limit are zero-filled without touching
  // memory.  Otherwise per-element bounds checks (workgroup-uniform choice).
  const bool VEC = aligned && (k0 + BK <= klim) && (ROWL || (ilim % 4 == 0));
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (ROWL) {
      const int row = (tid >> 3) + 32 * it, kq = (tid & 7) * 4;
      const int64_t i = i0 + row, k = k0 + kq;
      if (VEC) {
        r[it] = (i < ilim) ? ld4(X + i * ld + k) : f4(0.f);
      } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (i < ilim && k + e < klim) ? X[i * ld + k + e] : 0.f;
        r[it] = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
      const int krow = (tid >> 5) + 8 * it, iq = (tid & 31) * 4;
      const int64_t k = k0 + krow, i = i0 + iq;
      if (VEC) {
        r[it] = (i < ilim) ? ld4(X + k * ld + i) : f4(0.f);
      } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k < klim && i + e < ilim) ? X[k * ld + i + e] : 0.f;
        r[it] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
}

template <bool ROWL>
__device__ __forceinline__ void store_tile(float* __restrict__ s, const float4 (&r)[4]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    if (ROWL) {
      const int row = (tid >> 3) + 32 * it, kq = (tid & 7) * 4;
      st4(s + row * PITCH_R + kq, r[it]);
    } else {
      const int krow = (tid >> 5) + 8 * it, iq = (tid & 31) * 4;
      st4(s + krow * PITCH_C + iq, r[it]);
    }
  }
}

// Fragment for MFMA k-steps (q, r=0..3): lane (i = l&31, g = l>>5) takes k = 8q + 4g + r.
template <bool ROWL>
__device__ __forceinline__ float4 read_frag(const float* __restrict__ s, int row0, int q) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, g = lane >> 5;
  const int k = 8 * q + 4 * g;
  if (ROWL) {
    return ld4(s + (row0 + i) * PITCH_R + k);
  } else {
    const float* p = s + k * PITCH_C + row0 + i;
    return make_float4(p[0], p[PITCH_C], p[2 * PITCH_C], p[3 * PITCH_C]);
  }
}

template <bool A_ROWL, bool B_ROWL>
__global__ __launch_bounds__(kBlock) void gemm_f32_k(GemmArgs a) {
  __shared__ float As[LDS_OP];
  __shared__ float Bs[LDS_OP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.y * BM, j0 = (int64_t)blockIdx.x * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * a.k_per_split;
  const int64_t kend = min(a.K, kbeg + a.k_per_split);

  floatx16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  float4 ra[4], rb[4];
  if (kbeg < kend) {
    load_tile<A_ROWL>(a.A, a.lda, i0, a.M, kbeg, kend, a.a_aligned, ra);
    load_tile<B_ROWL>(a.B, a.ldb, j0, a.N, kbeg, kend, a.b_aligned, rb);
  }
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();   // previous tile's fragment reads are done
    store_tile<A_ROWL>(As, ra);
    store_tile<B_ROWL>(Bs, rb);
    __syncthreads();
    if (k0 + BK < kend) {   // prefetch the next tile; in flight during the MFMAs below
      load_tile<A_ROWL>(a.A, a.lda, i0, a.M, k0 + BK, kend, a.a_aligned, ra);
      load_tile<B_ROWL>(a.B, a.ldb, j0, a.N, k0 + BK, kend, a.b_aligned, rb);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 fa[2], fb[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) fa[mi] = read_frag<A_ROWL>(As, wm * 64 + mi * 32, q);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) fb[ni] = read_frag<B_ROWL>(Bs, wn * 64 + ni * 32, q);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].x, fb[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].y, fb[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].z, fb[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].w, fb[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
  }

  // epilogue.  C/D layout of 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const int g = lane >> 5, cl = lane & 31;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int64_t col = j0 + wn * 64 + ni * 32 + cl;
      if (col >= a.N) continue;
      const float bv = (a.bias && !a.slab) ? a.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int64_t row = i0 + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        if (row >= a.M) continue;
        float v = acc[mi][ni][e];
        if (a.slab) {
          a.slab[((int64_t)blockIdx.z * a.M + row) * a.N + col] = v;
        } else {
          v += bv;
          if (a.resid) v += a.resid[row * a.ldr + col];
          if (a.relu) v = fmaxf(v, 0.f);
          a.C[row * a.ldc + col] = v;
        }
      }
    }
}

// C = sum_z slab[z] (+bias +resid, relu), fixed summation order -> deterministic.
__global__ void splitk_reduce_k(GemmArgs a, int splits) {
  const int64_t total = a.M * a.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / a.N, col = i % a.N;
    float v = 0.f;
    for (int z = 0; z < splits; ++z) v += a.slab[(int64_t)z * total + i];
    if (a.bias) v += a.bias[col];
    if (a.resid) v += a.resid[row * a.ldr + col];
    if (a.relu) v = fmaxf(v, 0.f);
    a.C[row * a.ldc + col] = v;
  }
}

static void plan_split(int mode, int64_t M, int64_t N, int64_t K, int* splits, int64_t* kps) {
  *splits = 1;
  *kps = (K + BK - 1) / BK * BK;
  if (mode != GNM_GEMM_TN) return;
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int64_t ktiles = (K + BK - 1) / BK;
  int64_t want = ((int64_t)num_cus() * 4 + tiles - 1) / tiles;  // ~4 workgroups per CU
  int64_t maxs = (ktiles + 15) / 16;                            // >= 16 k-tiles (512 rows) per split
  if (want > maxs) want = maxs;
  if (want > 1024) want = 1024;
  if (want < 1) want = 1;
  int64_t tiles_per = (ktiles + want - 1) / want;
  *kps = tiles_per * BK;
  *splits = (int)((ktiles + tiles_per - 1) / tiles_per);
}

// ------------------------------------------------------------------------------------------
// Skinny weight gradient on the fp32 matrix cores:  C[128, N] = X[K, 128]^T Y[K, N]  (N <= 32, K rows huge) and the
// column sums of X -- e.g. linear_pe (full_graph.py:23): gW = gh^T pe with pe 18 wide, gb = sum gh.  The 128 x 128
// tiled kernel above spends 7/8 of its MFMAs on padding there.  Structure of gnm_encoder.hip's MFMA backward: one wave
// per 16-row tile of X (coalesced load, wave-private LDS image of pitch 132), v_mfma_f32_16x16x4_f32 with
//   A = X[row 4 g + s][16 cb + i]  (transposed read of the image),   B = Y[row 4 g + s][16 qb + i]  (read from global),
// lane (i = l & 15, g = l >> 4), step s = 0..3 -- the contraction order over the tile's rows is free.
// partials[chunk][128 x 32 | 128] in fp64, reduced in a fixed order by tn_skinny_finish_k.
// ------------------------------------------------------------------------------------------
typedef float floatx4s __attribute__((ext_vector_type(4)));
constexpr int SKT = 16, SKW = 128, SKP = SKW + 4, SKQ = 32;
constexpr int SKN = SKW * SKQ + SKW;                  // partial sums per workgroup
constexpr int SK_LDS = kWavesPerBlock * SKN;          // floats (>= the four tile images)
constexpr int kSkinnyMaxBlocks = 1024;

template <int NB>
__global__ __launch_bounds__(kBlock) void tn_skinny128_k(int64_t K, const float* __restrict__ X, int64_t ldx,
                                                         const float* __restrict__ Y, int64_t ldy, int N,
                                                         double* __restrict__ partials, int64_t tiles_per_block) {
  __shared__ __attribute__((aligned(16))) float lds[SK_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int cr = lane >> 5, cc4 = (lane & 31) * 4;
  float* tile = lds + wave * SKT * SKP;
  const int64_t ntiles = (K + SKT - 1) / SKT;
  const int64_t t0 = (int64_t)blockIdx.x * tiles_per_block;
  const int64_t t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  const int64_t Klast = K - 1;
  floatx4s acc[NB][8];
#pragma unroll
  for (int qb = 0; qb < NB; ++qb)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) acc[qb][cb] = (floatx4s){0.f, 0.f, 0.f, 0.f};
  float4 cs = f4(0.f);
  const int64_t niter = (tiles_per_block + kWavesPerBlock - 1) / kWavesPerBlock;
  auto tile_of = [&](int64_t it) __attribute__((always_inline)) { return t0 + wave + it * kWavesPerBlock; };
  auto clampr = [&](int64_t r) __attribute__((always_inline)) { return r < Klast ? r : Klast; };
  float4 xn[8];
  float yn[NB][4];
  auto load_tile = [&](int64_t t) __attribute__((always_inline)) {
    const int64_t r0 = t * SKT;
#pragma unroll
    for (int q = 0; q < 8; ++q) xn[q] = ld4_nt(X + clampr(r0 + 2 * q + cr) * ldx + cc4);
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const int col = 16 * qb + i;
        yn[qb][s_] = Y[clampr(r0 + 4 * g + s_) * ldy + (col < N ? col : 0)];
      }
  };
  load_tile(tile_of(0));
  for (int64_t it = 0; it < niter; ++it) {
    const int64_t t = tile_of(it);
    const int64_t r0 = t * SKT;
    const bool live_tile = t < t1;
    float y[NB][4];
#pragma unroll
    for (int qb = 0; qb < NB; ++qb)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) y[qb][s_] = (16 * qb + i < N) ? yn[qb][s_] : 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool live = live_tile && r0 + 2 * q + cr < K;
      const float4 v = live ? xn[q] : f4(0.f);        // rows past the end contribute nothing
      cs += v;
      st4(tile + (2 * q + cr) * SKP + cc4, v);
    }
    load_tile(tile_of(it + 1));
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        const float a = tile[(4 * g + s_) * SKP + 16 * cb + i];
#pragma unroll
        for (int qb = 0; qb < NB; ++qb) acc[qb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, y[qb][s_], acc[qb][cb], 0, 0, 0);
      }
    __syncthreads();
  }
  // C / D of the 16 x 16 MFMA: row 4 g + e (= X column 16 cb + 4 g + e), column l & 15 (= Y column 16 qb + i)
  float* r = lds + wave * SKN;
  for (int k = lane; k < SKN; k += 64) r[k] = 0.f;     // the unused 16-column block when NB == 1
#pragma unroll
  for (int qb = 0; qb < NB; ++qb)
#pragma unroll
    for (int cb = 0; cb < 8; ++cb)
#pragma unroll
      for (int e = 0; e < 4; ++e) r[(16 * cb + 4 * g + e) * SKQ + 16 * qb + i] = acc[qb][cb][e];
  cs += shfl_xor4(cs, 32);
  if (cr == 0) {
    r[SKW * SKQ + cc4 + 0] = cs.x; r[SKW * SKQ + cc4 + 1] = cs.y;
    r[SKW * SKQ + cc4 + 2] = cs.z; r[SKW * SKQ + cc4 + 3] = cs.w;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < SKN; k += kBlock) {
    double a = 0.0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) a += (double)lds[w * SKN + k];
    partials[(size_t)blockIdx.x * SKN + k] = a;
  }
}

// C[m * ldc + n] = sum_b partials[b][m * 32 + n] (n < N), colsum[m] = sum_b partials[b][128 * 32 + m]; fixed order
__global__ void tn_skinny_finish_k(const double* __restrict__ partials, int nblk, int N, float* __restrict__ C, int64_t ldc,
                                   float* __restrict__ colsum) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= SKN) return;
  const int m = k < SKW * SKQ ? k / SKQ : k - SKW * SKQ, n = k < SKW * SKQ ? k % SKQ : -1;
  if (n >= N || (n < 0 && !colsum)) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int b = 0;
  for (; b + 3 < nblk; b += 4) {
    a0 += partials[(size_t)b * SKN + k];
    a1 += partials[(size_t)(b + 1) * SKN + k];
    a2 += partials[(size_t)(b + 2) * SKN + k];
    a3 += partials[(size_t)(b + 3) * SKN + k];
  }
  for (; b < nblk; ++b) a0 += partials[(size_t)b * SKN + k];
  const float v = (float)((a0 + a1) + (a2 + a3));
  if (n >= 0) C[(int64_t)m * ldc + n] = v;
  else colsum[m] = v;
}

static bool tn_skinny_shape_ok(int64_t M, int64_t N, int64_t K) { return M == SKW && N >= 1 && N <= SKQ && K >= 4096; }
static size_t tn_skinny_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  return tn_skinny_shape_ok(M, N, K) ? (size_t)kSkinnyMaxBlocks * SKN * sizeof(double) : 0;
}
// 1 = done, 0 = not eligible
static int tn_skinny_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                         int64_t ldc, float* colsum, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!tn_skinny_shape_ok(M, N, K) || (uintptr_t)A % 16 != 0 || lda % 4 != 0 || !ws || ws_bytes < tn_skinny_workspace_bytes(M, N, K))
    return 0;
  const int64_t ntiles = (K + SKT - 1) / SKT;
  int grid = persistent_grid(ntiles, 16, N > 16 ? occ_blocks<tn_skinny128_k<2>>() : occ_blocks<tn_skinny128_k<1>>());
  if (grid > kSkinnyMaxBlocks) grid = kSkinnyMaxBlocks;
  const int64_t tpb = (ntiles + grid - 1) / grid;
  if (N > 16) hipLaunchKernelGGL(tn_skinny128_k<2>, dim3(grid), dim3(kBlock), 0, st, K, A, lda, B, ldb, (int)N, (double*)ws, tpb);
  else hipLaunchKernelGGL(tn_skinny128_k<1>, dim3(grid), dim3(kBlock), 0, st, K, A, lda, B, ldb, (int)N, (double*)ws, tpb);
  hipLaunchKernelGGL(tn_skinny_finish_k, dim3((SKN + 255) / 256), dim3(256), 0, st, (const double*)ws, grid, (int)N, C, ldc, colsum);
  return 1;
}

}  // namespace gnm

using namespace gnm;

namespace gnm {   // gnm_fused.hip: the split-mode (bf16x3) route for big-M shapes with 128-multiple other dimensions
size_t gemm_b3_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K);
int gemm_b3_try(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                float* C, int64_t ldc, const float* bias, const float* resid, int64_t ldr, int relu, void* ws,
                size_t ws_bytes, hipStream_t st);
int gemm_b3_tn_colsum_try(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                          int64_t ldc, float* colsum, void* ws, size_t ws_bytes, hipStream_t st);
}

extern "C" size_t gnm_gemm_f32_workspace_bytes(int mode, int64_t M, int64_t N, int64_t K) {
  int splits; int64_t kps;
  plan_split(mode, M, N, K, &splits, &kps);
  const size_t f32 = splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 0;
  const size_t b3 = gemm_b3_workspace_bytes(mode, M, N, K);
  const size_t sk = mode == GNM_GEMM_TN ? tn_skinny_workspace_bytes(M, N, K) : 0;
  const size_t m2 = f32 > b3 ? f32 : b3;
  return m2 > sk ? m2 : sk;
}

extern "C" int gnm_gemm_f32(int mode, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                            const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias,
                            const float* resid, int64_t ldr, int relu, void* ws, size_t ws_bytes,
                            void* stream) {
  GNM_CHECK_ARG(mode >= 0 && mode <= 2, "gemm_f32: mode %d", mode);
  GNM_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && A && B && C, "gemm_f32: null/neg argument");
  if (M == 0 || N == 0) return 0;
  if (mode == GNM_GEMM_TN && !bias && !resid && !relu &&
      tn_skinny_try(M, N, K, A, lda, B, ldb, C, ldc, nullptr, ws, ws_bytes, (hipStream_t)stream)) {
    GNM_LAUNCH_CHECK("gemm_f32 (skinny TN)");
    return 0;
  }
  {
    const int rc = gemm_b3_try(mode, M, N, K, A, lda, B, ldb, C, ldc, bias, resid, ldr, relu, ws, ws_bytes, (hipStream_t)stream);
    if (rc < 0) { GNM_LAUNCH_CHECK("gemm_f32 (split route)"); return rc; }
    if (rc > 0) return 0;
  }
  GemmArgs a;
  a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
  a.bias = bias; a.resid = resid; a.ldr = ldr; a.relu = relu; a.slab = nullptr;
  int splits;
  plan_split(mode, M, N, K, &splits, &a.k_per_split);
  if (splits > 1) {
    const size_t need = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
    GNM_CHECK_ARG(ws && ws_bytes >= need, "gemm_f32: workspace %zu < %zu bytes", ws_bytes, need);
    a.slab = (float*)ws;
  }
  const bool a_rowl = (mode != GNM_GEMM_TN);   // A[M,K] row-major vs A[K,M]
  const bool b_rowl = (mode == GNM_GEMM_NT);   // B[N,K] row-major vs B[K,N]
  a.a_aligned = ((uintptr_t)A % 16 == 0) && lda % 4 == 0;
  a.b_aligned = ((uintptr_t)B % 16 == 0) && ldb % 4 == 0;
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)splits);
  hipStream_t st = (hipStream_t)stream;
#define GNM_GEMM_LAUNCH(AR, BR) hipLaunchKernelGGL((gemm_f32_k<AR, BR>), grid, dim3(kBlock), 0, st, a)
  if (a_rowl && b_rowl) GNM_GEMM_LAUNCH(true, true);
  else if (a_rowl && !b_rowl) GNM_GEMM_LAUNCH(true, false);
  else GNM_GEMM_LAUNCH(false, false);
#undef GNM_GEMM_LAUNCH
  GNM_LAUNCH_CHECK("gemm_f32");
  if (splits > 1) {
    int64_t g = (M * N + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(splitk_reduce_k, dim3((unsigned)g), dim3(256), 0, st, a, splits);
    GNM_LAUNCH_CHECK("gemm_f32 split-K reduce");
  }
  return 0;
}

// C[M,N] = A[K,M]^T B[K,N] and colsum[m] = sum_k A[k][m]: weight and bias gradient of a Linear in one call
// (one pass over A where the split-mode kernel applies, gnm_gemm_f32 + gnm_colsum_f32 otherwise).
extern "C" size_t gnm_gemm_tn_colsum_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  const size_t a = gnm_gemm_f32_workspace_bytes(GNM_GEMM_TN, M, N, K), b = gnm_colsum_workspace_bytes(K, M);
  return a > b ? a : b;
}

extern "C" int gnm_gemm_tn_colsum(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B,
                                  int64_t ldb, float* C, int64_t ldc, float* colsum, void* ws, size_t ws_bytes,
                                  void* stream) {
  GNM_CHECK_ARG(M > 0 && N > 0 && K >= 0 && A && B && C && colsum, "gemm_tn_colsum: null/neg argument");
  GNM_CHECK_ARG(ws_bytes >= gnm_gemm_tn_colsum_workspace_bytes(M, N, K) && (ws || ws_bytes == 0),
                "gemm_tn_colsum: workspace too small");
  if (tn_skinny_try(M, N, K, A, lda, B, ldb, C, ldc, colsum, ws, ws_bytes, (hipStream_t)stream)) {
    GNM_LAUNCH_CHECK("gemm_tn_colsum (skinny)");
    return 0;
  }
  const int rc = gemm_b3_tn_colsum_try(M, N, K, A, lda, B, ldb, C, ldc, colsum, ws, ws_bytes, (hipStream_t)stream);
  if (rc < 0) { GNM_LAUNCH_CHECK("gemm_tn_colsum (split route)"); return rc; }
  if (rc > 0) return 0;
  const int r2 = gnm_gemm_f32(GNM_GEMM_TN, M, N, K, A, lda, B, ldb, C, ldc, nullptr, nullptr, 0, 0, ws, ws_bytes, stream);
  return r2 ? r2 : gnm_colsum_f32(K, M, A, lda, colsum, ws, ws_bytes, stream);
}
