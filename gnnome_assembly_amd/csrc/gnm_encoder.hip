// Edge-feature encoder (full_graph.py:24-26): e0 = linear2_edge(relu(linear1_edge(e_raw))),
// forward and backward, for the reference's sizes edge_features = 2, hidden_edge_features = 16
// (hyperparameters.py:9-10) and H = 128 / 256.  Both are single HBM passes over the [E,H] tensor on the fp32 matrix cores
// (the VALU kernels of round 1 -- 24 FMAs per output element forward, a 16-shuffle reduce-scatter backward: 1.3 / 2.3 ms
// against 0.86 / 0.85 -- were removed in round 5).
// Rows are in internal (destination-sorted) order; e_raw is read through `perm`.
#include "gnm_common.h"

namespace gnm {

constexpr int EH = 128;      // hidden width
constexpr int EQ = 16;       // hidden_edge_features
constexpr int EG = EH / 4;   // lanes per row
constexpr int ENP = EH * EQ + EH + 2 * EQ + EQ;   // gW2 | gb2 | gW1 | gb1 = 2224 partial sums

// ------------------------------------------------------------------------------------------
// Backward on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate): reads ge0 once and produces
// every encoder gradient; partials[chunk][ENP] (fp64) = per-workgroup sums of gW2[c][q], gb2[c], gW1[q][0..1], gb1[q].
// (A VALU formulation issues ~250 instructions per row and lane -- 16 hidden units x 8 FMAs + a 16-shuffle reduce-scatter -- and
// ran at 1.8 TB/s.)  Both contractions are small dense products per 16-row tile:
//   NN  ga1pre[16 rows][16 q]  = ge0[16 rows][128 c] W2[128 c][16 q]     32 MFMAs, contraction over c
//   TN  gW2[128 c][16 q]      += ge0[16 rows][128 c]^T a1[16 rows][16 q]  32 MFMAs, contraction over rows
// One wave per 16-row tile; the tile is loaded coalesced (512-byte rows), parked in a wave-private LDS image of
// pitch 132 floats and read back in the two operand layouts.  The order of a contraction is free, so both are
// arranged for reads without shuffles: lane (i = l & 15, g = l >> 4)
//   NN step (j, c): A = tile[i][16 j + 4 g + c] (the c-th word of ONE ds_read_b128), B = W2[16 j + 4 g + c][q = i]
//   TN step s, column block cb: A = tile[4 g + s][16 cb + i], B = a1[4 g + s][q = i] -- exactly the four rows the
//   lane holds of the NN result (C layout of the 16 x 16 MFMA: row 4 g + r, column l & 15).
// relu mask, gW1, gb1 in that C layout; gb2 from the coalesced load registers.  Partials as above.
// ------------------------------------------------------------------------------------------
typedef float floatx4m __attribute__((ext_vector_type(4)));
constexpr int ET = 16;            // rows per wave tile
// EHT = the hidden width: 128, or 256 (the reference's default dim_latent) = the same recipe with twice the column blocks
template <int EHT> struct EncDims {
  static constexpr int NCB = EHT / 16;                          // 16-column blocks of a row
  static constexpr int NH = EHT / 128;                          // 128-column passes of the coalesced row image
  static constexpr int EPL = EHT + 4;                           // LDS pitch (floats): conflict-free b128 rows and transposed b32 reads
  static constexpr int ENP = EHT * EQ + EHT + 2 * EQ + EQ;      // gW2 | gb2 | gW1 | gb1 partial sums
  static constexpr int LDS = kWavesPerBlock * ET * EPL > kWavesPerBlock * ENP ? kWavesPerBlock * ET * EPL : kWavesPerBlock * ENP;
};

template <int EHT>
__global__ __launch_bounds__(kBlock) void edge_encoder_bwd_mfma_k(int64_t E, const float* __restrict__ ge0,
                                                                  const float* __restrict__ e_raw,
                                                                  const int32_t* __restrict__ perm,
                                                                  const float* __restrict__ W1,
                                                                  const float* __restrict__ b1,
                                                                  const float* __restrict__ W2,
                                                                  double* __restrict__ partials,
                                                                  int64_t tiles_per_block) {
  using D = EncDims<EHT>;
  constexpr int NCB = D::NCB, NH = D::NH, EPL = D::EPL, ENPT = D::ENP;
  __shared__ __attribute__((aligned(16))) float lds[D::LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;           // MFMA operand / result coordinates
  const int cr = lane >> 5, cc4 = (lane & 31) * 4;  // coalesced load coordinates: rows cr, cr + 2, .. ; columns cc4 .. +3 of each 128-column pass
  float* tile = lds + wave * ET * EPL;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + ET - 1) / ET;
  const int64_t t0 = (int64_t)chunk * tiles_per_block;
  const int64_t t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  const int64_t Elast = E - 1;
  // B operand of the NN product: W2[16 j + 4 g + c][q = i], stationary
  float w2r[4 * NCB];
#pragma unroll
  for (int j = 0; j < NCB; ++j)
#pragma unroll
    for (int c = 0; c < 4; ++c) w2r[4 * j + c] = W2[(16 * j + 4 * g + c) * EQ + i];
  const float w1a = W1[2 * i], w1b = W1[2 * i + 1], bq = b1[i];
  floatx4m gw2[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) gw2[cb] = (floatx4m){0.f, 0.f, 0.f, 0.f};
  float4 gb2[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) gb2[h] = f4(0.f);
  float gw1_0 = 0.f, gw1_1 = 0.f, gb1_ = 0.f;
  // software pipeline: perm two tiles ahead, the tile's rows and features one tile ahead; every wave of the workgroup
  // runs the same number of iterations (tiles past the end are clamped and contribute zeros), so the barriers match
  const int64_t niter = (tiles_per_block + kWavesPerBlock - 1) / kWavesPerBlock;
  auto tile_of = [&](int64_t it) __attribute__((always_inline)) { return t0 + wave + it * kWavesPerBlock; };
  auto clampr = [&](int64_t r) __attribute__((always_inline)) { return r < Elast ? r : Elast; };
  float4 gn[NH][8];
  int kn[4];
  float2 xn[4];
  auto load_rows = [&](int64_t t) __attribute__((always_inline)) {
    const int64_t r0 = t * ET;
#pragma unroll
    for (int it = 0; it < 8; ++it)
#pragma unroll
      for (int h = 0; h < NH; ++h) gn[h][it] = ld4_nt(ge0 + clampr(r0 + 2 * it + cr) * EHT + 128 * h + cc4);
  };
  auto load_perm = [&](int64_t t) __attribute__((always_inline)) {
    const int64_t r0 = t * ET;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) kn[s_] = perm[clampr(r0 + 4 * g + s_)];
  };
  auto load_x = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) xn[s_] = *reinterpret_cast<const float2*>(e_raw + 2 * (int64_t)kn[s_]);
  };
  load_perm(tile_of(0));
  load_rows(tile_of(0));
  load_x();
  load_perm(tile_of(1));
  for (int64_t it = 0; it < niter; ++it) {
    const int64_t t = tile_of(it);
    const int64_t r0 = t * ET;
    const bool live_tile = t < t1;
    float2 x[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) x[s_] = xn[s_];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool live = live_tile && r0 + 2 * q + cr < E;
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const float4 v = live ? gn[h][q] : f4(0.f);   // rows past the end contribute nothing anywhere below
        gb2[h] += v;
        st4(tile + (2 * q + cr) * EPL + 128 * h + cc4, v);
      }
    }
    load_x();                                          // features of the next tile (its perm was requested a tile ago)
    load_rows(tile_of(it + 1));
    load_perm(tile_of(it + 2));
    __syncthreads();
    // ---- NN: ga1pre = ge0 W2 ----
    floatx4m cacc = (floatx4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCB; ++j) {
      const float4 a4 = ld4(tile + i * EPL + 16 * j + 4 * g);
      cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, w2r[4 * j + 0], cacc, 0, 0, 0);
      cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, w2r[4 * j + 1], cacc, 0, 0, 0);
      cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, w2r[4 * j + 2], cacc, 0, 0, 0);
      cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, w2r[4 * j + 3], cacc, 0, 0, 0);
    }
    // ---- relu of linear1_edge for rows 4 g + s, unit q = i; its backward; gW1, gb1 ----
    float a1[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const float ap = fmaf(w1a, x[s_].x, fmaf(w1b, x[s_].y, bq));
      a1[s_] = fmaxf(ap, 0.f);
      const float ga = ap > 0.f ? cacc[s_] : 0.f;
      gw1_0 = fmaf(ga, x[s_].x, gw1_0);
      gw1_1 = fmaf(ga, x[s_].y, gw1_1);
      gb1_ += ga;
    }
    // ---- TN: gW2[16 cb + 4 g + r][q] += sum_rows ge0[row][16 cb + ..] a1[row][q] ----
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
        gw2[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(tile[(4 * g + s_) * EPL + 16 * cb + i], a1[s_], gw2[cb], 0, 0, 0);
    }
    __syncthreads();                                   // the tile image is rewritten by the next iteration
  }
  // ---- per-wave results -> red[wave][ENP] (the tile images are dead), then the 4 waves in fp64 ----
  float* r = lds + wave * ENPT;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) r[(16 * cb + 4 * g + e) * EQ + i] = gw2[cb][e];
#pragma unroll
  for (int h = 0; h < NH; ++h) {
    gb2[h] += shfl_xor4(gb2[h], 32);                   // the two row slots of the coalesced layout
    if (cr == 0) {
      r[EHT * EQ + 128 * h + cc4 + 0] = gb2[h].x; r[EHT * EQ + 128 * h + cc4 + 1] = gb2[h].y;
      r[EHT * EQ + 128 * h + cc4 + 2] = gb2[h].z; r[EHT * EQ + 128 * h + cc4 + 3] = gb2[h].w;
    }
  }
  gw1_0 += __shfl_xor(gw1_0, 16, 64); gw1_1 += __shfl_xor(gw1_1, 16, 64); gb1_ += __shfl_xor(gb1_, 16, 64);
  gw1_0 += __shfl_xor(gw1_0, 32, 64); gw1_1 += __shfl_xor(gw1_1, 32, 64); gb1_ += __shfl_xor(gb1_, 32, 64);
  if (g == 0) {
    r[EHT * EQ + EHT + 2 * i + 0] = gw1_0;
    r[EHT * EQ + EHT + 2 * i + 1] = gw1_1;
    r[EHT * EQ + EHT + 2 * EQ + i] = gb1_;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ENPT; k += kBlock) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) acc += (double)lds[w * ENPT + k];
    partials[(size_t)chunk * ENPT + k] = acc;
  }
}

// ------------------------------------------------------------------------------------------
// Forward on the fp32 matrix cores, same recipe: per 16-row tile  e0[16][H] = a1[16][16] W2^T + b2  is H / 4
// v_mfma_f32_16x16x4_f32 (contraction over the 16 hidden units, 4 per instruction); lane (i = l & 15, g = l >> 4):
//   A = a1[row i][4 s + g] -- computed in place from the row's two features (same fmaf nesting as the VALU kernel, so
//       the relu decisions are bit-identical),   B = W2[16 cb + i][4 s + g] (stationary registers),
//   C = e0[row 4 g + r][16 cb + i] -> wave-private LDS image -> coalesced 512-byte pieces of the rows (+ b2) to HBM.
// ------------------------------------------------------------------------------------------
template <int EHT>
__global__ __launch_bounds__(kBlock) void edge_encoder_fwd_mfma_k(int64_t E, const float* __restrict__ e_raw,
                                                                  const int32_t* __restrict__ perm,
                                                                  const float* __restrict__ W1,
                                                                  const float* __restrict__ b1,
                                                                  const float* __restrict__ W2,
                                                                  const float* __restrict__ b2,
                                                                  float* __restrict__ e0, int64_t tiles_per_block) {
  using D = EncDims<EHT>;
  constexpr int NCB = D::NCB, NH = D::NH, EPL = D::EPL;
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock * ET * EPL];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int cr = lane >> 5, cc4 = (lane & 31) * 4;
  float* tile = lds + wave * ET * EPL;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t ntiles = (E + ET - 1) / ET;
  const int64_t t0 = (int64_t)chunk * tiles_per_block;
  const int64_t t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  const int64_t Elast = E - 1;
  float w2r[4 * NCB];                          // W2[16 cb + i][4 s + g]
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) w2r[4 * cb + s_] = W2[(16 * cb + i) * EQ + 4 * s_ + g];
  float w1a[4], w1b[4], bq[4];                 // hidden units 4 s + g
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) { w1a[s_] = W1[2 * (4 * s_ + g)]; w1b[s_] = W1[2 * (4 * s_ + g) + 1]; bq[s_] = b1[4 * s_ + g]; }
  float4 bb[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) bb[h] = ld4(b2 + 128 * h + cc4);
  const int64_t niter = (tiles_per_block + kWavesPerBlock - 1) / kWavesPerBlock;
  auto tile_of = [&](int64_t it) __attribute__((always_inline)) { return t0 + wave + it * kWavesPerBlock; };
  auto clampr = [&](int64_t r) __attribute__((always_inline)) { return r < Elast ? r : Elast; };
  // perm two tiles ahead, the row's features one tile ahead (a dependent pair of loads per row)
  int kn = perm[clampr(tile_of(0) * ET + i)];
  float2 xn = *reinterpret_cast<const float2*>(e_raw + 2 * (int64_t)kn);
  kn = perm[clampr(tile_of(1) * ET + i)];
  for (int64_t it = 0; it < niter; ++it) {
    const int64_t t = tile_of(it);
    const int64_t r0 = t * ET;
    const float x0 = xn.x, x1 = xn.y;
    xn = *reinterpret_cast<const float2*>(e_raw + 2 * (int64_t)kn);
    kn = perm[clampr(tile_of(it + 2) * ET + i)];
    float a1[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) a1[s_] = fmaxf(fmaf(w1a[s_], x0, fmaf(w1b[s_], x1, bq[s_])), 0.f);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      floatx4m c = (floatx4m){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s_], w2r[4 * cb + s_], c, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[(4 * g + e) * EPL + 16 * cb + i] = c[e];
    }
    __syncthreads();
    if (t < t1) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int64_t row = r0 + 2 * q + cr;
        if (row < E) {
#pragma unroll
          for (int h = 0; h < NH; ++h)
            st4_nt(e0 + row * EHT + 128 * h + cc4, ld4(tile + (2 * q + cr) * EPL + 128 * h + cc4) + bb[h]);
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace gnm

using namespace gnm;

extern "C" int gnm_edge_encoder_fwd(int64_t E, int H, int F, int Q, const float* e_raw, const int32_t* perm,
                                    const float* W1, const float* b1, const float* W2, const float* b2,
                                    float* e0, void* stream) {
  GNM_CHECK_ARG((H == EH || H == 2 * EH) && F == 2 && Q == EQ, "edge_encoder_fwd: built for H=128 or 256, edge_features=2, hidden=16 (got %d,%d,%d)", H, F, Q);
  GNM_CHECK_ARG(E >= 0 && e_raw && perm && W1 && b1 && W2 && b2 && e0, "edge_encoder_fwd: null/neg argument");
  if (H == 2 * EH) {
    if (E == 0) return 0;
    const int64_t ntiles = (E + ET - 1) / ET;
    const int grid = persistent_grid(ntiles, 16, occ_blocks<edge_encoder_fwd_mfma_k<2 * EH>>());
    hipLaunchKernelGGL(edge_encoder_fwd_mfma_k<2 * EH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, E, e_raw, perm, W1, b1,
                       W2, b2, e0, (ntiles + grid - 1) / grid);
    GNM_LAUNCH_CHECK("edge_encoder_fwd (256)");
    return 0;
  }
  if (E == 0) return 0;
  const int64_t ntiles = (E + ET - 1) / ET;
  const int grid = persistent_grid(ntiles, 16, occ_blocks<edge_encoder_fwd_mfma_k<EH>>());
  hipLaunchKernelGGL(edge_encoder_fwd_mfma_k<EH>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, E, e_raw, perm, W1, b1,
                     W2, b2, e0, (ntiles + grid - 1) / grid);
  GNM_LAUNCH_CHECK("edge_encoder_fwd");
  return 0;
}

// partials must hold gnm_max_partial_blocks() * (H * 16 + H + 48) doubles; the size returned covers H = 256
extern "C" size_t gnm_edge_encoder_bwd_workspace_bytes(void) { return (size_t)kMaxPartialBlocks * EncDims<2 * EH>::ENP * sizeof(double); }

extern "C" int gnm_edge_encoder_bwd(int64_t E, int H, int F, int Q, const float* ge0, const float* e_raw,
                                    const int32_t* perm, const float* W1, const float* b1, const float* W2,
                                    float* gW1, float* gb1, float* gW2, float* gb2, void* ws, size_t ws_bytes,
                                    void* stream) {
  GNM_CHECK_ARG((H == EH || H == 2 * EH) && F == 2 && Q == EQ, "edge_encoder_bwd: built for H=128 or 256, edge_features=2, hidden=16 (got %d,%d,%d)", H, F, Q);
  GNM_CHECK_ARG(E >= 0 && ge0 && e_raw && perm && W1 && b1 && W2 && gW1 && gb1 && gW2 && gb2, "edge_encoder_bwd: null/neg argument");
  hipStream_t st = (hipStream_t)stream;
  int grid;
  const int enp = H * EQ + H + 2 * EQ + EQ;
  if (H == 2 * EH) {
    const int64_t ntiles = (E + ET - 1) / ET;
    grid = persistent_grid(ntiles, 16, occ_blocks<edge_encoder_bwd_mfma_k<2 * EH>>());
    GNM_CHECK_ARG(ws && ws_bytes >= (size_t)grid * enp * sizeof(double), "edge_encoder_bwd: workspace too small");
    hipLaunchKernelGGL(edge_encoder_bwd_mfma_k<2 * EH>, dim3(grid), dim3(kBlock), 0, st, E, ge0, e_raw, perm, W1, b1, W2,
                       (double*)ws, (ntiles + grid - 1) / grid);
  } else {
    const int64_t ntiles = (E + ET - 1) / ET;
    grid = persistent_grid(ntiles, 16, occ_blocks<edge_encoder_bwd_mfma_k<EH>>());
    GNM_CHECK_ARG(ws && ws_bytes >= (size_t)grid * ENP * sizeof(double), "edge_encoder_bwd: workspace too small");
    hipLaunchKernelGGL(edge_encoder_bwd_mfma_k<EH>, dim3(grid), dim3(kBlock), 0, st, E, ge0, e_raw, perm, W1, b1, W2,
                       (double*)ws, (ntiles + grid - 1) / grid);
  }
  GNM_LAUNCH_CHECK("edge_encoder_bwd");
  // gW2 | gb2 | gW1 | gb1 are contiguous in the partial rows; reduce each piece into its tensor
  const double* p = (const double*)ws;
  struct Piece { int off, n; float* out; } pieces[4] = {
      {0, H * EQ, gW2}, {H * EQ, H, gb2}, {H * EQ + H, 2 * EQ, gW1}, {H * EQ + H + 2 * EQ, EQ, gb1}};
  for (const Piece& pc : pieces) {
    // strided view: row stride enp, `n` columns starting at `off`
    if (reduce_partials_strided(p, grid, enp, pc.off, pc.n, pc.out, stream)) return -3;
  }
  return 0;
}
