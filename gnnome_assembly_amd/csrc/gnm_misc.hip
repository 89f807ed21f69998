// Predictor epilogue, loss, and the small reductions / gathers around the layer stack (gfx950).
//
//   predictor_score_*   score_predictor.py:12-25 in the split-W1 form
//                       relu(Ps[src] + Pd[dst] + e W1e^T + b1) . W2 + b2   (no [E,3H] concat)
//   bce_fwd_bwd         train.py:210-211,253-255 BCEWithLogitsLoss(pos_weight), mean, + dloss/dlogit
//   colsum / seg_sum_rows / gather_rows / relu_mask: bias gradients, segmented sums of the
//                       predictor's [E,64] gradient rows, edge-id -> internal-order gather.
#include "gnm_common.h"

namespace gnm {

// hid[j] += Ps[isrc j] + Pd[idst j]; score[perm j] = W2 . relu(hid[j]) + b2
template <int HS>
__global__ __launch_bounds__(kBlock) void predictor_score_fwd_k(
    int64_t E, float* __restrict__ hid, const float* __restrict__ Pn,
    const int32_t* __restrict__ isrc, const int32_t* __restrict__ idst,
    const float* __restrict__ W2, const float* __restrict__ b2, const int32_t* __restrict__ perm,
    float* __restrict__ scores) {
  constexpr int G = HS / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, c4 = (lane % G) * 4;
  const float4 w2 = ld4(W2 + c4);
  const float bias = b2[0];
  const int64_t stride = (int64_t)gridDim.x * kWavesPerBlock * RPW;
  for (int64_t j = ((int64_t)blockIdx.x * kWavesPerBlock + wave) * RPW + sub; j < E; j += stride) {
    const int64_t s = isrc[j], d = idst[j];
    float4 v = ld4(hid + j * HS + c4);
    v = v + ld4(Pn + s * (2 * HS) + c4) + ld4(Pn + d * (2 * HS) + HS + c4);
    st4(hid + j * HS + c4, v);
    const float4 r = relu4(v);
    float dot = r.x * w2.x + r.y * w2.y + r.z * w2.z + r.w * w2.w;
#pragma unroll
    for (int off = 1; off < G; off <<= 1) dot += __shfl_xor(dot, off, 64);
    if ((lane % G) == 0) scores[perm[j]] = dot + bias;
  }
}

// ghid = gscore[perm j] * W2 * [hid > 0] (in place); partials: row0 = sum gscore*relu(hid), row1[0] = sum gscore
template <int HS>
__global__ __launch_bounds__(kBlock) void predictor_score_bwd_k(
    int64_t E, float* __restrict__ hid, const float* __restrict__ gscore,
    const float* __restrict__ W2, const int32_t* __restrict__ perm, double* __restrict__ partials,
    int64_t rows_per_block) {
  constexpr int G = HS / 4, RPW = 64 / G;
  __shared__ double lds[kWavesPerBlock * 2 * HS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t r0 = (int64_t)chunk * rows_per_block;
  const int64_t r1 = min(E, r0 + rows_per_block);
  const float4 w2 = ld4(W2 + c4);
  Stat4 st;
  st.zero();
  for (int64_t j = r0 + wave * RPW + sub; j < r1; j += kWavesPerBlock * RPW) {
    const float gs = gscore[perm[j]];
    const float4 v = ld4(hid + j * HS + c4);
    st4(hid + j * HS + c4, gate4(v, w2 * gs));
    // row 0: gs*relu(v) per channel; row 1: gs once per row (channel 0 only)
    st.add(relu4(v) * gs, make_float4(c4 == 0 ? gs : 0.f, 0.f, 0.f, 0.f));
  }
  block_stat_store<HS>(st, lds, partials, chunk);
}

// out[i] = sum_b partials[b*total + i] (fp64 -> fp32); one workgroup per 16 columns,
// 16 row-groups x 16 columns, fixed order -> deterministic
__global__ __launch_bounds__(256) void reduce_partials_k(const double* __restrict__ partials, int nblk,
                                                          int total, float* __restrict__ out,
                                                          int64_t stride = -1, int off = 0) {
  __shared__ double red[16][17];
  const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c;
  if (stride < 0) stride = total;          // dense rows
  // blockIdx.y = batch (gnm::reduce_partials_batched): batch g reads its own nblk rows and writes its own `total` outputs
  partials += (size_t)blockIdx.y * nblk * stride;
  out += (size_t)blockIdx.y * total;
  double acc = 0.0;
  if (col < total)
    for (int b = r; b < nblk; b += 16) acc += partials[(size_t)b * stride + off + col];
  red[r][c] = acc;
  __syncthreads();
  if (r == 0 && col < total) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][c];
    out[col] = (float)s;
  }
}

// out[v*ldo + c] = sum_{m in [ptr[v],ptr[v+1])} X[(pos ? pos[m] : m)*W + c]
template <int W>
__global__ __launch_bounds__(kBlock) void seg_sum_rows_k(int64_t N, const float* __restrict__ X,
                                                         const int32_t* __restrict__ ptr,
                                                         const int32_t* __restrict__ pos,
                                                         float* __restrict__ out, int64_t ldo,
                                                         int64_t nodes_per_block) {
  constexpr int G = W / 4, RPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sub = lane / G, c4 = (lane % G) * 4;
  const int chunk = xcd_chunk(blockIdx.x, gridDim.x);
  const int64_t v0 = (int64_t)chunk * nodes_per_block;
  const int64_t v1 = min(N, v0 + nodes_per_block);
  for (int64_t v = v0 + wave; v < v1; v += kWavesPerBlock) {
    const int a = ptr[v], b = ptr[v + 1];
    float4 acc = f4(0.f);
    for (int64_t m = a + sub; m < b; m += RPW) {
      const int64_t j = pos ? (int64_t)pos[m] : m;
      acc += ld4(X + j * W + c4);
    }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) acc += shfl_xor4(acc, off);
    if (sub == 0) st4(out + v * ldo + c4, acc);
  }
}

// stage 1 of column sums: ws[b][c] = sum over the block's row chunk (fp64 accumulation).
// VEC: W % 4 == 0 and 16-B aligned rows -> one float4 column group per thread, blockDim.x =
// W/4 column groups x blockDim.y rows; otherwise one column per thread.
template <bool VEC>
__global__ void colsum_stage1_k(int64_t M, int64_t W, const float* __restrict__ X, int64_t ld,
                                double* __restrict__ ws, int64_t rows_per_block) {
  extern __shared__ double sm[];   // [blockDim.y][W] (VEC only)
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  if (VEC) {
    const int cg = threadIdx.x, ry = threadIdx.y, ny = blockDim.y;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (int64_t r = r0 + ry; r < r1; r += ny) {
      const float4 v = ld4(X + r * ld + 4 * cg);
      a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    double* row = sm + (size_t)ry * W + 4 * cg;
    row[0] = a0; row[1] = a1; row[2] = a2; row[3] = a3;
    __syncthreads();
    const int tid = ry * blockDim.x + cg, nt = blockDim.x * ny;
    for (int64_t c = tid; c < W; c += nt) {
      double acc = 0.0;
      for (int y = 0; y < ny; ++y) acc += sm[(size_t)y * W + c];
      ws[(int64_t)blockIdx.x * W + c] = acc;
    }
  } else {
    for (int64_t c = threadIdx.x; c < W; c += blockDim.x) {
      double acc = 0.0;
      for (int64_t r = r0; r < r1; ++r) acc += (double)X[r * ld + c];
      ws[(int64_t)blockIdx.x * W + c] = acc;
    }
  }
}

__global__ void gather_rows_k(int64_t M, int64_t W, const float* __restrict__ X,
                              const int32_t* __restrict__ idx, float* __restrict__ out) {
  const int64_t total = M * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = i / W, c = i % W;
    out[i] = X[(int64_t)idx[j] * W + c];
  }
}

__global__ void relu_mask_k(int64_t n, float* __restrict__ x, const float* __restrict__ ref) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = ref[i] > 0.f ? x[i] : 0.f;
}

__device__ __forceinline__ float softplusf_(float x) {
  // log(1 + exp(x)) without overflow: max(x,0) + log1p(exp(-|x|))
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}

__global__ __launch_bounds__(kBlock) void bce_fwd_bwd_k(int64_t E, const float* __restrict__ x,
                                                        const float* __restrict__ y, float pw,
                                                        float inv_e, float* __restrict__ gscore,
                                                        double* __restrict__ ws) {
  __shared__ double red[kBlock];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < E; i += (int64_t)gridDim.x * kBlock) {
    const float xi = x[i], yi = y[i];
    const float p = sigmoid_ieee_(xi);
    acc += (double)(pw * yi * softplusf_(-xi) + (1.f - yi) * softplusf_(xi));
    gscore[i] = (-pw * yi * (1.f - p) + (1.f - yi) * p) * inv_e;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = kBlock / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
}

__global__ void bce_finalize_k(const double* __restrict__ ws, int nblk, double inv_e,
                               float* __restrict__ loss_out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) acc += ws[b];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = (float)(red[0] * inv_e);
}

}  // namespace gnm

using namespace gnm;

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int ew_grid2(int64_t items, int block) {
  int64_t g = cdiv(items, block);
  const int64_t cap = (int64_t)num_cus() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

#define GNM_DISPATCH_W(W, ...)                                     \
  switch (W) {                                                     \
    case 32: { constexpr int WW = 32; __VA_ARGS__; } break;        \
    case 64: { constexpr int WW = 64; __VA_ARGS__; } break;        \
    case 128: { constexpr int WW = 128; __VA_ARGS__; } break;      \
    case 256: { constexpr int WW = 256; __VA_ARGS__; } break;      \
    default: ::gnm::set_error("row width %d unsupported (32, 64, 128, 256)", (int)(W)); return -1; \
  }

extern "C" int gnm_predictor_score_fwd(int64_t E, int HS, float* hid, const float* Pn,
                                       const int32_t* isrc, const int32_t* idst, const float* W2,
                                       const float* b2, const int32_t* perm, float* scores,
                                       void* stream) {
  GNM_CHECK_ARG(E >= 0 && hid && Pn && isrc && idst && W2 && b2 && perm && scores, "predictor_score_fwd: null/neg argument");
  GNM_DISPATCH_W(HS, hipLaunchKernelGGL(predictor_score_fwd_k<WW>, dim3(ew_grid2(E * (WW / 4), kBlock)),
                                        dim3(kBlock), 0, (hipStream_t)stream, E, hid, Pn, isrc, idst,
                                        W2, b2, perm, scores));
  GNM_LAUNCH_CHECK("predictor_score_fwd");
  return 0;
}

extern "C" int gnm_predictor_score_bwd(int64_t E, int HS, float* hid, const float* gscore,
                                       const float* W2, const int32_t* perm, double* partials,
                                       int* nblk_out, void* stream) {
  GNM_CHECK_ARG(E >= 0 && hid && gscore && W2 && perm && partials && nblk_out, "predictor_score_bwd: null/neg argument");
  GNM_DISPATCH_W(HS, {
    const int grid = persistent_grid(E, 256, occ_blocks<predictor_score_bwd_k<WW>>());
    const int64_t rpb = cdiv(E, grid);
    hipLaunchKernelGGL(predictor_score_bwd_k<WW>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, E, hid,
                       gscore, W2, perm, partials, rpb);
    *nblk_out = grid;
  });
  GNM_LAUNCH_CHECK("predictor_score_bwd");
  return 0;
}

extern "C" int gnm_reduce_partials(const double* partials, int nblk, int rows, int W, float* out,
                                   void* stream) {
  GNM_CHECK_ARG(partials && nblk > 0 && rows > 0 && W > 0 && out, "reduce_partials: bad argument");
  const int total = rows * W;
  hipLaunchKernelGGL(reduce_partials_k, dim3((total + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                     partials, nblk, total, out, (int64_t)-1, 0);
  GNM_LAUNCH_CHECK("reduce_partials");
  return 0;
}

// `batch` independent reductions in one launch: out[g][i] = sum_b partials[(g*nblk + b)*W + i], i < W
int gnm::reduce_partials_batched(const double* partials, int batch, int nblk, int W, float* out, void* stream) {
  GNM_CHECK_ARG(partials && batch > 0 && nblk > 0 && W > 0 && out, "reduce_partials_batched: bad argument");
  hipLaunchKernelGGL(reduce_partials_k, dim3((W + 15) / 16, batch), dim3(256), 0, (hipStream_t)stream, partials, nblk, W,
                     out, (int64_t)-1, 0);
  GNM_LAUNCH_CHECK("reduce_partials_batched");
  return 0;
}

// out[i] = sum_b partials[b*row_stride + off + i], i < n (a column slice of wider partial rows)
int gnm::reduce_partials_strided(const double* partials, int nblk, int row_stride, int off, int n, float* out,
                                 void* stream) {
  GNM_CHECK_ARG(partials && nblk > 0 && n > 0 && out && row_stride >= off + n, "reduce_partials_strided: bad argument");
  hipLaunchKernelGGL(reduce_partials_k, dim3((n + 15) / 16), dim3(256), 0, (hipStream_t)stream, partials, nblk, n,
                     out, (int64_t)row_stride, off);
  GNM_LAUNCH_CHECK("reduce_partials_strided");
  return 0;
}

extern "C" int gnm_seg_sum_rows(int64_t N, int W, const float* X, const int32_t* ptr,
                                const int32_t* pos, float* out, int64_t ldo, void* stream) {
  GNM_CHECK_ARG(N >= 0 && X && ptr && out && ldo >= W, "seg_sum_rows: bad argument");
  GNM_DISPATCH_W(W, {
    const int grid = persistent_grid(N, 64, occ_blocks<seg_sum_rows_k<WW>>());
    const int64_t npb = cdiv(N, grid);
    hipLaunchKernelGGL(seg_sum_rows_k<WW>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, N, X, ptr, pos,
                       out, ldo, npb);
  });
  GNM_LAUNCH_CHECK("seg_sum_rows");
  return 0;
}

static int colsum_blocks(int64_t M) {
  int64_t b = cdiv(M, 256);
  if (b > kMaxPartialBlocks) b = kMaxPartialBlocks;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" size_t gnm_colsum_workspace_bytes(int64_t M, int64_t W) {
  return (size_t)colsum_blocks(M) * (size_t)W * sizeof(double);
}

extern "C" int gnm_colsum_f32(int64_t M, int64_t W, const float* X, int64_t ld, float* out, void* ws,
                              size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(M >= 0 && W > 0 && X && out && ld >= W, "colsum_f32: bad argument");
  const int nb = colsum_blocks(M);
  GNM_CHECK_ARG(ws && ws_bytes >= (size_t)nb * W * sizeof(double), "colsum_f32: workspace too small");
  const int64_t rpb = cdiv(M, nb);
  const bool vec = (W % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)X % 16 == 0) && W / 4 <= 256;
  if (vec) {
    const int tx = (int)(W / 4);
    int ty = 256 / tx;
    if (ty < 1) ty = 1;
    if (ty > 8) ty = 8;
    hipLaunchKernelGGL(colsum_stage1_k<true>, dim3(nb), dim3(tx, ty), (size_t)ty * W * sizeof(double),
                       (hipStream_t)stream, M, W, X, ld, (double*)ws, rpb);
  } else {
    hipLaunchKernelGGL(colsum_stage1_k<false>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, M, W, X, ld,
                       (double*)ws, rpb);
  }
  GNM_LAUNCH_CHECK("colsum stage 1");
  hipLaunchKernelGGL(reduce_partials_k, dim3((unsigned)((W + 15) / 16)), dim3(256), 0,
                     (hipStream_t)stream, (const double*)ws, nb, (int)W, out, (int64_t)-1, 0);
  GNM_LAUNCH_CHECK("colsum stage 2");
  return 0;
}

extern "C" int gnm_gather_rows_f32(int64_t M, int64_t W, const float* X, const int32_t* idx,
                                   float* out, void* stream) {
  GNM_CHECK_ARG(M >= 0 && W > 0 && X && idx && out, "gather_rows_f32: bad argument");
  hipLaunchKernelGGL(gather_rows_k, dim3(ew_grid2(M * W, 256)), dim3(256), 0, (hipStream_t)stream, M,
                     W, X, idx, out);
  GNM_LAUNCH_CHECK("gather_rows_f32");
  return 0;
}

extern "C" int gnm_relu_mask_f32(int64_t n, float* x, const float* ref, void* stream) {
  GNM_CHECK_ARG(n >= 0 && x && ref, "relu_mask_f32: bad argument");
  hipLaunchKernelGGL(relu_mask_k, dim3(ew_grid2(n, 256)), dim3(256), 0, (hipStream_t)stream, n, x, ref);
  GNM_LAUNCH_CHECK("relu_mask_f32");
  return 0;
}

extern "C" int gnm_bce_fwd_bwd(int64_t E, const float* scores, const float* y, float pos_weight,
                               float* loss_out, float* gscore, void* ws, size_t ws_bytes,
                               void* stream) {
  GNM_CHECK_ARG(E > 0 && scores && y && loss_out && gscore, "bce_fwd_bwd: bad argument");
  int grid = ew_grid2(E, kBlock);
  if (grid > kMaxPartialBlocks) grid = kMaxPartialBlocks;
  GNM_CHECK_ARG(ws && ws_bytes >= (size_t)grid * sizeof(double), "bce_fwd_bwd: workspace too small");
  hipLaunchKernelGGL(bce_fwd_bwd_k, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, E, scores, y,
                     pos_weight, (float)(1.0 / (double)E), gscore, (double*)ws);
  GNM_LAUNCH_CHECK("bce_fwd_bwd");
  hipLaunchKernelGGL(bce_finalize_k, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double*)ws,
                     grid, 1.0 / (double)E, loss_out);
  GNM_LAUNCH_CHECK("bce finalize");
  return 0;
}
