// Input feature preparation on the GPU ("next" row f-1 of SURVEY.md section 8): what the reference
// does on the CPU with scipy before every run and re-uploads every step (train.py:245-251).
//   pagerank_pe      utils.add_positional_encoding (utils.py:97-138): float in/out degrees and the
//                    16-step PageRank features x <- alpha * P x + (1-alpha)/n, P = (D^-1 A)^T,
//                    iterated in fp64 like the reference, emitted as the [N, 2+pe_dim] fp32 tensor
//                    that train.py:251 / inference.py:452 concatenate (in_deg | out_deg | pe)
//   edge_feats_zscore utils.preprocess_graph (utils.py:70-74): z-score of overlap length and
//                    similarity with the unbiased std (torch.std default), -> e[E,2]
// Pull-style SpMV over the destination-sorted index: one thread per node, fixed summation order,
// no atomics.  Negligible next to the layer stack (E*4 B of indices + 8-byte gathers per step).
#include "gnm_common.h"

namespace gnm {

// w[u] = x[u] / (outdeg[u] + 1e-9), 0 for nodes without out-edges (utils.py:126-127)
__global__ void pr_scale_k(int64_t N, const double* __restrict__ x, const int32_t* __restrict__ out_ptr,
                           double* __restrict__ w) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x) {
    const double d = (double)(out_ptr[v + 1] - out_ptr[v]);
    w[v] = d < 1e-9 ? 0.0 : x[v] * (1.0 / (d + 1e-9));
  }
}

// x_new[v] = alpha * sum_{j in in(v)} w[isrc j] + (1-alpha)/n ; pe[v][2+step] = (float)x_new[v]
__global__ void pr_step_k(int64_t N, const double* __restrict__ w, const int32_t* __restrict__ isrc,
                          const int32_t* __restrict__ in_ptr, double alpha, double teleport,
                          double* __restrict__ x_new, float* __restrict__ pe, int ldpe, int col) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x) {
    double acc = 0.0;
    for (int j = in_ptr[v]; j < in_ptr[v + 1]; ++j) acc += w[isrc[j]];
    const double xn = alpha * acc + teleport;
    x_new[v] = xn;
    pe[v * ldpe + col] = (float)xn;
  }
}

__global__ void pr_init_k(int64_t N, const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ out_ptr,
                          double* __restrict__ x, float* __restrict__ pe, int ldpe) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < N; v += (int64_t)gridDim.x * blockDim.x) {
    x[v] = 1.0 / (double)N;
    pe[v * ldpe + 0] = (float)(in_ptr[v + 1] - in_ptr[v]);     // in_deg  (utils.py:102)
    pe[v * ldpe + 1] = (float)(out_ptr[v + 1] - out_ptr[v]);   // out_deg (utils.py:103)
  }
}

// per-block (sum a, sum a^2, sum b, sum b^2) in fp64
__global__ __launch_bounds__(kBlock) void zs_stats_k(int64_t E, const float* __restrict__ a,
                                                     const float* __restrict__ b, double* __restrict__ part) {
  __shared__ double red[4][kBlock];
  double s[4] = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < E; i += (int64_t)gridDim.x * kBlock) {
    const double x = a[i], y = b[i];
    s[0] += x; s[1] += x * x; s[2] += y; s[3] += y * y;
  }
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int st = kBlock / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
      for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 4) part[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// e[perm? no: edge-id order][0] = (a - mean_a)/std_a, [1] likewise (unbiased std)
__global__ void zs_apply_k(int64_t E, const float* __restrict__ a, const float* __restrict__ b,
                           const double* __restrict__ part, int nblk, float* __restrict__ e) {
  __shared__ double st[4];
  if (threadIdx.x < 4) {
    double acc = 0.0;
    for (int k = 0; k < nblk; ++k) acc += part[(size_t)k * 4 + threadIdx.x];
    st[threadIdx.x] = acc;
  }
  __syncthreads();
  const double n = (double)E;
  const double ma = st[0] / n, mb = st[2] / n;
  const double va = (st[1] - n * ma * ma) / (n - 1.0), vb = (st[3] - n * mb * mb) / (n - 1.0);
  const double ra = 1.0 / sqrt(va), rb = 1.0 / sqrt(vb);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < E; i += (int64_t)gridDim.x * blockDim.x) {
    e[2 * i] = (float)(((double)a[i] - ma) * ra);
    e[2 * i + 1] = (float)(((double)b[i] - mb) * rb);
  }
}

}  // namespace gnm

using namespace gnm;

static inline int fgrid(int64_t n) {
  int64_t g = (n + 255) / 256;
  const int64_t cap = (int64_t)num_cus() * 8;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ws: 3*N doubles
extern "C" size_t gnm_pagerank_pe_workspace_bytes(int64_t N) { return (size_t)3 * (size_t)N * sizeof(double); }

extern "C" int gnm_pagerank_pe(int64_t N, int64_t E, const int32_t* isrc, const int32_t* in_ptr,
                               const int32_t* out_ptr, int pe_dim, double alpha, float* pe, void* ws,
                               size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(N > 0 && E >= 0 && isrc && in_ptr && out_ptr && pe_dim >= 0 && pe, "pagerank_pe: bad argument");
  GNM_CHECK_ARG(ws && ws_bytes >= gnm_pagerank_pe_workspace_bytes(N), "pagerank_pe: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  double* x = (double*)ws;
  double* xn = x + N;
  double* w = x + 2 * N;
  const int ld = pe_dim + 2, g = fgrid(N);
  hipLaunchKernelGGL(pr_init_k, dim3(g), dim3(256), 0, st, N, in_ptr, out_ptr, x, pe, ld);
  for (int s = 0; s < pe_dim; ++s) {
    hipLaunchKernelGGL(pr_scale_k, dim3(g), dim3(256), 0, st, N, (const double*)x, out_ptr, w);
    hipLaunchKernelGGL(pr_step_k, dim3(g), dim3(256), 0, st, N, (const double*)w, isrc, in_ptr, alpha,
                       (1.0 - alpha) / (double)N, xn, pe, ld, 2 + s);
    double* tmp = x; x = xn; xn = tmp;
  }
  GNM_LAUNCH_CHECK("pagerank_pe");
  return 0;
}

// ws: 4 * gnm_max_partial_blocks() doubles
extern "C" int gnm_edge_feats_zscore(int64_t E, const float* overlap_length, const float* overlap_similarity,
                                     float* e, void* ws, size_t ws_bytes, void* stream) {
  GNM_CHECK_ARG(E > 1 && overlap_length && overlap_similarity && e, "edge_feats_zscore: bad argument");
  int nb = fgrid(E);
  if (nb > kMaxPartialBlocks) nb = kMaxPartialBlocks;
  GNM_CHECK_ARG(ws && ws_bytes >= (size_t)nb * 4 * sizeof(double), "edge_feats_zscore: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(zs_stats_k, dim3(nb), dim3(kBlock), 0, st, E, overlap_length, overlap_similarity, (double*)ws);
  hipLaunchKernelGGL(zs_apply_k, dim3(fgrid(E)), dim3(256), 0, st, E, overlap_length, overlap_similarity,
                     (const double*)ws, nb, e);
  GNM_LAUNCH_CHECK("edge_feats_zscore");
  return 0;
}
