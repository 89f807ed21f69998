// A C++ host with no Python and no torch: one GatedGCN layer forward (gated_gcn_full.py:99-157,
// BatchNorm mode, H = 128) driven purely through the C ABI of libgnm.so with hipMalloc'd buffers, checked
// against a naive fp64 loop restatement of the same arithmetic.  Test infrastructure (built and run by
// tests/test_gpu_parity.py::test_cxx_host_through_the_c_abi); it shows the boundary INTEGRATION.md claims.
//   hipcc --offload-arch=gfx950 -std=c++17 -I include tests/cabi/host_layer.cpp -L gnnome_assembly_amd -lgnm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gnm.h"

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
  } while (0)
#define GNM_OK(x)                                                                  \
  do {                                                                             \
    int r_ = (x);                                                                  \
    if (r_ != 0) { std::printf("gnm error %d: %s (%s)\n", r_, gnm_last_error(), #x); return 3; } \
  } while (0)

template <class T>
static T* to_dev(const std::vector<T>& v) {
  T* p = nullptr;
  if (hipMalloc(&p, v.size() * sizeof(T)) != hipSuccess) std::abort();
  if (hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort();
  return p;
}
template <class T>
static T* dev_alloc(size_t n) {
  T* p = nullptr;
  if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) std::abort();
  return p;
}
template <class T>
static std::vector<T> to_host(const T* p, size_t n) {
  std::vector<T> v(n);
  if (hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::abort();
  return v;
}

int main() {
  const int H = 128;
  const int64_t N = 3001;
  std::mt19937 rng(7);
  std::normal_distribution<float> nrm(0.f, 1.f);
  // a banded graph with a few long edges, duplicate and self-loop included; edge ids in random order
  std::vector<int32_t> src, dst;
  for (int64_t v = 0; v < N; ++v) {
    const int k = 1 + (int)(rng() % 6);
    for (int j = 1; j <= k; ++j) { src.push_back((int32_t)v); dst.push_back((int32_t)((v + j * (1 + rng() % 3)) % N)); }
  }
  src.push_back(5); dst.push_back(5);
  src.push_back(src[3]); dst.push_back(dst[3]);
  const int64_t E = (int64_t)src.size();
  for (int64_t i = E - 1; i > 0; --i) { const int64_t j = rng() % (i + 1); std::swap(src[i], src[j]); std::swap(dst[i], dst[j]); }

  std::vector<float> h(N * H), e(E * H), W5(5 * H * H), b5(5 * H), W3(H * H), b3(H), ge(H), be(H), gh(H), bh(H);
  for (auto& x : h) x = nrm(rng);
  for (auto& x : e) x = nrm(rng);
  for (auto& x : W5) x = nrm(rng) / std::sqrt((float)H);
  for (auto& x : W3) x = nrm(rng) / std::sqrt((float)H);
  for (auto& x : b5) x = 0.1f * nrm(rng);
  for (auto& x : b3) x = 0.1f * nrm(rng);
  for (int c = 0; c < H; ++c) { ge[c] = 1.f + 0.1f * nrm(rng); be[c] = 0.1f * nrm(rng); gh[c] = 1.f + 0.1f * nrm(rng); bh[c] = 0.1f * nrm(rng); }

  // ---- index (host entry point), internal edge order ----
  std::vector<int32_t> perm(E), isrc(E), idst(E), in_ptr(N + 1), out_ptr(N + 1), out_pos(E), out_dst(E);
  GNM_OK(gnm_graph_build_index(src.data(), dst.data(), N, E, perm.data(), isrc.data(), idst.data(), in_ptr.data(),
                               out_ptr.data(), out_pos.data(), out_dst.data()));
  std::vector<float> e_int(E * H);
  for (int64_t j = 0; j < E; ++j)
    for (int c = 0; c < H; ++c) e_int[j * H + c] = e[(int64_t)perm[j] * H + c];

  // ---- device side, C ABI only ----
  float *d_h = to_dev(h), *d_e = to_dev(e_int), *d_W5 = to_dev(W5), *d_b5 = to_dev(b5), *d_W3 = to_dev(W3), *d_b3 = to_dev(b3);
  float *d_ge = to_dev(ge), *d_be = to_dev(be), *d_gh = to_dev(gh), *d_bh = to_dev(bh);
  int32_t *d_isrc = to_dev(isrc), *d_idst = to_dev(idst), *d_inp = to_dev(in_ptr), *d_outp = to_dev(out_ptr),
          *d_opos = to_dev(out_pos), *d_odst = to_dev(out_dst);
  float *d_P = dev_alloc<float>(N * 5 * H), *d_t = dev_alloc<float>(E * H), *d_eo = dev_alloc<float>(E * H);
  float *d_hf = dev_alloc<float>(N * H), *d_if = dev_alloc<float>(N * H), *d_hb = dev_alloc<float>(N * H),
        *d_ib = dev_alloc<float>(N * H), *d_z = dev_alloc<float>(N * H), *d_ho = dev_alloc<float>(N * H);
  float *d_se = dev_alloc<float>(4 * H), *d_sh = dev_alloc<float>(4 * H);
  double* d_part = dev_alloc<double>((size_t)(gnm_max_partial_blocks() + 1) * 2 * 256);
  const size_t wsb = gnm_rowtile_workspace_bytes(5 * H);
  void* d_ws = dev_alloc<char>(wsb);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int nblk = 0;
  GNM_OK(gnm_node_proj_fwd(N, H, 5 * H, d_h, d_W5, d_b5, d_P, d_ws, wsb, st));
  GNM_OK(gnm_edge_t_fused_fwd(E, H, d_e, d_W3, d_b3, d_P, d_isrc, d_idst, d_t, d_part, &nblk, d_ws, wsb, st));
  GNM_OK(gnm_bn_finalize(d_part, nblk, E, H, d_ge, d_be, 1e-5f, d_se, st));
  GNM_OK(gnm_edge_gate_fwd(N, E, H, d_t, d_e, d_se, d_P, d_isrc, d_inp, d_eo, d_hf, d_if, st));
  GNM_OK(gnm_node_agg_src_fwd(N, E, H, d_eo, d_P, d_outp, d_opos, d_odst, d_hf, d_hb, d_ib, d_z, d_part, &nblk, st));
  GNM_OK(gnm_bn_finalize(d_part, nblk, N, H, d_gh, d_bh, 1e-5f, d_sh, st));
  GNM_OK(gnm_node_update_fwd(N, H, d_z, d_sh, d_h, d_ho, st));
  HIP_OK(hipStreamSynchronize(st));
  const std::vector<float> got_h = to_host(d_ho, N * H), got_e_int = to_host(d_eo, E * H);

  // ---- naive fp64 restatement in the CALLER's edge order ----
  auto lin = [&](const std::vector<float>& x, int64_t row, const float* W, const float* b, std::vector<double>& out) {
    for (int n = 0; n < H; ++n) {
      double a = b[n];
      for (int k = 0; k < H; ++k) a += (double)x[row * H + k] * (double)W[n * H + k];
      out[n] = a;
    }
  };
  std::vector<double> P((size_t)N * 5 * H), tt((size_t)E * H), tmp(H);
  for (int64_t v = 0; v < N; ++v)
    for (int g = 0; g < 5; ++g) {
      lin(h, v, W5.data() + (size_t)g * H * H, b5.data() + g * H, tmp);
      for (int c = 0; c < H; ++c) P[(v * 5 + g) * H + c] = tmp[c];
    }
  std::vector<double> mean(H, 0.0), var(H, 0.0);
  for (int64_t k = 0; k < E; ++k) {
    lin(e, k, W3.data(), b3.data(), tmp);
    for (int c = 0; c < H; ++c) {
      const double v = tmp[c] + P[((int64_t)src[k] * 5 + 3) * H + c] + P[((int64_t)dst[k] * 5 + 4) * H + c];
      tt[k * H + c] = v;
      mean[c] += v;
    }
  }
  for (int c = 0; c < H; ++c) mean[c] /= (double)E;
  for (int64_t k = 0; k < E; ++k)
    for (int c = 0; c < H; ++c) var[c] += (tt[k * H + c] - mean[c]) * (tt[k * H + c] - mean[c]);
  std::vector<double> eo((size_t)E * H), fn((size_t)N * H, 0.0), fd((size_t)N * H, 0.0), bn_((size_t)N * H, 0.0), bd((size_t)N * H, 0.0);
  for (int64_t k = 0; k < E; ++k)
    for (int c = 0; c < H; ++c) {
      const double u = ge[c] * (tt[k * H + c] - mean[c]) / std::sqrt(var[c] / (double)E + 1e-5) + be[c];
      const double o = (u > 0 ? u : 0) + (double)e[k * H + c];
      eo[k * H + c] = o;
      const double sg = 1.0 / (1.0 + std::exp(-o));
      fn[(int64_t)dst[k] * H + c] += sg * P[((int64_t)src[k] * 5 + 1) * H + c];
      fd[(int64_t)dst[k] * H + c] += sg;
      bn_[(int64_t)src[k] * H + c] += sg * P[((int64_t)dst[k] * 5 + 2) * H + c];
      bd[(int64_t)src[k] * H + c] += sg;
    }
  std::vector<double> z((size_t)N * H), zm(H, 0.0), zv(H, 0.0);
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) {
      z[v * H + c] = P[(v * 5) * H + c] + fn[v * H + c] / (fd[v * H + c] + 1e-6) + bn_[v * H + c] / (bd[v * H + c] + 1e-6);
      zm[c] += z[v * H + c];
    }
  for (int c = 0; c < H; ++c) zm[c] /= (double)N;
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) zv[c] += (z[v * H + c] - zm[c]) * (z[v * H + c] - zm[c]);
  double num_h = 0, den_h = 0, num_e = 0, den_e = 0;
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) {
      const double w = gh[c] * (z[v * H + c] - zm[c]) / std::sqrt(zv[c] / (double)N + 1e-5) + bh[c];
      const double want = (w > 0 ? w : 0) + (double)h[v * H + c];
      num_h += (got_h[v * H + c] - want) * (got_h[v * H + c] - want);
      den_h += want * want;
    }
  for (int64_t j = 0; j < E; ++j)
    for (int c = 0; c < H; ++c) {
      const double want = eo[(int64_t)perm[j] * H + c];
      num_e += (got_e_int[j * H + c] - want) * (got_e_int[j * H + c] - want);
      den_e += want * want;
    }
  const double rh = std::sqrt(num_h / den_h), re = std::sqrt(num_e / den_e);
  std::printf("C-ABI host: N=%lld E=%lld H=%d  rel_l2(h_out)=%.3e rel_l2(e_out)=%.3e\n", (long long)N, (long long)E, H, rh, re);
  if (!(rh < 1e-5 && re < 1e-5)) { std::printf("FAIL\n"); return 1; }
  std::printf("OK\n");
  return 0;
}
