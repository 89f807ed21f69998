// A C++ host with no Python and no torch that drives the path bench.py MEASURES -- the default schedule of the engine -- purely
// through the C ABI of libgnm.so on hipMalloc'd buffers: two stacked GatedGCN layers (gated_gcn_full.py:99-157, BatchNorm mode,
// H = 128), forward through the sweep plans and the two-sided gate kernel, backward through the two-sided top sweep, the chained
// edge kernel and the fused node-side kernels, each result against a naive fp64 loop restatement of the same arithmetic
// (SURVEY.md section 8a row 8).  Test infrastructure (built and run by tests/test_gpu_parity.py::test_cxx_host_through_the_c_abi).
//
// Entry points exercised (include/gnm.h), in call order:
//   gnm_graph_build_index, gnm_sweep_partition, gnm_graph_build_sweep_plan (both partitions),
//   forward x2:  gnm_node_proj_fwd, gnm_edge_t_fused_fwd, gnm_bn_finalize, gnm_edge_gate2_fwd, gnm_node_update_fwd
//   backward:    gnm_node_bwd_stats, gnm_bn_bwd_finalize, gnm_node_bwd_apply, gnm_edge_bwd_top, gnm_edge_bwd_src_fix,
//                gnm_tn128_bgrad, gnm_node_proj_bwd_nn_stats, gnm_tn128, gnm_edge_bwd_chain_src, gnm_node_proj_bwd_nn,
//                gnm_edge_bwd_fused
//   and once more through the composite entry points gnm_layer_forward / gnm_stack_backward (bit-identical by construction).
// The relu decisions of the backward are the DEVICE's (read back from t / z and the BatchNorm statistics): the network is
// piecewise linear in them, so the fp64 gradients of those branches are the meaningful reference on a 3 k-node graph.
//   hipcc --offload-arch=gfx950 -std=c++17 -I include tests/cabi/host_step.cpp -L gnnome_assembly_amd -lgnm
#include <hip/hip_runtime.h>

#include <algorithm>
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
  if (hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) std::abort();
  if (!v.empty() && hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::abort();
  return p;
}
template <class T>
static T* dev_alloc(size_t n) {
  T* p = nullptr;
  if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) std::abort();
  return p;
}
template <class T>
static std::vector<T> to_host(const T* p, size_t n) {
  std::vector<T> v(n);
  if (hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::abort();
  return v;
}

constexpr int H = 128;
typedef std::vector<double> vd;
typedef std::vector<float> vf;

struct Params {
  vf W5, b5, W3, b3, gam_e, bet_e, gam_h, bet_h;      // [5H,H] [5H] [H,H] [H] [H] x4
};

// everything the fp64 restatement keeps of one layer (caller's edge order)
struct Ref {
  vd P, th, u, e_out, sig, inv_f, inv_b, hf, hb, zh, w, h_out;
  vd rstd_e, rstd_h;
};

static void ref_forward(int64_t N, int64_t E, const std::vector<int32_t>& src, const std::vector<int32_t>& dst, const Params& p,
                        const vd& h, const vd& e, Ref& r) {
  r.P.assign((size_t)N * 5 * H, 0.0);
  for (int64_t v = 0; v < N; ++v)
    for (int n = 0; n < 5 * H; ++n) {
      double a = p.b5[n];
      for (int k = 0; k < H; ++k) a += h[v * H + k] * (double)p.W5[(size_t)n * H + k];
      r.P[v * 5 * H + n] = a;
    }
  vd t((size_t)E * H), mean(H, 0.0), var(H, 0.0);
  for (int64_t k = 0; k < E; ++k)
    for (int n = 0; n < H; ++n) {
      double a = p.b3[n];
      for (int c = 0; c < H; ++c) a += e[k * H + c] * (double)p.W3[(size_t)n * H + c];
      a += r.P[(int64_t)src[k] * 5 * H + 3 * H + n] + r.P[(int64_t)dst[k] * 5 * H + 4 * H + n];
      t[k * H + n] = a;
      mean[n] += a;
    }
  for (int c = 0; c < H; ++c) mean[c] /= (double)E;
  for (int64_t k = 0; k < E; ++k)
    for (int c = 0; c < H; ++c) var[c] += (t[k * H + c] - mean[c]) * (t[k * H + c] - mean[c]);
  r.rstd_e.resize(H);
  for (int c = 0; c < H; ++c) r.rstd_e[c] = 1.0 / std::sqrt(var[c] / (double)E + 1e-5);
  r.th.resize((size_t)E * H); r.u.resize((size_t)E * H); r.e_out.resize((size_t)E * H); r.sig.resize((size_t)E * H);
  vd fn((size_t)N * H, 0.0), fd((size_t)N * H, 0.0), bn((size_t)N * H, 0.0), bd((size_t)N * H, 0.0);
  for (int64_t k = 0; k < E; ++k)
    for (int c = 0; c < H; ++c) {
      const double th = (t[k * H + c] - mean[c]) * r.rstd_e[c];
      const double u = p.gam_e[c] * th + p.bet_e[c];
      const double o = (u > 0 ? u : 0) + e[k * H + c];
      const double sg = 1.0 / (1.0 + std::exp(-o));
      r.th[k * H + c] = th; r.u[k * H + c] = u; r.e_out[k * H + c] = o; r.sig[k * H + c] = sg;
      fn[(int64_t)dst[k] * H + c] += sg * r.P[(int64_t)src[k] * 5 * H + H + c];
      fd[(int64_t)dst[k] * H + c] += sg;
      bn[(int64_t)src[k] * H + c] += sg * r.P[(int64_t)dst[k] * 5 * H + 2 * H + c];
      bd[(int64_t)src[k] * H + c] += sg;
    }
  r.inv_f.resize((size_t)N * H); r.inv_b.resize((size_t)N * H); r.hf.resize((size_t)N * H); r.hb.resize((size_t)N * H);
  vd z((size_t)N * H), zm(H, 0.0), zv(H, 0.0);
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) {
      r.inv_f[v * H + c] = 1.0 / (fd[v * H + c] + 1e-6);
      r.inv_b[v * H + c] = 1.0 / (bd[v * H + c] + 1e-6);
      r.hf[v * H + c] = fn[v * H + c] * r.inv_f[v * H + c];
      r.hb[v * H + c] = bn[v * H + c] * r.inv_b[v * H + c];
      z[v * H + c] = r.P[v * 5 * H + c] + r.hf[v * H + c] + r.hb[v * H + c];
      zm[c] += z[v * H + c];
    }
  for (int c = 0; c < H; ++c) zm[c] /= (double)N;
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) zv[c] += (z[v * H + c] - zm[c]) * (z[v * H + c] - zm[c]);
  r.rstd_h.resize(H);
  for (int c = 0; c < H; ++c) r.rstd_h[c] = 1.0 / std::sqrt(zv[c] / (double)N + 1e-5);
  r.zh.resize((size_t)N * H); r.w.resize((size_t)N * H); r.h_out.resize((size_t)N * H);
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) {
      r.zh[v * H + c] = (z[v * H + c] - zm[c]) * r.rstd_h[c];
      r.w[v * H + c] = p.gam_h[c] * r.zh[v * H + c] + p.bet_h[c];
      r.h_out[v * H + c] = (r.w[v * H + c] > 0 ? r.w[v * H + c] : 0) + h[v * H + c];
    }
}

struct RefGrads { vd gW5, gb5, gW3, gb3, g_gam_e, g_bet_e, g_gam_h, g_bet_h, gh_in, ge_in; };

// BNbwd(gy) = w rstd (gy - mean(gy) - xh mean(gy xh)); ggamma = sum gy xh; gbeta = sum gy
static void bn_bwd(int64_t M, const vd& gy, const vd& xh, const vf& w, const vd& rstd, vd& gx, vd& gg, vd& gb) {
  gg.assign(H, 0.0); gb.assign(H, 0.0);
  for (int64_t r = 0; r < M; ++r)
    for (int c = 0; c < H; ++c) { gb[c] += gy[r * H + c]; gg[c] += gy[r * H + c] * xh[r * H + c]; }
  gx.resize((size_t)M * H);
  for (int64_t r = 0; r < M; ++r)
    for (int c = 0; c < H; ++c)
      gx[r * H + c] = (double)w[c] * rstd[c] * (gy[r * H + c] - gb[c] / (double)M - xh[r * H + c] * (gg[c] / (double)M));
}

// mask_u [E,H] (caller order), mask_w [N,H]: the relu branches of THIS evaluation (the device's)
static void ref_backward(int64_t N, int64_t E, const std::vector<int32_t>& src, const std::vector<int32_t>& dst, const Params& p,
                         const vd& h_in, const vd& e_in, const Ref& r, const std::vector<char>& mask_u, const std::vector<char>& mask_w,
                         const vd& gh, const vd& ge, RefGrads& g) {
  vd gw((size_t)N * H), gz;
  for (size_t i = 0; i < gw.size(); ++i) gw[i] = mask_w[i] ? gh[i] : 0.0;
  bn_bwd(N, gw, r.zh, p.gam_h, r.rstd_h, gz, g.g_gam_h, g.g_bet_h);
  vd gP((size_t)N * 5 * H, 0.0), ge_tot((size_t)E * H), gu((size_t)E * H), gt;
  for (int64_t v = 0; v < N; ++v)
    for (int c = 0; c < H; ++c) gP[v * 5 * H + c] = gz[v * H + c];
  for (int64_t k = 0; k < E; ++k) {
    const int64_t s = src[k], d = dst[k];
    for (int c = 0; c < H; ++c) {
      const double qf = gz[d * H + c] * r.inv_f[d * H + c], qb = gz[s * H + c] * r.inv_b[s * H + c];
      const double gsig = qf * r.P[s * 5 * H + H + c] - qf * r.hf[d * H + c] + qb * r.P[d * 5 * H + 2 * H + c] - qb * r.hb[s * H + c];
      const double sg = r.sig[k * H + c];
      gP[s * 5 * H + H + c] += sg * qf;
      gP[d * 5 * H + 2 * H + c] += sg * qb;
      ge_tot[k * H + c] = ge[k * H + c] + gsig * sg * (1.0 - sg);
      gu[k * H + c] = mask_u[k * H + c] ? ge_tot[k * H + c] : 0.0;
    }
  }
  bn_bwd(E, gu, r.th, p.gam_e, r.rstd_e, gt, g.g_gam_e, g.g_bet_e);
  g.gW3.assign((size_t)H * H, 0.0); g.gb3.assign(H, 0.0); g.ge_in.assign((size_t)E * H, 0.0);
  for (int64_t k = 0; k < E; ++k) {
    const int64_t s = src[k], d = dst[k];
    for (int n = 0; n < H; ++n) {
      const double v = gt[k * H + n];
      gP[s * 5 * H + 3 * H + n] += v;
      gP[d * 5 * H + 4 * H + n] += v;
      g.gb3[n] += v;
      for (int c = 0; c < H; ++c) {
        g.gW3[(size_t)n * H + c] += v * e_in[k * H + c];
        g.ge_in[k * H + c] += v * (double)p.W3[(size_t)n * H + c];
      }
    }
    for (int c = 0; c < H; ++c) g.ge_in[k * H + c] += ge_tot[k * H + c];
  }
  g.gW5.assign((size_t)5 * H * H, 0.0); g.gb5.assign(5 * H, 0.0); g.gh_in = gh;
  for (int64_t v = 0; v < N; ++v)
    for (int n = 0; n < 5 * H; ++n) {
      const double x = gP[v * 5 * H + n];
      g.gb5[n] += x;
      for (int c = 0; c < H; ++c) {
        g.gW5[(size_t)n * H + c] += x * h_in[v * H + c];
        g.gh_in[v * H + c] += x * (double)p.W5[(size_t)n * H + c];
      }
    }
}

static double rel_l2(const vf& got, const vd& want, double* maxabs = nullptr) {
  double num = 0, den = 0, mx = 0;
  for (size_t i = 0; i < want.size(); ++i) {
    const double d = (double)got[i] - want[i];
    num += d * d; den += want[i] * want[i]; mx = std::max(mx, std::fabs(d));
  }
  if (maxabs) *maxabs = mx;
  return std::sqrt(num / std::max(den, 1e-300));
}

struct DevLayer {
  float *W5, *b5, *W3, *b3, *gam_e, *bet_e, *gam_h, *bet_h;
  float *h_in, *e_in, *P, *t, *e_out, *hf, *inv_f, *hb, *inv_b, *z, *h_out, *stat_e, *stat_h;
  float *gW5, *gb5, *gW3, *gb3, *g_gam_e, *g_bet_e, *g_gam_h, *g_bet_h, *bstat_e, *bstat_h;
};

int main() {
  const int64_t N = 3001;
  std::mt19937 rng(11);
  std::normal_distribution<float> nrm(0.f, 1.f);
  // a banded graph with a few long ("repeat") edges, a duplicate, a self loop, isolated tail nodes; edge ids in random order
  std::vector<int32_t> src, dst;
  for (int64_t v = 0; v < N - 3; ++v) {
    const int k = 1 + (int)(rng() % 6);
    for (int j = 1; j <= k; ++j) { src.push_back((int32_t)v); dst.push_back((int32_t)((v + j * (1 + rng() % 3)) % (N - 3))); }
  }
  for (int q = 0; q < 15; ++q) { src.push_back((int32_t)(rng() % (N - 3))); dst.push_back((int32_t)(rng() % (N - 3))); }
  src.push_back(5); dst.push_back(5);
  src.push_back(src[3]); dst.push_back(dst[3]);
  const int64_t E = (int64_t)src.size();
  for (int64_t i = E - 1; i > 0; --i) { const int64_t j = rng() % (i + 1); std::swap(src[i], src[j]); std::swap(dst[i], dst[j]); }

  Params prm[2];
  for (auto& p : prm) {
    p.W5.resize((size_t)5 * H * H); p.b5.resize(5 * H); p.W3.resize((size_t)H * H); p.b3.resize(H);
    p.gam_e.resize(H); p.bet_e.resize(H); p.gam_h.resize(H); p.bet_h.resize(H);
    for (auto& x : p.W5) x = nrm(rng) / std::sqrt((float)H);
    for (auto& x : p.W3) x = nrm(rng) / std::sqrt((float)H);
    for (auto& x : p.b5) x = 0.1f * nrm(rng);
    for (auto& x : p.b3) x = 0.1f * nrm(rng);
    for (int c = 0; c < H; ++c) { p.gam_e[c] = 1.f + 0.1f * nrm(rng); p.bet_e[c] = 0.1f * nrm(rng); p.gam_h[c] = 1.f + 0.1f * nrm(rng); p.bet_h[c] = 0.1f * nrm(rng); }
  }
  vf h0((size_t)N * H), e0((size_t)E * H), gh_top((size_t)N * H), ge_top((size_t)E * H);
  for (auto& x : h0) x = nrm(rng);
  for (auto& x : e0) x = nrm(rng);
  for (auto& x : gh_top) x = 1e-3f * nrm(rng);
  for (auto& x : ge_top) x = 1e-3f * nrm(rng);

  // ---- index and sweep plans (host entry points) ----
  std::vector<int32_t> perm(E), isrc(E), idst(E), in_ptr(N + 1), out_ptr(N + 1), out_pos(E), out_dst(E);
  GNM_OK(gnm_graph_build_index(src.data(), dst.data(), N, E, perm.data(), isrc.data(), idst.data(), in_ptr.data(), out_ptr.data(),
                               out_pos.data(), out_dst.data()));
  HIP_OK(hipSetDevice(0));
  struct Plan { int64_t npb = 0, nfix = 0; std::vector<uint32_t> sinfo, dinfo; std::vector<int32_t> fix; uint32_t *d_s, *d_d; int32_t* d_fix; } plan[3];
  for (int wg = 1; wg <= 2; ++wg) {
    Plan& pl = plan[wg];
    int grid = 0, peak = 0;
    GNM_OK(gnm_sweep_partition(N, wg, &pl.npb, &grid));
    pl.sinfo.resize(E); pl.dinfo.resize(E); pl.fix.resize(N);
    GNM_OK(gnm_graph_build_sweep_plan(isrc.data(), idst.data(), in_ptr.data(), N, E, pl.npb, 16, 32, (int64_t)1 << 16, pl.sinfo.data(),
                                      pl.dinfo.data(), pl.fix.data(), &pl.nfix, &peak));
    pl.fix.resize(pl.nfix);
    pl.d_s = to_dev(pl.sinfo); pl.d_d = to_dev(pl.dinfo); pl.d_fix = to_dev(pl.fix);
    std::printf("sweep plan (%d workgroup(s) per CU): %lld nodes per workgroup, %lld of %lld nodes on the fix list, peak live slots %d\n", wg,
                (long long)pl.npb, (long long)pl.nfix, (long long)N, peak);
  }
  auto to_internal = [&](const vf& x) { vf y((size_t)E * H); for (int64_t j = 0; j < E; ++j) for (int c = 0; c < H; ++c) y[j * H + c] = x[(int64_t)perm[j] * H + c]; return y; };

  // ---- device buffers ----
  int32_t *d_isrc = to_dev(isrc), *d_idst = to_dev(idst), *d_inp = to_dev(in_ptr), *d_outp = to_dev(out_ptr), *d_opos = to_dev(out_pos),
          *d_odst = to_dev(out_dst);
  DevLayer L[2];
  for (int l = 0; l < 2; ++l) {
    DevLayer& d = L[l];
    const Params& p = prm[l];
    d.W5 = to_dev(p.W5); d.b5 = to_dev(p.b5); d.W3 = to_dev(p.W3); d.b3 = to_dev(p.b3);
    d.gam_e = to_dev(p.gam_e); d.bet_e = to_dev(p.bet_e); d.gam_h = to_dev(p.gam_h); d.bet_h = to_dev(p.bet_h);
    d.P = dev_alloc<float>((size_t)N * 5 * H); d.t = dev_alloc<float>((size_t)E * H); d.e_out = dev_alloc<float>((size_t)E * H);
    d.hf = dev_alloc<float>((size_t)N * H); d.inv_f = dev_alloc<float>((size_t)N * H); d.hb = dev_alloc<float>((size_t)N * H);
    d.inv_b = dev_alloc<float>((size_t)N * H); d.z = dev_alloc<float>((size_t)N * H); d.h_out = dev_alloc<float>((size_t)N * H);
    d.stat_e = dev_alloc<float>(4 * H); d.stat_h = dev_alloc<float>(4 * H); d.bstat_e = dev_alloc<float>(2 * H); d.bstat_h = dev_alloc<float>(2 * H);
    d.gW5 = dev_alloc<float>((size_t)5 * H * H); d.gb5 = dev_alloc<float>(5 * H); d.gW3 = dev_alloc<float>((size_t)H * H); d.gb3 = dev_alloc<float>(H);
    d.g_gam_e = dev_alloc<float>(H); d.g_bet_e = dev_alloc<float>(H); d.g_gam_h = dev_alloc<float>(H); d.g_bet_h = dev_alloc<float>(H);
  }
  L[0].h_in = to_dev(h0); L[0].e_in = to_dev(to_internal(e0));
  L[1].h_in = L[0].h_out; L[1].e_in = L[0].e_out;
  const size_t npart = (size_t)(gnm_max_partial_blocks() + 1) * 2 * 256;
  double *d_part = dev_alloc<double>(npart), *d_part_hi = dev_alloc<double>(npart), *d_part_tn = dev_alloc<double>(npart);
  const size_t wsb = std::max(std::max(gnm_node_proj_bwd_workspace_bytes(5 * H), gnm_edge_bwd_fused_workspace_bytes()),
                              std::max(gnm_tn128_workspace_bytes(), gnm_rowtile_workspace_bytes(5 * H)));
  void *d_ws = dev_alloc<char>(wsb), *d_ws_tn = dev_alloc<char>(wsb);
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  int nblk = 0;

  // ---- forward, two layers (what engine.layer_forward issues with a sweep plan) ----
  for (int l = 0; l < 2; ++l) {
    DevLayer& d = L[l];
    GNM_OK(gnm_node_proj_fwd(N, H, 5 * H, d.h_in, d.W5, d.b5, d.P, d_ws, wsb, st));
    GNM_OK(gnm_edge_t_fused_fwd(E, H, d.e_in, d.W3, d.b3, d.P, d_isrc, d_idst, d.t, d_part, &nblk, d_ws, wsb, st));
    GNM_OK(gnm_bn_finalize(d_part, nblk, E, H, d.gam_e, d.bet_e, 1e-5f, d.stat_e, st));
    GNM_OK(gnm_edge_gate2_fwd(N, E, H, d.t, d.e_in, d.stat_e, d.P, d_isrc, d_idst, d_inp, plan[2].d_s, plan[2].d_d, plan[2].npb, plan[2].nfix,
                              plan[2].d_fix, d_outp, d_opos, d_odst, d.e_out, d.hf, d.inv_f, d.hb, d.inv_b, d.z, d_part, &nblk, st));
    GNM_OK(gnm_bn_finalize(d_part, nblk, N, H, d.gam_h, d.bet_h, 1e-5f, d.stat_h, st));
    GNM_OK(gnm_node_update_fwd(N, H, d.z, d.stat_h, d.h_in, d.h_out, st));
  }
  HIP_OK(hipStreamSynchronize(st));

  // ---- fp64 forward ----
  vd h0d(h0.begin(), h0.end()), e0d(e0.begin(), e0.end());
  Ref R[2];
  ref_forward(N, E, src, dst, prm[0], h0d, e0d, R[0]);
  ref_forward(N, E, src, dst, prm[1], R[0].h_out, R[0].e_out, R[1]);
  int fails = 0;
  auto check = [&](const char* what, const vf& got, const vd& want, double tol, double floor_abs = 0.0) {
    double mx = 0;
    const double r = rel_l2(got, want, &mx);
    const bool ok = r <= tol || mx <= floor_abs;
    std::printf("  %-28s rel_l2=%.3e max_abs=%.3e %s\n", what, r, mx, ok ? "" : "  <-- FAIL");
    if (!ok) ++fails;
  };
  auto edges_to_caller = [&](const float* dptr) { const vf x = to_host(dptr, (size_t)E * H); vf y((size_t)E * H); for (int64_t j = 0; j < E; ++j) for (int c = 0; c < H; ++c) y[(int64_t)perm[j] * H + c] = x[j * H + c]; return y; };
  std::printf("forward (two layers through the sweep plans):\n");
  check("h_out(1)", to_host(L[1].h_out, (size_t)N * H), R[1].h_out, 1e-5);
  check("e_out(1)", edges_to_caller(L[1].e_out), R[1].e_out, 1e-5);
  check("hb(1) [two-sided sweep]", to_host(L[1].hb, (size_t)N * H), R[1].hb, 1e-5);

  // the device's relu branches (sign of fma(x, scale, shift) = sign of the exact fp64 value)
  std::vector<char> mu[2], mw[2];
  for (int l = 0; l < 2; ++l) {
    const vf t = edges_to_caller(L[l].t), se = to_host(L[l].stat_e, 4 * H), z = to_host(L[l].z, (size_t)N * H), sh = to_host(L[l].stat_h, 4 * H);
    mu[l].resize((size_t)E * H); mw[l].resize((size_t)N * H);
    for (int64_t k = 0; k < E; ++k) for (int c = 0; c < H; ++c) mu[l][k * H + c] = (double)t[k * H + c] * (double)se[2 * H + c] + (double)se[3 * H + c] > 0;
    for (int64_t v = 0; v < N; ++v) for (int c = 0; c < H; ++c) mw[l][v * H + c] = (double)z[v * H + c] * (double)sh[2 * H + c] + (double)sh[3 * H + c] > 0;
  }

  // ---- backward (what engine.layers_backward_chained issues for L = 2 with a sweep plan and NODE_FUSED) ----
  float *d_gh = to_dev(gh_top), *d_ge = to_dev(to_internal(ge_top));
  float *d_gh0 = dev_alloc<float>((size_t)N * H), *d_ghin = dev_alloc<float>((size_t)N * H);
  float *d_gP1 = dev_alloc<float>((size_t)N * 5 * H), *d_gP0 = dev_alloc<float>((size_t)N * 5 * H), *d_Q = dev_alloc<float>((size_t)N * 2 * H);
  float *d_UT = dev_alloc<float>((size_t)N * 2 * H), *d_DT = dev_alloc<float>((size_t)N * 2 * H);
  float *d_Ud = d_DT, *d_Td = d_DT + H;           // [Ud | Td] in one [N,2H] array: pitch 2H
  const int64_t udp = 2 * H;
  int nblk_h = 0;
  {
    DevLayer& d = L[1];
    GNM_OK(gnm_node_bwd_stats(N, H, d.z, d.stat_h, d_gh, d_part, &nblk, st));
    GNM_OK(gnm_bn_bwd_finalize(d_part, nblk, N, H, d.bstat_h, d.g_gam_h, d.g_bet_h, st));
    GNM_OK(gnm_node_bwd_apply(N, H, d.z, d.stat_h, d.bstat_h, d.gam_h, d_gh, d.inv_f, d.inv_b, d_gP1, d_Q, st));
    GNM_OK(gnm_edge_bwd_top(N, E, H, d_ge, d.e_out, d.t, d.stat_e, d.P, d_Q, d.hf, d.hb, d_isrc, d_idst, d_inp, d_gP1, d_Ud, d_Td, d_part,
                            plan[1].d_s, plan[1].npb, d_UT, &nblk, d_ws, wsb, st));
    GNM_OK(gnm_edge_bwd_src_fix(plan[1].nfix, plan[1].d_fix, N, E, H, d.e_out, d.t, d.stat_e, d_ge, d_Q, d_outp, d_opos, d_odst, d_gP1, d_UT, st));
    GNM_OK(gnm_bn_bwd_finalize(d_part, nblk, E, H, d.bstat_e, d.g_gam_e, d.g_bet_e, st));
    GNM_OK(gnm_tn128_bgrad(N, H, d_UT, d_Ud, d_Td, udp, d.stat_e, d.bstat_e, d.gam_e, d_inp, d_outp, d_gP1, d.h_in,
                           d.gW5 + (size_t)3 * H * H, d.gb5 + 3 * H, d_part, d_ws, wsb, st));
    GNM_OK(gnm_node_proj_bwd_nn_stats(N, H, 5 * H, d_gP1, d.W5, d_gh, d_gh0, L[0].z, L[0].stat_h, d_part, &nblk_h, d_ws, wsb, st));
    GNM_OK(gnm_tn128(N, d_gP1, 5 * H, 3, d.h_in, d.gW5, d.gb5, d_part_tn, d_ws_tn, wsb, st));
  }
  {
    DevLayer &d = L[0], &u = L[1];
    GNM_OK(gnm_bn_bwd_finalize(d_part, nblk_h, N, H, d.bstat_h, d.g_gam_h, d.g_bet_h, st));
    GNM_OK(gnm_node_bwd_apply(N, H, d.z, d.stat_h, d.bstat_h, d.gam_h, d_gh0, d.inv_f, d.inv_b, d_gP0, d_Q, st));
    GNM_OK(gnm_edge_bwd_chain_src(N, E, H, d_ge, d_ge, u.t, u.e_in, u.stat_e, u.bstat_e, u.gam_e, u.W3, u.gW3, u.gb3, d_part_hi,
                                  d.t, d.stat_e, d.P, d_Q, d.hf, d.hb, d_isrc, d_idst, d_inp, d_gP0, d_Ud, d_Td, d_part, plan[1].d_s,
                                  plan[1].npb, d_UT, &nblk, d_ws, wsb, st));
    GNM_OK(gnm_edge_bwd_src_fix(plan[1].nfix, plan[1].d_fix, N, E, H, d.e_out, d.t, d.stat_e, d_ge, d_Q, d_outp, d_opos, d_odst, d_gP0, d_UT, st));
    GNM_OK(gnm_bn_bwd_finalize(d_part, nblk, E, H, d.bstat_e, d.g_gam_e, d.g_bet_e, st));
    GNM_OK(gnm_tn128_bgrad(N, H, d_UT, d_Ud, d_Td, udp, d.stat_e, d.bstat_e, d.gam_e, d_inp, d_outp, d_gP0, d.h_in,
                           d.gW5 + (size_t)3 * H * H, d.gb5 + 3 * H, d_part, d_ws, wsb, st));
    GNM_OK(gnm_node_proj_bwd_nn(N, H, 5 * H, d_gP0, d.W5, d_gh0, d_ghin, d_ws, wsb, st));
    GNM_OK(gnm_tn128(N, d_gP0, 5 * H, 3, d.h_in, d.gW5, d.gb5, d_part_tn, d_ws_tn, wsb, st));
    GNM_OK(gnm_edge_bwd_fused(E, H, d_ge, d_ge, d.t, d.e_in, d.stat_e, d.bstat_e, d.gam_e, d.W3, d.gW3, d.gb3, d_part, d_ws, wsb, st));
  }
  HIP_OK(hipStreamSynchronize(st));

  // ---- fp64 backward on the device's branches ----
  vd ghd(gh_top.begin(), gh_top.end()), ged(ge_top.begin(), ge_top.end());
  RefGrads G[2];
  ref_backward(N, E, src, dst, prm[1], R[0].h_out, R[0].e_out, R[1], mu[1], mw[1], ghd, ged, G[1]);
  ref_backward(N, E, src, dst, prm[0], h0d, e0d, R[0], mu[0], mw[0], G[1].gh_in, G[1].ge_in, G[0]);
  double gmax = 0;
  for (int l = 0; l < 2; ++l) for (double x : G[l].gW5) gmax = std::max(gmax, std::fabs(x));
  const double fl = 2e-7 * std::max(gmax * 1e3, 1.0);      // biases in front of a BatchNorm have an analytically zero gradient
  for (int l = 1; l >= 0; --l) {
    std::printf("backward, layer %d:\n", l);
    const DevLayer& d = L[l];
    check("gW5", to_host(d.gW5, (size_t)5 * H * H), G[l].gW5, 5e-5);
    check("gb5", to_host(d.gb5, 5 * H), G[l].gb5, 5e-5, fl);
    check("gW3", to_host(d.gW3, (size_t)H * H), G[l].gW3, 5e-5);
    check("gb3", to_host(d.gb3, H), G[l].gb3, 5e-5, fl);
    check("g bn_e.weight", to_host(d.g_gam_e, H), G[l].g_gam_e, 5e-5);
    check("g bn_e.bias", to_host(d.g_bet_e, H), G[l].g_bet_e, 5e-5);
    check("g bn_h.weight", to_host(d.g_gam_h, H), G[l].g_gam_h, 5e-5);
    check("g bn_h.bias", to_host(d.g_bet_h, H), G[l].g_bet_h, 5e-5);
  }
  check("gh_in(0)", to_host(d_ghin, (size_t)N * H), G[0].gh_in, 5e-5);
  check("ge_in(0)", edges_to_caller(d_ge), G[0].ge_in, 5e-5);
  // ---- the same two layers through the COMPOSITE entry points (gnm_layer_forward x 2, gnm_stack_backward): one call per layer
  //      forward, one for the whole backward; same kernels in the same order -> bit-identical to the explicit sequence above ----
  {
    gnm_graph_view gv{};
    gv.N = N; gv.E = E; gv.isrc = d_isrc; gv.idst = d_idst; gv.in_ptr = d_inp; gv.out_ptr = d_outp; gv.out_pos = d_opos; gv.out_dst = d_odst;
    gv.fwd_sinfo = plan[2].d_s; gv.fwd_dinfo = plan[2].d_d; gv.fwd_nodes_per_block = plan[2].npb; gv.fwd_nfix = plan[2].nfix; gv.fwd_fix_nodes = plan[2].d_fix;
    gv.bwd_sinfo = plan[1].d_s; gv.bwd_nodes_per_block = plan[1].npb; gv.bwd_nfix = plan[1].nfix; gv.bwd_fix_nodes = plan[1].d_fix;
    gnm_layer_weights W[2];
    gnm_layer_state S[2];
    gnm_layer_grads Gr[2];
    for (int l = 0; l < 2; ++l) {
      const DevLayer& d = L[l];
      W[l] = gnm_layer_weights{d.W5, d.b5, d.W3, d.b3, d.gam_e, d.bet_e, d.gam_h, d.bet_h};
      gnm_layer_state& q = S[l];
      q.P = dev_alloc<float>((size_t)N * 5 * H); q.t = dev_alloc<float>((size_t)E * H); q.e_out = dev_alloc<float>((size_t)E * H);
      q.hf = dev_alloc<float>((size_t)N * H); q.inv_f = dev_alloc<float>((size_t)N * H); q.hb = dev_alloc<float>((size_t)N * H);
      q.inv_b = dev_alloc<float>((size_t)N * H); q.z = dev_alloc<float>((size_t)N * H); q.h_out = dev_alloc<float>((size_t)N * H);
      q.stat_e = dev_alloc<float>(4 * H); q.stat_h = dev_alloc<float>(4 * H);
      Gr[l] = gnm_layer_grads{dev_alloc<float>((size_t)5 * H * H), dev_alloc<float>(5 * H), dev_alloc<float>((size_t)H * H), dev_alloc<float>(H),
                              dev_alloc<float>(H), dev_alloc<float>(H), dev_alloc<float>(H), dev_alloc<float>(H)};
    }
    S[0].h_in = L[0].h_in; S[0].e_in = L[0].e_in; S[1].h_in = S[0].h_out; S[1].e_in = S[0].e_out;
    const size_t np2 = gnm_compose_partials_doubles(), wb2 = gnm_compose_workspace_bytes(H);
    gnm_scratch sc{dev_alloc<double>(np2), dev_alloc<double>(np2), dev_alloc<double>(np2), dev_alloc<char>(wb2), wb2, dev_alloc<char>(wb2), wb2};
    gnm_backward_work wk{};
    wk.gP[0] = dev_alloc<float>((size_t)N * 5 * H); wk.gP[1] = dev_alloc<float>((size_t)N * 5 * H);
    wk.Q = dev_alloc<float>((size_t)N * 2 * H); wk.UT = dev_alloc<float>((size_t)N * 2 * H); wk.DT = dev_alloc<float>((size_t)N * 2 * H);
    wk.gh_tmp[0] = dev_alloc<float>((size_t)N * H); wk.gh_tmp[1] = dev_alloc<float>((size_t)N * H);
    wk.bstat_e[0] = dev_alloc<float>(2 * H); wk.bstat_e[1] = dev_alloc<float>(2 * H); wk.bstat_h = dev_alloc<float>(2 * H);
    float *c_gh = to_dev(gh_top), *c_ge = to_dev(to_internal(ge_top)), *c_ghin = dev_alloc<float>((size_t)N * H);
    GNM_OK(gnm_layer_forward(&gv, H, &W[0], &S[0], &sc, st));
    GNM_OK(gnm_layer_forward(&gv, H, &W[1], &S[1], &sc, st));
    GNM_OK(gnm_stack_backward(&gv, H, 2, W, S, Gr, c_gh, c_ge, c_ghin, &wk, &sc, st));
    HIP_OK(hipStreamSynchronize(st));
    int diff = 0;
    auto same = [&](const char* what, const float* a, const float* b, size_t n) {
      const vf x = to_host(a, n), y = to_host(b, n);
      size_t bad = 0;
      for (size_t k = 0; k < n; ++k) bad += !(x[k] == y[k]);
      if (bad) { std::printf("  composite %-22s differs in %zu of %zu elements  <-- FAIL\n", what, bad, n); ++diff; }
    };
    same("h_out(1)", S[1].h_out, L[1].h_out, (size_t)N * H);
    same("e_out(1)", S[1].e_out, L[1].e_out, (size_t)E * H);
    for (int l = 0; l < 2; ++l) {
      same("gW5", Gr[l].gW5, L[l].gW5, (size_t)5 * H * H); same("gb5", Gr[l].gb5, L[l].gb5, 5 * H);
      same("gW3", Gr[l].gW3, L[l].gW3, (size_t)H * H); same("gb3", Gr[l].gb3, L[l].gb3, H);
      same("g bn_e.weight", Gr[l].g_gamma_e, L[l].g_gam_e, H); same("g bn_e.bias", Gr[l].g_beta_e, L[l].g_bet_e, H);
      same("g bn_h.weight", Gr[l].g_gamma_h, L[l].g_gam_h, H); same("g bn_h.bias", Gr[l].g_beta_h, L[l].g_bet_h, H);
    }
    same("gh_in(0)", c_ghin, d_ghin, (size_t)N * H);
    same("ge_in(0)", c_ge, d_ge, (size_t)E * H);
    std::printf("composite entry points (gnm_layer_forward x 2 + gnm_stack_backward): %s the explicit launch sequence\n",
                diff ? "DIFFER from" : "bit-identical to");
    fails += diff;
  }
  std::printf("C-ABI host, measured path: N=%lld E=%lld H=%d L=2: %d mismatches\n", (long long)N, (long long)E, H, fails);
  if (fails) { std::printf("FAIL\n"); return 1; }
  std::printf("OK\n");
  return 0;
}
