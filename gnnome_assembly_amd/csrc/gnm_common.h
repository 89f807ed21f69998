// Shared host/device helpers for libgnm.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "gnm.h"

namespace gnm {

constexpr int kWave = 64;             // CDNA wavefront
constexpr int kBlock = 256;           // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxPartialBlocks = 2048;  // 256 CUs x 8 resident workgroups
constexpr int kXcds = 8;
constexpr float kEpsDen = 1e-6f;      // gated_gcn_full.py:130,143

// ---- host side ---------------------------------------------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);   // records message, returns (int)e
int num_cus();
// out[i] = sum_b partials[b*row_stride + off + i], i < n (gnm_misc.hip)
int reduce_partials_strided(const double* partials, int nblk, int row_stride, int off, int n, float* out,
                            void* stream);
int reduce_partials_batched(const double* partials, int batch, int nblk, int W, float* out, void* stream);

#define GNM_CHECK_ARG(cond, ...)                     \
  do {                                               \
    if (!(cond)) {                                   \
      ::gnm::set_error(__VA_ARGS__);                 \
      return -1;                                     \
    }                                                \
  } while (0)

#define GNM_LAUNCH_CHECK(what)                                   \
  do {                                                           \
    hipError_t e__ = hipGetLastError();                          \
    if (e__ != hipSuccess) return ::gnm::hip_fail(e__, what);    \
  } while (0)

// Grid for the persistent row/segment kernels: a multiple of 8 (one slice per XCD),
// at most kMaxPartialBlocks, at least enough to give every block `min_items` items.
// `call_cap` > 0: the caller's per-call cap on workgroups per CU (entry points that take max_blocks_per_cu);
// otherwise the process-wide knob of gnm_set_occupancy_cap (tools only; 0 = none).
int occupancy_cap();
inline int persistent_grid(int64_t items, int64_t min_items_per_block, int blocks_per_cu, int call_cap = 0) {
  const int cap_ = call_cap > 0 ? call_cap : occupancy_cap();
  if (cap_ > 0 && blocks_per_cu > cap_) blocks_per_cu = cap_;
  int64_t want = (items + min_items_per_block - 1) / min_items_per_block;
  int64_t cap = (int64_t)num_cus() * blocks_per_cu;
  if (cap > kMaxPartialBlocks) cap = kMaxPartialBlocks;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  want = (want + kXcds - 1) / kXcds * kXcds;
  return (int)want;
}

// ---- device side -------------------------------------------------------------------
#ifdef __HIPCC__

// Workgroups of `Kern` (256 threads, static LDS only) that fit on one CU, capped at 8.  The
// persistent kernels size their grid to exactly one resident wave of workgroups so that the
// equal-sized chunks all finish together (no second, mostly idle, dispatch round).
template <auto Kern>
inline int occ_blocks() {
  static int v = 0;
  if (v == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, Kern, kBlock, 0) != hipSuccess || nb < 1) nb = 4;
    v = nb > 8 ? 8 : nb;
  }
  return v;
}

// Workgroup b is dispatched to XCD b % 8 (observed, speed only).  Give every XCD a
// contiguous slice of the chunk range so that neighbouring chunks (which share gathered
// node rows) meet in the same L2.  gridDim.x is a multiple of 8.
__device__ __forceinline__ int xcd_chunk(int b, int nb) {
  return (b % kXcds) * (nb / kXcds) + (b / kXcds);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Streaming ("read or written once by this launch") rows: non-temporal hint, so that the [E,H]
// streams do not push the gathered node rows -- which ARE re-used, by the ~5 edges of a node -- out of
// the 4 MB per-XCD L2.
typedef float floatx4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_nt(const float* p) {
  const floatx4_ v = __builtin_nontemporal_load(reinterpret_cast<const floatx4_*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4_nt(float* p, float4 v) {
  const floatx4_ w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<floatx4_*>(p));
}

// IEEE division + OCML expf (about 30 VALU instructions): for the once-per-edge uses (loss)
__device__ __forceinline__ float sigmoid_ieee_(float x) { return 1.0f / (1.0f + expf(-x)); }
// The gate sigmoid is evaluated for every element of every [E,H] pass (five kernels per layer); with IEEE division and
// OCML's expf that alone was ~0.7 ms of VALU issue per kernel.  Hardware forms instead: ea = 2^(-|x| log2 e) on
// v_exp_f32 (1 ulp; the product's rounding adds |x| * 4e-8 relative to ea, which only matters where sigma is
// saturated), r = v_rcp_f32(1 + ea) (1 ulp, argument in [1,2]); sigma = r or ea*r -- no overflow, no 1 - sigma
// cancellation.  ~7 instructions; forward and backward use the SAME expressions, so sigma' is consistent with sigma.
__device__ __forceinline__ float gate_exp_(float x) { return __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896340736f); }
__device__ __forceinline__ float sigmoidf_(float x) {
  const float ea = gate_exp_(x);
  const float r = __builtin_amdgcn_rcpf(1.0f + ea);
  return x >= 0.f ? r : ea * r;
}

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float b) { return make_float4(a.x * b, a.y * b, a.z * b, a.w * b); }
__device__ __forceinline__ float4& operator+=(float4& a, float4 b) { a = a + b; return a; }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 sigmoid4(float4 a) { return make_float4(sigmoidf_(a.x), sigmoidf_(a.y), sigmoidf_(a.z), sigmoidf_(a.w)); }
// sigma(x) and sigma'(x) = sigma(1-sigma) from one exp, without the 1 - sigma cancellation
// (for |x| ~ 10 the fp32 difference 1 - sigma keeps only ~3 digits):
//   ea = exp(-|x|), r = 1/(1+ea):  sigma = x >= 0 ? r : ea*r ;  sigma' = ea*r*r
__device__ __forceinline__ void sigmoid_grad_(float x, float& sg, float& dsg) {
  const float ea = gate_exp_(x);
  const float r = __builtin_amdgcn_rcpf(1.0f + ea);
  sg = x >= 0.f ? r : ea * r;
  dsg = ea * r * r;
}
__device__ __forceinline__ void sigmoid_grad4(float4 a, float4& sg, float4& dsg) {
  sigmoid_grad_(a.x, sg.x, dsg.x);
  sigmoid_grad_(a.y, sg.y, dsg.y);
  sigmoid_grad_(a.z, sg.z, dsg.z);
  sigmoid_grad_(a.w, sg.w, dsg.w);
}
// (m > 0) ? v : 0
__device__ __forceinline__ float4 gate4(float4 m, float4 v) {
  return make_float4(m.x > 0.f ? v.x : 0.f, m.y > 0.f ? v.y : 0.f, m.z > 0.f ? v.z : 0.f, m.w > 0.f ? v.w : 0.f);
}
__device__ __forceinline__ float4 shfl_xor4(float4 v, int off) {
  return make_float4(__shfl_xor(v.x, off, 64), __shfl_xor(v.y, off, 64), __shfl_xor(v.z, off, 64), __shfl_xor(v.w, off, 64));
}

// Per-lane fp64 accumulator for 4 channels x 2 statistics (BatchNorm column sums).  A
// float and its square are exact in fp64, so the column sums over millions of rows carry
// no fp32 accumulation error (the reference's BatchNorm reduces with a cascade sum).
struct Stat4 {
  double a[4];
  double b[4];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = 0.0; b[i] = 0.0; }
  }
  __device__ __forceinline__ void add(float4 x, float4 y) {  // a += x, b += y
    a[0] += (double)x.x; a[1] += (double)x.y; a[2] += (double)x.z; a[3] += (double)x.w;
    b[0] += (double)y.x; b[1] += (double)y.y; b[2] += (double)y.z; b[3] += (double)y.w;
  }
  __device__ __forceinline__ void add_prod(float4 x, float4 y) {  // a += x, b += x*y (fp64 product)
    a[0] += (double)x.x; a[1] += (double)x.y; a[2] += (double)x.z; a[3] += (double)x.w;
    b[0] += (double)x.x * (double)y.x; b[1] += (double)x.y * (double)y.y;
    b[2] += (double)x.z * (double)y.z; b[3] += (double)x.w * (double)y.w;
  }
};

// Reduce Stat4 over the sub-groups of a wave (lanes with equal lane % G), then over the
// block's 4 waves through LDS, and write partials[chunk][2][H] (fp64).  G = H/4 lanes per
// row.  `lds` must hold kWavesPerBlock*2*H doubles.
// HF > H: the partial rows are HF wide and this call fills columns [0, H) of them (`partials` already points at the
// first of those columns): a 256-wide layer run as two 128-column problems.
template <int H, int HF = H>
__device__ __forceinline__ void block_stat_store(Stat4& s, double* lds, double* partials, int chunk) {
  constexpr int G = H / 4;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = G; off < 64; off <<= 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s.a[i] += __shfl_xor(s.a[i], off, 64);
      s.b[i] += __shfl_xor(s.b[i], off, 64);
    }
  }
  if (lane < G) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds[(wave * 2 + 0) * H + lane * 4 + i] = s.a[i];
      lds[(wave * 2 + 1) * H + lane * 4 + i] = s.b[i];
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * H; idx += kBlock) {
    double acc = 0.0;
#pragma unroll
    for (int w = 0; w < kWavesPerBlock; ++w) acc += lds[w * 2 * H + idx];
    partials[(size_t)chunk * 2 * HF + (idx / H) * HF + (idx % H)] = acc;
  }
}

// out[i] = sum_b slab[b][i] for the 128 consecutive elements i0 .. i0+127 of this workgroup (256 threads), fixed order ->
// deterministic.  `total` (elements per slab) is a multiple of 128.
// Round 5: the weight-gradient kernels leave one [128,128] slab per workgroup (256-512 of them, 16-32 MB); the first version of
// this reduction gave every output element ONE thread that walked all slabs in four chains -- 49 k threads, ~42 dependent rounds
// of loads 64 KB apart: 0.44 ms per call, 24 calls per step = 10.6 ms of the 166 ms step (profiles/r04_kernel_stats.csv, 5.4 % of
// the GPU time, hidden inside the tn / chained-kernel op times).  Here a workgroup covers 128 elements with 32 float4 lanes x 8
// slab groups (each group walks every 8th slab in four chains, ~6 rounds), and LDS adds the 8 partial sums in a fixed order.
__device__ __forceinline__ float4 slab_reduce_128(const float* __restrict__ slab, int nslab, int64_t total, int i0, float* red /* [8][128] */) {
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const float* p = slab + i0 + c * 4;
  float4 a0 = f4(0.f), a1 = f4(0.f), a2 = f4(0.f), a3 = f4(0.f);
  int b = r;
  for (; b + 24 < nslab; b += 32) {
    a0 += ld4_nt(p + (int64_t)b * total);
    a1 += ld4_nt(p + (int64_t)(b + 8) * total);
    a2 += ld4_nt(p + (int64_t)(b + 16) * total);
    a3 += ld4_nt(p + (int64_t)(b + 24) * total);
  }
  for (; b < nslab; b += 8) a0 += ld4_nt(p + (int64_t)b * total);
  st4(red + r * 128 + c * 4, (a0 + a1) + (a2 + a3));
  __syncthreads();
  float4 s = f4(0.f);
  if (threadIdx.x < 32) {
#pragma unroll
    for (int k = 0; k < 8; ++k) s += ld4(red + k * 128 + c * 4);
  }
  return s;       // valid in threads 0-31: elements i0 + 4 c .. + 3
}

#endif  // __HIPCC__

}  // namespace gnm
