// LDS bank-conflict probe for the transposed split staging (ds_write_b128 patterns).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TP = 72, TIMG = 128 * TP;
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(long long* out, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 img[3 * TIMG];
  const int tid = threadIdx.x, lrow = tid >> 5, lc4 = (tid & 31) * 4;
  bf16x8 v;
  for (int i = 0; i < 8; ++i) v[i] = (__bf16)(float)(tid + i);
  int u = lrow;
  if (MODE == 1) u = lrow ^ ((lc4 >> 4) & 3);
  if (MODE == 2) u = lrow ^ ((lc4 >> 4) & 7);
  __bf16* b = img + lc4 * TP + 8 * u;
  if (MODE == 3) b = img + tid * 8;            // contiguous reference
  if (MODE == 4) b = img + (tid & 31) * TP + 8 * lrow;   // 1 column per lane (stride 9 units)
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = (MODE == 3) ? j * 2048 : (MODE == 4 ? j * 32 * TP : j * TP);
      *reinterpret_cast<bf16x8*>(b + o) = v;
      *reinterpret_cast<bf16x8*>(b + o + TIMG) = v;
      *reinterpret_cast<bf16x8*>(b + o + 2 * TIMG) = v;
    }
    __builtin_amdgcn_s_waitcnt(0);
  }
  long long t1 = clock64();
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = t1 - t0;
  if (img[tid] == (__bf16)12345.f) out[1] = 1;
}
int main() {
  long long* d; hipMalloc(&d, 64);
  long long h[2];
  const int iters = 1000;
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), 0, 0, d, iters); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); \
  printf("mode %d: %.1f clk64 ticks per 12 ds_write_b128 (4 waves)\n", M, (double)h[0] / iters);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
  return 0;
}
