// Host-side pieces of libgnm.so: error reporting, device query, graph index build.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "gnm_common.h"

namespace gnm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;  // MI355X
  }
  return cus;
}

}  // namespace gnm

extern "C" int gnm_abi_version(void) { return GNM_ABI_VERSION; }
extern "C" const char* gnm_last_error(void) { return gnm::g_err; }
extern "C" int gnm_num_cus(void) { return gnm::num_cus(); }
extern "C" int gnm_max_partial_blocks(void) { return gnm::kMaxPartialBlocks; }

// Stable counting sort of the edge list by destination (internal order) and by source.
// Replaces DGL's lazy CSR/CSC construction and dgl.reverse (gated_gcn_full.py:115): the
// reversed graph is the same edge set indexed by source, so one index serves both passes.
extern "C" int gnm_graph_build_index(const int32_t* src, const int32_t* dst, int64_t N, int64_t E,
                                     int32_t* perm, int32_t* isrc, int32_t* idst, int32_t* in_ptr,
                                     int32_t* out_ptr, int32_t* out_pos, int32_t* out_dst) {
  GNM_CHECK_ARG(N >= 0 && E >= 0 && N < INT32_MAX && E < INT32_MAX, "graph_build_index: N/E out of int32 range");
  GNM_CHECK_ARG((E == 0 || (src && dst)) && perm && isrc && idst && in_ptr && out_ptr && out_pos && out_dst,
                "graph_build_index: null argument");
  for (int64_t k = 0; k < E; ++k) {
    if (src[k] < 0 || src[k] >= N || dst[k] < 0 || dst[k] >= N) {
      gnm::set_error("graph_build_index: edge %lld = (%d -> %d) outside [0, %lld)", (long long)k, src[k],
                     dst[k], (long long)N);
      return -2;
    }
  }
  // by destination, stable in edge id
  std::memset(in_ptr, 0, sizeof(int32_t) * (size_t)(N + 1));
  for (int64_t k = 0; k < E; ++k) in_ptr[dst[k] + 1]++;
  for (int64_t v = 0; v < N; ++v) in_ptr[v + 1] += in_ptr[v];
  {
    std::vector<int32_t> cur(in_ptr, in_ptr + N);
    for (int64_t k = 0; k < E; ++k) {
      const int32_t j = cur[dst[k]]++;
      perm[j] = (int32_t)k;
      isrc[j] = src[k];
      idst[j] = dst[k];
    }
  }
  // by source, stable in internal position
  std::memset(out_ptr, 0, sizeof(int32_t) * (size_t)(N + 1));
  for (int64_t j = 0; j < E; ++j) out_ptr[isrc[j] + 1]++;
  for (int64_t v = 0; v < N; ++v) out_ptr[v + 1] += out_ptr[v];
  {
    std::vector<int32_t> cur(out_ptr, out_ptr + N);
    for (int64_t j = 0; j < E; ++j) {
      const int32_t m = cur[isrc[j]]++;
      out_pos[m] = (int32_t)j;
      out_dst[m] = idst[j];
    }
  }
  return 0;
}
