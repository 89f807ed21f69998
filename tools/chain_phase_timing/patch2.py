"""Per-role, per-segment clock stamps for edge_bwd_chain2_k (the shipped chained kernel): patches a COPY of gnm_tr.hip.
segments: matrix role  0 phase 0 | 1 wait at barrier 1 | 2 prefetch + MFMAs + og update | 3 wait at barrier 2
          gather role  0 walk    | 1 wait at barrier 1 | 2 arithmetic + prefetches      | 3 wait at barrier 2"""
import sys
p = sys.argv[1]
s = open(p).read()


def rep(old, new, cnt=1):
    global s
    assert s.count(old) == cnt, (old[:80], s.count(old))
    s = s.replace(old, new)


rep("__global__ __launch_bounds__(CT, 2) void edge_bwd_chain2_k(const ChainArgs a) {",
    """__device__ long long g_chain_dbg[256 * 8 * 12];
#define TS(n) { const long long t_ = clock64(); tacc[n] += t_ - tlast; tlast = t_; }
__global__ __launch_bounds__(CT, 2) void edge_bwd_chain2_k(const ChainArgs a) {
  long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tlast = clock64();""")
rep("      float* efb = ef + (int)(k & 1) * ER * SW;\n      // ---- phase 0:", "      float* efb = ef + (int)(k & 1) * ER * SW;\n      tlast = clock64();\n      // ---- phase 0:")
rep("      __syncthreads();   // (1) images and rows of tile k staged", "      TS(0)\n      __syncthreads();   // (1) images and rows of tile k staged")
rep("      prefetch(k + 1 < klast ? k + 1 : klast);          // a tile ahead", "      TS(1)\n      prefetch(k + 1 < klast ? k + 1 : klast);          // a tile ahead")
rep("      __syncthreads();   // (2) og(k) = ge(i-1) rows of tile k complete; the images may be restaged", "      TS(2)\n      __syncthreads();   // (2) og(k) = ge(i-1) rows of tile k complete; the images may be restaged\n      TS(3)")
rep("        chain2_walk(sd + (int)((k + 2) % 3) * 2 * ER + ER,", "        tlast = clock64();\n        chain2_walk(sd + (int)((k + 2) % 3) * 2 * ER + ER,")
rep("        __syncthreads();   // (1)\n", "        TS(0)\n        __syncthreads();   // (1)\n        TS(1)\n")
rep("        ring_put(k + 1);                       //", "        ring_put(k + 1);\n        TS(4)                                  //")
rep("        if ((k & 7) == 7) {                    // fp32 over sixteen rows", "        TS(5)\n        if ((k & 7) == 7) {                    // fp32 over sixteen rows")
rep("        prefetch_pl(k + 1 < klast ? k + 1 : klast);\n        gather();", "        TS(6)\n        prefetch_pl(k + 1 < klast ? k + 1 : klast);\n        gather();")
rep("        __syncthreads();   // (2) term images of tile k and the ring slot of tile k+1 complete", "        TS(2)\n        __syncthreads();   // (2) term images of tile k and the ring slot of tile k+1 complete\n        TS(3)")
rep("  // ---- results: gW3 slab (matrix role)", """  if ((tid & 63) == 0 && blockIdx.x < 256) {
    for (int q = 0; q < 10; ++q) g_chain_dbg[(blockIdx.x * 8 + wave) * 12 + q] = tacc[q];
    g_chain_dbg[(blockIdx.x * 8 + wave) * 12 + 10] = ntile;
  }
  // ---- results: gW3 slab (matrix role)""")
s = s.rstrip('\n') + """

extern "C" int gnm_debug_chain_timing(long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gnm::g_chain_dbg), sizeof(long long) * 256 * 8 * 12);
}
"""
open(p, 'w').write(s)
