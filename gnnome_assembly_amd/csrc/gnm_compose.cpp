// Composite entry points (ABI 5): the launch sequences of the measured path behind ONE call each, on one stream, for hosts that
// do not want to schedule the kernels themselves (a C++ trainer, a Go / Rust binding): gnm_layer_forward = what
// engine.layer_forward issues for a 128-wide BatchNorm layer, gnm_stack_backward = engine.layers_backward_chained (two-sided
// sweeps through the graph's plans, fused node side) without its side stream.  Host code only: every arithmetic step is one of
// the extern "C" kernels' entry points declared in gnm.h, called in the order the Python engine calls them, so the results are
// the engine's bit for bit (tests/cabi/host_step.cpp runs both and compares).  Memory stays the caller's: every output, every
// intermediate and every scratch buffer is passed in.
#include <algorithm>

#include "gnm_common.h"

using namespace gnm;

#define GNM_TRY(x)            \
  do {                        \
    const int rc__ = (x);     \
    if (rc__ != 0) return rc__; \
  } while (0)

extern "C" size_t gnm_compose_workspace_bytes(int H) {
  if (H != 128) return 0;
  return std::max(std::max(gnm_node_proj_bwd_workspace_bytes(5 * H), gnm_edge_bwd_fused_workspace_bytes()),
                  std::max(gnm_tn128_workspace_bytes(), gnm_rowtile_workspace_bytes(5 * H)));
}
extern "C" size_t gnm_compose_partials_doubles(void) { return (size_t)(kMaxPartialBlocks + 1) * 2 * 256; }

static int check_common(const char* who, const gnm_graph_view* g, int H, const gnm_scratch* sc) {
  GNM_CHECK_ARG(g && sc, "%s: null graph / scratch", who);
  GNM_CHECK_ARG(H == 128, "%s: H=%d (the composite entry points are built for the 128-wide fused kernels)", who, H);
  GNM_CHECK_ARG(g->N > 0 && g->E > 0 && g->isrc && g->idst && g->in_ptr && g->out_ptr && g->out_pos && g->out_dst,
                "%s: incomplete graph index", who);
  GNM_CHECK_ARG(sc->partials && sc->ws && sc->ws_bytes >= gnm_compose_workspace_bytes(H),
                "%s: scratch: partials (gnm_compose_partials_doubles()) and ws >= gnm_compose_workspace_bytes(H)", who);
  return 0;
}

extern "C" int gnm_layer_forward(const gnm_graph_view* g, int H, const gnm_layer_weights* w, const gnm_layer_state* s,
                                 const gnm_scratch* sc, void* stream) {
  GNM_TRY(check_common("layer_forward", g, H, sc));
  GNM_CHECK_ARG(w && s && w->W5 && w->b5 && w->W3 && w->b3 && w->gamma_e && w->beta_e && w->gamma_h && w->beta_h,
                "layer_forward: null weights");
  GNM_CHECK_ARG(s->h_in && s->e_in && s->P && s->t && s->e_out && s->hf && s->inv_f && s->hb && s->inv_b && s->z && s->h_out &&
                    s->stat_e && s->stat_h, "layer_forward: null state buffer");
  const int64_t N = g->N, E = g->E;
  int nblk = 0;
  GNM_TRY(gnm_node_proj_fwd(N, H, 5 * H, s->h_in, w->W5, w->b5, s->P, sc->ws, sc->ws_bytes, stream));
  GNM_TRY(gnm_edge_t_fused_fwd(E, H, s->e_in, w->W3, w->b3, s->P, g->isrc, g->idst, s->t, sc->partials, &nblk, sc->ws, sc->ws_bytes,
                               stream));
  GNM_TRY(gnm_bn_finalize(sc->partials, nblk, E, H, w->gamma_e, w->beta_e, 1e-5f, s->stat_e, stream));
  if (g->fwd_sinfo) {       // gate + both aggregations + z in one two-sided sweep
    GNM_CHECK_ARG(g->fwd_dinfo && (g->fwd_nfix == 0 || g->fwd_fix_nodes), "layer_forward: incomplete forward sweep plan");
    GNM_TRY(gnm_edge_gate2_fwd(N, E, H, s->t, s->e_in, s->stat_e, s->P, g->isrc, g->idst, g->in_ptr, g->fwd_sinfo, g->fwd_dinfo,
                               g->fwd_nodes_per_block, g->fwd_nfix, g->fwd_fix_nodes, g->out_ptr, g->out_pos, g->out_dst, s->e_out,
                               s->hf, s->inv_f, s->hb, s->inv_b, s->z, sc->partials, &nblk, stream));
  } else {                  // a graph without a plan: the separate passes
    GNM_TRY(gnm_edge_gate_fwd(N, E, H, s->t, s->e_in, s->stat_e, s->P, g->isrc, g->in_ptr, s->e_out, s->hf, s->inv_f, stream));
    GNM_TRY(gnm_node_agg_src_fwd(N, E, H, s->e_out, s->P, g->out_ptr, g->out_pos, g->out_dst, s->hf, s->hb, s->inv_b, s->z,
                                 sc->partials, &nblk, stream));
  }
  GNM_TRY(gnm_bn_finalize(sc->partials, nblk, N, H, w->gamma_h, w->beta_h, 1e-5f, s->stat_h, stream));
  return gnm_node_update_fwd(N, H, s->z, s->stat_h, s->h_in, s->h_out, stream);
}

extern "C" int gnm_stack_backward(const gnm_graph_view* g, int H, int L, const gnm_layer_weights* w, const gnm_layer_state* s,
                                  const gnm_layer_grads* gr, const float* gh, float* ge, float* gh_in,
                                  const gnm_backward_work* wk, const gnm_scratch* sc, void* stream) {
  GNM_TRY(check_common("stack_backward", g, H, sc));
  GNM_CHECK_ARG(gnm_get_matmul_mode() >= 1, "stack_backward: the chained schedule belongs to the split matmul modes");
  GNM_CHECK_ARG(L >= 1 && w && s && gr && gh && ge && gh_in && wk, "stack_backward: null argument");
  GNM_CHECK_ARG(g->bwd_sinfo && (g->bwd_nfix == 0 || g->bwd_fix_nodes), "stack_backward: needs the backward sweep plan "
                "(gnm_graph_build_sweep_plan over gnm_sweep_partition(N, 1))");
  GNM_CHECK_ARG(wk->gP[0] && wk->gP[1] && wk->Q && wk->UT && wk->DT && wk->gh_tmp[0] && wk->gh_tmp[1] && wk->bstat_e[0] &&
                    wk->bstat_e[1] && wk->bstat_h, "stack_backward: null work buffer");
  GNM_CHECK_ARG(sc->partials2 && sc->partials3 && sc->ws2 && sc->ws2_bytes >= gnm_tn128_workspace_bytes(),
                "stack_backward: scratch needs partials2, partials3 and ws2 >= gnm_tn128_workspace_bytes()");
  const int64_t N = g->N, E = g->E;
  float* const Ud = wk->DT;             // [Ud | Td] in one [N,2H] array
  float* const Td = wk->DT + H;
  const int64_t udp = 2 * H;
  int nblk = 0, nblk_h = 0;
  // ---- top layer: BatchNorm_h backward, then the two-sided sweep without a layer above ----
  int i = L - 1;
  float* gP = wk->gP[i & 1];
  GNM_TRY(gnm_node_bwd_stats(N, H, s[i].z, s[i].stat_h, gh, sc->partials, &nblk, stream));
  GNM_TRY(gnm_bn_bwd_finalize(sc->partials, nblk, N, H, wk->bstat_h, gr[i].g_gamma_h, gr[i].g_beta_h, stream));
  GNM_TRY(gnm_node_bwd_apply(N, H, s[i].z, s[i].stat_h, wk->bstat_h, w[i].gamma_h, gh, s[i].inv_f, s[i].inv_b, gP, wk->Q, stream));
  GNM_TRY(gnm_edge_bwd_top(N, E, H, ge, s[i].e_out, s[i].t, s[i].stat_e, s[i].P, wk->Q, s[i].hf, s[i].hb, g->isrc, g->idst, g->in_ptr,
                           gP, Ud, Td, sc->partials, g->bwd_sinfo, g->bwd_nodes_per_block, wk->UT, &nblk, sc->ws, sc->ws_bytes, stream));
  GNM_TRY(gnm_edge_bwd_src_fix(g->bwd_nfix, g->bwd_fix_nodes, N, E, H, s[i].e_out, s[i].t, s[i].stat_e, ge, wk->Q, g->out_ptr,
                               g->out_pos, g->out_dst, gP, wk->UT, stream));
  const float* gh_cur = gh;
  for (;; --i) {
    float* const bstat_e = wk->bstat_e[i & 1];
    GNM_TRY(gnm_bn_bwd_finalize(sc->partials, nblk, E, H, bstat_e, gr[i].g_gamma_e, gr[i].g_beta_e, stream));
    // gB1h | gB2h formed from the raw sums in the operand load of their weight gradient, written to gP for the projection backward
    GNM_TRY(gnm_tn128_bgrad(N, H, wk->UT, Ud, Td, udp, s[i].stat_e, bstat_e, w[i].gamma_e, g->in_ptr, g->out_ptr, gP, s[i].h_in,
                            gr[i].gW5 + (size_t)3 * H * H, gr[i].gb5 + 3 * H, sc->partials, sc->ws, sc->ws_bytes, stream));
    float* const gh_next = i == 0 ? gh_in : wk->gh_tmp[i & 1];
    if (i > 0)      // + the BatchNorm_h backward sums of the layer below in the epilogue
      GNM_TRY(gnm_node_proj_bwd_nn_stats(N, H, 5 * H, gP, w[i].W5, gh_cur, gh_next, s[i - 1].z, s[i - 1].stat_h, sc->partials, &nblk_h,
                                         sc->ws, sc->ws_bytes, stream));
    else
      GNM_TRY(gnm_node_proj_bwd_nn(N, H, 5 * H, gP, w[i].W5, gh_cur, gh_next, sc->ws, sc->ws_bytes, stream));
    GNM_TRY(gnm_tn128(N, gP, 5 * H, 3, s[i].h_in, gr[i].gW5, gr[i].gb5, sc->partials3, sc->ws2, sc->ws2_bytes, stream));
    gh_cur = gh_next;
    if (i == 0)
      return gnm_edge_bwd_fused(E, H, ge, ge, s[0].t, s[0].e_in, s[0].stat_e, bstat_e, w[0].gamma_e, w[0].W3, gr[0].gW3, gr[0].gb3,
                                sc->partials, sc->ws, sc->ws_bytes, stream);
    const int j = i - 1;
    float* const gPj = wk->gP[j & 1];
    GNM_TRY(gnm_bn_bwd_finalize(sc->partials, nblk_h, N, H, wk->bstat_h, gr[j].g_gamma_h, gr[j].g_beta_h, stream));
    GNM_TRY(gnm_node_bwd_apply(N, H, s[j].z, s[j].stat_h, wk->bstat_h, w[j].gamma_h, gh_cur, s[j].inv_f, s[j].inv_b, gPj, wk->Q, stream));
    // fused edge backward of layer i chained with the two-sided sweep of layer j
    GNM_TRY(gnm_edge_bwd_chain_src(N, E, H, ge, ge, s[i].t, s[i].e_in, s[i].stat_e, bstat_e, w[i].gamma_e, w[i].W3, gr[i].gW3, gr[i].gb3,
                                   sc->partials2, s[j].t, s[j].stat_e, s[j].P, wk->Q, s[j].hf, s[j].hb, g->isrc, g->idst, g->in_ptr, gPj,
                                   Ud, Td, sc->partials, g->bwd_sinfo, g->bwd_nodes_per_block, wk->UT, &nblk, sc->ws, sc->ws_bytes, stream));
    GNM_TRY(gnm_edge_bwd_src_fix(g->bwd_nfix, g->bwd_fix_nodes, N, E, H, s[j].e_out, s[j].t, s[j].stat_e, ge, wk->Q, g->out_ptr,
                                 g->out_pos, g->out_dst, gPj, wk->UT, stream));
    gP = gPj;
  }
}
