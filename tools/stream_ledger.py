#!/usr/bin/env python3
"""Stream ledger of one training step: every [E,H] / [N,H] tensor of a GatedGCN layer, the kernel that writes
it, every kernel that reads it -- and the same table turned round: per kernel, the streams it must move once ("algorithmic": whole
tensors, perfect reuse of gathered node rows), reconciled with the per-launch HBM traffic the PMC counters measured
(profiles/rNN_traffic.json: TCC FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md).

    python tools/stream_ledger.py profiles/r05_traffic.json > profiles/r06_stream_ledger.txt

Units: eh = 4 E H bytes (one [E,H] fp32 stream), nh = 4 N H bytes.  A gathered node tensor counts one nh (every node row is
needed by some edge; the rows of an edge's endpoints are near each other in the sweep order, so a perfect cache reads each once)."""
import json
import sys

# ---- the tensors of ONE layer (engine.layer_forward / layers_backward_chained), in the order they come to exist ----
#   (name, shape unit, streams, writer, readers)
TENSORS = [
    ("h_in", "nh", 1, "node_update_fwd(i-1) | linear_pe", ["node_proj_fwd", "node_update_fwd (residual)", "tn128_bgrad (B operand)", "tn128[3] (B operand)"]),
    ("P = A1h|A2h|A3h|B1h|B2h", "nh", 5, "node_proj_fwd", ["edge_t: B1h[src], B2h[dst]", "gate2: A2h[src], A3h[dst]", "node_z_stats: A1h",
                                                         "chain(i+1 -> i): A2h[src], A3h[dst]"]),
    ("e_in = e_out(i-1)", "eh", 1, "gate2(i-1) | edge_encoder_fwd", ["edge_t", "gate2 (residual)", "chain(i): TN operand of gW3(i) AND sigma of layer i-1"]),
    ("t", "eh", 1, "edge_t", ["gate2", "chain(i+1 -> i) as t_lo (by-destination half)", "chain(i -> i-1) as t_hi (gt of layer i)"]),
    ("hf, inv_f, hb, inv_b", "nh", 4, "gate2 (+ node_agg_src_fix, gate2_empty_segments)", ["node_z_stats: hf, hb", "node_bwd_apply: inv_f, inv_b",
                                                                                          "chain: hf[dst], hb[src]"]),
    ("z", "nh", 1, "node_z_stats", ["node_update_fwd", "nn2(i+1) epilogue (BatchNorm_h backward sums)", "node_bwd_apply"]),
    ("h_out = h_in(i+1)", "nh", 1, "node_update_fwd", ["(next layer)"]),
    ("gh_out", "nh", 1, "nn2(i+1) | predictor backward", ["node_bwd_apply", "nn2(i) (residual)"]),
    ("gP[:,0:H] = gz,  Q = Qf|Qb", "nh", 3, "node_bwd_apply", ["chain: Qf[dst], Qb[src]", "nn2 / tn128[3]: gz"]),
    ("ge' (in place)", "eh", 1, "chain(i+1 -> i) | predictor backward", ["chain(i -> i-1)"]),
    ("gP[:,H:3H] = gA2h|gA3h,  UT = Us|Ts,  DT = Ud|Td", "nh", 6, "chain (+ edge_bwd_src_fix, zero_empty_segments)", ["tn128_bgrad: the four raw sums",
                                                                                                                 "nn2 / tn128[3]: gA2h, gA3h"]),
    ("gP[:,3H:5H] = gB1h|gB2h", "nh", 2, "tn128_bgrad (formed in its operand load)", ["nn2"]),
    ("gh_in", "nh", 1, "nn2", ["(layer below)"]),
]

# ---- per kernel: algorithmic streams (eh, nh) per launch and launches per step; PMC name = the key of per_launch ----
L = 8
KERNELS = [
    # (display, pmc key prefix, eh, nh, launches/step, note)
    ("node_proj_fwd", "rowtile_nt_k<MmH2, false, 1>", 0, 6, L, "r h_in 1, w P 5"),
    ("edge_t (gnm_edge_t_fused_fwd)", "edge_t32_b3p_k<MmH2>", 2, 2, L, "r e_in, w t; gathers B1h[s] B2h[d]"),
    ("gate2 (gnm_edge_gate2_fwd)", "edge_gate2_fwd_k<true, true, 128, false>", 3, 6, L, "r t, e_in, w e_out; gathers A2h[s] A3h[d]; w hf inv_f hb inv_b"),
    ("node_agg_src_fix", "node_agg_src_fix_k<128>", 0.031 * 1, 0.026 * 2 + 0.15, L, "2.6 % of the nodes, 3.1 % of the edges by gathers"),
    ("node_z_stats", "node_z_stats_k<128>", 0, 4, L, "r A1h hf hb, w z"),
    ("node_update_fwd", "node_update_fwd_k<128, true>", 0, 3, L, "r z h_in, w h_out"),
    ("node_bwd_apply", "node_bwd_apply_k<128>", 0, 7, L, "r z gh_out inv_f inv_b, w gz Qf Qb"),
    # (the PMC pass names both instantiations edge_bwd_chain_k: 7 chained launches + the top sweep, averaged -> one row, 4.875 eh)
    ("chain x7 + top sweep x1 (average)", "edge_bwd_chain_k", (5 * (L - 1) + 4) / L, 12, L,
     "r ge' t_hi e_mid t_lo (top: no t_hi), w ge'; gathers A2h Qb hb [s], Qf hf A3h [d]; w gA2h gA3h Us Ts Ud Td"),
    ("edge_bwd_src_fix", "edge_bwd_src_fix_k<128>", 0.031 * 3, 0.026 * 3 + 0.1, L, "unserved sources by gathers"),
    ("tn128_bgrad", "tn_tr_k<true, 2, true>", 0, 7, L, "r Us Ts Ud Td, h_in; w gB1h gB2h"),
    ("nn2 (gnm_node_proj_bwd_nn_stats)", "rowtile_nn2_k<MmH2, 4>", 0, 8, L - 1, "r gP 5, gh_out, z(i-1); w gh_in"),
    ("nn (layer 0)", "rowtile_nn_group32_b3_k<MmH2, 4>", 0, 7, 1, "r gP 5, gh_out; w gh_in"),
    ("tn128[3] (side stream)", "tn_tr_k<false, 2, true>", 0, 4, L + 1, "r gP[:,0:3H], h_in (+1 launch: the predictor's node halves)"),
    ("fused(0) (gnm_edge_bwd_fused)", "edge_bwd_tr_k<3, true>", 4, 0, 1, "r ge' t e_in, w ge_in (layer 0: nothing below to chain with)"),
    ("edge_encoder_fwd", "edge_encoder_fwd_mfma_k<128>", 1, 0, 1, "w e"),
    ("edge_encoder_bwd", "edge_encoder_bwd_mfma_k<128>", 1, 0, 1, "r ge_in"),
    ("pred_fwd", "pred_fwd_k<true, 128>", 1.5, 1, 1, "r e, w hid [E,64]; gathers Ps[s] Pd[d] ([N,128])"),
    ("pred_bwd", "pred_bwd_k<128>", 3, 0, 1, "r e, r/w hid [E,64] (2 x 0.5), w ge"),
    ("seg_sum_rows (x2)", "seg_sum_rows_k<64>", 0.5, 0.5, 2, "r ghid [E,64] by gathers, w gPs | gPd [N,64]"),
]


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_traffic.json"
    tr = json.load(open(path))
    E, N, H = tr["workload"]["edges"], tr["workload"]["nodes"], tr["workload"]["hidden"]
    eh, nh = 4.0 * E * H / 1e9, 4.0 * N * H / 1e9
    print(f"# stream ledger -- {path} (commit {tr.get('commit')}); E = {E:,}, N = {N:,}, H = {H}, L = {L}; eh = {eh:.3f} GB, nh = {nh:.3f} GB")
    print("#\n# 1. the tensors of one layer: who writes, who reads\n#")
    print(f"{'tensor':54s} {'size':>8s}  written by -> read by")
    for name, unit, k, writer, readers in TENSORS:
        print(f"{name:54s} {k} {unit:>5s}  {writer}")
        for r in readers:
            print(f"{'':66s}<- {r}")
    print("#\n# 2. per kernel: algorithmic streams per launch against the PMC traffic per launch\n#")
    print(f"{'kernel':36s} {'[E,H]':>6s} {'[N,H]':>6s} {'model GB':>9s} {'PMC GB':>8s} {'PMC/model':>9s} {'x/step':>6s} {'GB/step':>8s}   streams")
    tot_model = tot_pmc = tot_e = tot_n = 0.0
    seen = set()
    per = tr["per_launch"]
    for disp, key, ne, nn, times, note in KERNELS:
        hit = [k for k in per if k.startswith(key)]
        pmc = per[hit[0]]["total_gb"] if hit else float("nan")
        seen.update(hit)
        model = ne * eh + nn * nh
        tot_model += model * times
        tot_pmc += (pmc if hit else 0.0) * times
        tot_e += ne * eh * times
        tot_n += nn * nh * times
        print(f"{disp:36s} {ne:6.2f} {nn:6.2f} {model:9.2f} {pmc:8.2f} {pmc / model if model else float('nan'):9.3f} {times:6d} {pmc * times:8.1f}   {note}")
    rest = sum(v["total_gb"] * v["launches"] / tr["steps_profiled"] for k, v in per.items() if k not in seen)
    print(f"{'(everything else: linear_pe and its gradient, the predictor node halves, packs, reductions, fills, Adam)':108s} {rest:8.1f}")
    print(f"#\n# step: model {tot_model:.1f} GB = {tot_e:.1f} edge-shaped ({tot_e / eh:.1f} eh) + {tot_n:.1f} node-shaped ({tot_n / nh:.1f} nh);  "
          f"PMC over the same kernels {tot_pmc:.1f} GB ({tot_pmc / tot_model:.3f} of the model), whole step {tr['per_step_total_gb']:.1f} GB")
    per_layer_e = (2 + 3) + 5
    per_layer_n = 6 + 2 + 6 + 4 + 3 + 7 + 12 + 7 + 8 + 4
    print(f"# per layer (a middle layer): {per_layer_e} [E,H] + {per_layer_n} [N,H] streams = {per_layer_e * eh + per_layer_n * nh:.1f} GB "
          f"(SURVEY 8(d) counts 8 [E,H] and no [N,H]: {8 * eh:.1f} GB)")
    print("#\n# 3. what each candidate removal is worth (streams per layer; GB per step at L = 8)\n#")
    cands = [
        ("t not read by gate2 (formed again in the sweep: +0.25 TFLOP per layer there)", 1, 0),
        ("t not written by edge_t AND not read by gate2 in a forward under no_grad (config 5)", 2, 0),
        ("node_update_fwd folded into the next node_proj_fwd prologue", 0, 1),
        ("z formed inside gate2 (A1h read there; hf, hb not re-read)", 0, 2),
        ("inv_f / inv_b not stored (recomputed from the gate in the backward sweep)", 0, 4),
        ("Qf, Qb not stored (chain gathers gz, inv_f, inv_b instead: +2 gathers per edge)", 0, 3),
        ("the two ends of the stack: predictor backward -> top sweep, fused(0) -> encoder backward", 4.0 / L, 0),
    ]
    for what, de, dn in cands:
        gb = (de * eh + dn * nh) * L
        print(f"  -{de:4.2f} eh -{dn:2d} nh per layer = -{gb:5.1f} GB per step ({100 * gb / tr['per_step_total_gb']:4.1f} %)   {what}")


if __name__ == "__main__":
    main()
