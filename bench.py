#!/usr/bin/env python3
"""bench.py -- GatedGCN edges/s, fwd+bwd (BASELINE.json metric) on N MI355X of one node.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full training step on one whole synthetic chr19-scale assembly graph per GPU
(BASELINE.json configs[1]: R=750k reads -> N=1.5M nodes, E~7.5M edges, hidden=128, 8 layers):
encoders + 8 GatedGCN layers + score predictor + BCE loss + backward of all of it + (N>1) one
RCCL all-reduce of the flat gradient + Adam step.  Graph, features and index are resident in
HBM before the timed region.  Weak scaling: every rank owns one graph (seed = rank), value =
sum of edges over ranks / max-over-ranks time.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      SURVEY.md section 8(d): the STEP against the HBM roofline -- achieved = algorithmic bytes
                (32*H*L per edge) / measured step time, peak 8 TB/s, frac = achieved / peak.  Inside it:
                `kernels` (every C-ABI op >= 2 % of the step: launches, average launch time from HIP events on
                the launch stream, its own algorithmic GB/s and TFLOP/s against the HBM / matrix-core peaks,
                PMC traffic where a committed rocprofv3 pass has it), `dominant_kernel` (the slowest of them),
                `traffic` = PMC HBM bytes of one step's layer kernels and `traffic_source` = where they
                were measured (the counters cannot be read from inside this process)
  cpu_baseline  the CPU oracle (torch restatement of the reference path; "port") timed on this
                box's host cores on a bounded, smaller sample of the same workload (ratio stated in `sample`)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL; must be set before the HIP runtime starts

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
F32_MFMA_PEAK = 157.3e12   # FLOP/s
BF16_MFMA_PEAK = 2516.6e12 # FLOP/s dense; the bf16x3 mode spends 6 bf16 MACs per fp32 MAC


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=750_000, help="R; N=2R nodes, E~10R edges")
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reads", type=int, default=75_000)
    ap.add_argument("--inference", action="store_true", help="forward only under no_grad (config 5)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) print the cpu_baseline object and exit")
    ap.add_argument("--matmul", default=None, choices=["f32", "bf16x3", "f16x2"],
                    help="fused-kernel matmul mode (default: GNM_MATMUL or the library default f16x2); see include/gnm.h")
    ap.add_argument("--no-alt-matmul", action="store_true", help="skip the extra measurement in the other matmul mode")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads for the CPU baseline (0 = usable cores)")
    ap.add_argument("--shuffle-nodes", action="store_true",
                    help="give the SAME graph randomly shuffled node ids (what the reference's pipeline produces: reads are "
                         "never sorted by position) for the main measurement; edge ids as usual")
    ap.add_argument("--node-order", default=None, choices=["auto", "keep", "bfs"],
                    help="internal node numbering policy of the graph index (default: the package default, 'auto')")
    ap.add_argument("--permute-edge-ids", action="store_true", help="seeded random edge-id permutation (SURVEY 8d second variant)")
    ap.add_argument("--no-alt-orders", action="store_true", help="skip the alt_node_order / alt_edge_ids measurements")
    return ap.parse_args()


def op_model(op: str, N: int, E: int, H: int):
    """(algorithmic HBM bytes, matrix flops) of ONE launch of `op` (fp32).  Bytes = the [E,H] / [N,H]
    streams it must read or write once, assuming perfect reuse of gathered node rows (DESIGN.md
    section 3).  (None, None) for ops not modelled."""
    eh, nh = 4.0 * E * H, 4.0 * N * H
    table = {
        "gnm_edge_t_stats_fwd": (2 * eh + 2 * nh, 0.0),       # t in/out, B1h/B2h rows
        "gnm_edge_gate_fwd": (3 * eh + 4 * nh, 0.0),          # t, e_in in; e_out out; A2h in; hf, inv_f, Td out
        "gnm_node_agg_src_fwd": (1 * eh + 6 * nh, 0.0),       # e_out in; A1h, A3h, hf in; hb, inv_b, z out
        "gnm_edge_bwd_dst": (4 * eh + 9 * nh, 0.0),           # e_out, t, ge in; ge out; Qf hf A3h (own) Qb hb A2h (src) in; gA3h Ud Td out
        "gnm_edge_bwd_src": (3 * eh + 6 * nh, 0.0),           # e_out, t, ge in; Qf Ud Td in; gA2h gB1h gB2h out
        "gnm_edge_bwd_gt": (3 * eh, 0.0),                     # ge, t in; gt out
        "gnm_edge_t_fused_fwd": (2 * eh + 2 * nh, 2.0 * E * H * H),        # e_in in, t out, B1h/B2h rows
        "gnm_edge_t_fused_fwd[256]": (2 * eh + 2 * nh, 2.0 * E * H * H),   # the 256-wide kernel (eight waves, one half of W3 stationary)
        "gnm_edge_bwd_gt_nn": (4 * eh, 2.0 * E * H * H),                   # H = 256: ge, t in; gt, ge_in out; NN
        "gnm_edge_bwd_fused": (4 * eh, 4.0 * E * H * H),                   # ge in/out, t, e_in; NN + TN
        # fused(i) chained with dst(i-1): ge'(i), t(i), e_out(i-1), t(i-1) in; ge'(i-1) out; node rows as edge_bwd_dst
        "gnm_edge_bwd_chain": (5 * eh + 9 * nh, 4.0 * E * H * H),
        # the two-sided sweep: + Qf[dst] is on chip already; gA2h, Us, Ts out
        "gnm_edge_bwd_chain_src": (5 * eh + 12 * nh, 4.0 * E * H * H),
        # gate + both aggregations: t, e_in in; e_out out; A2h, A3h rows in; hf, inv_f, hb, inv_b out; then z: A1h, hf, hb in, z out
        "gnm_edge_gate2_fwd": (3 * eh + 10 * nh, 0.0),
        "gnm_node_bgrad": (6 * nh, 0.0),                      # Us, Ts, Ud, Td in; gB1h, gB2h out
        "gnm_edge_bwd_top": (4 * eh + 12 * nh, 0.0),          # the top layer on the same sweep: e_out, t, ge in; ge out
        "gnm_node_proj_fwd": (6 * nh, 2.0 * N * H * 5 * H),                # h in, P out
        "gnm_node_proj_bwd_nn": (7 * nh, 2.0 * N * H * 5 * H),             # gP, gh_out in; gh_in out
        "gnm_node_proj_bwd_tn": (6 * nh, 2.0 * N * H * 5 * H),             # gP, h_in in
        # round 5 (engine.NODE_FUSED): node_bgrad in the operand load of the gB1h | gB2h weight gradient; node_bwd_stats of the layer
        # below in the projection backward's epilogue; the other three column groups' weight gradient
        "gnm_tn128_bgrad": (7 * nh, 2.0 * N * H * 2 * H),                  # Us, Ts, Ud, Td, h_in in; gB1h, gB2h out
        "gnm_node_proj_bwd_nn_stats": (8 * nh, 2.0 * N * H * 5 * H),       # gP, gh_out, z(below) in; gh_in out
        "gnm_tn128[3]": (4 * nh, 2.0 * N * H * 3 * H),                     # gz | gA2h | gA3h, h_in in
        "gnm_edge_encoder_fwd": (eh, 0.0),
        "gnm_edge_encoder_bwd": (eh, 0.0),
        "gnm_node_bwd_apply": (7 * nh, 0.0),                  # z, gh_out, inv_f, inv_b in; gz, Qf, Qb out
        "gnm_node_bwd_stats": (2 * nh, 0.0),
        "gnm_node_update_fwd": (3 * nh, 0.0),
        "gnm_predictor_fused_fwd": (1.5 * eh + 2 * 4.0 * N * 64, 2.0 * E * H * 64),    # e in, hid out ([E,64])
        "gnm_predictor_fused_bwd": (2 * eh + 4.0 * E * 64, 4.0 * E * H * 64),          # e in, ge out, hid in place
    }
    if op in table:
        return table[op]
    if op.startswith("gemm_"):
        kind = op[5:7]
        M, Nn, K = (int(x) for x in op[op.index("[") + 1:-1].split("x"))
        fl = 2.0 * M * Nn * K
        if kind == "TN":
            return 4.0 * K * (M + Nn), fl                    # both operands stream over the contraction
        return 4.0 * M * (K + Nn) + (4.0 * M * Nn if kind == "NN" else 0.0), fl   # A in, C out (+resid in)
    return None, None


# The committed PMC pass (tools/collect_traffic.sh + tools/traffic_summary.py): HBM bytes per kernel launch on
# this workload.  PMC counters cannot be collected from inside this process, so the bench line carries the
# number together with `traffic_source`; C-ABI op -> rocprof kernel name(s) of the op in each matmul mode.
TRAFFIC_FILE = os.path.join("profiles", "r06_traffic.json")
OP_KERNELS = {
    "gnm_edge_bwd_fused": {"f32": ["edge_bwd_fused32_k"], "bf16x3": ["edge_bwd_tr_k<3, false>"], "f16x2": ["edge_bwd_tr_k<3, true>"]},
    "gnm_edge_bwd_chain": ["edge_bwd_chain_k"], "gnm_edge_bwd_chain_src": ["edge_bwd_chain_k"],
    "gnm_edge_gate2_fwd": ["edge_gate2_fwd_k<true, true, 128, false>"], "gnm_node_bgrad": ["node_bgrad_k<128>"], "gnm_edge_bwd_top": ["edge_bwd_chain_k"],
    "gnm_edge_bwd_dst": ["edge_bwd_dst_k<128>"], "gnm_edge_bwd_src": ["edge_bwd_src_k<128>"],
    "gnm_edge_gate_fwd": ["edge_gate_fwd_k<128, true>"], "gnm_node_agg_src_fwd": ["node_agg_src_fwd_k<128>"],
    "gnm_edge_t_fused_fwd": {"f32": ["rowtile_nt_k<MmF32, true, 1>"], "bf16x3": ["edge_t32_b3p_k<MmB3>"], "f16x2": ["edge_t32_b3p_k<MmH2>"]},
    "gnm_node_proj_fwd": {"f32": ["rowtile_nt_k<MmF32, false, 5>"], "bf16x3": ["rowtile_nt_k<MmB3, false, 1>"], "f16x2": ["rowtile_nt_k<MmH2, false, 1>"]},
    "gnm_node_proj_bwd_nn": {"f32": ["rowtile_nn_acc_k<MmF32>"], "bf16x3": ["rowtile_nn_group32_b3_k<MmB3, 4>"], "f16x2": ["rowtile_nn_group32_b3_k<MmH2, 4>"]},
    "gnm_node_proj_bwd_tn": {"f32": ["tn_colgroup_k<MmF32>"], "bf16x3": ["tn_tr_k<false, 2, false>"], "f16x2": ["tn_tr_k<false, 2, true>"]},
    "gnm_tn128_bgrad": {"bf16x3": ["tn_tr_k<true, 2, false>"], "f16x2": ["tn_tr_k<true, 2, true>"]},
    "gnm_tn128[3]": {"bf16x3": ["tn_tr_k<false, 2, false>"], "f16x2": ["tn_tr_k<false, 2, true>"]},
    "gnm_node_proj_bwd_nn_stats": {"bf16x3": ["rowtile_nn2_k<MmB3, 4>"], "f16x2": ["rowtile_nn2_k<MmH2, 4>"]},
    "gnm_node_bwd_apply": ["node_bwd_apply_k<128>"], "gnm_node_bwd_stats": ["node_bwd_stats_k<128>"],
    "gnm_node_update_fwd": ["node_update_fwd_k<128, true>"],
}


def load_traffic(N, E, H, mode):
    """({op: bytes per launch}, bytes per step or None, source string) from the committed PMC pass -- if it was taken on
    this workload, in this matmul mode AND on these sources (csrc_sha: a kernel change without a new PMC pass must not
    report the old bytes); ({}, None, why not) otherwise."""
    from gnnome_assembly_amd._lib import csrc_sha
    path = os.path.join(REPO, TRAFFIC_FILE)
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return {}, None, f"no PMC pass committed ({TRAFFIC_FILE} missing)"
    w = d.get("workload", {})
    if (w.get("edges"), w.get("nodes"), w.get("hidden")) != (E, N, H) or w.get("matmul", "f32") != mode:
        return {}, None, f"{TRAFFIC_FILE} was taken on another workload / matmul mode"
    sha = csrc_sha()
    if d.get("csrc_sha") != sha:
        return {}, None, (f"stale: {TRAFFIC_FILE} was taken on sources {d.get('csrc_sha')}, the library in use is built from "
                          f"{sha} (re-run tools/collect_traffic.sh)")
    out = {}
    for op, ks in OP_KERNELS.items():
        ks = ks.get(mode, ks.get("bf16x3", [])) if isinstance(ks, dict) else ks
        hit = [d["per_launch"][k]["total_gb"] * 1e9 for k in ks if k in d["per_launch"]]
        if hit:
            out[op] = hit[0]
    step = d.get("per_step_total_gb")
    return out, (step * 1e9 if step else None), (f"{TRAFFIC_FILE} @ {d.get('commit', 'unknown commit')}, sources {sha} (rocprofv3 --pmc "
                                                   "FETCH_SIZE/WRITE_SIZE passes over one bench step, every kernel; FETCH_SIZE x2 per MI355X_MICROARCH.md)")


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(reads, H, L, budget_s=240.0, threads=0, full_reads=750_000):
    """fwd+bwd edges/s of the CPU oracle on this host (all cores torch gives us), on a bounded
    sample: the graph is shrunk until one step fits the time budget (edges/s is size-normalised)."""
    from gnnome_assembly_amd import synth
    from oracle import gatedgcn_oracle as orc
    threads = threads or min(usable_cores(), 64)   # torch-CPU stops scaling well before 64 threads
    torch.set_num_threads(threads)
    t_start = time.time()

    def make(r):
        src, dst, n = synth.make_graph(r, seed=0)
        inp = synth.make_inputs(src, dst, n, seed=0)
        sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.synth_state_dict(H, L, 0).items()}
        ts, td = torch.from_numpy(src).long(), torch.from_numpy(dst).long()
        e, pe, y = torch.from_numpy(inp["e"]), torch.from_numpy(inp["pe"]), torch.from_numpy(inp["y"])
        pw = float(inp["pos_weight"])

        def step():
            for p in sd.values():
                p.grad = None
            loss = orc.bce_loss(orc.model_forward(sd, ts, td, n, e, pe), y, pw)
            loss.backward()
            return loss.item()
        return step, int(src.size), n

    # the same graph, or a 1/10-size one (R = 75 k) when a step exceeds 60 s, 1 warm-up + 3 timed steps.
    # The sample doubles from 1/8 of the requested size while warm-up + 3 timed steps at the NEXT size are predicted to fit
    # the budget (host throughput is strongly size-dependent once the working set leaves the caches, so a size is priced by
    # the one measured before it); the largest size measured is kept.
    r = max(1000, reads // 8)
    while True:
        step, E, n = make(r)
        step()                                   # warm-up at this size
        t0 = time.time()
        step()
        times = [time.time() - t0]
        if r >= reads:
            break
        nxt = min(reads, 2 * r)
        if (time.time() - t_start) + 4 * 1.15 * times[0] * nxt / r > budget_s:
            break
        r = nxt
    while len(times) < 3 and (time.time() - t_start) + times[0] < budget_s:
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    med = float(np.median(times))
    tenth = full_reads // 10
    note = (f"the 1/10-size sample of SURVEY 8(d) (R={tenth}) would take ~{med * tenth / r:.0f} s/step x 4 steps on this host: "
            f"over the {budget_s:.0f} s CPU budget of the default run" if r < tenth else "the 1/10-size sample of SURVEY 8(d)")
    return {"value": E / med, "unit": "edges/s", "cores": threads, "kind": "port",
            "sample": f"R={r} (N={n}, E={E}) = 1/{full_reads / r:.0f} of the GPU workload's R={full_reads} (edges/s is "
                      f"size-normalised; a smaller graph is cache-friendlier, which favours the CPU; {note}), H={H} L={L} "
                      f"fwd+bwd, torch-CPU oracle fp32, median of {len(times)} steps after warm-up ({med:.2f} s/step)"}


def cpu_baseline_subprocess(args, timeout_s=420):
    """Run the CPU leg in its own process with a hard wall-clock bound: the bench line must
    come out within minutes whatever the host does."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-reads", str(args.cpu_reads),
           "--hidden", str(args.hidden), "--layers", str(args.layers), "--cpu-threads", str(args.cpu_threads),
           "--reads", str(args.reads)]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "edges/s", "cores": usable_cores(), "kind": "port",
                "sample": "failed: " + (out.stderr.strip().splitlines() or ["no output"])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "edges/s", "cores": usable_cores(), "kind": "port",
                "sample": f"timed out after {timeout_s} s"}


def dbg(msg):
    if os.environ.get("GNM_BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.cpu_reads, args.hidden, args.layers, threads=args.cpu_threads,
                                      full_reads=args.reads)), flush=True)
        return
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    # GNM_BENCH_DEVICE / GNM_BENCH_BACKEND exist only to exercise the N>1 code path on a 1-GPU box
    # (all ranks on one device, gloo); the driver's multi-GPU runs use one device per rank and RCCL.
    local = int(os.environ.get("GNM_BENCH_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth, engine, dp
    if args.matmul:
        G._lib.set_matmul_mode(args.matmul)
    backend_name = None
    # GNM_FORCE_COLLECTIVE=1: take the distributed path at world size 1 too (process group up, the flat gradient buffer
    # all-reduced through RCCL, per_rank reported): the 8-GPU node is the driver's, this is how the collective path
    # gets executed on the one GPU a lease has (tests/test_gpu_dp.py::test_rccl_all_reduce_executes_at_world_size_one)
    dist_on = world > 1 or dp.FORCE_COLLECTIVE
    if dist_on:
        dp.init_process_group(os.environ.get("GNM_BENCH_BACKEND", "nccl"))
        backend_name = {"nccl": "RCCL"}.get(dist.get_backend(), dist.get_backend())     # what actually carries the all-reduce
        dbg("process group up")

    H, L, R = args.hidden, args.layers, args.reads

    def workload(shuffle_nodes, permute_edges, node_order):
        """The rank's synthetic graph + inputs resident in HBM.  shuffle_nodes: the same graph under a seeded random
        renumbering of its nodes (pe rows follow their nodes); permute_edges: a seeded random edge-id permutation."""
        src, dst, n = synth.make_graph(R, seed=rank, permute_edge_ids=permute_edges)
        inp = synth.make_inputs(src, dst, n, seed=rank)
        pe_np = inp["pe"]
        if shuffle_nodes:
            p = np.random.default_rng(rank + 4242).permutation(n).astype(np.int32)
            src, dst = p[src], p[dst]
            pe_np = np.empty_like(inp["pe"])
            pe_np[p] = inp["pe"]
        t0 = time.perf_counter()
        g = G.AssemblyGraph(src, dst, n, node_order=node_order).to(dev)
        g.index()                                   # index resident in HBM before timing
        t_index = time.perf_counter() - t0
        return {"graph": g, "n": n, "E": int(src.size), "e": torch.from_numpy(inp["e"]).to(dev),
                "pe": torch.from_numpy(pe_np).to(dev), "y": torch.from_numpy(inp["y"]).to(dev),
                "crit": G.BCEWithLogitsLoss(float(inp["pos_weight"])), "index_seconds": t_index,
                "relabel": dict(g.relabel_info)}

    W = workload(args.shuffle_nodes, args.permute_edge_ids, args.node_order)
    n, E = W["n"], W["E"]
    model = G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(H, L, 0, randomize_norm=False).items()})
    model.to(dev)
    model.flatten_parameters()                      # one parameter buffer in the engine's layout; the gradients follow it
    flat = dp.FlatGradients(model.parameters(), direct_write=True)
    opt = dp.make_adam(model.parameters(), 1e-3)

    opt_ev = None   # (start, end) events around optimizer.step() of the per-op timing step: SURVEY 8(d) "optimizer step reported separately"
    own = []        # world > 1: (event before the step, event before the gradient exchange) = this rank's OWN compute

    def step():
        if args.inference:
            with torch.no_grad():
                return model(W["graph"], None, W["e"], W["pe"])
        if dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        flat.zero_()
        scores = model(W["graph"], None, W["e"], W["pe"])
        loss = W["crit"](scores.squeeze(-1), W["y"])
        loss.backward()
        if dist_on:
            e1.record()
            own.append((e0, e1))
        flat.all_reduce_mean()
        if opt_ev is not None:
            opt_ev[0].record()
        opt.step()
        if opt_ev is not None:
            opt_ev[1].record()
        return loss

    dbg(f"setup done E={E}")
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dbg("warmup done")
    per_rank = None

    def timed_run(nsteps):
        """EXACTLY nsteps steps between barrier + synchronize on both sides; max over ranks."""
        nonlocal per_rank
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        own.clear()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        tt = torch.tensor([dt_, float(E)], dtype=torch.float64, device=dev)
        if dist_on:
            tmax = tt[0:1].clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            esum = tt[1:2].clone()
            dist.all_reduce(esum, op=dist.ReduceOp.SUM)
            # per-rank edges, own compute time per step (HIP events: forward + backward up to the gradient
            # exchange) and the fraction of the step the rank spends waiting for the slowest one / in the collective
            mine = sum(a.elapsed_time(b) for a, b in own) / max(len(own), 1)
            rows = torch.zeros(world, 3, dtype=torch.float64, device=dev)
            rows[rank] = torch.tensor([float(E), mine, dt_ / nsteps * 1e3], dtype=torch.float64)
            dist.all_reduce(rows, op=dist.ReduceOp.SUM)
            step_ms = float(tmax.item()) / nsteps * 1e3
            per_rank = [{"rank": r, "edges": int(rows[r, 0].item()), "own_compute_ms_per_step": round(rows[r, 1].item(), 3),
                         "idle_fraction": round(max(0.0, 1.0 - rows[r, 1].item() / step_ms), 4)} for r in range(world)]
            return float(tmax.item()), float(esum.item())
        return dt_, float(E)

    dt, total_edges = timed_run(args.steps)
    per_rank_main = per_rank
    dbg(f"timed region done {dt:.3f}s")
    ms = dt / args.steps * 1e3
    value = total_edges * args.steps / dt

    # per-op timing of ONE extra (untimed) step: HIP events on the launch stream.  Every rank runs
    # the step (it contains the gradient all-reduce); only rank 0 records and reports.
    if rank == 0:
        engine.profile_ops(True)
        opt_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    step()
    ops = engine.profile_ops(False) if rank == 0 else None
    optimizer_ms = None
    if rank == 0 and not args.inference:
        torch.cuda.synchronize()
        optimizer_ms = opt_ev[0].elapsed_time(opt_ev[1])
    opt_ev = None
    dbg("profile step done")
    # the other matmul mode of the fused kernels (include/gnm.h), measured the same way right after the
    # default run; reported beside `value`, never as `value`
    alt = None
    mode = G._lib.get_matmul_mode()
    if args.matmul is None and H == 128 and not args.no_alt_matmul and not args.inference:
        alt = []
        for other in [m for m in ("f16x2", "bf16x3", "f32") if m != mode]:
            G._lib.set_matmul_mode(other)
            step()
            adt, aedges = timed_run(args.steps)
            alt.append({"matmul": other, "ms_per_step": adt / args.steps * 1e3, "value": aedges * args.steps / adt, "unit": "edges/s"})
        G._lib.set_matmul_mode(mode)
        alt[0]["note"] = ("f16x2: fp32 operands as two fp16 terms of a power-of-two multiple (22 significand bits), 3 MFMAs per product, in "
                          "every split-mode matrix kernel; bf16x3: fp32 operands "
                          "split exactly into 3 bf16 terms, 6 bf16 MFMAs per product; f32: every contraction on v_mfma_f32_32x32x2_f32.  "
                          "fp32 accumulate in all three; all three pass the whole of tests/test_gpu_parity.py; distance to the fp64 oracle "
                          "per mode: profiles/r06_f16x2_accuracy.txt")
        dbg("alt matmul run done")
    # the "lean" activation mode (engine.set_activation_mode): one step, for its time and its peak memory
    alt_act = None
    peak_saved = torch.cuda.max_memory_allocated()
    if not args.inference and args.matmul is None and not args.no_alt_matmul and engine.ACTIVATIONS == "saved":
        engine.set_activation_mode("lean")
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        step()
        ldt, ledges = timed_run(max(2, args.steps // 2))
        alt_act = {"activations": "lean", "ms_per_step": ldt / max(2, args.steps // 2) * 1e3,
                   "peak_mem_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                   "note": "P [N,5H] and t [E,H] are rebuilt in the backward by the kernels that made them instead of being "
                           "kept; bit-identical results (tests/test_gpu_parity.py::test_lean_activation_mode...)"}
        engine.set_activation_mode("saved")
    # The same graph under other numberings (never `value`): SURVEY 8(d) asks for a seeded random edge-id permutation
    # beside the src-major ids; VERDICT r2 for randomly shuffled NODE ids -- what the reference's pipeline produces (reads
    # are never sorted by position: pipeline.py:46-61,160-169, graph_parser.py:297-304) -- with the index's internal
    # renumbering ('auto') and without it ('keep' = what every gather kernel sees when node ids are taken as they come).
    alt_orders = {}
    if world == 1 and not args.inference and not args.no_alt_orders and not args.shuffle_nodes and args.matmul is None:
        main_W = dict(W)
        nalt = max(2, args.steps // 2)
        for key, (shuf, perm_e, order) in {"shuffled_node_ids": (True, False, None),
                                           "shuffled_node_ids_kept": (True, False, "keep"),
                                           "alt_edge_ids": (False, True, None)}.items():
            W.clear()
            torch.cuda.empty_cache()
            W.update(workload(shuf, perm_e, order))
            step()
            adt, aedges = timed_run(nalt)
            alt_orders[key] = {"ms_per_step": adt / nalt * 1e3, "value": aedges * nalt / adt, "unit": "edges/s",
                               "steps": nalt, "index_seconds": round(W["index_seconds"], 3), "node_order": W["relabel"]}
        W.clear()
        W.update(main_W)
        dbg("alt order runs done")
    res = None
    if rank == 0:
        tot = sum(t for _, t in ops.values())
        ranked = sorted(ops.items(), key=lambda kv: -kv[1][1])
        # fp32-equivalent FLOP/s of an op's matrix arithmetic: six bf16 MFMAs per product (bf16x3), three fp16 (f16x2), or the fp32 MFMA rate
        def mm_peak_of(op):
            if mode == "f32" or op.startswith(("gnm_predictor", "gnm_edge_encoder")):       # those run fp32 MFMAs in every mode
                return F32_MFMA_PEAK
            return BF16_MFMA_PEAK / (3 if mode == "f16x2" else 6)
        traffic, step_traffic, traffic_src = load_traffic(n, E, H, mode)
        if args.inference:
            step_traffic = None
        kernels = []
        for k, (c, tms) in ranked:
            if tms < 0.02 * tot:
                continue
            ab, fl = op_model(k, n, E, H)
            mm_peak = mm_peak_of(k)
            avg_s = tms / c / 1e3
            row = {"op": k, "launches_per_step": c, "avg_launch_ms": round(avg_s * 1e3, 4), "share_of_step": round(tms / tot, 4)}
            if ab:
                row["hbm"] = {"algorithmic_gb": round(ab / 1e9, 3), "achieved_gbps": round(ab / avg_s / 1e9, 1),
                              "frac": round(ab / avg_s / HBM_PEAK, 4)}
            if fl:
                row["mfma"] = {"tflop": round(fl / 1e12, 4), "achieved_tflops": round(fl / avg_s / 1e12, 1),
                               "peak_tflops": round(mm_peak / 1e12, 1), "frac": round(fl / avg_s / mm_peak, 4)}
            if ab is not None:
                row["bound"] = "mfma" if (fl or 0.0) / mm_peak > ab / HBM_PEAK else "hbm"
            if k in traffic:
                row["traffic_gb"] = round(traffic[k] / 1e9, 3)
            kernels.append(row)
        per_edge = (12 if args.inference else 32) * H * L          # algorithmic B / edge / step
        alg_bytes = per_edge * total_edges / world                  # per GPU and step
        achieved = alg_bytes / (ms / 1e3)
        roof = {"scope": "whole training step on one GPU (SURVEY.md 8d)" if not args.inference else "whole forward pass",
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "algorithmic_bytes_per_edge_step": per_edge,
                "traffic": step_traffic, "traffic_source": traffic_src,
                "rocprof_tables": "profiles/r06_kernel_stats_serial.csv (rocprofv3 --kernel-trace --stats of this command with GNM_TN_SIDE=0: one stream, "
                                  "as the per-op HIP-event step below runs -- its average durations are the ones that agree with avg_launch_ms and sum to "
                                  "the step); profiles/r06_kernel_stats.csv (the two-stream schedule `value` is measured on: a kernel that waits for CUs "
                                  "beside the side-stream weight gradient shows the wait as its duration)",
                "kernels": kernels, "dominant_kernel": kernels[0] if kernels else None}
        if kernels:     # scalars: a parser that keeps only scalar fields still has the kernel-level roofline
            k0 = kernels[0]
            roof.update(dominant_kernel_name=k0["op"], dominant_kernel_ms=k0["avg_launch_ms"],
                        dominant_kernel_launches=k0["launches_per_step"], dominant_kernel_share_of_step=k0["share_of_step"],
                        dominant_kernel_hbm_frac=k0.get("hbm", {}).get("frac"),
                        dominant_kernel_traffic_gb=k0.get("traffic_gb"))
            # the same kernel priced with SURVEY.md 8(d)'s own per-edge figure instead of this repo's stream count (VERDICT r4
            # weak #6): a layer's whole backward = 20 H bytes per edge, its whole forward = 12 H, whichever side the op is on
            bwd_side = "bwd" in k0["op"] or "tn128" in k0["op"]
            b8d = (20 if bwd_side else 12) * H * E
            roof.update(dominant_kernel_8d_bytes_per_edge=(20 if bwd_side else 12) * H,
                        dominant_kernel_hbm_frac_8d=round(b8d / (k0["avg_launch_ms"] / 1e3) / HBM_PEAK, 4))
        res = {
            "metric": "GatedGCN edges/sec fwd+bwd, chr19 assembly graph" if not args.inference
                      else "GatedGCN edges/sec fwd only (inference)",
            "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "f32 (bf16x3 split products, f32 accumulate)", "f16x2": "f32 (f16x2 split products, f32 accumulate)"}.get(mode, "f32"), "data": "synthetic",
            "config": {"workload": f"synthetic chr19-scale assembly graph per GPU: R={R} reads, N={n} nodes, "
                                   f"E={E} edges, hidden={H}, layers={L}, BCE fwd+bwd + Adam"
                                   + (f", {backend_name} grad all-reduce" if dist_on else ""),
                       "reads": R, "nodes": n, "edges": E, "edges_total": int(total_edges), "hidden": H, "layers": L,
                       "parallelism": f"dp{world}", "edge_layers_per_s": value * L, "matmul": mode,
                       "activations": engine.ACTIVATIONS,
                       "node_ids": "randomly shuffled" if args.shuffle_nodes else "position-sorted (SURVEY 8d generator)",
                       "edge_ids": "seeded random permutation" if args.permute_edge_ids else "src-major",
                       "node_order": W["relabel"], "index_seconds": round(W["index_seconds"], 3)},
            "roofline": roof,
            "op_ms": {k: round(v[1], 3) for k, v in ranked},
            "op_total_ms": round(tot, 3),
            "optimizer_ms": None if optimizer_ms is None else round(optimizer_ms, 3),
            "peak_mem_gib": round(peak_saved / 2 ** 30, 1),
        }
        if alt:
            res["alt_matmul"] = alt
        if alt_act:
            res["alt_activations"] = alt_act
        if alt_orders:
            res["alt_node_order"] = {k: v for k, v in alt_orders.items() if k != "alt_edge_ids"}
            res["alt_edge_ids"] = alt_orders.get("alt_edge_ids")
        if per_rank_main:
            res["per_rank"] = per_rank_main
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline_subprocess(args)
        print(json.dumps(res), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
