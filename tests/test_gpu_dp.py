"""The data-parallel path with the HIP kernels under N > 1 (the reference's per-graph
step is train.py:238-258, it has no DP of its own): two processes, each running the real model on its own
graph, exchange ONE flat gradient.  A 1-GPU box has one device, so both ranks share it and the collective goes
through gloo (GNM_BENCH_DEVICE=0, GNM_BENCH_BACKEND=gloo); with one device per rank the same code uses RCCL.

DP parity contract: exchanged gradient == mean of the two single-rank HIP gradients == mean of the two
single-graph oracle gradients; replicas bit-equal after the Adam step."""
import json
import os
import signal
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import rel_l2

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_group(cmds, envs, timeout):
    """Start the ranks in their own process groups; on timeout kill exactly those groups."""
    procs = [subprocess.Popen(c, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
             for c, e in zip(cmds, envs)]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            outs.append(o)
    except subprocess.TimeoutExpired:
        for p in procs:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        raise AssertionError(f"data-parallel ranks did not finish within {timeout} s:\n" + "\n".join(outs))
    return procs, outs


def _log(name, text):
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", name), "w") as f:
        f.write(text)


@pytest.mark.parametrize("world", [2, 4])      # 4 = the rank count of BASELINE config 3 (4 x chr19 graphs), here on one device
def test_two_ranks_hip_gradients_average_and_replicas_stay_equal(tmp_path, world):
    H, L, R = 128, 3, 1500
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                GNM_BENCH_DEVICE="0", GNM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "tests", "dp_worker.py"), str(tmp_path), str(H), str(L), str(R)]
    procs, outs = _run_group([cmd] * world, [dict(base, RANK=str(r), LOCAL_RANK=str(r)) for r in range(world)], 300)
    _log(f"dp{world}_hip_ranks.log", "\n".join(outs))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # (1) one collective, same result everywhere, replicas bit-equal after Adam
    assert all(np.array_equal(z[0]["reduced"], z[r]["reduced"]) and np.array_equal(z[0]["w"], z[r]["w"]) for r in range(1, world))
    # (2) == mean of the single-rank HIP gradients (gloo sums in fp32 in an order of its own: bitwise for two ranks, to
    #     fp32 round-off for more)
    mean_hip = sum(z[r]["own"].astype(np.float64) for r in range(world)) / world
    if world == 2:
        assert np.array_equal(z[0]["reduced"], ((z[0]["own"] + z[1]["own"]) / np.float32(2)).astype(np.float32))
    assert np.abs(z[0]["reduced"] - mean_hip).max() <= 1e-6 * max(1e-30, np.abs(mean_hip).max())
    # (3) == mean of the two single-graph ORACLE gradients (fp64), per parameter tensor, at the gradient bars of
    #     test_gpu_parity (rel-L2 2e-4, or within 3x the fp32 oracle's own distance: relu-kink noise)
    from gnnome_assembly_amd import synth
    from oracle import gatedgcn_oracle as orc
    sd = synth.synth_state_dict(H, L, seed=0)
    pw = float(z[0]["pos_weight"])

    def oracle(dtype):
        tot = None
        for r in range(world):
            src, dst, n = synth.make_graph(R, seed=r, permute_edge_ids=True)
            inp = synth.make_inputs(src, dst, n, seed=r)
            p = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in sd.items()}
            s = orc.model_forward(p, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).to(dtype),
                                  torch.from_numpy(inp["pe"]).to(dtype))
            orc.bce_loss(s, torch.from_numpy(inp["y"]).to(dtype), pw).backward()
            g = {k: v.grad.double().numpy() / world for k, v in p.items()}
            tot = g if tot is None else {k: tot[k] + g[k] for k in g}
        return tot
    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    o, bad = 0, []
    gmax = max(float(np.linalg.norm(v)) for v in g64.values())
    for k in [str(s) for s in z[0]["order"]]:
        want = g64[k].reshape(-1)
        got = z[0]["reduced"][o:o + want.size].astype(np.float64)
        o += want.size
        r, r32, mx = rel_l2(got, want), rel_l2(g32[k].reshape(-1), want), float(np.abs(got - want).max())
        if not (r <= 2e-4 or r <= 3.0 * r32 + 1e-6 or mx <= max(2e-7, 1e-6 * gmax)):
            bad.append((k, r, r32, mx))
    assert o == z[0]["reduced"].size and not bad, bad


def test_bench_two_ranks_on_one_device(tmp_path):
    """bench.py --gpus 2 exactly as the driver launches it (torch.distributed.run, one JSON line from rank 0),
    with both ranks on the one device of this box and gloo instead of RCCL."""
    port = _free_port()
    env = dict(os.environ, GNM_BENCH_DEVICE="0", GNM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--reads", "30000", "--no-cpu-baseline"]
    procs, outs = _run_group([cmd], [env], 600)
    _log("dp2_bench.log", outs[0])
    assert procs[0].returncode == 0, outs[0][-4000:]
    lines = [ln for ln in outs[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[0][-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak" and r["value"] > 0
    assert r["steps"] == 2 and r["warmup"] == 1
    # whole-job aggregate: the edges of BOTH ranks per max-over-ranks step time
    assert abs(r["value"] - r["config"]["edges_total"] / (r["ms_per_step"] / 1e3)) <= 1e-6 * r["value"]


def test_train_loop_under_two_ranks_mixed_chromosomes(tmp_path):
    """BASELINE.json configs[3] in miniature, EXECUTED: gnnome_assembly_amd.train.train (the counterpart of the
    reference's loop, train.py:232-281,379-529) under world = 2 on the HIP path, three training graphs at
    chr19 : chr20 : chr21 relative sizes (evaluate.py:28-30) sharded 1 + 2 (one padded step per epoch), one
    validation graph on rank 0 only, two epochs, patience 0.  Checks: broadcast initial weights + matched collectives
    (replicas bit-equal after every epoch), the first exchanged gradient == mean of the oracle gradients of the two
    graphs of that step, identical validation losses / LR schedule / best epoch on both ranks, files from rank 0 only,
    no hang."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import train_dp_worker as W
    world, H, L, R, epochs = 2, 128, 3, 1200, 2
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
                GNM_BENCH_DEVICE="0", GNM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(REPO, "tests", "train_dp_worker.py"), str(tmp_path), str(H), str(L), str(R), str(epochs)]
    procs, outs = _run_group([cmd] * world, [dict(base, RANK=str(r), LOCAL_RANK=str(r)) for r in range(world)], 300)
    _log("dp2_train_config4.log", "\n".join(outs))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    j = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    # shards: size-sorted, dealt in alternating directions, of (chr19, chr20, chr21) = sizes (1, 1.073, 0.731): step 0 = (chr20, chr19)
    # to ranks (0, 1), step 1 = (chr21) dealt from the other end: rank 0 [chr20], rank 1 [chr19, chr21]
    assert j[0]["shard"] == [1] and j[1]["shard"] == [0, 2] and j[0]["valid_shard"] == [0] and j[1]["valid_shard"] == []
    assert len(j[0]["step_graph"]) == epochs and len(j[1]["step_graph"]) == 2 * epochs      # rank 0 pads one step per epoch
    # replicas bit-equal after every epoch and at the end
    assert len(j[0]["epoch_hashes"]) == epochs and j[0]["epoch_hashes"] == j[1]["epoch_hashes"]
    assert np.array_equal(z[0]["final"], z[1]["final"]) and list(z[0]["order"]) == list(z[1]["order"])
    # the reduced scalars are the same everywhere
    for k in ("loss_train", "loss_valid", "lr", "final_lr", "best_epoch", "tfpn_train", "tfpn_valid"):
        assert j[0][k] == j[1][k], (k, j[0][k], j[1][k])
    assert all(np.isfinite(j[0]["loss_valid"])) and len(j[0]["loss_valid"]) == epochs
    assert sum(j[0]["tfpn_valid"][0]) > 0                                   # the validation graph was scored (by rank 0 only)
    # only rank 0 wrote files
    assert "checkpoints/cfg4.pt" in j[0]["files"] and j[1]["files"] == []
    assert ("pretrained/model_cfg4.pt" in j[0]["files"]) == (j[0]["best_epoch"] >= 0)
    # first exchanged gradient == mean of the ORACLE gradients of the two graphs of step 0 (bars of the test above)
    assert np.array_equal(z[0]["grad0"], z[1]["grad0"])
    import gnnome_assembly_amd as G
    from gnnome_assembly_amd import synth
    from oracle import gatedgcn_oracle as orc
    train_set, _ = W.dataset(R)
    first = [train_set[j[r]["shard"][j[r]["step_graph"][0]]] for r in range(world)]     # (reads, seed) of each rank's step-0 graph
    torch.manual_seed(0)                                                     # train.train: utils.set_seed, then the model
    sd0 = {k: v.detach().numpy() for k, v in G.GraphGatedGCNModel(1, 2, H, 16, L, 64, True, 16).state_dict().items()}
    ratios = []
    for reads, seed in train_set:                                            # train.py:181: dataset mean of #pos / #neg
        src, dst, n = synth.make_graph(reads, seed=seed, permute_edge_ids=True)
        y = synth.make_inputs(src, dst, n, seed=seed)["y"]
        ratios.append(np.float32((y == 1).sum()) / np.float32((y == 0).sum()))
    pw = 1.0 / float(np.mean(ratios))

    def oracle(dtype):
        tot = None
        for reads, seed in first:
            src, dst, n = synth.make_graph(reads, seed=seed, permute_edge_ids=True)
            inp = synth.make_inputs(src, dst, n, seed=seed)
            p = {k: torch.from_numpy(v).to(dtype).requires_grad_(True) for k, v in sd0.items()}
            s = orc.model_forward(p, torch.from_numpy(src), torch.from_numpy(dst), n, torch.from_numpy(inp["e"]).to(dtype),
                                  torch.from_numpy(inp["pe"]).to(dtype))
            orc.bce_loss(s, torch.from_numpy(inp["y"]).to(dtype), pw).backward()
            g = {k: v.grad.double().numpy() / world for k, v in p.items()}
            tot = g if tot is None else {k: tot[k] + g[k] for k in g}
        return tot
    g64, g32 = oracle(torch.float64), oracle(torch.float32)
    o, bad = 0, []
    gmax = max(float(np.linalg.norm(v)) for v in g64.values())
    for k in [str(s) for s in z[0]["order"]]:
        want = g64[k].reshape(-1)
        got = z[0]["grad0"][o:o + want.size].astype(np.float64)
        o += want.size
        r, r32, mx = rel_l2(got, want), rel_l2(g32[k].reshape(-1), want), float(np.abs(got - want).max())
        if not (r <= 2e-4 or r <= 3.0 * r32 + 1e-6 or mx <= max(2e-7, 1e-6 * gmax)):
            bad.append((k, r, r32, mx))
    assert o == z[0]["grad0"].size and not bad, bad


def test_rccl_all_reduce_executes_at_world_size_one():
    """The RCCL path end to end on the hardware a lease has (one GPU; RCCL refuses two ranks on one device): bench.py
    under torch.distributed.run with ONE rank, backend nccl (= RCCL on ROCm), GNM_FORCE_COLLECTIVE=1 so that the world-size-1
    shortcuts of dp.py / bench.py are not taken: init_process_group("nccl"), device placement from LOCAL_RANK, the
    stream-ordered all-reduce of the flat gradient buffer (826,033 + 1 floats at H = 128 / L = 8), barriers, per_rank.
    NCCL_DEBUG=INFO's tail is kept (gpurun_out/rccl_smoke.log -> profiles/).  The per-graph step this averages over ranks is
    the reference's train.py:238-258."""
    port = _free_port()
    env = dict(os.environ, NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,COLL", GNM_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GNM_BENCH_DEVICE", None)
    env.pop("GNM_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "1", "--reads", "30000", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline", "--no-alt-matmul", "--no-alt-orders"]
    procs, outs = _run_group([cmd], [env], 600)
    out = outs[0]
    lines = out.splitlines()
    nccl = [ln for ln in lines if ("NCCL" in ln or "RCCL" in ln) and " Channel " not in ln]
    _log("rccl_smoke.log", "\n".join(nccl[:30] + ["..."] + nccl[-60:] + [ln for ln in lines if ln.startswith("{")]))
    assert procs[0].returncode == 0, out[-4000:]
    res = json.loads([ln for ln in lines if ln.startswith("{")][-1])
    assert res["n_gpus"] == 1 and "RCCL grad all-reduce" in res["config"]["workload"], res["config"]["workload"]
    assert res["per_rank"] and res["per_rank"][0]["edges"] == res["config"]["edges"]
    assert any("Init COMPLETE" in ln or "init complete" in ln.lower() for ln in nccl), "no RCCL communicator came up"
    assert any("AllReduce" in ln for ln in nccl) or any("opCount" in ln for ln in nccl), "no RCCL collective in the debug log"
