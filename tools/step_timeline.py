#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: every kernel of the LAST profiled step with start / end relative
to the step's first kernel, its stream (queue) and which kernels ran beside it.  tools/step_timeline.py kernel_trace.csv [out.txt]"""
import csv
import sys


def short(n):
    n = n.replace("void ", "").replace("gnm::", "")
    return n.split("(")[0][:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows]
    ks.sort()
    # steps: split at the BCE kernel (one per step); keep the kernels between the last two
    cuts = [i for i, k in enumerate(ks) if k[2].startswith("bce_fwd_bwd_k")]
    if len(cuts) < 2:
        print("fewer than two steps in the trace")
        return
    # a step = from the first encoder kernel after the previous bce ... find edge_encoder_fwd
    enc = [i for i, k in enumerate(ks) if k[2].startswith("edge_encoder_fwd")]
    a, b = enc[-2], enc[-1]
    step = ks[a:b]
    t0 = step[0][0]
    out = []
    out.append(f"one step: {len(step)} kernels, {(step[-1][1] - t0) / 1e6:.2f} ms from the first kernel's start to the last kernel's end")
    busy = 0
    cur_end = t0
    for s, e, *_ in step:
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    out.append(f"time with at least one kernel running: {busy / 1e6:.2f} ms; sum of kernel durations: {sum(e - s for s, e, *_ in step) / 1e6:.2f} ms")
    bce = [i for i, k in enumerate(step) if k[2].startswith("bce_fwd_bwd_k")][0]
    out.append(f"forward part: {(step[bce][0] - t0) / 1e6:.2f} ms; backward part: {(step[-1][1] - step[bce][0]) / 1e6:.2f} ms")
    out.append("")
    out.append(f"{'start ms':>9s} {'dur us':>8s} {'queue':>5s}  kernel    [beside: kernels on another queue that overlap it, with the overlap in us]")
    for i, (s, e, n, q, st) in enumerate(step):
        if i < bce - 2:
            continue
        if e - s < 20000 and "tn_tr" not in n:
            continue
        ov = []
        for s2, e2, n2, q2, _ in step:
            if q2 != q and s2 < e and e2 > s and e2 - s2 >= 20000:
                ov.append(f"{n2[:28]} {(min(e, e2) - max(s, s2)) / 1e3:.0f}")
        out.append(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {q:>5s}  {n}" + (f"    [beside: {'; '.join(ov)}]" if ov else ""))
    out.append("")
    out.append("idle gaps of more than 30 us (no kernel running on any queue), with the kernels on either side:")
    cur_end, last = step[0][1], step[0][2]
    nxt = ks[b:b + 40]
    for s, e, n, q, st in step[1:] + nxt:
        if s - cur_end > 30000:
            out.append(f"  {(cur_end - t0) / 1e6:9.3f} ms: {(s - cur_end) / 1e3:7.1f} us between {last} and {n}")
        if e > cur_end:
            cur_end, last = e, n
    out.append("")
    out.append("every kernel from 1.2 ms before the step's last kernel ends to 1 ms into the next step:")
    tend = step[-1][1]
    for s, e, n, q, st in step + nxt:
        if s > tend - 1200000 and s < tend + 1000000:
            out.append(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e3:8.1f} {q:>5s}  {n}")
    txt = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print("\n".join(out[:5]))


if __name__ == "__main__":
    main()
