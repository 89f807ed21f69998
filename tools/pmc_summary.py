#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, fraction of wave time spent
issuing (active), parked on s_waitcnt/barrier (wait_any), stalled on issue dependencies such as a
busy matrix pipe (wait_inst), MFMA pipe busy fraction, LDS bank-conflict ratio."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter(); waves = {}; dur = collections.defaultdict(float)
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void gnm::", "").replace("gnm::", "")[:34]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        n[k] += 1
        waves[k] = int(r["Grid_Size"]) // 64
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"]):
    wc = v["SQ_WAVE_CYCLES"]
    if wc < 1e8:
        continue
    simds = 1024.0
    print(f"{k:34s} calls={n[k]} ms={dur[k]/n[k]:6.2f} waves={waves[k]:6d} active={v['SQ_ACTIVE_INST_ANY']/wc:5.2f} "
          f"wait_any={v['SQ_WAIT_ANY']/wc:5.2f} wait_inst={v['SQ_WAIT_INST_ANY']/wc:5.2f} "
          f"clk_GHz={wc*4/waves[k]/ (dur[k]*1e6):5.2f} mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES']/simds/(dur[k]*1e6)/ (wc*4/waves[k]/(dur[k]*1e6)):5.2f} "
          f"lds_conf={v['SQ_LDS_BANK_CONFLICT']/max(v['SQ_LDS_IDX_ACTIVE'],1):5.3f}")
