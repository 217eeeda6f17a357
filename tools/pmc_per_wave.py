"""Per-wave averages of a rocprofv3 --pmc counter_collection.csv: python tools/pmc_per_wave.py <csv> [top]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES":
        cnt[k] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:int(sys.argv[2]) if len(sys.argv) > 2 else 6]:
    n = max(cnt[k], 1)
    w = d.get("SQ_WAVES", 1) or 1
    wc = d.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:44s} launches {n:5d} waves/launch {w / n:7.0f} VALU/wave {d.get('SQ_INSTS_VALU', 0) / w:7.0f} SALU/wave {d.get('SQ_INSTS_SALU', 0) / w:7.0f} "
          f"VMEM/wave {d.get('SQ_INSTS_VMEM', 0) / w:6.1f} wave-cycles/wave {wc / w:8.0f} active VALU {d.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f} "
          f"wait_inst {d.get('SQ_WAIT_INST_ANY', 0) / wc:.3f} wait_any {d.get('SQ_WAIT_ANY', 0) / wc:.3f}")
