#!/bin/bash
# Run ON THE GPU BOX: matrix-core utilisation counters of the K2 E-step (both forms) at the C5 shape.
#   gpurun --timeout 600 -- 'bash tools/collect_mfma_pmc.sh'   -> gpurun_out/mfma_pmc/
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/mfma_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CREG_KM_SHAPE=1048576,128
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o a -- python $R/tools/bench_kmeans_assign.py > $O/run_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $O -o b -- python $R/tools/bench_kmeans_assign.py > $O/run_b.log 2>&1
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob("$O/**/%s_counter_collection.csv" % tag, recursive=True)
    if not fs:
        print(tag, "no counter file; log tail:", open("$O/run_%s.log" % tag).read()[-400:]); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        n = r["Kernel_Name"]
        if "k_km_assign" not in n: continue
        key = ("mfma" if "mfma" in n else "valu") + " grid=" + r.get("Grid_Size", "?")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in sorted(agg):
        m = {c: sum(v) / len(v) for c, v in agg[k].items()}
        line = tag + " " + k + " " + str({c: round(v, 1) for c, v in m.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
            line += "  => matrix-pipe utilisation %.3f (busy cycles / (1024 SIMDs x kernel cycles at 2.4 GHz))" % (
                m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * m["dur_ns"] * 2.4))
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in m and m["SQ_INSTS_VALU_MFMA_MOPS_F64"] > 0:
            line += "  => %.2f TFLOP/s fp64 on the matrix cores (MOPS x 512 flops / kernel time)" % (
                m["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / m["dur_ns"] / 1e3)
        print(line)
PY
