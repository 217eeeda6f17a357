"""Turn the rocprofv3 outputs of a gpurun (gpurun_out/final/) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py gpurun_out/final r01
"""
import collections
import csv
import json
import shutil
import sys


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        name = "k_nn_plan" if "k_nn_plan" in n else "k_nn_l1" if "k_nn_l1" in n else n.split("(")[0].split("::")[-1][:28]
        a[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        a[name]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return a


def main(src, tag):
    f = agg(f"{src}/pmc_FETCH_SIZE_counter_collection.csv")
    w = agg(f"{src}/pmc_WRITE_SIZE_counter_collection.csv")
    sq = agg(f"{src}/pmc_sq_counter_collection.csv")
    lines, out = [], {}
    for k in ["k_l2<8>", "k_head<8>", "k_nn_plan", "k_gradc", "k_bwd2<8, 48>", "k_dw<8>", "k_km_small", "k_sort_y", "k_sort_p"]:
        if k not in f:
            continue
        fs = sum(f[k]["FETCH_SIZE"]) / len(f[k]["FETCH_SIZE"])
        ws = sum(w[k]["WRITE_SIZE"]) / len(w[k]["WRITE_SIZE"])
        m = {c: sum(v) / len(v) for c, v in sq[k].items()}
        wv = m["SQ_WAVES"]
        lines.append(f"{k:14s} FETCH_SIZE={fs:9.1f} KB WRITE_SIZE={ws:9.1f} KB | waves={wv:6.0f} VALU/wave={m['SQ_INSTS_VALU'] / wv:6.0f} "
                     f"LDS/wave={m['SQ_INSTS_LDS'] / wv:5.0f} wave_cycles/wave={4 * m['SQ_WAVE_CYCLES'] / wv:7.0f} "
                     f"active_valu={4 * m['SQ_ACTIVE_INST_VALU'] / wv:6.0f} wait_inst={4 * m['SQ_WAIT_INST_ANY'] / wv:6.0f} "
                     f"wait_any={4 * m['SQ_WAIT_ANY'] / wv:6.0f} dur_us={m['dur'] / 1e3:6.2f}")
        out[k] = (fs, ws)
    header = ("rocprofv3 PMC, `bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-icp-variant` (5 sequences as two graph branches: launches carry "
              "3 or 2 problems, the values below average over both; N=4096, K=20, H=512); one counter set "
              "per pass (FETCH_SIZE | WRITE_SIZE | SQ_*); per-wave values are cycles (quad-cycle counters x4); FETCH_SIZE / WRITE_SIZE are the "
              "raw rocprofv3 values in KB.  MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests of wide (16 B/lane) coalesced reads as 64 B -> "
              "double it for the dwordx4 / LDS-DMA streams (k_nn_plan block reads, k_dw, k_l2 staging); dword-wide reads are uncorrected.\n")
    open(f"profiles/{tag}_pmc_summary.txt", "w").write(header + "\n".join(lines) + "\n")
    fs, ws = out["k_nn_plan"]
    json.dump({"kernel": "k_nn_plan<true>", "problems_per_launch_avg": 2.5, "FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws,
               "hbm_bytes_per_launch_avg": (2 * fs + ws) * 1024, "hbm_bytes_per_problem": (2 * fs + ws) * 1024 / 2.5,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 5 --warmup 5 --no-cpu-baseline), "
                         "averaged over all launches of the kernel; FETCH_SIZE doubled per MI355X_MICROARCH.md (the block reads are 16 B/lane coalesced), "
                         "WRITE_SIZE uncorrected",
               "algorithmic_bytes_per_problem": 2 * 16 * 4096 + 16 * 4096 + 4 * 4096}, open(f"profiles/{tag}_nn_l1_pmc.json", "w"), indent=1)
    shutil.copy(f"{src}/r01_kernel_stats.csv", f"profiles/{tag}_final_kernel_stats.csv")
    shutil.copy(f"{src}/bench.log", f"profiles/{tag}_final_bench.log")
    open(f"profiles/{tag}_final_bench_b1_b8.log", "w").write(open(f"{src}/bench_b1.log").read() + open(f"{src}/bench_b8.log").read())
    open(f"profiles/{tag}_final_bench_other_workloads.log", "w").write(open(f"{src}/bench_franka.log").read() + open(f"{src}/bench_allegro.log").read())
    print("\n".join(lines))
    for r in list(csv.DictReader(open(f"{src}/r01_kernel_stats.csv")))[:8]:
        print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
