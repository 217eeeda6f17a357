"""Turn the rocprofv3 outputs of a gpurun (gpurun_out/final/) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py gpurun_out/final r02
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
    for k in ["k_l2<8>", "k_head<8>", "k_nn_plan", "k_gradc", "k_bwd2<8, 48>", "k_dw<8>", "k_km_small", "k_sort_y", "k_sort_p", "k_masked_icp"]:
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
    dfs, dws = out["k_dw<8>"]
    nnm = {c: sum(v) / len(v) for c, v in sq["k_nn_plan"].items()}
    dwm = {c: sum(v) / len(v) for c, v in sq["k_dw<8>"].items()}
    n_params = 425991
    json.dump({"source": f"rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE / WRITE_SIZE / the SQ set, `bench.py --steps 5 --warmup 5 "
                         "--no-cpu-baseline --no-icp-variant --no-roofline` (tools/collect_profiles.sh); averages over all launches of a kernel, which carry "
                         "3 or 2 problems (2.5 on average); FETCH_SIZE doubled per MI355X_MICROARCH.md for the 16 B/lane coalesced streams, WRITE_SIZE raw",
               "problems_per_launch_avg": 2.5,
               "k_nn_FETCH_SIZE_KB": fs, "k_nn_WRITE_SIZE_KB": ws, "k_nn_hbm_bytes_per_problem": (2 * fs + ws) * 1024 / 2.5,
               "k_nn_algorithmic_bytes_per_problem": 2 * 16 * 4096 + 16 * 4096 + 4 * 4096,
               "k_nn_valu_active_frac": nnm["SQ_ACTIVE_INST_VALU"] / nnm["SQ_WAVE_CYCLES"],
               "k_nn_wait_any_frac": nnm["SQ_WAIT_ANY"] / nnm["SQ_WAVE_CYCLES"],
               "k_dw_FETCH_SIZE_KB": dfs, "k_dw_WRITE_SIZE_KB": dws, "k_dw_hbm_bytes_per_problem": (2 * dfs + dws) * 1024 / 2.5,
               "k_dw_algorithmic_bytes_per_problem": 24 * n_params,
               "k_dw_valu_active_frac": dwm["SQ_ACTIVE_INST_VALU"] / dwm["SQ_WAVE_CYCLES"]},
              open(f"profiles/{tag}_pmc.json", "w"), indent=1)
    shutil.copy(f"{src}/{tag}_kernel_stats.csv", f"profiles/{tag}_final_kernel_stats.csv")
    shutil.copy(f"{src}/c5_kernel_stats.csv", f"profiles/{tag}_c5_kernel_stats.csv")
    shutil.copy(f"{src}/bench.log", f"profiles/{tag}_final_bench.log")
    open(f"profiles/{tag}_final_bench_b1_b8.log", "w").write(open(f"{src}/bench_b1.log").read() + open(f"{src}/bench_b8.log").read())
    open(f"profiles/{tag}_final_bench_other_workloads.log", "w").write(open(f"{src}/bench_franka.log").read() + open(f"{src}/bench_allegro.log").read())
    open(f"profiles/{tag}_final_bench_replay_and_c5.log", "w").write(open(f"{src}/bench_replay_allegro.log").read() + open(f"{src}/bench_c5.log").read())
    shutil.copy(f"{src}/c5_resegment.log", f"profiles/{tag}_c5_resegment_bench.log")
    shutil.copy(f"{src}/icp_frame_phases.log", f"profiles/{tag}_icp_frame_phases.log")
    print("\n".join(lines))
    for r in list(csv.DictReader(open(f"{src}/{tag}_kernel_stats.csv")))[:8]:
        print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
