"""Turn the rocprofv3 outputs of a gpurun (gpurun_out/final/) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py gpurun_out/final r04 [output directory, default profiles/]

tools/collect_profiles.sh runs it ON THE GPU BOX into gpurun_out/r04_profiles/ (the raw counter CSVs are tens of MB each and
stay there); copy that directory's files into profiles/ afterwards.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROLES = (("bd", "k_bd<"), ("nn_l1", "k_nn_"), ("l2", "k_l2<"), ("gradc", "k_gradc"), ("head", "k_head<"))
EXTRA = ("k_km_small", "k_sort_y", "k_sort_p", "k_masked_icp", "k_l1", "k_params_home")


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    names = {}
    if not os.path.exists(path):
        return a, names
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        key = next((role for role, pat in ROLES if pat in n), None) or next((e for e in EXTRA if e in n), None)
        if key is None:
            continue
        names[key] = n.split("(")[0].replace("void ", "").replace("creg::", "")
        a[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        a[key]["dur"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return a, names


def mean(v):
    return sum(v) / len(v) if v else float("nan")


def main(src, tag, dst="profiles"):
    os.makedirs(dst, exist_ok=True)
    lines, out = [], {}
    for wl, per in (("wx200_5", 2.5), ("franka", 2.5), ("allegro", 2.5)):
        f, _ = agg(f"{src}/pmc_{wl}_FETCH_SIZE_counter_collection.csv")
        w, _ = agg(f"{src}/pmc_{wl}_WRITE_SIZE_counter_collection.csv")
        sq, names = agg(f"{src}/pmc_{wl}_sq_counter_collection.csv")
        mf, _ = agg(f"{src}/pmc_{wl}_mfma_counter_collection.csv")
        if not sq:
            continue
        lines.append(f"--- {wl}: bench.py --workload {wl} --steps 5 --warmup 5 (5 sequences as two chains: launches carry 3 or 2 problems)")
        ks = {}
        for key in [r for r, _ in ROLES] + list(EXTRA):
            if key not in sq:
                continue
            m = {c: mean(v) for c, v in sq[key].items()}
            wv = m["SQ_WAVES"]
            fs, ws = mean(f[key]["FETCH_SIZE"]) if key in f else float("nan"), mean(w[key]["WRITE_SIZE"]) if key in w else float("nan")
            lines.append(f"{names[key]:22s} FETCH_SIZE={fs:9.1f} KB WRITE_SIZE={ws:9.1f} KB | waves={wv:6.0f} VALU/wave={m['SQ_INSTS_VALU'] / wv:6.0f} "
                         f"LDS/wave={m['SQ_INSTS_LDS'] / wv:5.0f} wave_cycles/wave={4 * m['SQ_WAVE_CYCLES'] / wv:7.0f} "
                         f"active_valu={4 * m['SQ_ACTIVE_INST_VALU'] / wv:6.0f} wait_inst={4 * m['SQ_WAIT_INST_ANY'] / wv:6.0f} "
                         f"wait_any={4 * m['SQ_WAIT_ANY'] / wv:6.0f} dur_us={m['dur'] / 1e3:6.2f}")
            if key in [r for r, _ in ROLES]:
                ks[key] = {"kernel": names[key], "FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "waves": wv,
                           "valu_insts_per_wave": m["SQ_INSTS_VALU"] / wv, "active_valu_cycles_per_wave": 4 * m["SQ_ACTIVE_INST_VALU"] / wv,
                           "wave_cycles_per_wave": 4 * m["SQ_WAVE_CYCLES"] / wv, "wait_any_cycles_per_wave": 4 * m["SQ_WAIT_ANY"] / wv,
                           "dur_us_profiled": m["dur"] / 1e3}
                if key in mf and mean(mf[key].get("SQ_INSTS_MFMA", [0])) > 0:
                    mm = {c: mean(v) for c, v in mf[key].items()}
                    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (MI355X_MICROARCH.md); v_mfma_f32_16x16x4_f32 = 2048 flops, MOPS = 512-flop units
                    ks[key].update({"mfma_insts_per_wave": mm["SQ_INSTS_MFMA"] / mm["SQ_WAVES"], "mfma_mops_f32": mm.get("SQ_INSTS_VALU_MFMA_MOPS_F32", float("nan")),
                                    "mfma_busy_cycles": mm["SQ_VALU_MFMA_BUSY_CYCLES"],
                                    "matrix_pipe_utilisation": mm["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * mm["dur"] * 2.4)})
                    lines.append(f"{'':22s} matrix cores: {mm['SQ_INSTS_MFMA'] / mm['SQ_WAVES']:.1f} MFMA per wave, MOPS_F32 {mm.get('SQ_INSTS_VALU_MFMA_MOPS_F32', float('nan')):.0f} per launch, "
                                 f"busy cycles {mm['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} = {mm['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * mm['dur'] * 2.4):.4f} of the launch's 1024 SIMD x cycles")
        out[wl] = {"problems_per_launch_avg": per, "kernels": ks}
    header = ("rocprofv3 PMC per workload (tools/collect_profiles.sh); one counter set per pass (FETCH_SIZE | WRITE_SIZE | SQ_*), every pass with "
              "--kernel-trace only; averages over all launches of a kernel (5 sequences as two chains: 3 or 2 problems per launch, 2.5 "
              "on average); per-wave values are cycles (quad-cycle counters x4); FETCH_SIZE / WRITE_SIZE are the raw rocprofv3 values in KB.  "
              "MI355X_MICROARCH.md: FETCH_SIZE counts 128-B requests of wide (16 B/lane) coalesced reads as 64 B -> bench.py doubles it "
              "before comparing with a byte count; narrower reads are uncalibrated, so `traffic` of the latency-bound kernels is an upper bound.\n")
    open(f"{dst}/{tag}_pmc_summary.txt", "w").write(header + "\n".join(lines) + "\n")
    out["source"] = ("rocprofv3 --kernel-trace --pmc, separate passes for FETCH_SIZE / WRITE_SIZE / the SQ set per workload, `bench.py --workload W "
                     "--steps 5 --warmup 5 --no-cpu-baseline --no-icp-variant --no-roofline` (tools/collect_profiles.sh)")
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from autourdf_amd.build import kernel_source_sha256
        out["kernel_source_sha256"] = kernel_source_sha256()       # of the tree the counters were collected from (bench.py: `stale`)
    except Exception as e:
        out["kernel_source_sha256"] = None
        print("no source fingerprint:", e)
    json.dump(out, open(f"{dst}/{tag}_pmc.json", "w"), indent=1)
    cp = lambda a, b: os.path.exists(f"{src}/{a}") and shutil.copy(f"{src}/{a}", f"{dst}/{b}")
    cp("stats_kernel_stats.csv", f"{tag}_final_kernel_stats.csv")
    cp("c5_kernel_stats.csv", f"{tag}_c5_kernel_stats.csv")
    cp("bench.log", f"{tag}_final_bench.log")
    dst_dir = dst
    cat = lambda names, dst: open(f"{dst_dir}/{dst}", "w").write("".join(open(f"{src}/{n}").read() for n in names if os.path.exists(f"{src}/{n}")))
    cat(["bench_b1.log", "bench_b8.log"], f"{tag}_final_bench_b1_b8.log")
    cat(["bench_franka.log", "bench_allegro.log"], f"{tag}_final_bench_other_workloads.log")
    cat(["bench_replay_allegro.log", "bench_c5.log"], f"{tag}_final_bench_replay_and_c5.log")
    cp("bench_rot_modes.log", f"{tag}_final_bench_rot_modes.log")
    cp("c5_resegment.log", f"{tag}_c5_resegment_bench.log")
    cp("icp_frame_phases.log", f"{tag}_icp_frame_phases.log")
    cp("handoff_stress.log", f"{tag}_handoff_stress.log")
    cp("km_quick.log", f"{tag}_kmeans_lloyd_iteration.log")
    cp("headline_repeats.log", f"{tag}_headline_repeats.log")
    cp("divergence_envelope_gpu.log", f"{tag}_divergence_envelope_gpu.log")
    cp("nn_rows_waves.log", f"{tag}_nn_rows_waves.log")
    cp("teacher_forced.log", f"{tag}_teacher_forced.log")
    cp("bench_wx200_5_real.log", f"{tag}_final_bench_wx200_5_real.log")
    print("\n".join(lines))
    if os.path.exists(f"{src}/stats_kernel_stats.csv"):
        for r in list(csv.DictReader(open(f"{src}/stats_kernel_stats.csv")))[:8]:
            print(r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"])


if __name__ == "__main__":
    main(*sys.argv[1:4])
