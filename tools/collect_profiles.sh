#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root: bench lines + rocprofv3 evidence into gpurun_out/final/.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r06'
# then here:  cp gpurun_out/r06_profiles/* profiles/
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-icp-variant --no-roofline --no-other-workloads --repeats 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- $CMD > $O/prof_run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c5 -- python $R/bench.py --workload c5 --steps 4 --warmup 1 > $O/prof_c5.log 2>&1
# counters: one set per pass (FETCH_SIZE | WRITE_SIZE | the SQ set), per workload, never together with a trace domain other than --kernel-trace
for wl in wx200_5 franka allegro; do
  steps=5; [ $wl = franka ] && steps=5
  W="python $R/bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --no-icp-variant --no-roofline --no-other-workloads --repeats 1"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_${wl}_$c -- $W > $O/pmc_${wl}_$c.log 2>&1
  done
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O -o pmc_${wl}_sq -- $W > $O/pmc_${wl}_sq.log 2>&1
  # round 4: the K-row GEMMs of k_bd / k_l2 run on the matrix cores -- their instruction count, MOPS and busy cycles (own pass)
  rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --output-format csv -d $O -o pmc_${wl}_mfma -- $W > $O/pmc_${wl}_mfma.log 2>&1
done
# the ICP-style configs[4] frame: the SQ set for its kernels (k_icp_nn, k_km_persist, ...)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O -o pmc_c5_sq -- python $R/bench.py --workload c5 --steps 3 --warmup 1 > $O/pmc_c5_sq.log 2>&1
# round 6: the L2 of the same frame (hit rate of the search's staged targets) and its memory-side bytes, own passes
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O -o pmc_c5_tcc -- python $R/bench.py --workload c5 --steps 3 --warmup 1 > $O/pmc_c5_tcc.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o pmc_c5_fetch -- python $R/bench.py --workload c5 --steps 3 --warmup 1 > $O/pmc_c5_fetch.log 2>&1
python - <<PYEOF > $R/gpurun_out/${TAG}_c5_pmc_summary.txt
import csv, glob, collections
f = glob.glob("$O/**/pmc_c5_sq*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])) if f else []:
    agg[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
print("# bench.py --workload c5 --steps 3 --warmup 1 under rocprofv3 --pmc (SQ set), summed over all launches of a kernel; ratios per wave-cycle")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    wc = d.get("SQ_WAVE_CYCLES", 1) or 1
    print(f"{k:48s} waves {d.get('SQ_WAVES', 0):.3g}  VALU insts {d.get('SQ_INSTS_VALU', 0):.3g}  LDS insts {d.get('SQ_INSTS_LDS', 0):.3g}  "
          f"active VALU / wave-cycle {d.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  active LDS {d.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}  "
          f"wait any {d.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait inst {d.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}")
for tagf, what in (("tcc", "L2 (TCC) requests of the same command, summed over all launches"), ("fetch", "FETCH_SIZE (KB as reported: x2 for wide coalesced reads, MI355X_MICROARCH.md), summed over all launches")):
    f = glob.glob("$O/**/pmc_c5_%s*counter_collection.csv" % tagf, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])) if f else []:
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"][:48]] += 1
    print("#", what)
    for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:5]:
        if tagf == "tcc":
            h, m = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
            print(f"{k:48s} TCC_HIT {h:.4g}  TCC_MISS {m:.4g}  hit rate {h / max(h + m, 1):.4f}  TCC_REQ {d.get('TCC_REQ_sum', 0):.4g}")
        else:
            print(f"{k:48s} FETCH_SIZE {d.get('FETCH_SIZE', 0):.4g} KB over {n[k]} launches")
PYEOF
rm -f $O/*_kernel_trace.csv $O/*_agent_info.csv $O/*_domain_stats.csv
# counters first, bench lines after: `roofline.traffic` / the measured VALU utilisation of a bench line come from profiles/${TAG}_pmc.json,
# which must describe the kernels that line runs (a line benched before its counters were re-collected divides old cycles by new times)
cd $R && python tools/summarize_profiles.py $O ${TAG} $R/gpurun_out/${TAG}_profiles > /dev/null 2>&1 && cp $R/gpurun_out/${TAG}_profiles/${TAG}_pmc.json $R/profiles/${TAG}_pmc.json
cd /tmp
python $R/bench.py > $O/bench.log 2>$O/bench.err
python $R/bench.py --sequences 1 --steps 8 --warmup 2 --no-cpu-baseline --no-icp-variant > $O/bench_b1.log 2>/dev/null
python $R/bench.py --sequences 8 --steps 40 --warmup 8 --no-cpu-baseline --no-icp-variant > $O/bench_b8.log 2>/dev/null
python $R/bench.py --workload franka --steps 10 --warmup 5 --no-cpu-baseline --no-icp-variant > $O/bench_franka.log 2>/dev/null
python $R/bench.py --workload allegro --steps 20 --warmup 5 --no-cpu-baseline --no-icp-variant > $O/bench_allegro.log 2>/dev/null
# the reference's other --r choices on the fused plan (rpy: RegMLP(6, 3), hidden 3, as mlp_reg.py:285 builds it -- it does not converge)
(for r in dq 6d rpy; do timeout 300 python $R/bench.py --r $r --steps 20 --warmup 5 --no-icp-variant 2>/dev/null < /dev/null | grep '^{'; done) > $O/bench_rot_modes.log
# BASELINE configs[3] / [4] in replay (independent-frame) mode, one GPU: the items an 8-GPU job would deal out
python $R/bench.py --mode replay --workload allegro --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_replay_allegro.log 2>/dev/null
python $R/bench.py --workload c5 --steps 12 --warmup 2 > $O/bench_c5.log 2>/dev/null
# six back-to-back headline runs (one process each): the default (chain streams, two chains at five sequences), the two-branch graph of
# rounds 2-4 and three chain streams -- run-to-run spread of each
(for r in 1 2 3 4 5 6; do for g in 0 2 -3; do echo -n "--graph-branches $g (0 = default: 2 chain streams; 2 = one graph, two branches; -3 = 3 chain streams), run $r: "; timeout 300 python $R/bench.py --steps 20 --warmup 5 --graph-branches $g --no-cpu-baseline --no-icp-variant --no-roofline 2>/dev/null < /dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(d["value"], "frames/s", d["ms_per_step"], "ms per frame")'; done; done) > $O/headline_repeats.log
(for so in 0 10 20 30; do echo -n "seed offset $so: "; python $R/bench.py --steps 20 --warmup 5 --seed-offset $so --no-cpu-baseline --no-icp-variant --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(d["value"], "frames/s")'; done) >> $O/headline_repeats.log
python $R/tests/measure/divergence_envelope.py gpu $O/divergence.json 2>/dev/null | grep -v amdgpu.ids > $O/divergence_envelope_gpu.log
python $R/tests/measure/bench_c5_resegment.py > $O/c5_resegment.log 2>/dev/null
python $R/tests/measure/profile_icp_frame.py both 2>/dev/null | grep -v amdgpu.ids > $O/icp_frame_phases.log
(cd $R && CREG_EXTRA_FLAGS=-DCREG_NN_WAVE_STAMPS python -m autourdf_amd.build --variant wstamp > $O/wstamp_build.log 2>&1)      # (the measurement build is made here: .gpurunignore keeps variant libraries out of the snapshot)
(CREG_LIB_VARIANT=wstamp python $R/tests/measure/nn_rows_waves.py franka; CREG_LIB_VARIANT=wstamp python $R/tests/measure/nn_rows_waves.py wx200_5) 2>/dev/null | grep -v "amdgpu.ids\|Warn\|ret = ret" > $O/nn_rows_waves.log
python $R/tests/measure/teacher_forced.py 2>/dev/null | grep -v "amdgpu.ids\|Warn\|current =\|Consider" > $O/teacher_forced.log
python $R/bench.py --workload wx200_5_real --no-cpu-baseline > $O/bench_wx200_5_real.log 2>/dev/null
python $R/tests/measure/stress_handoffs.py 2>/dev/null | grep -v amdgpu.ids > $O/handoff_stress.log
(for pe in 1 0; do for pr in 1 0; do [ $pe = 1 ] && [ $pr = 0 ] && continue; echo "# CREG_KM_PRUNE=$pr CREG_KM_PERSIST=$pe"; CREG_KM_PRUNE=$pr CREG_KM_PERSIST=$pe python $R/tests/measure/km_quick.py 2>/dev/null | grep Lloyd; done; done) > $O/km_quick.log
# summaries on the box (gpurun brings back at most 64 MiB; the raw counter CSVs are ~18 MB each), raw files dropped
cd $R && python tools/summarize_profiles.py $O ${TAG} $R/gpurun_out/${TAG}_profiles > $R/gpurun_out/${TAG}_profiles_summary.log 2>&1
rm -f $O/*_counter_collection.csv
cp $R/gpurun_out/${TAG}_c5_pmc_summary.txt $R/gpurun_out/${TAG}_profiles/ 2>/dev/null
ls -la $R/gpurun_out/${TAG}_profiles | head -40
tail -c 600 $O/bench.log
