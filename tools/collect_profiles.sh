#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repository root: bench lines + rocprofv3 evidence into gpurun_out/final/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
# then here:  python tools/summarize_profiles.py gpurun_out/final r01
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.log 2>$O/bench.err
python $R/bench.py --sequences 1 --steps 8 --warmup 2 --no-cpu-baseline --no-icp-variant > $O/bench_b1.log 2>/dev/null
python $R/bench.py --sequences 8 --steps 40 --warmup 8 --no-cpu-baseline --no-icp-variant > $O/bench_b8.log 2>/dev/null
python $R/bench.py --workload franka --steps 10 --warmup 5 --no-cpu-baseline --no-icp-variant > $O/bench_franka.log 2>/dev/null
python $R/bench.py --workload allegro --steps 20 --warmup 5 --no-cpu-baseline --no-icp-variant > $O/bench_allegro.log 2>/dev/null
CMD="python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-icp-variant"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o r01 -- $CMD > $O/prof_run.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$c -- python $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-icp-variant > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O -o pmc_sq -- python $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-icp-variant > $O/pmc_sq.log 2>&1
rm -f $O/*_kernel_trace.csv.bak
ls -la $O | head -40
tail -c 600 $O/bench.log
