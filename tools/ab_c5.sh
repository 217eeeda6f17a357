# configs[4] frame, variants alternating inside one gpurun call:  bash tools/ab_c5.sh "" old
for rep in 1 2; do for v in "$@"; do
  echo -n "c5 variant=[$v] rep $rep: "; CREG_LIB_VARIANT=$v timeout 300 python bench.py --workload c5 --steps 12 --warmup 2 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); r=d["roofline"]; print(d["ms_per_step"], "ms/frame; k_icp_nn", r["avg_launch_us"], "us x", r["launches_per_frame"], "; f32 pairs/src-it", r["pairs_per_source_and_iteration"]["float32_screen"], "fp64", r["pairs_per_source_and_iteration"]["fp64"], "fp64 trip frac", r["fp64_trip_fraction"], "; checksum", d["pose_checksum"])'
done; done
