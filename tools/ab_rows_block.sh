mkdir -p gpurun_out
for rep in 1 2; do for v in "" rows512; do
  echo -n "franka variant=[$v] rep $rep: "; CREG_LIB_VARIANT=$v timeout 200 python bench.py --workload franka --steps 20 --warmup 5 --no-cpu-baseline --no-icp-variant --no-parity --repeats 1 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); k=d["roofline"]["kernels"]["nn_l1"]; print(d["value"], "frames/s; nn", k["kernel"], k["avg_launch_us"], "us b2b; checksum", d["pose_checksum"])'
done; done
