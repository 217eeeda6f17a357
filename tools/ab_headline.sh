for rep in 1 2 3; do for v in "" old; do
  echo -n "headline variant=[$v] rep $rep: "; CREG_LIB_VARIANT=$v timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-icp-variant --no-parity --no-roofline --no-other-workloads --repeats 1 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(d["value"], "frames/s; checksum", d["pose_checksum"])'
done; done
