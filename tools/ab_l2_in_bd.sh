# A/B inside one gpurun call: the next hidden activation inside the backward launch (CREG_L2_IN_BD=1) against the five-launch epoch
for rep in 1 2 3; do for v in 0 1; do
  echo -n "headline CREG_L2_IN_BD=$v rep $rep: "; CREG_L2_IN_BD=$v timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-icp-variant --no-parity --no-other-workloads --repeats 1 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); k=d["roofline"]["kernels"]; print(d["value"], "frames/s; b2b us bd", k["bd"]["avg_launch_us"], "l2", k["l2"]["avg_launch_us"], "; checksum", d["pose_checksum"])'
done; done
for v in 0 1; do
  for seqs in 1 8; do echo -n "sequences $seqs CREG_L2_IN_BD=$v: "; CREG_L2_IN_BD=$v timeout 200 python bench.py --sequences $seqs --steps $((8*seqs)) --warmup $seqs --no-cpu-baseline --no-icp-variant --no-parity --no-roofline --no-other-workloads --repeats 1 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(d["value"], "frames/s; checksum", d["pose_checksum"])'; done
  for wl in franka allegro; do echo -n "$wl CREG_L2_IN_BD=$v: "; CREG_L2_IN_BD=$v timeout 200 python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-icp-variant --no-parity --no-roofline --repeats 1 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][0]); print(d["value"], "frames/s; checksum", d["pose_checksum"])'; done
done
