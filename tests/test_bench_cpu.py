"""CPU, gloo: bench.py's rank logic -- sharding, warm-up / timed loops, padding, the final gather and the JSON line --
executed at world_size 1 and 2 with the stand-in registrar (CREG_BENCH_STUB=1), in both modes, before any GPU sees
`--gpus 2`.  Replay mode deals the SAME items to however many ranks there are, so the gathered poses must not depend
on the world size."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra):
    env = dict(os.environ, CREG_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--gpus", str(world), "--sequences", "3"] + extra
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + common
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # exactly one JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("mode,steps", [("sequences", 6), ("replay", 7)])
def test_bench_rank_logic_two_ranks_gloo(mode, steps):
    one = _run(1, ["--steps", str(steps), "--warmup", "3", "--mode", mode])
    two = _run(2, ["--steps", str(steps), "--warmup", "3", "--mode", mode])
    for d, w in ((one, 1), (two, 2)):
        assert d["n_gpus"] == w and d["steps"] == steps and d["warmup"] == 3 and d["value"] > 0
        assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
        assert d["scaling"] == ("strong" if mode == "replay" else "weak")
        assert ("replay" in d["config"]["mode"]) == (mode == "replay")
    if mode == "replay":
        # the same --steps items whatever the world size (7 items over 2 ranks: 4 + 3, ragged gather, padded last batch)
        assert abs(one["pose_checksum"] - two["pose_checksum"]) < 1e-3 * max(1.0, abs(one["pose_checksum"]))
        assert two["config"]["padded_steps_timed_not_counted"] >= 0
    else:
        assert two["pose_checksum"] != one["pose_checksum"]          # twice the frames: other sequences on rank 1
