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


def _run(world, extra, self_launch=False):
    env = dict(os.environ, CREG_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    common = ["--gpus", str(world), "--sequences", "3"] + extra
    if world == 1 or self_launch:
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
        # every rank sizes its own batches: 7 items on one rank = one round of 7; 4 + 3 on two ranks = one round each, nothing padded
        assert one["config"]["batch_sizes_per_rank"] == [[7]] and two["config"]["batch_sizes_per_rank"] == [[4], [3]]
        assert two["config"]["rounds_per_rank"] == [1, 1] and two["config"]["padded_steps_per_rank"] == [0, 0]
        assert two["config"]["padded_steps_timed_not_counted"] == 0
    else:
        assert two["pose_checksum"] != one["pose_checksum"]          # twice the frames: other sequences on rank 1


def test_bench_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK in the environment (the way the driver starts `--gpus 1`) must not die on an
    assertion: it re-executes itself under torch.distributed.run and prints the same single JSON line."""
    via_torchrun = _run(2, ["--steps", "6", "--warmup", "2"])
    direct = _run(2, ["--steps", "6", "--warmup", "2"], self_launch=True)
    assert direct["n_gpus"] == 2 and direct["steps"] == 6 and direct["scaling"] == "weak"
    assert abs(direct["pose_checksum"] - via_torchrun["pose_checksum"]) < 1e-6 * max(1.0, abs(direct["pose_checksum"]))


def test_bench_gpus_mismatch_is_a_clear_error():
    env = dict(os.environ, CREG_BENCH_STUB="1", OMP_NUM_THREADS="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in p.stderr


def test_replay_batches_fewest_rounds_no_padding():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.replay_batches(0) == [] and bench.replay_batches(7) == [7] and bench.replay_batches(6) == [6]
    assert bench.replay_batches(9) == [5, 4] and bench.replay_batches(50) == [8, 7, 7, 7, 7, 7, 7]
    # BASELINE configs[3]: 50 frames round-robin over 8 ranks = ONE round per rank (round 3: two padded rounds of 5)
    counts = [len(range(r, 50, 8)) for r in range(8)]
    assert [bench.replay_batches(c) for c in counts] == [[7], [7], [6], [6], [6], [6], [6], [6]]
    for n in range(1, 200):
        b = bench.replay_batches(n)
        assert sum(b) == n and max(b) <= 8 and max(b) - min(b) <= 1 and len(b) == -(-n // 8)


def test_replay_many_rounds_ragged_two_ranks():
    """19 items over 2 ranks = 10 + 9: rank 0 runs [5, 5], rank 1 [5, 4] -- two plans on rank 1, ragged gather."""
    one = _run(1, ["--steps", "19", "--warmup", "2", "--mode", "replay"])
    two = _run(2, ["--steps", "19", "--warmup", "2", "--mode", "replay"])
    assert one["config"]["batch_sizes_per_rank"] == [[7, 6, 6]] and two["config"]["batch_sizes_per_rank"] == [[5, 5], [5, 4]]
    assert abs(one["pose_checksum"] - two["pose_checksum"]) < 1e-3 * max(1.0, abs(one["pose_checksum"]))


def test_bench_gpus_2_line_carries_the_strong_scaling_legs():
    """`bench.py --gpus N` (N > 1) answers the north star's strong-scaling questions by itself -- after the weak-scaling
    headline every rank runs BASELINE configs[3] (--mode replay --workload allegro --steps 50) and configs[4] (--workload c5 --steps 200);
    the one JSON line carries both under `strong_scaling`, each with value / scaling "strong" / what every rank ran / the gather."""
    env = dict(os.environ, CREG_BENCH_STUB="1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "5"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 10            # the headline stays the weak-scaling line
    legs = d["strong_scaling"]
    assert set(legs) == {"configs[3]", "configs[4]"}
    c3, c4 = legs["configs[3]"], legs["configs[4]"]
    for leg, steps in ((c3, 50), (c4, 200)):
        assert "error" not in leg, leg
        assert leg["scaling"] == "strong" and leg["steps"] == steps and leg["n_gpus"] == 2 and leg["value"] > 0 and leg["unit"] == "frames/s"
        assert leg["gather"]["world"] == 2 and leg["gather"]["backend"] == "gloo" and leg["gather"]["s"] >= 0
        assert "pose_checksum" in leg and leg["leg_wall_s"] >= 0
    assert c3["config"]["batch_sizes_per_rank"] == [[7, 6, 6, 6], [7, 6, 6, 6]]          # 25 items per rank: four rounds of <= 8, no padding
    assert c4["config"]["frames_per_rank"] == [100, 100]
    assert "allegro" in c3["config"]["workload"] and "replay" in c3["config"]["mode"]
    # a single rank's line has no such legs (the divisor is the same command at --gpus 1 with the leg's own arguments)
    one = _run(1, ["--steps", "6", "--warmup", "3"])
    assert "strong_scaling" not in one
