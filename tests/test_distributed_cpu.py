"""CPU, world_size 2, gloo: the N>1 path (sequence sharding + the single final pose gather)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_seq, frames, K, q):
    import torch.distributed as dist
    from autourdf_amd.distributed import gather_poses, scatter_order, shard_sequences
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_sequences(n_seq, rank, world)
    # a pose that encodes (sequence, frame, cluster) so misplaced rows are detectable
    local = torch.stack([torch.full((K, 4, 4), float(1000 * s + f)) + torch.arange(K).view(K, 1, 1)
                         for s in mine for f in range(frames)]) if mine else torch.zeros(0, K, 4, 4)
    counts = [len(shard_sequences(n_seq, r, world)) * frames for r in range(world)]
    allp = gather_poses(local, counts=counts)
    order = scatter_order(n_seq, world, frames)
    ok = allp.shape[0] == n_seq * frames
    for row, (s, f) in zip(allp, order):
        ok &= bool((row[:, 0, 0] == torch.arange(K) + 1000 * s + f).all())
    if counts[0] == counts[-1]:                               # equal shards: the un-padded fast path
        ok &= torch.equal(gather_poses(local), allp)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_seq", [4, 5])                      # 5 sequences over 2 ranks: ragged shards
def test_two_rank_gather_gloo(n_seq):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_seq, 3, 4, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


def test_shard_sequences_partition():
    from autourdf_amd.distributed import shard_sequences
    for n, w in ((5, 8), (50, 8), (7, 2), (1, 1)):
        parts = [shard_sequences(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        shard_sequences(4, 2, 2)


def test_gather_is_identity_without_process_group():
    from autourdf_amd.distributed import gather_poses
    x = torch.randn(3, 2, 4, 4)
    assert gather_poses(x) is x


def _main_worker(rank, world, port, root, q):
    """autourdf_amd.mlp_reg.main()'s torchrun prologue (_shard_for_rank) at world size 2 on gloo: the registration itself
    is stubbed (no GPU here) -- what runs is the drop-in's own code for: who writes the frame-0 state, the barrier before
    anybody reads it, which sequences a rank takes."""
    os.chdir(root)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      CREG_DIST_BACKEND="gloo")
    from autourdf_amd import mlp_reg
    calls = []

    def fake_frame0(first_dir):
        assert rank == 0
        os.makedirs("data/part/frame0_written", exist_ok=True)
        open("data/part/frame0_written/by_rank_%d" % rank, "w").write(first_dir)

    mlp_reg._ensure_frame0 = fake_frame0
    dirs = ["data/raw/toy/4_deg_20_cams/V%04d/" % i for i in range(5)]
    mine = mlp_reg._shard_for_rank(dirs)
    saw_frame0 = os.path.exists("data/part/frame0_written/by_rank_0")      # after the barrier every rank must see it
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, saw_frame0))


def test_mlp_reg_main_sharding_prologue_two_ranks_gloo(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_main_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = ["data/raw/toy/4_deg_20_cams/V%04d/" % i for i in range(5)]
    assert res[0][1] == [d[0], d[2], d[4]] and res[1][1] == [d[1], d[3]]
    assert res[0][2] and res[1][2]
    assert os.listdir(tmp_path / "data/part/frame0_written") == ["by_rank_0"]
