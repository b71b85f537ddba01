"""world_size-2 gloo test of the frame sharding + pose-record gather (CPU, two processes)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sam6d_amd.utils import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_indices(n_frames, rank, world)
    g = torch.Generator().manual_seed(1234)
    R_all = torch.randn(n_frames, 3, 3, generator=g)
    t_all = torch.randn(n_frames, 3, generator=g)
    idx = torch.tensor(mine, dtype=torch.long)
    rec = shard.pack_records(7, idx.float(), 5, torch.full((len(mine),), 0.5), R_all[idx], t_all[idx], 0.01)
    full = shard.gather_records(rec)
    q.put((rank, full))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_reassembles_every_frame_once():
    world, n_frames = 2, 7            # uneven shards: 4 + 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(got[0], got[1])                       # every rank holds the same table
    full = got[0]
    assert full.shape == (n_frames, shard.RECORD_WIDTH)
    assert sorted(full[:, 1].tolist()) == [float(i) for i in range(n_frames)]   # each frame exactly once
    g = torch.Generator().manual_seed(1234)
    R_all = torch.randn(n_frames, 3, 3, generator=g)
    order = full[:, 1].long()
    assert torch.equal(full[:, 4:13], R_all[order].reshape(-1, 9))
    lines = shard.to_bop_csv_lines(full)
    assert len(lines) == n_frames and lines[0].startswith("7,0,5,0.5,")


def test_shard_indices_partition():
    for n, w in ((10, 1), (10, 3), (5, 8)):
        parts = [shard.shard_indices(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))


def _sharded_worker(rank, world, port, out, n_frames):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tools import run_sharded
    run_sharded.main(["--frames", str(n_frames), "--group", "3", "--stand-in", "--fixed-time", "0.5", "--out", out])


def test_sharded_frame_loop_csv_equals_the_single_rank_run(tmp_path, monkeypatch):
    """tools/run_sharded.py (BASELINE configs[2]) with stand-in models: 23 frames with different instance counts, sharded round-
    robin over two gloo ranks, run in groups of 3, gathered with ONE variable-length all_gather -> the csv rank 0 writes is BYTE
    FOR BYTE the csv of the single-process run over the same frame list (rows in split order, float32 forms of test_bop.py)."""
    from tools import run_sharded
    n = 23
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    single = str(tmp_path / "single.csv")
    res = run_sharded.main(["--frames", str(n), "--group", "3", "--stand-in", "--fixed-time", "0.5", "--out", single])
    assert res["world"] == 1 and res["records"].shape[0] > n          # several instances per frame
    ctx = mp.get_context("spawn")
    port = _free_port()
    out = str(tmp_path / "sharded.csv")
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, out, n)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    a, b = open(single, "rb").read(), open(out, "rb").read()
    assert a == b and a.count(b"\n") == res["records"].shape[0]
    # split order: (scene_id, im_id) non-decreasing down the file
    keys = [tuple(int(v) for v in ln.split(",")[:2]) for ln in a.decode().splitlines()]
    assert keys == sorted(keys)


def test_prefetch_thread_changes_nothing_and_balance_report():
    """run_sharded loads the next group's frames on a background thread while the current group computes (round 6): the records are
    those of the serial order; assignment_efficiency reproduces hand-computed balances of the static round-robin assignment and of
    the cost-sorted bound."""
    import torch
    from sam6d_amd.utils import shard
    from tools import run_sharded as rs
    pipe = rs.StandInPipeline()

    def load(s, i):
        P, K = rs.frame_shape(s, i)
        g = torch.Generator().manual_seed(s * 7919 + i)
        return (torch.randint(0, 256, (8, 8, 3), generator=g, dtype=torch.uint8), torch.rand(8, 8, generator=g),
                torch.eye(3, dtype=torch.float64), torch.rand(K, 64, generator=g), torch.rand(K, 18000, generator=g))
    ids = rs.frame_list(13)
    a = shard.run_sharded(ids, load, pipe, group_size=4, fixed_time=0.0, prefetch=True)
    b = shard.run_sharded(ids, load, pipe, group_size=4, fixed_time=0.0, prefetch=False)
    assert torch.equal(a["records"], b["records"]) and a["csv_lines"] == b["csv_lines"]
    assert len(a["group_seconds"]) == 4 and a["load_wait_seconds"] >= 0.0
    costs = [4, 1, 1, 1, 1, 1, 1, 1, 5]                     # 2 ranks, round robin: 4+1+1+1+5 = 12 | 1+1+1+1 = 4 -> mean 8 / max 12
    assert abs(shard.assignment_efficiency(costs, 2, "round_robin") - 8 / 12) < 1e-9
    assert abs(shard.assignment_efficiency(costs, 2, "lpt") - 1.0) < 1e-9      # 5+1+1+1 | 4+1+1+1+1 = 8 | 8
