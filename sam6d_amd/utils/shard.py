"""Frame sharding and the final gather of pose records.

Frames (and instances within a frame) are independent in both ISM and PEM, so the path
shards with NO data-path collective: rank r of W owns frames i = r (mod W) of the sorted
(scene_id, im_id) list; template data is replicated.  The only collective is one
all_gather of fixed-width records at the end (RCCL over xGMI on MI355X: backend "nccl";
"gloo" in the CPU tests).  It replaces the reference's file-glob merge of per-frame .npz files
(Instance_Segmentation_Model/model/detector.py:425-462, which avoids all_gather on purpose)
and the CSV append of Pose_Estimation_Model/test_bop.py:166-185.

Record layout (17 x float32 = 68 B):
  [scene_id, im_id, obj_id, score, R(9 row-major), t(3), time]
Integers up to 2^24 are exact in float32 (BOP scene / image / object ids are far below).
"""
import torch

RECORD_WIDTH = 17


def shard_indices(n_items, rank, world):
    """Indices of the items rank `rank` owns (round-robin, deterministic)."""
    return list(range(rank, n_items, world))


def pack_records(scene_id, im_id, obj_id, score, R, t, time_s):
    n = R.shape[0]
    rec = torch.empty(n, RECORD_WIDTH, dtype=torch.float32, device=R.device)
    rec[:, 0] = torch.as_tensor(scene_id, dtype=torch.float32, device=R.device)
    rec[:, 1] = torch.as_tensor(im_id, dtype=torch.float32, device=R.device)
    rec[:, 2] = torch.as_tensor(obj_id, dtype=torch.float32, device=R.device)
    rec[:, 3] = score
    rec[:, 4:13] = R.reshape(n, 9)
    rec[:, 13:16] = t
    rec[:, 16] = torch.as_tensor(time_s, dtype=torch.float32, device=R.device)
    return rec


def gather_records(rec, group=None):
    """all_gather of per-rank record blocks with different row counts -> (sum_n, 17) on every
    rank, ordered by rank.  Two collectives: the counts (1 int each), then the padded blocks."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world = dist.get_world_size(group)
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts) if counts else 0
    pad = torch.zeros(m, RECORD_WIDTH, dtype=torch.float32, device=rec.device)
    pad[: rec.shape[0]] = rec
    blocks = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad, group=group)
    return torch.cat([b[:c] for b, c in zip(blocks, counts)], dim=0)


def to_bop_csv_lines(records):
    """BOP result lines `scene,im,obj,score,R,t(mm),time` from the (n,17) float32 record table, printed the way
    test_bop.py:155-181 prints them: float32 values in their shortest round-tripping decimal form (`str(numpy.float32)`),
    t scaled to millimetres in float32.  (sam6d_amd/pem/results.py is the per-frame writer; this one serves the gathered
    table.)"""
    import numpy as np

    r = records.detach().cpu().numpy().astype(np.float32)
    t = r[:, 13:16] * 1000
    out = []
    for k in range(r.shape[0]):
        out.append(",".join((str(int(r[k, 0])), str(int(r[k, 1])), str(int(r[k, 2])), str(r[k, 3]),
                             " ".join(str(v) for v in r[k, 4:13]), " ".join(str(v) for v in t[k]), f"{float(r[k, 16])}\n")))
    return out


def frame_records(det, poses, dataset_name, time_s=0.0):
    """The (n, 17) record block of one frame from what FramePipeline.run_group returned for it: one row per instance the PEM
    kept, BOP category ids (sam6d_amd.ism.handoff: LM-O id table / obj + 1), pose score x detection score in float32
    (test_bop.py:157).  Empty when the frame has no pose."""
    import numpy as np

    from ..ism.handoff import LMO_OBJECT_IDS
    from ..pem import results
    if poses is None or poses["pred_R"].shape[0] == 0:
        return torch.zeros(0, RECORD_WIDTH, dtype=torch.float32)
    kept = poses["kept"]
    obj = det.object_ids[kept].cpu().numpy()
    cat = LMO_OBJECT_IDS[obj] if dataset_name == "lmo" else obj + 1
    s = torch.from_numpy(np.ascontiguousarray(results.combined_scores(poses["pred_pose_score"], det.scores[kept])))
    return pack_records(det.scene_id, det.image_id, torch.from_numpy(np.asarray(cat, np.float32)), s,
                        poses["pred_R"].detach().float().cpu(), poses["pred_t"].detach().float().cpu(), time_s)


def assignment_efficiency(costs, world, policy="round_robin"):
    """Shard-balance efficiency mean(busy) / max(busy) of `world` ranks for per-frame costs (seconds, in split order) under the static
    round-robin assignment run_sharded uses (frame i -> rank i % world) or under "lpt" (longest processing time first: frames sorted
    by cost, each to the least-loaded rank -- the bound a cost-aware assignment could reach; costs are only known after the ISM stage,
    so this is reported, not used).  Lets a world-1 run say what a world-8 run of the same split would lose to imbalance."""
    costs = [float(c) for c in costs]
    busy = [0.0] * world
    if policy == "round_robin":
        for i, c in enumerate(costs):
            busy[i % world] += c
    elif policy == "lpt":
        for c in sorted(costs, reverse=True):
            busy[busy.index(min(busy))] += c
    else:
        raise ValueError(policy)
    return (sum(busy) / world) / max(busy) if max(busy) > 0 else 1.0


def run_sharded(frame_ids, load_frame, pipeline, group_size=8, dataset_name="ycbv", device=None, fixed_time=None, prefetch=True):
    """BASELINE configs[2]: the frame loop of a test split sharded over the ranks of one node (the reference: one Lightning
    test step per frame + a per-frame .npz + a file-glob merge, ISM model/detector.py:425-462; PEM test_bop.py:123-185).

    frame_ids: the sorted list of (scene_id, im_id) of the split, identical on every rank.  Rank r of W takes frames r, r + W,
    ... (shard_indices), loads them with ``load_frame(scene_id, im_id)`` -> the tuple FramePipeline.run_group takes, runs them
    in groups of ``group_size`` (one SAM pass and one PEM pass per group), packs one 68-byte record per estimated pose and takes
    part in ONE variable-length all_gather at the end (gather_records; RCCL when the process group is "nccl").  No other
    collective.  -> dict(records = the whole split's table sorted by (scene_id, im_id) with each frame's rows in the
    pipeline's order, identical on every rank; csv_lines; stats = per-rank [busy seconds, frames, instances] + the
    shard-balance efficiency mean(busy) / max(busy)).  ``fixed_time``: value of the time column instead of the measured
    per-frame seconds (the byte-for-byte tests)."""
    import time

    import torch.distributed as dist
    dist_on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist_on else (0, 1)
    sync = (lambda: torch.cuda.synchronize()) if (device is not None and torch.device(device).type == "cuda") else (lambda: None)
    mine = shard_indices(len(frame_ids), rank, world)
    blocks, busy, n_inst, group_s, load_wait = [], 0.0, 0, [], 0.0
    groups = [[frame_ids[i] for i in mine[g0:g0 + group_size]] for g0 in range(0, len(mine), group_size)]
    # The next group's frames are loaded (disk / decode / upload: the caller's load_frame) by ONE background thread while this
    # group computes (round 6; before, loading sat on the compute thread between two groups).  One group ahead, no queue: the
    # loader holds at most one group of host / device buffers.  prefetch=False: the old order (the byte-for-byte tests use both).
    pool = None
    if prefetch and len(groups) > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="s6d-load")
    load = lambda ids: [load_frame(s, i) for (s, i) in ids]            # noqa: E731
    pending = pool.submit(load, groups[0]) if pool is not None and groups else None
    for gi, ids in enumerate(groups):
        tl = time.perf_counter()
        frames = pending.result() if pending is not None else load(ids)
        load_wait += time.perf_counter() - tl
        pending = pool.submit(load, groups[gi + 1]) if pool is not None and gi + 1 < len(groups) else None
        sync()
        t0 = time.perf_counter()
        res = pipeline.run_group(frames)
        sync()
        dt = time.perf_counter() - t0
        busy += dt
        group_s.append(dt)
        for (s, i), (det, poses) in zip(ids, res):
            det.scene_id, det.image_id = s, i
            rec = frame_records(det, poses, dataset_name, fixed_time if fixed_time is not None else dt / len(ids))
            n_inst += rec.shape[0]
            blocks.append(rec)
    rec = torch.cat(blocks) if blocks else torch.zeros(0, RECORD_WIDTH)
    dev = torch.device(device) if device is not None else rec.device
    full = gather_records(rec.to(dev)).cpu()
    # rank-major -> split order; a frame's rows come from one rank and stay in its order (stable sort on the frame key)
    key = full[:, 0].double() * (1 << 24) + full[:, 1].double()
    full = full[torch.sort(key, stable=True).indices]
    st = torch.tensor([[busy, float(len(mine)), float(n_inst)]], dtype=torch.float32)
    stats = gather_records(torch.nn.functional.pad(st, (0, RECORD_WIDTH - 3)).to(dev)).cpu()[:, :3] if dist_on else st
    eff = float(stats[:, 0].mean() / stats[:, 0].max()) if stats[:, 0].max() > 0 else 1.0
    if pool is not None:
        pool.shutdown(wait=True)
    return dict(records=full, csv_lines=to_bop_csv_lines(full), stats=stats, balance_efficiency=eff, rank=rank, world=world,
                group_seconds=group_s, load_wait_seconds=load_wait)
