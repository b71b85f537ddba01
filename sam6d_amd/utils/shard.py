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
