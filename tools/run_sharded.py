"""BASELINE configs[2] as a runnable program: a test split's frame list sharded per frame over the GPUs of one node, every rank
running sam6d_amd.pipeline.FramePipeline.run_group on its frames, ONE variable-length all_gather of the 68-byte pose records
at the end (RCCL over xGMI under backend "nccl"), rank 0 writing the BOP csv.  Reference loop: Instance_Segmentation_Model/
run_inference.py:46-51,74-77 + model/detector.py:425-462 (per-frame .npz files merged by a glob) and Pose_Estimation_Model/
test_bop.py:123-185.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/run_sharded.py --frames 256 --group 8 --out results/ycbv.csv

Synthetic split (there is no dataset in the image): --frames frames of 480 x 640 RGB-D whose proposal count P (32..128) and
instance count K (2..16) vary from frame to frame, seeded by (scene_id, im_id) -- the imbalance a real split has.  Weights are
seeded (no checkpoint offline).  Prints one JSON line: frames/s of the whole job, per-rank busy seconds / frames / instances
and the shard-balance efficiency mean(busy) / max(busy).  `--stand-in` replaces the five models by a deterministic function of
the frame (host tensors, backend "gloo"): what tests/test_dist_gloo.py runs at world size 2 to check that the gathered csv is
byte for byte the single-rank one.  NOT a scaling measurement: no multi-GPU node has run this yet."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam6d_amd.utils import shard  # noqa: E402


def frame_list(n, per_scene=50):
    """Sorted (scene_id, im_id) pairs of a synthetic split."""
    return [(48 + i // per_scene, 1 + i % per_scene) for i in range(n)]


def frame_shape(scene_id, im_id):
    """(P, K) of a frame: proposals entering the descriptor stage, instances handed to the PEM."""
    g = torch.Generator().manual_seed(scene_id * 100003 + im_id)
    return int(torch.randint(32, 129, (1,), generator=g)), int(torch.randint(2, 17, (1,), generator=g))


class StandInPipeline:
    """run_group with the interface of FramePipeline and no model behind it: detections and poses are a deterministic function
    of the frame tensors (so any rank computes the same rows for the same frame)."""

    def run_group(self, frames):
        from sam6d_amd.ism.handoff import Detections
        out = []
        for (img, depth, K, keys, ru) in frames:
            k = keys.shape[0]
            g = torch.Generator().manual_seed(int(img.long().sum()) % (2 ** 31))
            q = torch.linalg.qr(torch.randn(k, 3, 3, generator=g))[0]
            det = Detections(0, 0, torch.zeros(k, 4, 4, dtype=torch.bool), torch.zeros(k, 4), torch.rand(k, generator=g),
                             torch.randint(0, 21, (k,), generator=g))
            drop = torch.rand(k, generator=g) < 0.2                       # the pre-processing drops some detections
            kept = (~drop).nonzero().flatten()
            poses = None if kept.numel() == 0 else dict(pred_R=q[kept], pred_t=torch.randn(k, 3, generator=g)[kept] * 0.3,
                                                        pred_pose_score=torch.rand(k, generator=g)[kept], kept=kept)
            out.append((det, poses))
        return out


def real_pipeline(dev):
    """The five released-size models on seeded weights + a loader of synthetic frames whose proposal count P and instance count K
    vary from frame to frame (frame_shape) -> (pipe, load_frame)."""
    import collections

    from sam6d_amd.utils import synth
    from tools import frame_demo
    counts = collections.deque()          # one P per frame, consumed by the proposal stage in frame order
    pipe, (img0, depth0, K0, _, _) = frame_demo.build(dev, top_k="keys", sync_stages=False, proposal_counts=counts)

    def load(s, i):
        P, K = frame_shape(s, i)
        g = torch.Generator().manual_seed(s * 7919 + i)
        img = (img0.int() + torch.randint(-8, 9, img0.shape, generator=g).to(dev)).clamp(0, 255).to(torch.uint8)
        counts.append(P)
        return (img, depth0, K0, torch.rand(K, 480 * 640, generator=g).to(dev), synth.coarse_uniforms(K, s * 1000 + i).to(dev))
    return pipe, load


def measure_world1(dev, frames=24, group=8):
    """BASELINE configs[2]'s program on ONE rank with the warm-up separated (the round-4 number, 86 ms busy per
    frame over 16 frames, included the first group's allocator growth, library autotuning and graph captures): one untimed group of
    other frames, then `frames` frames in groups of `group`.  -> dict for bench.py's `sharded_world1` block."""
    pipe, load = real_pipeline(dev)
    warm = [(9, 1 + i) for i in range(group)]
    pipe.run_group([load(s, i) for (s, i) in warm])
    torch.cuda.synchronize()
    ids = frame_list(frames)
    t0 = time.perf_counter()
    res = shard.run_sharded(ids, load, pipe, group_size=group, dataset_name="ycbv", device=dev, fixed_time=0.0)
    wall = time.perf_counter() - t0
    busy = float(res["stats"][0, 0])
    shapes = [frame_shape(s, i) for (s, i) in ids]
    # what a world-8 run of this split would lose to imbalance, from per-frame busy times measured here (frames alone, groups of 1):
    # the static round-robin assignment run_sharded uses against the bound of a cost-sorted one
    one = shard.run_sharded(ids, load, pipe, group_size=1, dataset_name="ycbv", device=dev, fixed_time=0.0)
    costs = one["group_seconds"]
    balance = {"frames": len(costs), "frame_ms_min": round(min(costs) * 1e3, 1), "frame_ms_max": round(max(costs) * 1e3, 1),
               "world8_round_robin": round(shard.assignment_efficiency(costs, 8, "round_robin"), 3),
               "world8_cost_sorted_bound": round(shard.assignment_efficiency(costs, 8, "lpt"), 3)}
    return {"load_wait_ms_per_frame_behind_the_prefetch_thread": round(res["load_wait_seconds"] / frames * 1e3, 2),
            "world8_balance_from_world1_frame_times": balance,"workload": f"{frames} synthetic 480x640 frames, proposals per frame {min(p for p, _ in shapes)}..{max(p for p, _ in shapes)} "
                        f"(mean {sum(p for p, _ in shapes) / frames:.0f}), instances per frame {min(k for _, k in shapes)}..{max(k for _, k in shapes)} "
                        f"(mean {sum(k for _, k in shapes) / frames:.1f}), groups of {group}, one warm-up group untimed",
            "frames_per_s": round(frames / busy, 2), "busy_ms_per_frame": round(busy / frames * 1e3, 2),
            "frames_per_s_incl_frame_synthesis": round(frames / wall, 2), "poses": int(res["records"].shape[0])}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--group", type=int, default=8)
    ap.add_argument("--out", default="")
    ap.add_argument("--dataset", default="ycbv")
    ap.add_argument("--stand-in", action="store_true")
    ap.add_argument("--fixed-time", type=float, default=None)
    a = ap.parse_args(argv)
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.stand_in:
        dev = torch.device("cpu")
        backend = "gloo"
    else:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        backend = "nccl"
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)
    ids = frame_list(a.frames)
    if a.stand_in:
        pipe = StandInPipeline()

        def load(s, i):
            P, K = frame_shape(s, i)
            g = torch.Generator().manual_seed(s * 7919 + i)
            return (torch.randint(0, 256, (8, 8, 3), generator=g, dtype=torch.uint8), torch.rand(8, 8, generator=g),
                    torch.eye(3, dtype=torch.float64), torch.rand(K, 64, generator=g), torch.rand(K, 18000, generator=g))
    else:
        pipe, load = real_pipeline(dev)
        pipe.run_group([load(9, 1 + i) for i in range(a.group)])          # warm-up group: allocator, library autotuning, graph captures
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = shard.run_sharded(ids, load, pipe, group_size=a.group, dataset_name=a.dataset, device=dev, fixed_time=a.fixed_time)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if rank == 0:
        if a.out:
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            with open(a.out, "w+") as f:
                f.writelines(res["csv_lines"])
        busy = [float(x) for x in res["stats"][:, 0]]
        print(json.dumps(dict(tool="run_sharded", world=world, frames=a.frames, group=a.group, poses=int(res["records"].shape[0]),
                              frames_per_s=a.frames / wall, wall_s=wall, frames_per_s_busy=a.frames / max(busy) if max(busy) > 0 else None,
                              balance_efficiency=res["balance_efficiency"],
                              per_rank_busy_s=[round(x, 4) for x in busy],
                              per_rank_frames=[int(x) for x in res["stats"][:, 1]],
                              per_rank_instances=[int(x) for x in res["stats"][:, 2]], stand_in=bool(a.stand_in),
                              scaling="unmeasured on hardware" if a.stand_in or world == 1 else "measured")))
    if world > 1:
        dist.destroy_process_group()
    return res


if __name__ == "__main__":
    main()
