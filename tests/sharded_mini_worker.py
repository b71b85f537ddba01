"""One rank of the sharded frame loop on the REAL (mini, seeded) models: builds tests/test_gpu_zz_pipeline.build_mini on cuda:0, runs
sam6d_amd.utils.shard.run_sharded over the four mini frames with the process group the environment describes (gloo: the record
gather runs on host tensors, so two ranks can share the one GPU of the test box) and rank 0 writes the BOP csv.
Launched by tests/test_gpu_zz_sharded.py:  python -m tests.sharded_mini_worker <out.csv> <group_size>"""
import os
import sys

import torch


def frame_table(frames):
    ids = [(7, 10 + i) for i in range(len(frames))]
    table = {k: f for k, f in zip(ids, frames)}
    return ids, (lambda s, i: table[(s, i)])


def main():
    out, group = sys.argv[1], int(sys.argv[2])
    import torch.distributed as dist

    from sam6d_amd.utils import shard
    from tests.test_gpu_zz_pipeline import build_mini, mini_frames
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe, frame = build_mini(torch.device("cuda", 0), top_k="keys", sync_stages=False)
    ids, load = frame_table(mini_frames(frame))
    res = shard.run_sharded(ids, load, pipe, group_size=group, dataset_name="ycbv", device=None, fixed_time=0.0)
    if rank == 0:
        with open(out, "w+") as f:
            f.writelines(res["csv_lines"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
