"""BASELINE configs[2] as a program, on the REAL (mini, seeded) models: the four mini frames through
  (a) frame-by-frame FramePipeline calls,
  (b) utils/shard.run_sharded at world size 1 in groups of 2 (frames 0,1 | 2,3),
  (c) run_sharded at world size 2 (two processes sharing this GPU; the record gather runs over gloo on host tensors: rank 0 owns
      frames 0,2, rank 1 frames 1,3)
must write the SAME BOP csv, byte for byte (benched dtypes): every frame's poses are independent of which frames share its SAM / PEM batch and of
which rank computed them.  (tests/test_dist_gloo.py checks the gather itself with a stand-in pipeline on the CPU.)"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sharded_csv_is_byte_identical_across_groupings_and_ranks(tmp_path, monkeypatch):
    # the benched extractor (this library's GEMMs): with the fp32 extractor the ViT-B runs rocBLAS kernels chosen by row count and a
    # frame's pose follows its group in the last bits (tests/test_gpu_zz_pipeline.py)
    monkeypatch.setenv("S6D_PEM_VIT_DTYPE", "fp16")
    from sam6d_amd.utils import shard
    from tests.sharded_mini_worker import frame_table
    from tests.test_gpu_zz_pipeline import build_mini, mini_frames
    pipe, frame = build_mini(torch.device("cuda", 0), top_k="keys", sync_stages=False)
    frames = mini_frames(frame)
    ids, load = frame_table(frames)
    # (a) frame by frame
    blocks = []
    for (s, i), f in zip(ids, frames):
        det, poses = pipe(*f)
        det.scene_id, det.image_id = s, i
        blocks.append(shard.frame_records(det, poses, "ycbv", 0.0))
    csv_a = shard.to_bop_csv_lines(torch.cat(blocks))
    assert len(csv_a) >= 4, "the mini frames should give at least one pose each"
    # (b) one rank, groups of two
    csv_b = shard.run_sharded(ids, load, pipe, group_size=2, dataset_name="ycbv", device=None, fixed_time=0.0)["csv_lines"]
    assert csv_b == csv_a
    del pipe
    torch.cuda.empty_cache()
    # (c) two ranks
    out = str(tmp_path / "w2.csv")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.sharded_mini_worker", out, "2"], cwd=ROOT, env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert open(out).readlines() == csv_a
