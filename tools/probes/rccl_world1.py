"""The collective calls of bench.py and utils/shard.py under the nccl (= RCCL) backend on the ONE GPU of a gpurun box, world size 1:
init_process_group with device_id, barrier, all_gather_into_tensor of the pose-record block, all_reduce(MAX) of a float64 scalar,
shard.gather_records (all_gather of counts and padded blocks), destroy_process_group -- what a single-GPU box can execute of the
multi-GPU path (no peer transport is involved at world 1; N > 1 runs are the driver's).  Prints one JSON line."""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sam6d_amd.utils import shard  # noqa: E402


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "init_s": round(time.perf_counter() - t0, 2)}
    dist.barrier()
    rec = shard.pack_records(0, torch.arange(32, device=dev), 5, torch.rand(32, device=dev), torch.eye(3, device=dev).expand(32, 3, 3).contiguous(),
                             torch.rand(32, 3, device=dev), 0.0)
    g = torch.empty_like(rec)
    dist.all_gather_into_tensor(g, rec.contiguous())
    out["all_gather_into_tensor_equal"] = bool(torch.equal(g, rec))
    t = torch.tensor([1.25], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out["all_reduce_max_f64"] = t.item()
    out["gather_records_equal"] = bool(torch.equal(shard.gather_records(rec[:7]), rec[:7]))
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    out["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    line = json.dumps(out)
    print(line, flush=True)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(line + "\n")


if __name__ == "__main__":
    main()
