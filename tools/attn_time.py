"""Timing of the fused attention kernels at the SAM shapes (run on the GPU box).
usage: attn_time.py [g|w|gw] dbg..."""
import sys

sys.path.insert(0, ".")
from tools.attn_ablate import run  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "gw"
    dbgs = [int(a) for a in sys.argv[2:]] or [0]
    for dbg in dbgs:
        if "g" in which:
            run(8, 64, 16, 80, 0, dbg, n=10)
        if "w" in which:
            run(8, 64, 16, 80, 14, dbg, n=20)
