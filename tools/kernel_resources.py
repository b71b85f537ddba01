"""Static resource usage (VGPR / AGPR / SGPR / scratch / occupancy / static LDS / spills) of every kernel in csrc/*.hip,
from hipcc's -Rpass-analysis=kernel-resource-usage remarks with the flags the library is built with.  No GPU needed.

    python tools/kernel_resources.py > profiles/rNN_kernel_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sam6d_amd import _lib  # noqa: E402

KEYS = (("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"TotalSGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("LDS", r"LDS Size \[bytes/block\]"), ("sSpill", r"SGPRs Spill"),
        ("vSpill", r"VGPRs Spill"))


def main():
    flags = [f for f in _lib.FLAGS if f not in ("-shared", "-fPIC")]
    print("# hipcc " + " ".join(flags) + " -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.py)")
    print("# LDS is the STATIC allocation: kernels that size their LDS at launch (attention, patch_scores, tok2img ...) show")
    print("# 0 or a small number here; their launch-time sizes are in DESIGN.md section 4.  occ = waves/SIMD allowed by registers.")
    print(f"{'file':11s} {'kernel':60s} " + " ".join(f"{k:>7s}" for k, _ in KEYS))
    with tempfile.TemporaryDirectory() as tmp:
        for src in _lib.sources():
            r = subprocess.run([_lib.HIPCC] + flags + ["-c", src, "-o", os.path.join(tmp, "o.o"),
                                                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stderr)
            for blk in re.split(r"(?=remark: Function Name:)", r.stderr):
                m = re.search(r"Function Name: (\S+)", blk)
                if not m:
                    continue
                name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name)).replace("s6d::", "")
                vals = []
                for _, pat in KEYS:
                    mm = re.search(pat + r": (\d+)", blk)
                    vals.append(int(mm.group(1)) if mm else -1)
                print(f"{os.path.basename(src)[4:-4]:11s} {name[:60]:60s} " + " ".join(f"{v:7d}" for v in vals))


if __name__ == "__main__":
    main()
