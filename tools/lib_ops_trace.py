"""Which lines of this package still launch library (ATen / rocBLAS / runtime copy) kernels inside the benched step?

Runs each stage of bench.py's HotPath once under torch.profiler (with Python stacks) and prints, per stage, the kernels whose
names are not `s6d::…`, summed by the innermost frame of this repository that launched them.  Output goes to stdout and,
with --out, to a text file kept under profiles/.

    python tools/lib_ops_trace.py --out gpurun_out/prof/r04_library_ops_by_line.txt
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def site_of(stack):
    for fr in stack:
        if "/sam6d_amd/" in fr or "bench.py" in fr:
            if "/utils/linear.py" in fr or "_lib.py" in fr:
                continue
            return fr.replace(ROOT + "/", "").strip()
    return stack[0].strip() if stack else "?"


def trace(name, fn, lines):
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        fn()
        torch.cuda.synchronize()
    by = collections.defaultdict(lambda: [0, 0.0])
    tot = lib = 0.0
    for ev in prof.events():
        ks = getattr(ev, "kernels", None)
        if not ks:
            continue
        for k in ks:
            d = float(k.duration)
            tot += d
            if "s6d" in k.name:
                continue
            lib += d
            short = k.name.split("(")[0].split("<")[0].replace("void ", "").replace("at::native::", "")
            shapes = str(getattr(ev, "input_shapes", ""))[:90]
            key = (site_of(ev.stack or []) if ev.stack else shapes, ev.name, short[:48])
            by[key][0] += 1
            by[key][1] += d
    lines.append(f"== {name}: library kernels {lib / 1e3:.3f} ms of {tot / 1e3:.3f} ms GPU time ({100 * lib / max(tot, 1e-9):.1f} %)")
    for (site, op, kern), (n, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:110]:
        lines.append(f"  {d / 1e3:7.3f} ms {n:4d}x  {kern[:34]:34s} {op[:24]:24s} {site}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--stages", default="pem,ism,sam")
    ap.add_argument("--pipeline", action="store_true", help="trace one whole frame of tools/frame_demo.py's FramePipeline instead")
    a = ap.parse_args()
    if a.pipeline:
        os.environ.setdefault("S6D_PEM_VIT_DTYPE", "fp16"); __import__("sam6d_amd.policy").policy.reload()
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import frame_demo
        pipe, args = frame_demo.build(torch.device("cuda", 0))
        lines = []
        for _ in range(2):
            pipe(*args)
        trace("whole frame (FramePipeline, one frame, graphs as the pipeline uses them)", lambda: pipe(*args), lines)
        os.environ["S6D_AMG_GRAPH"] = "0"; __import__("sam6d_amd.policy").policy.reload()
        os.environ["S6D_PEM_GRAPH"] = "0"; __import__("sam6d_amd.policy").policy.reload()
        pipe.invalidate_graphs()
        pipe(*args)
        trace("whole frame, hipGraph replay off (every launch visible to the profiler)", lambda: pipe(*args), lines)
        txt = "\n".join(lines)
        print(txt)
        if a.out:
            os.makedirs(os.path.dirname(a.out), exist_ok=True)
            open(a.out, "w").write(txt + "\n")
        return
    import bench
    dev = torch.device("cuda:0")
    bench.benched_policy()
    hp = bench.HotPath(dev, a.frames, 16)
    lines = []
    with torch.no_grad():
        if "pem" in a.stages:
            trace("PEM (batch of %d instances)" % a.frames, hp.pem_stage, lines)
        if "ism" in a.stages:
            trace("ISM scoring (%d frames)" % a.frames, hp.ism_stage, lines)
        if "sam" in a.stages:
            trace("SAM encoder (%d frames)" % a.frames, hp.sam_stage, lines)
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
