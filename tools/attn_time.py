"""Time the SAM encoder's attention launches per build variant of csrc/s6d_attn.hip (tools/attn_variants.sh), all in ONE process so
that the comparison is free of box-to-box clock variance: global attention over the 64 x 64 grid (B frames x 16 heads x 80) and
the 14 x 14 windowed attention, against the base build's output (layout / schedule variants must reproduce it bit for bit; the
ablation builds are timing probes and are not compared).
Usage: python tools/attn_time.py [B] [name ...]      (writes gpurun_out/attn_time.json)"""
import ctypes
import glob
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vp = ctypes.c_void_p
def is_probe(name):
    return any(k in name for k in ("noload", "nostore", "nostage", "nosoftmax", "nopv", "noqk", "nomath", "probe", "timing"))


def event_ms(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    names = [a for a in sys.argv[1:] if not a.isdigit()]
    libs = sorted(glob.glob(os.path.join(ROOT, "tools", "attn_variants", "libattn_*.so")))
    libs = [p for p in libs if not names or os.path.basename(p)[8:-3] in names or os.path.basename(p) == "libattn_base.so"]
    libs.sort(key=lambda p: os.path.basename(p) != "libattn_base.so")
    nh, hd, H = 16, 80, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(B, H, H, 3 * nh * hd, generator=g, device="cuda").to(torch.bfloat16)
    bias = (0.3 * torch.randn(3 * nh * hd, generator=g, device="cuda")).to(torch.bfloat16)
    out = {}
    ref = {}
    for ws in (0, 14):
        S = ws if ws else H
        rh = (0.3 * torch.randn(2 * S - 1, hd, generator=g, device="cuda")).to(torch.bfloat16)
        rw = (0.3 * torch.randn(2 * S - 1, hd, generator=g, device="cuda")).to(torch.bfloat16)
        flop = 4.0 * B * nh * hd * (H * H) ** 2 if ws == 0 else 0.0
        for path in libs:
            name = os.path.basename(path)[8:-3]
            L = ctypes.CDLL(path)
            L.s6d_win_attention_scratch_bytes.restype = ctypes.c_long
            scratch = torch.empty(int(L.s6d_win_attention_scratch_bytes(H, ws, hd)), dtype=torch.uint8, device="cuda")
            o_full = torch.zeros(B * H * H * nh * hd + 2048, dtype=torch.bfloat16, device="cuda")     # + 4 KiB for the timing probe's dump
            o = o_full[:B * H * H * nh * hd].view(B, H, H, nh * hd)

            def run(hm=None):
                fn = L.s6d_win_attention_layout_bf16
                rc = fn(vp((qkv if hm is None else hm).data_ptr()), 0 if hm is None else 1, vp(bias.data_ptr()), vp(rh.data_ptr()), vp(rw.data_ptr()),
                        B, H, H, nh, hd, ws, ctypes.c_float(hd ** -0.5), vp(scratch.data_ptr()), vp(o.data_ptr()),
                        vp(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            ms = min(event_ms(run, n=20) for _ in range(3))
            row = {"ms": round(ms, 4)}
            if flop:
                row["tflops"] = round(flop / ms / 1e9, 1)
            if name == "base":
                ref[ws] = o.clone()
            elif not is_probe(name):
                row["equals_base"] = bool(torch.equal(o, ref[ws]))
                if not row["equals_base"]:
                    row["max_abs_diff_vs_base"] = float((o.float() - ref[ws].float()).abs().max())
            if ws == 14 and "timing" in name:
                t = o_full[B * H * H * nh * hd:].view(torch.int64)[:48].view(8, 6).cpu().tolist()
                items = (B * 25 * nh + 255) // 256
                row["phase_cycles_per_item"] = {f"wave{w}": [round(v / items) for v in t[w]] for w in (0, 3, 4, 7)}
            if ws == 0 and "timing" in name:
                t = o_full[B * H * H * nh * hd:].view(torch.int64)[:40].view(8, 5).cpu().tolist()
                row["phase_cycles_per_tile"] = {f"wave{w}": [round(v / 64) for v in t[w]] for w in (0, 3, 4, 7)}
            if ws == 0 and not is_probe(name):             # the same launch on the head-major tensor (3 heads, B H W, hd)
                hm = qkv.view(B * H * H, 3 * nh, hd).permute(1, 0, 2).contiguous()
                o_tok = o.clone()
                row["head_major_ms"] = round(event_ms(lambda: run(hm)), 4)
                row["head_major_equal"] = bool(torch.equal(o, o_tok))
                del hm
            out[f"{'global' if ws == 0 else 'window14'}.{name}"] = row
            print(f"{'global' if ws == 0 else 'window14':9s} {name:18s} {json.dumps(row)}", flush=True)
    # the all-resident kernel behind s6d_seq_attention_bf16 at the DINOv2 shape: 64 crops x 16 heads x 257 tokens x 64
    Bs, Ns, nhs, hds = 64, 257, 16, 64
    qs = torch.randn(Bs, Ns, 3 * nhs * hds, generator=g, device="cuda").to(torch.bfloat16)
    os_ = torch.empty(Bs, Ns, nhs * hds, dtype=torch.bfloat16, device="cuda")
    flop = 4.0 * Bs * nhs * hds * Ns * Ns
    for path in libs:
        name = os.path.basename(path)[8:-3]
        if "timing" in name:                  # the phase-clock build writes through a null clock array in kernels that do not pass one
            continue
        L = ctypes.CDLL(path)

        def run_seq():
            rc = L.s6d_seq_attention_bf16(vp(qs.data_ptr()), Bs, Ns, nhs, hds, ctypes.c_float(hds ** -0.5), vp(os_.data_ptr()),
                                          vp(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
        ms = min(event_ms(run_seq, n=20) for _ in range(3))
        out[f"seq257.{name}"] = {"ms": round(ms, 4), "tflops": round(flop / ms / 1e9, 1)}
        print(f"seq257    {name:18s} {json.dumps(out[f'seq257.{name}'])}", flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "attn_time.json"), "w") as f:
        json.dump({"B": B, "heads": nh, "head_dim": hd, "grid": H, "results": out}, f, indent=1)


if __name__ == "__main__":
    main()
