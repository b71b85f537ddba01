"""Static instruction mix of the gfx950 code of one kernel, per innermost loop: how many MFMA / VALU / transcendental / LDS /
VMEM / SALU / wait instructions a loop iteration issues (no GPU needed).  With the issue-rate figures of
tools/probes/valu_rate.hip this gives a lower bound for a wave's time per iteration and shows which class dominates.

    python tools/isa_mix.py s6d_attn.hip 'attn_global_kernel<80, 4, 1>'
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sam6d_amd import _lib  # noqa: E402

# measured on MI355X, saturated, per SIMD: ns per wave-instruction
COST_NS = {"mfma": 7.36, "trans": 3.47, "valu": 1.34}


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    src, want = sys.argv[1], sys.argv[2]
    flags = [f for f in _lib.FLAGS if f not in ("-shared", "-fPIC")]
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.check_call([_lib.HIPCC] + flags + ["-S", "--cuda-device-only", "-o", asm, os.path.join(_lib._CSRC, src)],
                              stderr=subprocess.DEVNULL)
        text = open(asm).read()
    # function bodies: "<mangled>: ; @<mangled>" ... "s_endpgm"
    for m in re.finditer(r"^(_Z\w+):\s*;.*?$(.*?)^\s*s_endpgm", text, flags=re.M | re.S):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"^void ", "", re.sub(r"\(.*", "", name)).replace("s6d::", "")
        if name != want:
            continue
        lines = m.group(2).split("\n")
        # loops: a label that a later branch jumps back to; innermost = no other loop header inside
        labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
        loops = []
        for i, l in enumerate(lines):
            b = re.match(r"\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)|\s*s_branch\s+(\.LBB\d+_\d+)", l)
            if b:
                tgt = b.group(1) or b.group(2)
                if tgt in labels and labels[tgt] < i:
                    loops.append((labels[tgt], i))
        inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
        print(f"{name}: {len(lines)} lines, {len(loops)} loops, {len(inner)} innermost")
        for a, b in sorted(inner, key=lambda lp: lp[0] - lp[1])[:4]:
            c = collections.Counter()
            ops = collections.Counter()
            for l in lines[a:b + 1]:
                l = l.strip()
                if not l or l.startswith((".", ";")) or l.endswith(":"):
                    continue
                op = l.split()[0]
                c[classify(op)] += 1
                ops[op] += 1
            lower = sum(COST_NS.get(k, 0.0) * v for k, v in c.items())
            print(f"  loop lines {a}-{b} ({b - a + 1} lines): " + ", ".join(f"{k} {v}" for k, v in c.most_common()))
            print(f"    issue-time sum (no overlap) {lower:.0f} ns; MFMA alone {COST_NS['mfma'] * c['mfma']:.0f} ns; top ops: "
                  + ", ".join(f"{k} x{v}" for k, v in ops.most_common(10)))
        return
    raise SystemExit(f"kernel {want!r} not found in {src}")


if __name__ == "__main__":
    main()
