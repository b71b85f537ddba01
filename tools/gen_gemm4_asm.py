"""Generates sam6d_amd/csrc/s6d_gemm4_asm.inc: the K loop of one output tile of the four-wave GEMM (csrc/s6d_gemm4.hip) as ONE
inline-asm block with a fixed register map.  hipcc cannot allocate 256 accumulator registers next to two fragment sets (it scatters
fragments into the accumulator file and spills inside the loop), and a lone wave per SIMD needs its LDS reads, DMA issues and scalar
address arithmetic placed BETWEEN its own matrix instructions, so the stream is laid out here, instruction by instruction.

    python tools/gen_gemm4_asm.py            # rewrites the .inc (committed; the build does not run this script)

Register map (all clobbered by the block):
    a[0:255]      accumulators, tile (mt, n4) at a[16 (4 mt + n4) : +15]  (transposed product: W fragment = A operand)
    v[128:143]    W fragments of set 0 (n4 = 0..3), v[144:159] activation fragments of set 0 (mt = 0..3)
    v[160:175]    W fragments of set 1,               v[176:191] activation fragments of set 1
    v[192:193]    LDS addresses of the fragment reads being issued
    s84 j (K tile of the output tile)   s85 / s86 LDS offsets of the A / B half-tiles being read
    s87 s88 s90 s91 scratch   s92 / s93 16 rows of A / W in bytes
    two scalar sets of a stream element (64-bit DMA source bases of pieces 0, 1 / 2, 3 and the LDS destination), used alternately
    so that element k + 1 is prepared while element k's pieces go out:  s[76:77] s[78:79] s80  and  s[94:95] s[96:97] s89

Operands (see csrc/s6d_gemm4.hip for the values):
    %[s0b] (in/out) ring byte offset of B0 of the current K tile;  %[cA] %[cB] byte offsets of this wave's A / B half-tile inside a
    K tile's four slots;  %[w4k] LDS byte offset of this wave's 4 KiB of a slot (+ the array's base);  %[alo] %[ahi] %[wlo] %[whi]
    operand base pointers;  %[acur] %[anxt] %[wcur] %[wnxt] byte offsets of this / the next output tile's first row;  %[lda128]
    %[ldw128] 128 rows in bytes;  %[nk] K tiles per output tile;  %[xf0..3] %[wf0..3] fragment read offsets per k step (array base
    included);  %[va0] %[va1] %[vb0] %[vb1] DMA source offsets of piece 0 / 1 (pieces 2 / 3 = + 16 rows).
"""
import argparse
import os

RING = 10 * 16384
KT = 4 * 16384

# Measurement variants (tools/gemm4_variants.sh builds one library per entry; the product uses "base"):
#   dma / reads / barrier / mfma = False drop that class of instructions (timing only: the results are wrong);
#   read_gaps / dma_gaps: after which matrix instruction of a k step (0..15) each read / DMA piece is placed
OPT = dict(buf=False, dma=True, reads=True, barrier=True, mfma=True, vmwait=True, lgkmwait=True, pieces=4, read_gaps=(0, 1, 2, 3, 4, 5, 6, 7), dma_gaps=(8, 10, 12, 14),
           salu_gaps=16)
VARIANTS = {
    "base": {},
    "nodma": dict(dma=False),
    "noreads": dict(reads=False),
    "nobarrier": dict(barrier=False),
    "nomfma": dict(mfma=False),
    "mfmaonly": dict(dma=False, reads=False, barrier=False),
    "buf": dict(buf=True),
    "novmwait": dict(vmwait=False),
    "nolgkm": dict(lgkmwait=False),
    "dma2": dict(pieces=2),
    "nodma_nobar": dict(dma=False, barrier=False),
    "spread": dict(read_gaps=(0, 2, 4, 6, 8, 10, 12, 14), dma_gaps=(3, 7, 11, 15)),
    "dmafirst": dict(read_gaps=(6, 7, 8, 9, 10, 11, 12, 13), dma_gaps=(0, 1, 2, 3)),
    "dmaearly": dict(read_gaps=(0, 1, 2, 3, 4, 5, 6, 7), dma_gaps=(1, 3, 5, 7)),
    "readslate": dict(read_gaps=(4, 5, 6, 7, 8, 9, 10, 11), dma_gaps=(0, 1, 2, 3)),
}


def mfma(op, mt, n4, cur):
    acc = 16 * (4 * mt + n4)
    w = 128 + 32 * cur + 4 * n4
    x = 144 + 32 * cur + 4 * mt
    return f"v_mfma_f32_32x32x16_{op} a[{acc}:{acc + 15}], v[{w}:{w + 3}], v[{x}:{x + 3}], a[{acc}:{acc + 15}]"


def reads(nxt):
    """the 8 fragment reads of a k step into set nxt, in the order the first matrix instructions of the next step need them"""
    w = lambda n4: f"ds_read_b128 v[{128 + 32 * nxt + 4 * n4}:{128 + 32 * nxt + 4 * n4 + 3}], v193 offset:{(n4 >> 1) * 8192 + (n4 & 1) * 2048}"
    x = lambda mt: f"ds_read_b128 v[{144 + 32 * nxt + 4 * mt}:{144 + 32 * nxt + 4 * mt + 3}], v192 offset:{mt * 4096}"
    return [w(0), x(0), w(1), w(2), w(3), x(1), x(2), x(3)]


def ring(reg, tmp="s90"):
    """reg >= RING ? reg - RING : reg.  A tuple = instructions that pass SCC to each other: they stay together when the scalar
    work is dealt out over the gaps (the m0 writes of the DMA pieces and other scalar adds in between would overwrite SCC)"""
    return [(f"s_cmp_ge_u32 {reg}, {RING}", f"s_cselect_b32 {tmp}, {RING}, 0"), f"s_sub_u32 {reg}, {reg}, {tmp}"]


def flat(groups):
    out = []
    for g in groups:
        out += list(g) if isinstance(g, tuple) else [g]
    return out


SETS = ((76, 78, 80), (94, 96, 89))          # scalar register sets of a stream element: source bases s[p0:p0+1], s[p1:p1+1], LDS dst


def element_salu(rs, kind, ahead, half, slot):
    """scalar work of one stream element into register set rs: source bases (pieces 0, 1 / pieces 2, 3) and the LDS destination.
    kind 'a' / 'w'; ahead = K tiles ahead of j (s84); half = 0 / 1 (rows 0-127 / 128-255); slot = ring slot relative to s0b"""
    p0, p1, dst = rs
    cur, nxt, lo, hi, ld128, ld16 = (("%[acur]", "%[anxt]", "%[alo]", "%[ahi]", "%[lda128]", "s92") if kind == "a" else
                                     ("%[wcur]", "%[wnxt]", "%[wlo]", "%[whi]", "%[ldw128]", "s93"))
    out = [f"s_add_u32 s88, s84, {ahead}",
           ("s_cmp_ge_u32 s88, %[nk]", f"s_cselect_b32 s87, {nxt}, {cur}", "s_cselect_b32 s91, %[nk], 0"),
           "s_sub_u32 s88, s88, s91",
           "s_lshl_b32 s88, s88, 7",
           "s_add_u32 s87, s87, s88"]
    if half:
        out.append(f"s_add_u32 s87, s87, {ld128}")
    if OPT["buf"]:                                          # buffer_load ... lds: 32-bit scalar offsets beside a descriptor
        out += [f"s_mov_b32 s{p0}, s87", f"s_add_u32 s{p1}, s87, {ld16}"]
    else:
        out += [(f"s_add_u32 s{p0}, {lo}, s87", f"s_addc_u32 s{p0 + 1}, {hi}, 0"),
                (f"s_add_u32 s{p1}, s{p0}, {ld16}", f"s_addc_u32 s{p1 + 1}, s{p0 + 1}, 0")]
    if slot:
        out += [f"s_add_u32 s{dst}, %[s0b], {slot * 16384}"] + ring(f"s{dst}", "s91") + [f"s_add_u32 s{dst}, s{dst}, %[w4k]"]
    else:
        out += [f"s_add_u32 s{dst}, %[s0b], %[w4k]"]
    return out


# the four stream elements of an iteration, in issue order (step 0 .. 3): (kind, K tiles ahead of j, half, ring slot from s0b)
ELEMS = (("a", 1, 1, 7), ("w", 2, 0, 8), ("w", 2, 1, 9), ("a", 2, 0, 0))


def step(op, k, extra_salu=()):
    """k step k of an iteration: 16 matrix instructions from fragment set k & 1; in their gaps the 8 reads of the next k step into the
    other set, the four DMA pieces of stream element k (register set k & 1) and the scalar work of element k + 1 (other set; for
    k = 3: element 0 of the NEXT iteration -- one K tile further, four slots further)"""
    cur, nxt, ks_next = k & 1, (k & 1) ^ 1, (k + 1) & 3
    kind = ELEMS[k][0]
    vo = ("%[va0]", "%[va1]") if kind == "a" else ("%[vb0]", "%[vb1]")
    p0, p1, dst = SETS[k & 1]
    out = [f"v_add_u32 v192, s85, %[xf{ks_next}]", f"v_add_u32 v193, s86, %[wf{ks_next}]"]
    if OPT["lgkmwait"]:
        out.append("s_waitcnt lgkmcnt(0)")
    rd = reads(nxt)
    if k < 3:
        salu = element_salu(SETS[(k + 1) & 1], *ELEMS[k + 1])
    else:
        e = ELEMS[0]
        salu = element_salu(SETS[0], e[0], e[1] + 1, e[2], e[3] + 4)
    salu += list(extra_salu)
    gaps = [[] for _ in range(16)]                          # gaps[i] = after matrix instruction i
    if OPT["reads"]:
        for q, g in enumerate(OPT["read_gaps"]):
            gaps[g].append(rd[q])
    n = OPT["salu_gaps"]
    per = (len(salu) + n - 1) // n                          # groups per gap
    for g in range(n):
        gaps[g] += flat(salu[g * per:(g + 1) * per])
    pre = []
    for pc, g in enumerate(OPT["dma_gaps"]):
        (gaps[g - 1] if g > 0 else pre).append(f"s_add_u32 m0, s{dst}, {pc * 1024}")
        if OPT["dma"] and pc < OPT["pieces"]:
            if OPT["buf"]:
                gaps[g].append(f"buffer_load_dwordx4 {vo[pc & 1]}, {'%[ra]' if kind == 'a' else '%[rw]'}, s{(p0, p1)[pc >> 1]} offen lds")
            else:
                gaps[g].append(f"global_load_lds_dwordx4 {vo[pc & 1]}, s[{(p0, p1)[pc >> 1]}:{(p0, p1)[pc >> 1] + 1}]")
    out += pre
    i = 0
    for mt in range(4):
        for n4 in range(4):
            if OPT["mfma"]:
                out.append(mfma(op, mt, n4, cur))
            out += gaps[i]
            i += 1
    return out


def block(op):
    o = ["s_lshr_b32 s92, %[lda128], 3", "s_lshr_b32 s93, %[ldw128], 3", "s_mov_b32 s84, 0",
         "s_add_u32 s85, %[s0b], %[cA]"] + flat(ring("s85")) + ["s_add_u32 s86, %[s0b], %[cB]",
         "v_add_u32 v192, s85, %[xf0]", "v_add_u32 v193, s86, %[wf0]"] + (reads(0) if OPT["reads"] else [])
    o += flat(element_salu(SETS[0], *ELEMS[0]))                # the first iteration's element 0 (later ones: step 3 of the iteration before)
    o.append("L_g4_loop_%=:")
    o += step(op, 0)                                        # A1 of K tile j + 1
    o += step(op, 1)                                        # B0 of K tile j + 2
    # step 2 also prepares the read offsets of K tile g + 1 (s85 / s86 are consumed by the v_adds at the head of the step)
    nxt_rd = ([f"s_add_u32 s85, %[s0b], {KT}", "s_add_u32 s85, s85, %[cA]"] + ring("s85") +
              [f"s_add_u32 s86, %[s0b], {KT}", "s_add_u32 s86, s86, %[cB]"] + ring("s86"))
    o += step(op, 2, nxt_rd)                                # B1 of K tile j + 2
    # every fragment of K tile g is in registers; this wave's pieces of K tile g + 1 have landed (8 younger pieces stay in flight)
    o += ([f"s_waitcnt vmcnt({2 * OPT['pieces']})"] if OPT["vmwait"] else []) + ["s_waitcnt lgkmcnt(0)"]
    o += ["s_barrier"] if OPT["barrier"] else []
    tail = flat([f"s_add_u32 %[s0b], %[s0b], {KT}"] + ring("%[s0b]") + ["s_add_u32 s84, s84, 1"])
    o += step(op, 3) + tail                                 # A0 of K tile j + 2 into the slot B0(g) just left
    o += ["s_cmp_lt_u32 s84, %[nk]", "s_cbranch_scc1 L_g4_loop_%=",
          "s_nop 15", "s_nop 15"]                           # the last matrix results are in a[] before anything reads them
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", default="base", choices=sorted(VARIANTS))
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    OPT.update(VARIANTS[args.variant])
    here = os.path.dirname(os.path.abspath(__file__))
    dst = args.out or os.path.join(here, "..", "sam6d_amd", "csrc", "s6d_gemm4_asm.inc")
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm4_asm.py -- do not edit; the K loop of one output tile of csrc/s6d_gemm4.hip.\n")
        for name, op in (("S6D_G4_ASM_BF16", "bf16"), ("S6D_G4_ASM_F16", "f16")):
            lines = block(op)
            f.write(f"#define {name} \\\n")
            f.write(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
            f.write("\n\n")
        cl = [f'"a{i}"' for i in range(256)] + [f'"v{i}"' for i in range(128, 194)] + [f'"s{i}"' for i in list(range(76, 81)) + list(range(84, 98))]
        f.write("#define S6D_G4_ASM_CLOBBERS \\\n  " + ", ".join(cl) + ', "scc", "memory"\n')
    n = len(block("bf16"))
    print(f"wrote {dst}: {n} instructions per block (loop body {n - 20})")


if __name__ == "__main__":
    main()
