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
    s87 .. s91 scratch   s92 / s93 16 rows of A / W in bytes   s[94:95], s[96:97] 64-bit DMA source bases

Operands (see csrc/s6d_gemm4.hip for the values):
    %[s0b] (in/out) ring byte offset of B0 of the current K tile;  %[cA] %[cB] byte offsets of this wave's A / B half-tile inside a
    K tile's four slots;  %[w4k] LDS byte offset of this wave's 4 KiB of a slot (+ the array's base);  %[alo] %[ahi] %[wlo] %[whi]
    operand base pointers;  %[acur] %[anxt] %[wcur] %[wnxt] byte offsets of this / the next output tile's first row;  %[lda128]
    %[ldw128] 128 rows in bytes;  %[nk] K tiles per output tile;  %[xf0..3] %[wf0..3] fragment read offsets per k step (array base
    included);  %[va0] %[va1] %[vb0] %[vb1] DMA source offsets of piece 0 / 1 (pieces 2 / 3 = + 16 rows).
"""
import os

RING = 10 * 16384
KT = 4 * 16384


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
    return [f"s_cmp_ge_u32 {reg}, {RING}", f"s_cselect_b32 {tmp}, {RING}, 0", f"s_sub_u32 {reg}, {reg}, {tmp}"]


def element_salu(kind, ahead, half, slot):
    """scalar work of one stream element: source bases s[94:95] (pieces 0, 1) and s[96:97] (pieces 2, 3), LDS destination s89.
    kind 'a' / 'w'; ahead = K tiles ahead of j (1 or 2); half = 0 / 1 (rows 0-127 / 128-255); slot = ring slot relative to s0b"""
    cur, nxt, lo, hi, ld128, ld16 = (("%[acur]", "%[anxt]", "%[alo]", "%[ahi]", "%[lda128]", "s92") if kind == "a" else
                                     ("%[wcur]", "%[wnxt]", "%[wlo]", "%[whi]", "%[ldw128]", "s93"))
    out = [f"s_add_u32 s88, s84, {ahead}",
           "s_cmp_ge_u32 s88, %[nk]",
           f"s_cselect_b32 s87, {nxt}, {cur}",
           "s_cselect_b32 s91, %[nk], 0",
           "s_sub_u32 s88, s88, s91",
           "s_lshl_b32 s88, s88, 7",
           "s_add_u32 s87, s87, s88"]
    if half:
        out.append(f"s_add_u32 s87, s87, {ld128}")
    out += [f"s_add_u32 s94, {lo}, s87", f"s_addc_u32 s95, {hi}, 0",
            f"s_add_u32 s96, s94, {ld16}", "s_addc_u32 s97, s95, 0"]
    if slot:
        out += [f"s_add_u32 s89, %[s0b], {slot * 16384}"] + ring("s89", "s91") + ["s_add_u32 s89, s89, %[w4k]"]
    else:
        out += ["s_add_u32 s89, %[s0b], %[w4k]"]
    return out


def step(op, cur, ks_next, kind, ahead, half, slot, extra_salu=()):
    """one k step: 16 matrix instructions from set cur; in their gaps the reads of k step ks_next into the other set (gaps 0-7), the
    scalar work of one stream element (gaps 0-7) and its four DMA pieces (gaps 8-15)"""
    nxt = cur ^ 1
    vo = ("%[va0]", "%[va1]") if kind == "a" else ("%[vb0]", "%[vb1]")
    out = [f"v_add_u32 v192, s85, %[xf{ks_next}]", f"v_add_u32 v193, s86, %[wf{ks_next}]", "s_waitcnt lgkmcnt(0)"]
    rd = reads(nxt)
    salu = element_salu(kind, ahead, half, slot) + list(extra_salu)
    per = (len(salu) + 7) // 8
    gaps = [[] for _ in range(16)]
    for g in range(8):
        gaps[g].append(rd[g])
        gaps[g] += salu[g * per:(g + 1) * per]
    assert len(salu) <= 8 * per
    for pc in range(4):
        gaps[7 + 2 * pc].append(f"s_add_u32 m0, s89, {pc * 1024}")
        gaps[8 + 2 * pc].append(f"global_load_lds_dwordx4 {vo[pc & 1]}, s[{94 + 2 * (pc >> 1)}:{95 + 2 * (pc >> 1)}]")
    i = 0
    for mt in range(4):
        for n4 in range(4):
            out.append(mfma(op, mt, n4, cur))
            out += gaps[i]
            i += 1
    return out


def block(op):
    o = ["s_lshr_b32 s92, %[lda128], 3", "s_lshr_b32 s93, %[ldw128], 3", "s_mov_b32 s84, 0",
         "s_add_u32 s85, %[s0b], %[cA]"] + ring("s85") + ["s_add_u32 s86, %[s0b], %[cB]",
         "v_add_u32 v192, s85, %[xf0]", "v_add_u32 v193, s86, %[wf0]"] + reads(0)
    o.append("L_g4_loop_%=:")
    o += step(op, 0, 1, "a", 1, 1, 7)                       # A1 of K tile j + 1
    o += step(op, 1, 2, "w", 2, 0, 8)                       # B0 of K tile j + 2
    # step 2 also prepares the read offsets of K tile g + 1 (s85 / s86 are consumed by the v_adds at the head of the step)
    nxt_rd = ([f"s_add_u32 s85, %[s0b], {KT}", "s_add_u32 s85, s85, %[cA]"] + ring("s85") +
              [f"s_add_u32 s86, %[s0b], {KT}", "s_add_u32 s86, s86, %[cB]"] + ring("s86"))
    o += step(op, 0, 3, "w", 2, 1, 9, nxt_rd)               # B1 of K tile j + 2
    # every fragment of K tile g is in registers; this wave's pieces of K tile g + 1 have landed (8 younger pieces stay in flight)
    o += ["s_waitcnt vmcnt(8)", "s_waitcnt lgkmcnt(0)", "s_barrier"]
    tail = [f"s_add_u32 %[s0b], %[s0b], {KT}"] + ring("%[s0b]") + ["s_add_u32 s84, s84, 1"]
    s3 = step(op, 1, 0, "a", 2, 0, 0)                       # A0 of K tile j + 2 into the slot B0(g) just left
    o += s3 + tail
    o += ["s_cmp_lt_u32 s84, %[nk]", "s_cbranch_scc1 L_g4_loop_%=",
          "s_nop 15", "s_nop 15"]                           # the last matrix results are in a[] before anything reads them
    return o


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "sam6d_amd", "csrc", "s6d_gemm4_asm.inc")
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_gemm4_asm.py -- do not edit; the K loop of one output tile of csrc/s6d_gemm4.hip.\n")
        for name, op in (("S6D_G4_ASM_BF16", "bf16"), ("S6D_G4_ASM_F16", "f16")):
            lines = block(op)
            f.write(f"#define {name} \\\n")
            f.write(" \\\n".join('  "' + ln + '\\n\\t"' for ln in lines))
            f.write("\n\n")
        cl = [f'"a{i}"' for i in range(256)] + [f'"v{i}"' for i in range(128, 194)] + [f'"s{i}"' for i in range(84, 98)]
        f.write("#define S6D_G4_ASM_CLOBBERS \\\n  " + ", ".join(cl) + ', "scc", "memory"\n')
    n = len(block("bf16"))
    print(f"wrote {dst}: {n} instructions per block (loop body {n - 20})")


if __name__ == "__main__":
    main()
