#!/usr/bin/env python3
"""Generates multirate.jl_amd/csrc/decim_lane_group.inc: the hand-scheduled body of decim_lane_kernel (kernels_decim_lane.hip) -- FIRDecimator
(src/Filters.jl:598-631) 1//4 with 128 taps, ComplexF32 samples x Float32 taps, a lane per channel, in TRANSPOSED form: the 32 outputs
whose windows contain an input sample are in flight in 32 accumulators of the lane, and ONE inline-assembly statement adds the four
samples of a group (the M = 4 inputs between two outputs) to all of them.

Output j (newest sample m'_j = 4 j) receives sample m' = 4 g - 3 + i (i = 0 .. 3) of group g with tap t = m' - (4 j - 127) = 124 + i - 4 s',
s' = j - g = 0 .. 31 the output's AGE (0: complete after this group).  Every accumulator meets its 128 products oldest sample first --
the order of the reference's dot (src/support.jl:33-42) -- and starts from -0.0, which is "the first product initialises"
((-0.0) + p == p for every p, bit for bit); outputs whose window reaches into the history start from +0.0 (support.jl:46).

Accumulator sigma holds output j = sigma (mod 32) for its whole life: the age of slot sigma in group g is (sigma - g) mod 32, so the
taps the 32 slots need for sample i are a ROTATION of the column R_i[s'] = h[124 + i - 4 s']: the kernel keeps the four columns
doubled (D_i[k] = R_i[k mod 32], k = 0 .. 63) and hands the statement the pointer D_0 + 32 - (g mod 32): slot sigma's tap is then
scalar register sigma of a plain 32-dword load.  Tap blocks (16 taps: s_load_dwordx16) are double-buffered in fixed scalar registers:
block b + 1 is requested in front of block b's 32 packed instructions and waited for behind them.

Per sample i and slot sigma:  STRICT  t = x_i * tap (v_pk_mul_f32: (re, im) packed, the tap broadcast); acc = acc + t (v_pk_add_f32);
FUSED  acc = fma(x_i, tap, acc).

    python scripts/gen_decim_lane_asm.py      # rewrites the .inc (committed; the build does not run this script)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "multirate.jl_amd", "csrc", "decim_lane_group.inc")

M, SLOTS, NBLK = 4, 32, 16           # samples per group, outputs in flight, taps per block
BUF = (68, 84)                       # first SGPR of the two tap buffers (16 each): s[68:99] clobbered


def tap_operand(first_sgpr, idx, fused):
    pair = f"s[{first_sgpr + (idx // 2) * 2}:{first_sgpr + (idx // 2) * 2 + 1}]"
    if fused:
        mod = "op_sel_hi:[1,0,1]" if idx % 2 == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]"
    else:
        mod = "op_sel_hi:[1,0]" if idx % 2 == 0 else "op_sel:[0,1]"
    return pair, mod


def gen(fused):
    lines = []
    emit = lines.append
    blocks = [(i, h) for i in range(M) for h in range(SLOTS // NBLK)]          # (sample, half of the slots)

    def load(buf, blk):
        i, h = blk
        emit(f"s_load_dwordx16 s[{BUF[buf]}:{BUF[buf] + 15}], %[tp], 0x{(i * 2 * SLOTS + h * NBLK) * 4:x}")

    load(0, blocks[0])
    emit("s_waitcnt lgkmcnt(0)")
    for n, (i, h) in enumerate(blocks):
        cur = n % 2
        if n + 1 < len(blocks):
            load(1 - cur, blocks[n + 1])
        for k0 in range(0, NBLK, 4):
            adds = []
            for q in range(4):
                k = k0 + q
                sigma = h * NBLK + k
                pair, mod = tap_operand(BUF[cur], k, fused)
                if fused:
                    emit(f"v_pk_fma_f32 %[a{sigma}], %[x{i}], {pair}, %[a{sigma}] {mod}")
                else:
                    emit(f"v_pk_mul_f32 %[t{q}], %[x{i}], {pair} {mod}")
                    adds.append(f"v_pk_add_f32 %[a{sigma}], %[a{sigma}], %[t{q}]")
            for a in adds:
                emit(a)
        if n + 1 < len(blocks):
            emit("s_waitcnt lgkmcnt(0)")
    body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
    outs = ", ".join(f'[a{s}] "+v"(acc[{s}])' for s in range(SLOTS))
    if not fused:
        outs += ", " + ", ".join(f'[t{q}] "=&v"(t[{q}])' for q in range(4))
    ins = ", ".join(f'[x{i}] "v"(x[{i}])' for i in range(M)) + ', [tp] "s"(tp)'
    clob = ", ".join(f'"s{i}"' for i in range(BUF[0], BUF[1] + 16))
    return (f"    if constexpr (FUSED == {'true' if fused else 'false'}) {{\n"
            f"        asm volatile(\n{body}\n"
            f"        : {outs}\n        : {ins}\n        : {clob});\n"
            f"    }}\n")


ROT = 4                              # the kernel handles ROT groups per iteration with the slots in place, then moves every accumulator ROT registers down


def gen_rotate():
    """acc[i] = acc[(i + ROT) mod 32], in place (tied operands: written in C++ the compiler builds the rotated set in other registers first)"""
    lines = [f"v_mov_b64 %[r{k}], %[a{k}]" for k in range(ROT)]
    lines += [f"v_mov_b64 %[a{i}], %[a{i + ROT}]" for i in range(SLOTS - ROT)]
    lines += [f"v_mov_b64 %[a{SLOTS - ROT + k}], %[r{k}]" for k in range(ROT)]
    body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
    outs = ", ".join(f'[a{s}] "+v"(acc[{s}])' for s in range(SLOTS)) + ", " + ", ".join(f'[r{k}] "=&v"(t[{k}])' for k in range(ROT))
    return (f"    if constexpr (ROTATE) {{\n"
            f"        asm volatile(\n{body}\n"
            f"        : {outs});\n"
            f"    }}\n")


def render():
    return ("// GENERATED by scripts/gen_decim_lane_asm.py -- do not edit; see that script for what the statements do and why they are assembly.\n"
            "// Included inside decim_lane_group<FUSED, ROTATE>(acc, x, tp) with `v2f_t t[4];` declared -- acc[0 .. 31]: the outputs in flight; x[0 .. 3]:\n"
            "// the group's samples; tp: the doubled tap columns + 32 - (the group's place in the iteration).  ROTATE: no arithmetic, the accumulators\n"
            f"// move {ROT} registers down.\n"
            "    if constexpr (!ROTATE) {\n"
            + gen(False) + gen(True) +
            "    }\n"
            + gen_rotate())


def main():
    text = render()
    with open(OUT, "w") as fh:
        fh.write(text)
    print("wrote", OUT, text.count("\n"), "lines")


if __name__ == "__main__":
    main()
