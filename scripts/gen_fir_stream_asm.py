#!/usr/bin/env python3
"""Generates multirate.jl_amd/csrc/fir_stream_pair_c64_m4.inc: the hand-scheduled pair of outputs of fir_stream_kernel
(fir_stream_kernel.inc) for BASELINE config 3b's shape -- FIRDecimator 1//4, 128 taps, ComplexF32 samples x Float32 taps (src/Filters.jl:598-631;
the dot product of src/support.jl:33-42) -- one inline-assembly statement per pair.

Why assembly: the compiler's loop over the blocks of 8 samples (4 x ds_read_b128, the block's taps by s_load_dwordx8 / x4, 32 packed
multiplies + 32 packed adds for the lane's two outputs) requests the first block of every iteration at its top and waits for it at once:
one LDS + scalar-cache latency exposed per two blocks, VALU 65 % busy.  Written out in C++ as a software pipeline the
compiler makes it slower (profiles/r04/experiments.md B).  Here block b + 1's samples and taps are requested in front of block b's
arithmetic and waited for behind it, for all 16 blocks of the window: two sample buffers in fixed VGPRs (a 16-byte read delivers two
samples: their halves are named registers, which an asm operand's sub-registers cannot be) and three tap buffers in fixed SGPRs
(output 1's taps trail output 0's by M = 4: half of them are the block before's).

Arithmetic and order are the compiler's form's, bit for bit: per sample e of a block, output 0 takes tap 8 b + e and output 1 tap
8 b + e - 4 on the same sample; STRICT p = t * x (v_pk_mul_f32, (re, im) packed), acc = acc + p (v_pk_add_f32); FUSED
acc = fma(t, x, acc); the first product of an output initialises its sum (support.jl:35).  The start-from-zero seam
(support.jl:46) is not in here: tiles that contain it take the C++ path.

LDS layout of a lane's run (StreamGeo<8, 4>: 16-byte reads, one pad chunk behind every 4): read i at byte 16 (i + i / 4).

    python scripts/gen_fir_stream_asm.py      # rewrites the .inc (committed; the build does not run this script)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "multirate.jl_amd", "csrc", "fir_stream_pair_c64_m4.inc")

M, BS, T = 4, 8, 128
NB = T // BS
XBUF = (32, 48)                 # first VGPR of the two sample buffers (16 registers each: 8 complex samples)
TBUF = (68, 76, 84)             # first SGPR of the three tap buffers (8 taps each)
ACC = {0: (64, 66), 1: (68, 70)}   # (sum, product) of output 0 / 1, in fixed registers of DIFFERENT banks (register number mod 4): left
                                     # to the allocator, sum and product of an output landed on the same two banks and every
                                     # v_pk_add_f32 read both operands from them -- the statement ran 11 % slower than the compiler's loop
V_CLOB = range(32, 72)
S_CLOB = range(68, 92)


def read_off(i):
    return 16 * (i + i // 4)


def tap_operand(buf, idx, fused):
    """(register pair, modifiers) for tap `idx` (0..7) of tap buffer `buf`, as src1 of a packed op, broadcast to (re, im)"""
    pair = f"s[{TBUF[buf] + (idx // 2) * 2}:{TBUF[buf] + (idx // 2) * 2 + 1}]"
    if fused:
        mod = "op_sel_hi:[1,0,1]" if idx % 2 == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]"
    else:
        mod = "op_sel_hi:[1,0]" if idx % 2 == 0 else "op_sel:[0,1]"
    return pair, mod


def gen(fused):
    lines = []
    emit = lines.append
    started = [False, False]

    def reads(buf, first_read, n):
        for i in range(n):
            r = XBUF[buf] + 4 * i
            emit(f"ds_read_b128 v[{r}:{r + 3}], %[ad] offset:{read_off(first_read + i)}")

    def taps(block):
        b = TBUF[block % 3]
        emit(f"s_load_dwordx8 s[{b}:{b + 7}], %[tp], 0x{block * BS * 4:x}")

    def mac(out, xbuf, e, tbuf, tidx):
        x = f"v[{XBUF[xbuf] + 2 * e}:{XBUF[xbuf] + 2 * e + 1}]"
        pair, mod = tap_operand(tbuf, tidx, fused and started[out])
        acc, tmp = f"v[{ACC[out][0]}:{ACC[out][0] + 1}]", f"v[{ACC[out][1]}:{ACC[out][1] + 1}]"
        if not started[out]:                                  # the first product initialises
            emit(f"v_pk_mul_f32 {acc}, {x}, {pair} {tap_operand(tbuf, tidx, False)[1]}")
            started[out] = True
        elif fused:
            emit(f"v_pk_fma_f32 {acc}, {x}, {pair}, {acc} {mod}")
        else:
            emit(f"v_pk_mul_f32 {tmp}, {x}, {pair} {mod}")
            return [f"v_pk_add_f32 {acc}, {acc}, {tmp}"]
        return []

    # prologue
    reads(0, 0, 4)
    taps(0)
    emit("s_waitcnt lgkmcnt(0)")
    for b in range(NB):
        xb = b % 2
        if b + 1 < NB:
            reads(1 - xb, 4 * (b + 1), 4)
            taps(b + 1)
        else:
            reads(1 - xb, 4 * NB, 2)                         # the M samples past the first output's window: the second output alone
        for e in range(BS):
            adds = []
            adds += mac(0, xb, e, b % 3, e)
            k1 = b * BS + e - M                               # output 1's tap on this sample
            if k1 >= 0:
                if e < M:
                    adds += mac(1, xb, e, (b - 1) % 3, e + M)
                else:
                    adds += mac(1, xb, e, b % 3, e - M)
            for a in adds:
                emit(a)
        emit("s_waitcnt lgkmcnt(0)")
    xb = NB % 2
    for e in range(M):
        for a in mac(1, xb, e, (NB - 1) % 3, e + M):
            emit(a)
    emit(f"v_mov_b64 %[a0], v[{ACC[0][0]}:{ACC[0][0] + 1}]")
    emit(f"v_mov_b64 %[a1], v[{ACC[1][0]}:{ACC[1][0] + 1}]")
    body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
    clob = ", ".join([f'"v{i}"' for i in V_CLOB] + [f'"s{i}"' for i in S_CLOB])
    return (f"    if constexpr (FUSED_ == {'true' if fused else 'false'}) {{\n"
            f"        asm volatile(\n{body}\n"
            f'        : [a0] "=&v"(a0), [a1] "=&v"(a1)\n'
            f'        : [ad] "v"(lds_addr), [tp] "s"(taps)\n'
            f'        : {clob}, "memory");\n'
            f"    }}\n")


def render():
    return ("// GENERATED by scripts/gen_fir_stream_asm.py -- do not edit; see that script for what the statements do and why they are assembly.\n"
            "// Included inside stream_pair_c64_m4_t128<FUSED_>(lds_addr, taps, a0, a1).\n"
            + gen(False) + gen(True))


def main():
    text = render()
    with open(OUT, "w") as fh:
        fh.write(text)
    print("wrote", OUT, text.count("\n"), "lines")


if __name__ == "__main__":
    main()
