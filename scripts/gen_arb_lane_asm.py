#!/usr/bin/env python3
"""Generates multirate.jl_amd/csrc/arb_lane_pair.inc: the hand-scheduled body of arb_lane_kernel (kernels_arb_lane.hip) -- ONE
inline-assembly statement per pair of outputs, for every (tapsPerPhi, window offset D, numerics) the kernel is instantiated for.

Why assembly: the taps are wave-uniform and come by SCALAR loads (s_load_dwordx16: eight Float64 taps of one PFB column).  Left to
the compiler the loads of a whole output are hoisted in front of its arithmetic (256 SGPRs: spilled to VGPR lanes); issued and
waited for block by block from separate statements every block pays the scalar cache's latency (measured: the kernel ran at 31 %
VALU utilisation).  A load that is still in flight when a statement ends is not expressible (the compiler takes an asm output for
valid at once and may move or re-use it), so the whole pair is one statement with its own double buffer of tap blocks in fixed
scalar registers (declared clobbered: s[68:99] with blocks of four taps, the default; s[36:99] with blocks of eight): the loads of block b + 1 are issued in front of the arithmetic of block b and
waited for behind it.

What a statement does (T taps per phase, halves of HT = T/2 taps so that a window costs HT + 1 registers, not T + 1):
    half 0: ds_read_b64 x HT+1 (samples 0 .. HT of the window at `addr`)
            output 0 taps 0 .. HT-1 over samples i, output 1 taps 0 .. HT-1 over samples i + D
    half 1: ds_read_b64 x HT+1 (samples HT .. T)
            output 0 taps HT .. T-1, output 1 taps HT .. T-1 over samples i + D
per tap and PFB column: STRICT  p = tap * x (v_mul_f64) ; acc = acc + p (v_add_f64) -- the first tap's product initialises the sum
(src/support.jl:7); FUSED  acc = fma(tap, x, acc).  Oldest sample first, exactly the order of arb_pipe_kernel / arb_generic_kernel
and of the oracle.

    python scripts/gen_arb_lane_asm.py        # rewrites the .inc (committed; the build does not run this script)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "multirate.jl_amd", "csrc", "arb_lane_pair.inc")

NB = int(os.environ.get("LANE_BLOCK_TAPS", "4"))       # taps per block: 8 (s_load_dwordx16, 64 fixed SGPRs) or 4 (s_load_dwordx8, 32)
if NB == 8:
    BUF = {"A": (36, 52), "B": (68, 84)}   # first SGPR of the yLower / yUpper block of taps
    CLOBBER_LO, CLOBBER_HI = 36, 99
else:
    BUF = {"A": (68, 76), "B": (84, 92)}
    CLOBBER_LO, CLOBBER_HI = 68, 99


def sreg(first, j):
    return f"s[{first + 2 * j}:{first + 2 * j + 1}]"


def gen(T, D, fused):
    HT = T // 2
    nblk = HT // NB                         # blocks of NB taps per output and half
    lines = []
    emit = lines.append
    # the blocks of a pair in the order they are computed: (half, output, block within the half)
    blocks = [(h, o, b) for h in range(2) for o in range(2) for b in range(nblk)]

    def load(buf, blk):
        h, o, b = blk
        off = (h * HT + b * NB) * 8
        lo, up = BUF[buf]
        emit(f"s_load_dwordx{2 * NB} s[{lo}:{lo + 2 * NB - 1}], %[p{o}l], 0x{off:x}")
        emit(f"s_load_dwordx{2 * NB} s[{up}:{up + 2 * NB - 1}], %[p{o}u], 0x{off:x}")

    def reads(h):
        for i in range(HT + 1):
            emit(f"ds_read_b64 %[x{i}], %[ad] offset:{(h * HT + i) * 8}")

    def macs(buf, blk, first_of_output):
        h, o, b = blk
        lo, up = BUF[buf]
        d = D if o == 1 else 0
        for j in range(NB):
            x = f"%[x{b * NB + j + d}]"
            L, U = f"%[l{o}]", f"%[u{o}]"
            if first_of_output and j == 0:                    # the first product initialises (no add, no fma)
                emit(f"v_mul_f64 {L}, {sreg(lo, j)}, {x}")
                emit(f"v_mul_f64 {U}, {sreg(up, j)}, {x}")
            elif fused:
                emit(f"v_fma_f64 {L}, {sreg(lo, j)}, {x}, {L}")
                emit(f"v_fma_f64 {U}, {sreg(up, j)}, {x}, {U}")
            else:
                emit(f"v_mul_f64 %[t0], {sreg(lo, j)}, {x}")
                emit(f"v_mul_f64 %[t1], {sreg(up, j)}, {x}")
                emit(f"v_add_f64 {L}, {L}, %[t0]")
                emit(f"v_add_f64 {U}, {U}, %[t1]")

    bufs = ["A", "B"]
    reads(0)
    load("A", blocks[0])
    emit("s_waitcnt lgkmcnt(0)")
    for n, blk in enumerate(blocks):
        cur = bufs[n % 2]
        nxt = bufs[(n + 1) % 2]
        if n + 1 < len(blocks):
            load(nxt, blocks[n + 1])                          # in flight behind this block's arithmetic
        macs(cur, blk, first_of_output=(blk[0] == 0 and blk[2] == 0))
        if n + 1 < len(blocks) and blocks[n + 1][0] != blk[0]:
            reads(1)                                          # the first half's samples are used up: their registers take the second half
        if n + 1 < len(blocks):
            emit("s_waitcnt lgkmcnt(0)")
    body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
    xs = ", ".join(f'[x{i}] "=&v"(w[{i}])' for i in range(HT + 1))
    outs = '[l0] "=&v"(lo0), [u0] "=&v"(up0), [l1] "=&v"(lo1), [u1] "=&v"(up1), [t0] "=&v"(t0), [t1] "=&v"(t1), ' + xs
    ins = '[ad] "v"(addr), [p0l] "s"(tl0), [p0u] "s"(tu0), [p1l] "s"(tl1), [p1u] "s"(tu1)'
    clob = ", ".join(f'"s{i}"' for i in range(CLOBBER_LO, CLOBBER_HI + 1))
    return (f"    if constexpr (T == {T} && D == {D} && FUSED == {'true' if fused else 'false'}) {{\n"
            f"        v2u_t w[{HT + 1}];\n"
            f"        asm volatile(\n{body}\n"
            f"        : {outs}\n        : {ins}\n        : {clob}, \"memory\");\n"
            f"    }}\n")


def render():
    parts = ["// GENERATED by scripts/gen_arb_lane_asm.py -- do not edit; see that script for what the statements do and why they are assembly.\n"
             "// Included inside lane_pair<FUSED, T, D>(addr, tl0, tu0, tl1, tu1, lo0, up0, lo1, up1) with `double t0, t1;` declared.\n"]
    for T in (32, 16):
        for D in (0, 1):
            for fused in (False, True):
                parts.append(gen(T, D, fused))
    return "".join(parts)


def main():
    text = render()
    with open(OUT, "w") as fh:
        fh.write(text)
    print("wrote", OUT, text.count("\n"), "lines")


if __name__ == "__main__":
    main()
