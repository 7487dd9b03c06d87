#!/usr/bin/env python3
"""Generates multirate.jl_amd/csrc/interp_lane_quad.inc: the hand-scheduled body of interp_lane_kernel (kernels_interp_lane.hip) -- the L
outputs one input sample of a FIRInterpolator yields (src/Filters.jl:505-512: for each input m, phases 1..L: y[(m-1)L + phi] =
dot(pfb[:, phi], window ending at m)), ComplexF32 samples x Float32 taps, a lane per channel -- ONE inline-assembly statement per input.

The 64 lanes of a wave are 64 channels at the same input index: the L phases' tap columns are wave-uniform and come by scalar loads
(s_load_dwordx4: four taps of one column) into fixed scalar registers, double-buffered: block b + 1's taps are requested in front of
block b's arithmetic and waited for behind it.  The window's T ComplexF32 samples are REGISTERS of the wave (operands w0 ... w{T-1}: the
kernel keeps the sliding window of its channels in VGPRs from input to input) -- no LDS read, no barrier.  A statement covers FOUR
consecutive inputs (operands w0 ... w{T+2}): the tap blocks cycle, block 0 of the next input is requested behind block 7 of this one.

Per tap i (oldest sample first) and phase p:  STRICT  t_p = x_i * tap[p][i] (v_pk_mul_f32: (re, im) packed, the tap broadcast);
acc_p = acc_p + t_p (v_pk_add_f32);  FUSED  acc_p = fma(x_i, tap[p][i], acc_p).  The first product of a phase initialises its sum
(src/support.jl:7).  Sums and products of a phase sit in fixed registers of different VGPR banks; the L sums leave by two 16-byte LDS
writes (the wave's own output patch: the kernel stores it as whole lines).

    python scripts/gen_interp_lane_asm.py      # rewrites the .inc (committed; the build does not run this script)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "multirate.jl_amd", "csrc", "interp_lane_quad.inc")

NB = 4                            # taps per block
SHAPES = [(32, 4)]                # (tapsPerPhi, L) the kernel is instantiated for
ACC0, TMP0 = 112, 120             # sums v[112:119] (phase p: 112 + 2 p), products v[120:127]: a phase's two on different banks (register mod 4)


def acc_reg(L, p):
    return ACC0 + 2 * p, TMP0 + 2 * ((p + 1) % L)


def tap_operand(first_sgpr, idx, fused):
    pair = f"s[{first_sgpr + (idx // 2) * 2}:{first_sgpr + (idx // 2) * 2 + 1}]"
    if fused:
        mod = "op_sel_hi:[1,0,1]" if idx % 2 == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]"
    else:
        mod = "op_sel_hi:[1,0]" if idx % 2 == 0 else "op_sel:[0,1]"
    return pair, mod


NINS = (4,)                       # inputs per statement: the tap blocks cycle (every input uses the same columns), so block 0 of input q + 1 is requested
                                  # behind block 7 of input q -- with four inputs ONE exposed wait for the scalar cache instead of four.  (Measured against one input per
                                  # statement on config 3a, same box, alternating: 2.90-2.92 ms both -- the kernel runs at the power limit, not at its waits.)


def gen(T, L, fused, NIN):
    nblk = T // NB
    s_per_buf = NB * L            # SGPRs of one tap buffer: NB taps of each of the L columns
    sbuf = (100 - 2 * s_per_buf, 100 - s_per_buf)          # two buffers ending at s99
    lines = []
    emit = lines.append

    def taps(buf, blk):
        for p in range(L):
            r = sbuf[buf] + NB * p
            emit(f"s_load_dwordx{NB} s[{r}:{r + NB - 1}], %[tp], 0x{(p * T + blk * NB) * 4:x}")

    taps(0, 0)
    emit("s_waitcnt lgkmcnt(0)")
    total = NIN * nblk
    for n in range(total):
        q, b = divmod(n, nblk)
        cur = n % 2
        if n + 1 < total:
            taps(1 - cur, (b + 1) % nblk)
        for j in range(NB):
            i = b * NB + j
            x = f"%[w{q + i}]"
            adds = []
            for p in range(L):
                acc, tmp = acc_reg(L, p)
                A, Tm = f"v[{acc}:{acc + 1}]", f"v[{tmp}:{tmp + 1}]"
                first = sbuf[cur] + NB * p
                if i == 0:                                    # the first product initialises
                    pair, mod = tap_operand(first, j, False)
                    emit(f"v_pk_mul_f32 {A}, {x}, {pair} {mod}")
                elif fused:
                    pair, mod = tap_operand(first, j, True)
                    emit(f"v_pk_fma_f32 {A}, {x}, {pair}, {A} {mod}")
                else:
                    pair, mod = tap_operand(first, j, False)
                    emit(f"v_pk_mul_f32 {Tm}, {x}, {pair} {mod}")
                    adds.append(f"v_pk_add_f32 {A}, {A}, {Tm}")
            for a in adds:
                emit(a)
        if b == nblk - 1:                                     # input q is complete: its L sums into the patch (32 bytes per input)
            assert L == 4
            emit(f"ds_write_b128 %[pa], v[{ACC0}:{ACC0 + 3}] offset:{32 * q}")
            emit(f"ds_write_b128 %[pa], v[{ACC0 + 4}:{ACC0 + 7}] offset:{32 * q + 16}")
            if q + 1 < NIN:
                emit("s_nop 1")                               # (the next input's first products overwrite the registers the writes take their data from)
        if n + 1 < total:
            emit("s_waitcnt lgkmcnt(0)")
    body = "\n".join(f'        "{ln}\\n\\t"' for ln in lines)
    vclob = list(range(ACC0, ACC0 + 2 * L)) + ([] if fused else list(range(TMP0, TMP0 + 2 * L)))
    sclob = range(sbuf[0], 100)
    clob = ", ".join([f'"v{i}"' for i in vclob] + [f'"s{i}"' for i in sclob])
    ins = ", ".join(f'[w{i}] "v"(w[{i}])' for i in range(T + NIN - 1))
    return (f"    if constexpr (T == {T} && L == {L} && NIN == {NIN} && FUSED == {'true' if fused else 'false'}) {{\n"
            f"        asm volatile(\n{body}\n"
            f"        :\n"
            f'        : {ins}, [tp] "s"(taps), [pa] "v"(patch)\n'
            f'        : {clob}, "memory");\n'
            f"    }}\n")


def render():
    parts = ["// GENERATED by scripts/gen_interp_lane_asm.py -- do not edit; see that script for what the statements do and why they are assembly.\n"
             "// Included inside interp_lane_quad<FUSED, T, L, NIN>(w, taps, patch) -- the outputs of NIN consecutive inputs: w[0 .. T + NIN - 2]: the windows (input q:\n"
             "// w[q .. q + T - 1], oldest sample first); patch: LDS byte address of this lane's NIN x L outputs (NIN x 32 bytes).\n"]
    for (T, L) in SHAPES:
        for nin in NINS:
            for fused in (False, True):
                parts.append(gen(T, L, fused, nin))
    return "".join(parts)


def main():
    text = render()
    with open(OUT, "w") as fh:
        fh.write(text)
    print("wrote", OUT, text.count("\n"), "lines")


if __name__ == "__main__":
    main()
