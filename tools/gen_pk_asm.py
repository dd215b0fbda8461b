#!/usr/bin/env python
"""Generates the hand-scheduled gfx950 instruction streams of gemm_pk.hip's COMPUTE phase (bf16x6 with promoted
accumulation) as C string literals:  python tools/gen_pk_asm.py > nabu_amd/csrc/gemm_pk_asm.inc

Why a generator: the stream is ~180 instructions whose ORDER is the point (which VALU add sits in which MFMA
gap), and inline-asm operands cannot name one register of a 16-register tuple, so every register is physical:

  v[0:127]    accumulators, tile n = (i, j) = (n >> 1, n & 1) at v[16n : 16n+15]
  v[128:175]  A fragments, (i, plane q) at v[128 + 4(3i+q) : +3]
  v[176:199]  B fragments, (j, plane q) at v[176 + 4(3j+q) : +3]
  v[200:231]  two scratch tiles T0, T1 (16 registers each)

Schedule "S1" (the shipped one).  The six plane products of tile n are a dependent chain into scratch tile n % 2
that starts from 0, smallest terms first (a dependent v_mfma_f32_32x32x16_bf16 issues back to back with its
predecessor at the full rate: measured, tools/experiments/ub/pk_sched.hip).  The 16 v_add_f32 that promote the
finished chain of tile n-1 into its accumulator tile sit in the gaps after the 2nd .. 5th MFMA of chain n, four per
gap (a wave issues ~5 independent instructions under one 32-cycle MFMA; the first add comes two MFMAs = 64 cycles
after the chain's last MFMA issued: the matrix result is written 8 passes + 3 states after issue, and nothing
inside an asm statement is padded by the compiler).  Tile 7's partial sum is promoted at the start of the NEXT
stage (T1 is carried across the stage boundary; one add pass after the loop), so every stage runs the same stream.
Measured one wave per SIMD: 653 ns per stage = the bare 48-MFMA stream (655 ns): the adds are free.

Schedules "S0" (direct accumulation, no promotion) and "S2" (three scratch tiles, two interleaved chains, wrapping
around the stage: a throughput probe only) are emitted with --bench for the micro-benchmark."""
import os
import sys

ACC, FA, FB, TMP = 0, 128, 176, 200
PRODUCTS = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]     # (plane of A, plane of B), smallest first


def acc(n):
    return 'v[%d:%d]' % (ACC + 16 * n, ACC + 16 * n + 15)


def fa(i, q):
    b = FA + 4 * (3 * i + q)
    return 'v[%d:%d]' % (b, b + 3)


def fb(j, q):
    b = FB + 4 * (3 * j + q)
    return 'v[%d:%d]' % (b, b + 3)


def tmp(r):
    return 'v[%d:%d]' % (TMP + 16 * r, TMP + 16 * r + 15)


def mfma(dst, a, b, c):
    return 'v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s' % (dst, a, b, c)


def chain(n, r):
    """the six MFMAs of tile n into scratch tile r"""
    i, j = n >> 1, n & 1
    out = []
    for k, (qa, qb) in enumerate(PRODUCTS):
        # operands swapped (B fragment first): the result tile is C^T — a lane holds 4 consecutive columns n of one
        # row m per register quad, which the epilogue stores as 16-byte pieces
        out.append(mfma(tmp(r), fb(j, qb), fa(i, qa), '0' if k == 0 else tmp(r)))
    return out


def adds(n, r):
    return ['v_add_f32 v%d, v%d, v%d' % (ACC + 16 * n + e, ACC + 16 * n + e, TMP + 16 * r + e) for e in range(16)]


def stream_s0():
    out = []
    for qa, qb in PRODUCTS:
        for n in range(8):
            out.append(mfma(acc(n), fb(n & 1, qb), fa(n >> 1, qa), acc(n)))
    return out


def pk_adds(n, r):
    return ['v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]' % (ACC + 16 * n + e, ACC + 16 * n + e + 1, ACC + 16 * n + e,
                                                          ACC + 16 * n + e + 1, TMP + 16 * r + e, TMP + 16 * r + e + 1)
            for e in range(0, 16, 2)]


def stream_s1(share=(0, 0, 4, 4, 4, 4), packed=False):
    """two scratch tiles; chain n alone, the adds of chain n-1 (tile 7 of the previous stage for n = 0) in its
    gaps: share[k] adds behind the chain's MFMA k+1"""
    out = []
    for n in range(8):
        c = chain(n, n % 2)
        a = (pk_adds if packed else adds)((n - 1) % 8, (n - 1) % 2)
        at = 0
        for k in range(6):
            out.append(c[k])
            cnt = share[k] // 2 if packed else share[k]
            out += a[at:at + cnt]
            at += cnt
        assert at == len(a)
    return out


def stream_s2():
    """three scratch tiles, two interleaved chains; position p of the MFMA sequence: chain n owns positions
    6n + {0,2,..,10} (n even) or 6n + {1,3,..,11} (n odd); chain n's adds go into the gaps after positions
    6n + 13 .. 6n + 17, wrapping around the stage (chains 6 and 7 are promoted at the start of the next stage)"""
    npos = 48
    slot = {}
    for n in range(8):
        c = chain(n, n % 3)
        for k in range(6):
            p = (6 * n + 2 * k + (n & 1)) % npos
            assert p not in slot
            slot[p] = (n, k, c[k])
    gaps = {p: [] for p in range(npos)}
    for n in range(8):
        a = adds(n, n % 3)
        first = 6 * n + 13
        share = [4, 3, 3, 3, 3]
        at = 0
        for g, cnt in enumerate(share):
            gaps[(first + g) % npos] += a[at:at + cnt]
            at += cnt
    # chains whose positions wrap (n = 7: positions 43..53 -> 43,45,47,1,3,5) belong to TWO stages: the stream is
    # periodic, the kernel runs one extra "drain" pass of the wrapped part after the loop
    out = []
    for p in range(npos):
        out.append(slot[p][2])
        out += gaps[p]
    return out, slot, gaps


# ---------------------------------------------------------------------------------------------------------------
# NP = 2 ("f16x3": two scaled fp16 planes x = (h + l) / scale, products l.h, h.l, h.h on v_mfma_f32_32x32x16_f16).
#   v[0:127]    accumulators (as above)
#   v[128:159]  A fragments, (i, plane q) at v[128 + 4(2i+q) : +3]
#   v[160:175]  B fragments, (j, plane q) at v[160 + 4(2j+q) : +3]
#   v[176:239]  four scratch tiles T0..T3, chain n runs in T[n % 4]
# A chain is only three MFMAs long, so the 16 adds of a tile need all three gaps of a chain's length: the adds of
# tile n-1 sit behind the 2nd and 3rd MFMA of chain n (6 + 5; the 2nd MFMA of chain n issues 64 cycles after the last
# one of chain n-1, whose result is written 8 passes + 3 states after issue) and behind the 1st MFMA of chain n+1
# (5).  T[(n-1) % 4] is rewritten by chain n+3 at the earliest.  Tiles 6 and 7 of a stage finish in the next stage's
# first gaps (T2 the last elements, T3 whole: one partial add pass after the loop), so every stage runs one stream.
FA2, FB2, TMP2 = 128, 160, 176
PRODUCTS2 = [(1, 0), (0, 1), (0, 0)]


def fa2(i, q):
    b = FA2 + 4 * (2 * i + q)
    return 'v[%d:%d]' % (b, b + 3)


def fb2(j, q):
    b = FB2 + 4 * (2 * j + q)
    return 'v[%d:%d]' % (b, b + 3)


def tmp2(r):
    return 'v[%d:%d]' % (TMP2 + 16 * r, TMP2 + 16 * r + 15)


def stream_f16(split=(6, 5, 5), packed=False):
    """split = adds behind MFMA 1 (the LAST ones of tile n-2), behind MFMA 2 and MFMA 3 (the first ones of tile n-1)"""
    assert sum(split) == 16
    out = []
    for n in range(8):
        i, j = n >> 1, n & 1
        r = n % 4
        ms = []
        for k, (qa, qb) in enumerate(PRODUCTS2):
            ms.append('v_mfma_f32_32x32x16_f16 %s, %s, %s, %s' % (tmp2(r), fb2(j, qb), fa2(i, qa), '0' if k == 0 else tmp2(r)))
        def adds_of(tile, lo, hi):
            t = tile % 8
            a, b = ACC + 16 * t, TMP2 + 16 * (t % 4)
            if packed:      # v_pk_add_f32 on aligned register pairs (lo, hi even)
                assert lo % 2 == 0 and hi % 2 == 0
                return ['v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]' % (a + e, a + e + 1, a + e, a + e + 1, b + e, b + e + 1)
                        for e in range(lo, hi, 2)]
            return ['v_add_f32 v%d, v%d, v%d' % (a + e, a + e, b + e) for e in range(lo, hi)]
        first = split[1] + split[2]
        out.append(ms[0])
        out += adds_of(n - 2, first, 16)
        out.append(ms[1])
        out += adds_of(n - 1, 0, split[1])
        out.append(ms[2])
        out += adds_of(n - 1, split[1], first)
    return out


def stream_load2(va, vb):
    out = []
    for q in range(2):
        for i in range(4):
            out.append('ds_read_b128 %s, %s offset:%d' % (fa2(i, q), va, q * 8192 + i * 1024))
        for j in range(2):
            out.append('ds_read_b128 %s, %s offset:%d' % (fb2(j, q), vb, q * 8192 + j * 1024))
    return out


def operand_lists2():
    accs = ', '.join('"+{%s}"(acc[%d][%d])' % (acc(n), n >> 1, n & 1) for n in range(8))
    tmps = ', '.join('"+{%s}"(ts[%d])' % (tmp2(r), r) for r in range(4))
    fas_o = ', '.join('"=&{%s}"(ha[%d][%d])' % (fa2(i, q), i, q) for i in range(4) for q in range(2))
    fbs_o = ', '.join('"=&{%s}"(hb[%d][%d])' % (fb2(j, q), j, q) for j in range(2) for q in range(2))
    fas_i = ', '.join('"{%s}"(ha[%d][%d])' % (fa2(i, q), i, q) for i in range(4) for q in range(2))
    fbs_i = ', '.join('"{%s}"(hb[%d][%d])' % (fb2(j, q), j, q) for j in range(2) for q in range(2))
    print('#define PK2_ASM_LOAD_OUTPUTS %s, %s\n' % (fas_o, fbs_o))
    print('#define PK2_ASM_COMPUTE_INOUT %s, %s\n' % (accs, tmps))
    print('#define PK2_ASM_COMPUTE_INPUTS %s, %s\n' % (fas_i, fbs_i))
    # number of asm operands in front of the two LDS addresses of the LOAD statement
    print('// PK2_STREAM_LOAD: 12 outputs, then %12 = A address, %13 = B address\n')


def as_c_string(lines):
    return '\n'.join('      "%s\\n\\t"' % l for l in lines)


def stream_load(va, vb):
    """the 18 operand reads of a stage: A (i, q) at va + q * 8192 + i * 1024, B (j, q) at vb + q * 8192 + j * 1024
    (va / vb = the asm operand names of the two per-lane LDS addresses; vb already points at the B pieces)"""
    out = []
    for q in range(3):
        for i in range(4):
            out.append('ds_read_b128 %s, %s offset:%d' % (fa(i, q), va, q * 8192 + i * 1024))
        for j in range(2):
            out.append('ds_read_b128 %s, %s offset:%d' % (fb(j, q), vb, q * 8192 + j * 1024))
    return out


def macro(name, lines):
    print('#define %s \\' % name)
    print(' \\\n'.join('  "%s\\n\\t"' % l for l in lines))
    print()


def operand_lists():
    """constraint lists matching the physical registers above (C macros)"""
    accs = ', '.join('"+{%s}"(acc[%d][%d])' % (acc(n), n >> 1, n & 1) for n in range(8))
    tmps = ', '.join('"+{%s}"(%s)' % (tmp(r), nm) for r, nm in ((0, 't0'), (1, 'tpend')))
    fas_o = ', '.join('"=&{%s}"(fa[%d][%d])' % (fa(i, q), i, q) for i in range(4) for q in range(3))
    fbs_o = ', '.join('"=&{%s}"(fb[%d][%d])' % (fb(j, q), j, q) for j in range(2) for q in range(3))
    fas_i = ', '.join('"{%s}"(fa[%d][%d])' % (fa(i, q), i, q) for i in range(4) for q in range(3))
    fbs_i = ', '.join('"{%s}"(fb[%d][%d])' % (fb(j, q), j, q) for j in range(2) for q in range(3))
    print('#define PK_ASM_LOAD_OUTPUTS %s, %s\n' % (fas_o, fbs_o))
    print('#define PK_ASM_COMPUTE_INOUT %s, %s\n' % (accs, tmps))
    print('#define PK_ASM_COMPUTE_INPUTS %s, %s\n' % (fas_i, fbs_i))


def main():
    print('// generated by tools/gen_pk_asm.py — do not edit (register map and schedules: see the generator)')
    if len(sys.argv) > 1 and sys.argv[1] == '--bench':
        macro('PK_STREAM_S0', stream_s0())
        macro('PK_STREAM_S2', stream_s2()[0])
    macro('PK_STREAM_S1', stream_s1())
    # measured and dropped (sum over the cfg2 product shapes, direct accumulation 7.78 ms): S1 8.14 ms; the adds as
    # (0,0,3,3,3,7) 8.23, as (0,0,0,5,5,6) 8.15, as 8 v_pk_add_f32 per tile 9.39, S1 without s_setprio 8.13
    macro('PK_STREAM_LOAD', stream_load('%18', '%19'))
    operand_lists()
    # measured, sum over the cfg2 product shapes: adds as (6,5,5) 5.09 ms, (5,6,5) 5.04, (4,6,6) 5.37; direct accumulation
    # (no adds, error 1.24 x exact fp32 at K = 2048) 4.56
    split2 = tuple(int(v) for v in os.environ.get('PK2_SPLIT', '6,5,5').split(','))
    macro('PK2_STREAM', stream_f16(split2, packed=os.environ.get('PK2_PACKED') == '1'))
    print('#define PK2_PEND_FIRST %d   // adds of tile 6 the stream leaves to the next stage: elements PK2_PEND_FIRST..15\n'
          % (split2[1] + split2[2]))
    macro('PK2_STREAM_LOAD', stream_load2('%12', '%13'))
    operand_lists2()


if __name__ == '__main__':
    main()
