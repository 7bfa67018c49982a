#!/usr/bin/env python3
"""Generates stage_asm.inc: the butterfly stages of an E = 16 tile step as gfx950 assembly.

    python sve_ntt_amd/csrc/gen_stage_asm.py > sve_ntt_amd/csrc/stage_asm.inc

Why assembly.  Both 2^24 kernels are bound by VALU issue.  hipcc turns every
conditional "+N" of the canonical 64-bit-modulus arithmetic (reference: modmul/sve/p-adic-64.hpp:
44-62, 90-92: compare-and-add) into two v_cndmask_b32 and a 64-bit add.  Here the borrow / carry
SGPR pair that v_subb_co / v_addc_co produce is moved straight into EXEC and the correction is ONE
v_lshl_add_u64 on the lanes that need it: 24 instead of 30 full-rate-class VALU instructions per
butterfly (profiles/r02/ubench_bfly_asm.txt: 135 against 155 cycles at 4 waves per SIMD).  That
needs the 32-bit halves of 64-bit values by name, which inline-asm operands cannot give, so the 16
tile elements of a thread live in FIXED registers (x_i = v[2i:2i+1], bound with "{v[a:b]}"
constraints) and the temporaries are fixed too.  Four butterflies are interleaved per asm statement
so that every dependent instruction has three independent ones in front of it.

Arithmetic is exactly csrc/field64.h's (montmul / addmod / submod / butterfly_fwd / butterfly_inv):
same values in, same canonical values out; tests compare the kernels with the oracle bit for bit.

EXEC invariant.  Every statement narrows EXEC for its corrections and restores it from `c.save`, the
EXEC read once at the entry of TileNTT::step_asm -- NOT from the EXEC it found ("exec" is not in the
clobber lists: hipcc would have to re-materialise it around ~140 statements per kernel).  That is
correct because every statement runs in the kernel's uniform control flow: a workgroup either
returns whole (tile_kernel: !t.live) or runs every step with all lanes; loads are selected, not
branched around, and the only divergent regions are plain C++ stores behind in_range() -- no asm
statement may ever be placed inside one.  tile_ntt.h repeats this where the statements are used.

Register map (per thread): data v0..v31; slot s of an asm statement uses v[32+12s .. 43+12s] and
s[60+4s .. 63+4s]; vcc is the unused carry-out of v_mad_u64_u32.  All of them are declared clobbered.
"""
import sys

E = 16
TEMP_BASE = 32
TEMP_PER_SLOT = 12
SG_BASE = 60
MAXW = 4


def pair(r):
    return 'v[%d:%d]' % (r, r + 1)


class Slot:
    def __init__(self, s):
        t = TEMP_BASE + TEMP_PER_SLOT * s
        self.m0, self.m1, self.m2, self.h, self.z, self.d = (t, t + 2, t + 4, t + 6, t + 8, t + 10)
        self.sb = 's[%d:%d]' % (SG_BASE + 4 * s, SG_BASE + 4 * s + 1)
        self.sc = 's[%d:%d]' % (SG_BASE + 4 * s + 2, SG_BASE + 4 * s + 3)


def zero_ext_init(S):
    """The high half of the slot's zero-extension pair stays 0 for the whole step: it is an
    in/out operand ("+{v41}"(zr[0]) ...) that no instruction writes, so nothing to do here."""
    return []


def hi_chain(S, x0, x1, y0, y1, low, mid, top, first):
    """top <- hi64(x * y), mid.lo <- bits 32..63 of the product (`first`: the instruction(s) that leave
    low = x0*y0 [+0], whose high word feeds the chain -- or put bits 32..63 of x0*y0 into z directly).
    The two cross products are summed in ONE 64-bit accumulator; the carry of that sum (SGPR pair
    S.sc, the carry-out of v_mad_u64_u32) is worth 2^32 in the top product and goes in with a
    v_addc on its high word.  Against adding the cross products' halves separately this saves two
    zero-extending v_mov per product and the 64-bit add (r02: 7 -> 3 v_mov per Montgomery product)."""
    z = S.z
    return first + [
        'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(mid), x0, y1, pair(z)),
        'v_mad_u64_u32 %s, %s, %s, %s, %s' % (pair(mid), S.sc, x1, y0, pair(mid)),
        'v_mov_b32 v%d, v%d' % (z, mid + 1),
        'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(top), x1, y1, pair(z)),
        'v_addc_co_u32 v%d, vcc, 0, v%d, %s' % (top + 1, top + 1, S.sc),
    ]


def mont(S, al, ah, w0, w1):
    """h <- hi64(a*w), m2 <- hi64(q*N) with q = lo64(a*w) * N^-1 mod 2^64 (field64.h: montmul).
    al/ah: the multiplicand's halves (VGPR operand texts); w0/w1: the multiplier's (VGPR or SGPR)."""
    m0, m1, m2, h, z = S.m0, S.m1, S.m2, S.h, S.z
    return (
        # a * w: t0 = m0.lo, t1 = m2.lo, h = (t2, t3)
        hi_chain(S, al, ah, w0, w1, m0, m2, h,
                 ['v_mad_u64_u32 %s, vcc, %s, %s, 0' % (pair(m0), al, w0),
                  'v_mov_b32 v%d, v%d' % (z, m0 + 1)]) +
        # q = (t0, t1) * N^-1 mod 2^64 -> (m1.lo, m1.hi)
        ['v_mad_u64_u32 %s, vcc, v%d, %%[ni0], 0' % (pair(m1), m0),
         'v_mul_lo_u32 v%d, v%d, %%[ni1]' % (m0 + 1, m0),
         'v_mul_lo_u32 v%d, v%d, %%[ni0]' % (m2, m2),
         'v_add3_u32 v%d, v%d, v%d, v%d' % (m1 + 1, m1 + 1, m0 + 1, m2)] +
        # m2 <- hi64(q * N); the low half of q * N is a * w's and is never formed
        hi_chain(S, 'v%d' % m1, 'v%d' % (m1 + 1), '%[n0]', '%[n1]', None, m0, m2,
                 ['v_mul_hi_u32 v%d, v%d, %%[n0]' % (z, m1)]))


def fix_if(S_mask, reg, addend='%[N]'):
    """+N (or another SGPR-pair addend) on the lanes whose bit is set in the SGPR pair"""
    return ('fix', 's_and_b64 exec, %%[save], %s' % S_mask,
            'v_lshl_add_u64 %s, %s, 0, %s' % (pair(reg), pair(reg), addend))


def fix_ifnot(S_mask, reg):
    return ('fix', 's_andn2_b64 exec, %%[save], %s' % S_mask, 'v_lshl_add_u64 %s, %s, 0, %%[N]' % (pair(reg), pair(reg)))


def sub_into(S, dl, al, ah, bl, bh):
    """v[dl:dl+1] = a - b, borrow in S.sb"""
    return ['v_sub_co_u32 v%d, %s, v%d, v%d' % (dl, S.sb, al, bl),
            'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (dl + 1, S.sb, ah, bh, S.sb)]


def mont_result(S, dl):
    """v[dl:dl+1] = h - m2 (+N on borrow): the canonical Montgomery product"""
    return sub_into(S, dl, S.h, S.h + 1, S.m2, S.m2 + 1) + [fix_if(S.sb, dl)]


ARITHS = ('ARITH_MONT', 'ARITH_GOLD', 'ARITH_SHOUP')


def product128(S, al, ah, w0, w1):
    """m0.lo = t0, m2.lo = t1, h = (t2, t3) of a * w"""
    m0, m2, h, z = S.m0, S.m2, S.h, S.z
    return hi_chain(S, al, ah, w0, w1, m0, m2, h,
                    ['v_mad_u64_u32 %s, vcc, %s, %s, 0' % (pair(m0), al, w0),
                     'v_mov_b32 v%d, v%d' % (z, m0 + 1)])


def gold_mul(S, al, ah, w0, w1, dst):
    """v[dst:dst+1] = a * w mod 2^64 - 2^32 + 1, canonical (field64.h: gold_mul).  negN = eps."""
    m0, m2, h = S.m0, S.m2, S.h
    return (product128(S, al, ah, w0, w1) +
            ['v_mov_b32 v%d, v%d' % (m0 + 1, m2),                                  # lo = {t0, t1}
             'v_sub_co_u32 v%d, %s, v%d, v%d' % (dst, S.sb, m0, h + 1),             # lo - t3
             'v_subb_co_u32 v%d, %s, v%d, 0, %s' % (dst + 1, S.sb, m0 + 1, S.sb),
             fix_if(S.sb, dst),                                                     # wrapped: -eps = +N
             'v_mad_u64_u32 %s, %s, v%d, %%[nn0], %s' % (pair(dst), S.sc, h, pair(dst)),  # + t2 * eps
             fix_if(S.sc, dst, '%[negN]'),                                          # wrapped: +eps
             'v_cmp_ge_u64 %s, %s, %%[N]' % (S.sb, pair(dst)),
             fix_if(S.sb, dst, '%[negN]')])                                         # >= N: -N = +eps


def shoup_mul(S, al, ah, w0, w1, p0, p1, dst):
    """v[dst:dst+1] = a*w - hi64(a*w')*N, then -N if >= N (field64.h: shoup_mul); dst must not
    hold a.  negN = 2^64 - N as (nn0, nn1)."""
    m1, m2, h, z = S.m1, S.m2, S.h, S.z
    return hi_chain(S, al, ah, p0, p1, None, m1, h,                                 # q = hi64(a * w')
                    ['v_mul_hi_u32 v%d, %s, %s' % (z, al, p0)]) + [
            'v_mad_u64_u32 %s, vcc, %s, %s, 0' % (pair(dst), al, w0),
            'v_mad_u64_u32 %s, vcc, v%d, %%[nn0], %s' % (pair(dst), h, pair(dst)),  # + q0 * negN0
            'v_mov_b32 v%d, v%d' % (m2, dst + 1),                                   # high word: only m2.lo counts
            'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(m2), al, w1, pair(m2)),
            'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(m2), ah, w0, pair(m2)),
            'v_mad_u64_u32 %s, vcc, v%d, %%[nn1], %s' % (pair(m2), h, pair(m2)),
            'v_mad_u64_u32 %s, vcc, v%d, %%[nn0], %s' % (pair(m2), h + 1, pair(m2)),
            'v_mov_b32 v%d, v%d' % (dst + 1, m2),
            'v_cmp_ge_u64 %s, %s, %%[N]' % (S.sb, pair(dst)),
            fix_if(S.sb, dst, '%[negN]')]


def mulmod(arith, S, al, ah, W, dst):
    """v[dst:dst+1] = a * w in the back end's domain, canonical.  W = (w0, w1[, p0, p1])."""
    if arith == 'ARITH_MONT':
        return mont(S, al, ah, W[0], W[1]) + mont_result(S, dst)
    if arith == 'ARITH_GOLD':
        return gold_mul(S, al, ah, W[0], W[1], dst)
    return shoup_mul(S, al, ah, W[0], W[1], W[2], W[3], dst)


def bf_fwd_a(arith, S, x, y, W):
    """(x, y) <- (x + y, (x - y) * w)"""
    e = S.m2
    return (zero_ext_init(S) +
            ['v_sub_co_u32 v%d, %s, v%d, v%d' % (S.d, S.sb, x, y),
             'v_lshl_add_u64 %s, %s, 0, %%[negN]' % (pair(e), pair(y)),
             'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (S.d + 1, S.sb, x + 1, y + 1, S.sb),
             'v_add_co_u32 v%d, %s, v%d, v%d' % (x, S.sc, x, e),
             fix_if(S.sb, S.d),
             'v_addc_co_u32 v%d, %s, v%d, v%d, %s' % (x + 1, S.sc, x + 1, e + 1, S.sc),
             fix_ifnot(S.sc, x)] +
            mulmod(arith, S, 'v%d' % S.d, 'v%d' % (S.d + 1), W, y))


def bf_inv_a(arith, S, x, y, W):
    """(x, y) <- (x + y*w, x - y*w)"""
    e = S.m0
    return (zero_ext_init(S) + mulmod(arith, S, 'v%d' % y, 'v%d' % (y + 1), W, S.d) +
            ['v_lshl_add_u64 %s, %s, 0, %%[negN]' % (pair(e), pair(S.d)),
             'v_sub_co_u32 v%d, %s, v%d, v%d' % (y, S.sb, x, S.d),
             'v_add_co_u32 v%d, %s, v%d, v%d' % (x, S.sc, x, e),
             'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (y + 1, S.sb, x + 1, S.d + 1, S.sb),
             'v_addc_co_u32 v%d, %s, v%d, v%d, %s' % (x + 1, S.sc, x + 1, e + 1, S.sc),
             fix_if(S.sb, y),
             fix_ifnot(S.sc, x)])


def twist_a(arith, S, x, H, L):
    """x <- x * hi * lo;  H, L = (lo word, hi word[, companion lo, companion hi]) operand texts"""
    xr = ('v%d' % x, 'v%d' % (x + 1))
    dr = ('v%d' % S.d, 'v%d' % (S.d + 1))
    if arith == 'ARITH_SHOUP':
        return (zero_ext_init(S) + mulmod(arith, S, xr[0], xr[1], H, S.d) + mulmod(arith, S, dr[0], dr[1], L, x))
    return (zero_ext_init(S) + mulmod(arith, S, H[0], H[1], L, S.d) + mulmod(arith, S, xr[0], xr[1], dr, x))


def bf_plain(S, x, y):
    """(x, y) <- (x + y, x - y)            field64.h: twiddle-less butterfly"""
    e = S.m2
    return ['v_lshl_add_u64 %s, %s, 0, %%[negN]' % (pair(e), pair(y)),
            'v_sub_co_u32 v%d, %s, v%d, v%d' % (y, S.sb, x, y),
            'v_add_co_u32 v%d, %s, v%d, v%d' % (x, S.sc, x, e),
            'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (y + 1, S.sb, x + 1, y + 1, S.sb),
            'v_addc_co_u32 v%d, %s, v%d, v%d, %s' % (x + 1, S.sc, x + 1, e + 1, S.sc),
            fix_if(S.sb, y),
            fix_ifnot(S.sc, x)]


def mul_inplace(S, x, w0, w1):
    """x <- x * w"""
    return zero_ext_init(S) + mont(S, 'v%d' % x, 'v%d' % (x + 1), w0, w1) + mont_result(S, x)



# ---- scheduling -------------------------------------------------------------------------------
def interleave(streams):
    """Round-robin merge of the slots' instruction streams.  Corrections run with EXEC narrowed;
    neighbouring ones share the restore."""
    out = []
    idx = [0] * len(streams)
    while any(i < len(s) for i, s in zip(idx, streams)):
        for k, s in enumerate(streams):
            if idx[k] < len(s):
                out.append(s[idx[k]])
                idx[k] += 1
    flat = []
    narrowed = False
    for it in out:
        if isinstance(it, tuple):
            flat += [it[1], it[2]]
            narrowed = True
        else:
            if narrowed:
                flat.append('s_mov_b64 exec, %[save]')
                narrowed = False
            flat.append(it)
    if narrowed:
        flat.append('s_mov_b64 exec, %[save]')
    return hazards(flat)


def sgpr_written(ins):
    op = ins.split()[0]
    if op in ('v_sub_co_u32', 'v_subb_co_u32', 'v_add_co_u32', 'v_addc_co_u32'):
        return ins.split(',')[1].strip()
    if op == 'v_mad_u64_u32':  # carry-out: "v[a:b], s[c:d], ..." -- the first comma is inside v[..]
        sdst = ins.split('],', 1)[1].split(',')[0].strip()
        return sdst if sdst.startswith('s[') else None
    if op == 'v_cmp_ge_u64':
        return ins.split()[1].rstrip(',')
    return None


def sgpr_read(ins):
    op = ins.split()[0]
    if op in ('v_subb_co_u32', 'v_addc_co_u32'):
        return ins.split(',')[-1].strip()
    if op in ('s_and_b64', 's_andn2_b64'):
        return ins.split(',')[-1].strip()
    return None


def hazards(seq):
    """gfx950: two wait states between a VALU write of an SGPR pair and a VALU read of it as carry-in
    (hipcc's own code: v_sub_co / s_nop 1 / v_subb_co); the same distance is kept before SALU reads."""
    out = []
    for ins in seq:
        r = sgpr_read(ins)
        if r is not None:
            for back in (1, 2):  # two other instructions between the write and the read
                if len(out) >= back and sgpr_written(out[-back]) == r:
                    out.append('s_nop %d' % (2 - back))
                    break
        out.append(ins)
    return out


# ---- C++ wrappers -------------------------------------------------------------------------------
CONSTS = {
    'N': '[N] "s"(c.N)', 'negN': '[negN] "s"(c.negN)', 'n0': '[n0] "s"(c.n0)', 'n1': '[n1] "s"(c.n1)',
    'ni0': '[ni0] "s"(c.ni0)', 'ni1': '[ni1] "s"(c.ni1)', 'save': '[save] "s"(c.save)',
    'nn0': '[nn0] "s"(c.nn0)', 'nn1': '[nn1] "s"(c.nn1)',
}


def clobbers(W):
    cl = ['"vcc"', '"scc"']
    zeros = {Slot(k).z + 1 for k in range(W)}
    cl += ['"v%d"' % r for r in range(TEMP_BASE, TEMP_BASE + TEMP_PER_SLOT * W) if r not in zeros]
    cl += ['"s%d"' % r for r in range(SG_BASE, SG_BASE + 4 * W)]
    return ', '.join(cl)


def asm_stmt(streams, data_regs, inputs):
    body = interleave(streams)
    text = ''.join('        "%s\\n\\t"\n' % l for l in body)
    used = [k for k in CONSTS if any('%%[%s]' % k in l for l in body)]
    outs = ', '.join('"+{v[%d:%d]}"(x[%d])' % (2 * i, 2 * i + 1, i) for i in data_regs)
    # the slots' persistent zeros (high halves of the zero-extension pairs)
    outs += ''.join(', "+{v%d}"(zr[%d])' % (Slot(k).z + 1, k) for k in range(len(streams)))
    return ('    asm volatile(\n%s        : %s\n        : %s\n        : %s);\n'
            % (text, outs, ', '.join(inputs + [CONSTS[k] for k in used]), clobbers(len(streams))))


def halves(name, expr):
    """asm input operands for the two halves of a 64-bit C expression"""
    return ['[%s0] "v"((u32)(%s))' % (name, expr), '[%s1] "v"((u32)((%s) >> 32))' % (name, expr)]


def butterflies(r):
    return [(i, i + (1 << r)) for i in range(E) if not i & (1 << r)]


def trivial(i, r):
    return (i & ((1 << r) - 1)) == 0


def gen_bfly_group(arith, mode, r, grp, triv):
    bfs = butterflies(r)[4 * grp:4 * grp + 4]
    streams, inputs, regs = [], [], []
    for k, (a, b) in enumerate(bfs):
        S = Slot(k)
        regs += [a, b]
        if triv and trivial(a, r):
            streams.append(bf_plain(S, 2 * a, 2 * b))
            continue
        W = ['%%[w%d0]' % k, '%%[w%d1]' % k]
        inputs += halves('w%d' % k, 'w%d' % k)
        if arith == 'ARITH_SHOUP':
            W += ['%%[p%d0]' % k, '%%[p%d1]' % k]
            inputs += halves('p%d' % k, 'p%d' % k)
        fn = bf_fwd_a if mode == 'MODE_FWD' else bf_inv_a
        streams.append(fn(arith, S, 2 * a, 2 * b, W))
    s = 'template <> struct BflyGroup<%s, %s, %d, %d, %s> {\n' % (arith, mode, r, grp, 'true' if triv else 'false')
    s += ('  static __device__ __forceinline__ void run(u64 (&x)[16], u64 w0, u64 w1, u64 w2, u64 w3,\n'
          '                                             u64 p0, u64 p1, u64 p2, u64 p3, u32 (&zr)[4],\n'
          '                                             const AsmConsts &c) {\n')
    s += asm_stmt(streams, sorted(regs), inputs)
    s += '  }\n};\n'
    return s


def twist_width(arith):
    """elements per twist statement: the FixedPoint64 back end brings four words per element"""
    return 2 if arith == 'ARITH_SHOUP' else 4


def gen_twist_group(arith, grp):
    streams, inputs = [], []
    tw = twist_width(arith)
    for k in range(tw):
        H = ['%%[h%d0]' % k, '%%[h%d1]' % k]
        L = ['%%[l%d0]' % k, '%%[l%d1]' % k]
        inputs += halves('h%d' % k, 'h[%d]' % k) + halves('l%d' % k, 'l[%d]' % k)
        if arith == 'ARITH_SHOUP':
            H += ['%%[hp%d0]' % k, '%%[hp%d1]' % k]
            L += ['%%[lp%d0]' % k, '%%[lp%d1]' % k]
            inputs += halves('hp%d' % k, 'hp[%d]' % k) + halves('lp%d' % k, 'lp[%d]' % k)
        streams.append(twist_a(arith, Slot(k), 2 * (tw * grp + k), H, L))
    s = 'template <> struct TwistGroup<%s, %d> {\n' % (arith, grp)
    s += ('  static __device__ __forceinline__ void run(u64 (&x)[16], const u64 (&h)[%d], const u64 (&l)[%d],\n'
          '                                             const u64 (&hp)[%d], const u64 (&lp)[%d], u32 (&zr)[4],\n'
          '                                             const AsmConsts &c) {\n' % (tw, tw, tw, tw))
    s += asm_stmt(streams, range(tw * grp, tw * grp + tw), inputs)
    s += '  }\n};\n'
    return s


def gen_mont_group(grp):
    streams, inputs = [], []
    for k in range(4):
        streams.append(mul_inplace(Slot(k), 2 * (4 * grp + k), '%%[w%d0]' % k, '%%[w%d1]' % k))
        inputs += halves('w%d' % k, 'w%d' % k)
    s = 'template <> struct MontGroup<%d> {\n' % grp
    s += ('  static __device__ __forceinline__ void run(u64 (&x)[16], u64 w0, u64 w1, u64 w2, u64 w3,\n'
          '                                             u32 (&zr)[4], const AsmConsts &c) {\n')
    s += asm_stmt(streams, range(4 * grp, 4 * grp + 4), inputs)
    s += '  }\n};\n'
    return s


def gen_scale_group(r, grp):
    """x0 of every butterfly of BflyGroup<., ., r, grp, .> times a wave-uniform factor"""
    bfs = butterflies(r)[4 * grp:4 * grp + 4]
    streams = [mul_inplace(Slot(k), 2 * a, '%[s0]', '%[s1]') for k, (a, _) in enumerate(bfs)]
    s = 'template <> struct ScaleGroup<%d, %d> {\n' % (r, grp)
    s += '  static __device__ __forceinline__ void run(u64 (&x)[16], u64 s, u32 (&zr)[4], const AsmConsts &c) {\n'
    s += asm_stmt(streams, [a for a, _ in bfs], ['[s0] "s"((u32)s)', '[s1] "s"((u32)(s >> 32))'])
    s += '  }\n};\n'
    return s


HEADER = """// GENERATED by gen_stage_asm.py -- do not edit; regenerate with
//     python sve_ntt_amd/csrc/gen_stage_asm.py > sve_ntt_amd/csrc/stage_asm.inc
// The butterfly stages of an E = 16 tile step as gfx950 assembly on fixed registers (x_i =
// v[2i:2i+1]); see the generator's docstring for the why and the register map.  Included by
// tile_ntt.h inside namespace sventt_hip, device compilation only.

struct AsmConsts {
  u64 N, negN;
  u32 n0, n1, ni0, ni1;
  u64 save;  // EXEC at kernel entry (all lanes of a live workgroup)
  u32 nn0, nn1;  // the words of negN = 2^64 - N (Goldilocks: nn0 = eps = 2^32 - 1)
};

// zr: four words that hold 0 and live in v41/v53/v65/v77 (the high halves of the slots'
// zero-extension pairs) -- operands of every group so that the compiler keeps them there.
// four butterflies (x_a, x_a + 2^R) of stage bit R: the GRP-th four in ascending a, in the
// arithmetic back end ARITH (field64.h); w_k / p_k: twiddle of the k-th and, for ARITH_SHOUP,
// its precomputed companion.  TRIV: the stage's twiddles with (a mod 2^R) == 0 are omega^0
// (lowest step of a transform) and those butterflies multiply by nothing.
template <int ARITH, int MODE, int R, int GRP, bool TRIV> struct BflyGroup;
// x[W GRP + k] *= hi_k * lo_k, k < W, in the back end's domain; W = 4, or 2 for ARITH_SHOUP
// (hp / lp: its precomputed companions -- four words per element)
template <int ARITH, int GRP> struct TwistGroup;
// x[4 GRP + k] *= w_k * 2^-64 (Montgomery product, whatever the back end)
template <int GRP> struct MontGroup;
// the first elements of BflyGroup<., ., R, GRP, .>'s butterflies times s * 2^-64
template <int R, int GRP> struct ScaleGroup;
"""


def main():
    only = sys.argv[1:] or ARITHS
    out = [HEADER]
    for arith in ARITHS:
        if arith not in only:
            continue
        for mode in ('MODE_FWD', 'MODE_INV'):
            for r in range(4):
                for grp in range(2):
                    for triv in (False, True):
                        out.append(gen_bfly_group(arith, mode, r, grp, triv))
        for grp in range(16 // twist_width(arith)):
            out.append(gen_twist_group(arith, grp))
    for grp in range(4):
        out.append(gen_mont_group(grp))
    for r in range(4):
        for grp in range(2):
            out.append(gen_scale_group(r, grp))
    sys.stdout.write('\n'.join(out))


if __name__ == '__main__':
    main()
