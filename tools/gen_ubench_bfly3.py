#!/usr/bin/env python3
"""Generates tools/ubench_bfly3.hip: the radix-16 register step of ubench_bfly2 with W butterflies
interleaved per asm statement (W = 1, 2, 4), to find how much instruction-level parallelism the
EXEC-masked assembly needs at 4 waves per SIMD.  Not part of the product.

    python tools/gen_ubench_bfly3.py > tools/ubench_bfly3.hip
"""
import sys

DATA_BASE = 0          # x_i = v[2i : 2i+1]
TEMP_BASE = 32
TEMP_PER_SLOT = 12
SG_BASE = 60           # carries: slot s uses s[SG_BASE+4s : +1] (borrow) and s[+2 : +3] (carry)


class Slot:
    def __init__(self, s, xa, xb, w0, w1):
        t = TEMP_BASE + TEMP_PER_SLOT * s
        self.m0, self.m1, self.m2, self.h, self.z, self.d = (t, t + 2, t + 4, t + 6, t + 8, t + 10)
        self.e = self.m2
        self.sb = 's[%d:%d]' % (SG_BASE + 4 * s, SG_BASE + 4 * s + 1)
        self.sc = 's[%d:%d]' % (SG_BASE + 4 * s + 2, SG_BASE + 4 * s + 3)
        self.xl, self.xh = DATA_BASE + 2 * xa, DATA_BASE + 2 * xa + 1
        self.yl, self.yh = DATA_BASE + 2 * xb, DATA_BASE + 2 * xb + 1
        self.w0, self.w1 = w0, w1


def pair(r):
    return 'v[%d:%d]' % (r, r + 1)


def mont_core(S, dl, dh):
    """thi -> S.h, g -> S.m2 ; plain list of single instructions (all full-EXEC VALU)."""
    m0, m1, m2, h, z = S.m0, S.m1, S.m2, S.h, S.z
    return [
        'v_mad_u64_u32 %s, vcc, v%d, %s, 0' % (pair(m0), dl, S.w0),
        'v_mov_b32 v%d, v%d' % (z, m0 + 1),
        'v_mad_u64_u32 %s, vcc, v%d, %s, %s' % (pair(m1), dl, S.w1, pair(z)),
        'v_mov_b32 v%d, v%d' % (z, m1 + 1),
        'v_mad_u64_u32 %s, vcc, v%d, %s, %s' % (pair(h), dh, S.w1, pair(z)),
        'v_mov_b32 v%d, v%d' % (z, m1),
        'v_mad_u64_u32 %s, vcc, v%d, %s, %s' % (pair(m2), dh, S.w0, pair(z)),
        'v_mov_b32 v%d, v%d' % (z, m2 + 1),
        'v_mad_u64_u32 %s, vcc, v%d, %%[ni0], 0' % (pair(m1), m0),
        'v_mul_lo_u32 v%d, v%d, %%[ni1]' % (m0 + 1, m0),
        'v_mul_lo_u32 v%d, v%d, %%[ni0]' % (m2, m2),
        'v_lshl_add_u64 %s, %s, 0, %s' % (pair(h), pair(h), pair(z)),
        'v_add3_u32 v%d, v%d, v%d, v%d' % (m1 + 1, m1 + 1, m0 + 1, m2),
        'v_mul_hi_u32 v%d, v%d, %%[n0]' % (z, m1),
        'v_mad_u64_u32 %s, vcc, v%d, %%[n1], %s' % (pair(m0), m1, pair(z)),
        'v_mov_b32 v%d, v%d' % (z, m0 + 1),
        'v_mad_u64_u32 %s, vcc, v%d, %%[n1], %s' % (pair(m2), m1 + 1, pair(z)),
        'v_mov_b32 v%d, v%d' % (z, m0),
        'v_mad_u64_u32 %s, vcc, v%d, %%[n0], %s' % (pair(m0), m1 + 1, pair(z)),
        'v_mov_b32 v%d, v%d' % (z, m0 + 1),
        'v_lshl_add_u64 %s, %s, 0, %s' % (pair(m2), pair(m2), pair(z)),
    ]


def mont_core_chain(S, dl, dh):
    """the shipped form (csrc/gen_stage_asm.py: hi_chain): the two cross products share one 64-bit
    accumulator, its carry (the SGPR carry-out of v_mad_u64_u32) enters the top word with v_addc"""
    m0, m1, m2, h, z = S.m0, S.m1, S.m2, S.h, S.z

    def chain(x0, x1, y0, y1, mid, top, first):
        return first + [
            'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(mid), x0, y1, pair(z)),
            'v_mad_u64_u32 %s, %s, %s, %s, %s' % (pair(mid), S.sc, x1, y0, pair(mid)),
            'v_mov_b32 v%d, v%d' % (z, mid + 1),
            'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(top), x1, y1, pair(z)),
            'NOPGAP',
            'v_addc_co_u32 v%d, vcc, 0, v%d, %s' % (top + 1, top + 1, S.sc),
        ]
    return (chain('v%d' % dl, 'v%d' % dh, S.w0, S.w1, m2, h,
                  ['v_mad_u64_u32 %s, vcc, v%d, %s, 0' % (pair(m0), dl, S.w0),
                   'v_mov_b32 v%d, v%d' % (z, m0 + 1)]) +
            ['v_mad_u64_u32 %s, vcc, v%d, %%[ni0], 0' % (pair(m1), m0),
             'v_mul_lo_u32 v%d, v%d, %%[ni1]' % (m0 + 1, m0),
             'v_mul_lo_u32 v%d, v%d, %%[ni0]' % (m2, m2),
             'v_add3_u32 v%d, v%d, v%d, v%d' % (m1 + 1, m1 + 1, m0 + 1, m2)] +
            chain('v%d' % m1, 'v%d' % (m1 + 1), '%[n0]', '%[n1]', m0, m2,
                  ['v_mul_hi_u32 v%d, v%d, %%[n0]' % (z, m1)]))


def mont_core_prime(S, dl, dh):
    """VERDICT r02 item 5 (r01 item 1b): hi64(q*N) from the shape of the BASELINE prime instead of a
    64 x 64 product.  N = 2^64 - k*2^31 + 1 (k = 1827), so q*N = q*2^64 - (E << 31) + q with E = q*k
    (75 bits: two v_mad_u64_u32 instead of v_mul_hi + three v_mad), and with
    A = E >> 33, B = (E << 31) mod 2^64:   hi64(q*N) = q - A - [B > q].
    Costs two multiplier instructions less and four shift/compare/subtract instructions more than the
    generic chain (9 + 1 v_mov against 5 + 1 v_mov).  q itself stays generic: from the shape of
    N^-1 = 1 + k*2^31 + 2^62 it would be one multiplier instruction + 4 slow + 5 fast against
    3 multiplier + 1 slow (gen_stage_asm.py: mont)."""
    m0, m1, m2, h, z, d = S.m0, S.m1, S.m2, S.h, S.z, S.d

    def chain(x0, x1, y0, y1, mid, top, first):
        return first + [
            'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(mid), x0, y1, pair(z)),
            'v_mad_u64_u32 %s, %s, %s, %s, %s' % (pair(mid), S.sc, x1, y0, pair(mid)),
            'v_mov_b32 v%d, v%d' % (z, mid + 1),
            'v_mad_u64_u32 %s, vcc, %s, %s, %s' % (pair(top), x1, y1, pair(z)),
            'NOPGAP',
            'v_addc_co_u32 v%d, vcc, 0, v%d, %s' % (top + 1, top + 1, S.sc),
        ]
    return (chain('v%d' % dl, 'v%d' % dh, S.w0, S.w1, m2, h,
                  ['v_mad_u64_u32 %s, vcc, v%d, %s, 0' % (pair(m0), dl, S.w0),
                   'v_mov_b32 v%d, v%d' % (z, m0 + 1)]) +
            ['v_mad_u64_u32 %s, vcc, v%d, %%[ni0], 0' % (pair(m1), m0),
             'v_mul_lo_u32 v%d, v%d, %%[ni1]' % (m0 + 1, m0),
             'v_mul_lo_u32 v%d, v%d, %%[ni0]' % (m2, m2),
             'v_add3_u32 v%d, v%d, v%d, v%d' % (m1 + 1, m1 + 1, m0 + 1, m2),
             # E = q * k: (e0, e1, e2) = (m0, m2, m2+1)
             'v_mad_u64_u32 %s, vcc, v%d, %%[k], 0' % (pair(m0), m1),
             'v_mov_b32 v%d, v%d' % (z, m0 + 1),
             'v_mad_u64_u32 %s, vcc, v%d, %%[k], %s' % (pair(m2), m1 + 1, pair(z)),
             # B = (E << 31) mod 2^64 -> d pair (the multiplicand is dead by now)
             'v_lshlrev_b32 v%d, 31, v%d' % (d, m0),
             'v_alignbit_b32 v%d, v%d, v%d, 1' % (d + 1, m2, m0),
             # A = E >> 33 -> m0 pair
             'v_alignbit_b32 v%d, v%d, v%d, 1' % (m0, m2 + 1, m2),
             'v_lshrrev_b32 v%d, 1, v%d' % (m0 + 1, m2 + 1),
             'v_cmp_gt_u64 %s, %s, %s' % (S.sb, pair(d), pair(m1)),
             'NOPGAP',
             'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (m2, S.sc, m1, m0, S.sb),
             'NOPGAP',
             'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (m2 + 1, S.sc, m1 + 1, m0 + 1, S.sc)])


import os
FIXMODE = os.environ.get('FIXMODE', 'exec')
MONT = os.environ.get('MONT', 'chain')
mont_core_movs = mont_core
if MONT == 'chain':  # MONT=movs: the r02a form with seven zero-extending v_mov
    mont_core = mont_core_chain
if MONT == 'prime':
    mont_core = mont_core_prime
def fix(mask_expr, reg):
    if FIXMODE == 'none':
        return []
    if FIXMODE == 'uncond':
        return ['v_lshl_add_u64 %s, %s, 0, %%[N]' % (pair(reg), pair(reg))]
    return [mask_expr, 'v_lshl_add_u64 %s, %s, 0, %%[N]' % (pair(reg), pair(reg))]


def bfly_fwd(S, core=None):
    """list of items; an item is a str (one instruction) or a tuple ('fix', [instrs]) that must
    run with EXEC narrowed and is followed by a restore."""
    core = core or mont_core
    x, y = S.xl, S.yl
    items = [
        'v_mov_b32 v%d, 0' % (S.z + 1),
        'v_sub_co_u32 v%d, %s, v%d, v%d' % (S.d, S.sb, S.xl, S.yl),
        'v_lshl_add_u64 %s, %s, 0, %%[negN]' % (pair(S.e), pair(y)),
        'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (S.d + 1, S.sb, S.xh, S.yh, S.sb),
        'v_add_co_u32 v%d, %s, v%d, v%d' % (S.xl, S.sc, S.xl, S.e),
        ('fix', fix('s_mov_b64 exec, %s' % S.sb, S.d)),
        'v_addc_co_u32 v%d, %s, v%d, v%d, %s' % (S.xh, S.sc, S.xh, S.e + 1, S.sc),
        'NOPGAP',
        ('fix', fix('s_andn2_b64 exec, %%[save], %s' % S.sc, x)),
    ]
    items += core(S, S.d, S.d + 1)
    items += [
        'v_sub_co_u32 v%d, %s, v%d, v%d' % (S.yl, S.sb, S.h, S.m2),
        'NOPGAP',
        'v_subb_co_u32 v%d, %s, v%d, v%d, %s' % (S.yh, S.sb, S.h + 1, S.m2 + 1, S.sb),
        'NOPGAP',
        ('fix', fix('s_mov_b64 exec, %s' % S.sb, y)),
    ]
    return items


def interleave(streams):
    """round-robin merge; consecutive fix groups share one EXEC restore."""
    out = []
    idx = [0] * len(streams)
    in_fix = False
    W = len(streams)
    while any(i < len(s) for i, s in zip(idx, streams)):
        for k, s in enumerate(streams):
            if idx[k] >= len(s):
                continue
            it = s[idx[k]]
            idx[k] += 1
            if it == 'NOPGAP':
                if W == 1:
                    if in_fix:
                        out.append('s_mov_b64 exec, %[save]')
                        in_fix = False
                    out.append('s_nop 0')
                continue
            if isinstance(it, tuple):
                out.extend(it[1])
                in_fix = True
            else:
                if in_fix:
                    out.append('s_mov_b64 exec, %[save]')
                    in_fix = False
                out.append(it)
    if in_fix:
        out.append('s_mov_b64 exec, %[save]')
    return out


def asm_stmt(pairs, wnames, W, core=None):
    """one asm statement for W butterflies; pairs = [(a, b)], wnames = C expressions"""
    slots = []
    ops_in = []
    for s, ((a, b), wn) in enumerate(zip(pairs, wnames)):
        slots.append(Slot(s, a, b, '%%[w%d0]' % s, '%%[w%d1]' % s))
        ops_in.append('[w%d0] "v"((u32)(%s))' % (s, wn))
        ops_in.append('[w%d1] "v"((u32)((%s) >> 32))' % (s, wn))
    body = interleave([bfly_fwd(S, core) for S in slots])
    text = ''.join('      "%s\\n\\t"\n' % l for l in body)
    outs = []
    for (a, b) in pairs:
        outs.append('"+{v[%d:%d]}"(x%d)' % (2 * a, 2 * a + 1, a))
        outs.append('"+{v[%d:%d]}"(x%d)' % (2 * b, 2 * b + 1, b))
    consts = ['[N] "s"(f.N)', '[negN] "s"(f.negN)', '[n0] "s"((u32)f.N)', '[n1] "s"((u32)(f.N >> 32))',
              '[ni0] "s"((u32)f.Ninv)', '[ni1] "s"((u32)(f.Ninv >> 32))', '[save] "s"(save)']
    if core is mont_core_prime:
        consts.append('[k] "s"((u32)((0ull - f.N + 1ull) >> 31))')  # N = 2^64 - k*2^31 + 1
    clob = ['"vcc"', '"scc"']
    clob += ['"v%d"' % r for r in range(TEMP_BASE, TEMP_BASE + TEMP_PER_SLOT * W)]
    clob += ['"s%d"' % r for r in range(SG_BASE, SG_BASE + 4 * W)]
    return ('    asm volatile(\n%s      : %s\n      : %s\n      : %s);\n'
            % (text, ', '.join(outs), ', '.join(ops_in + consts), ', '.join(clob)))


def kernel(W, core=None, name=None):
    lines = ['__global__ __launch_bounds__(512) void %s(u64 *out, Field f, u64 seed) {' % (name or 'k_asm_w%d' % W),
             '  u64 x[16], w[8];', '  init(x, w, f, seed);']
    lines.append('  u64 ' + ', '.join('x%d = x[%d]' % (i, i) for i in range(16)) + ';')
    lines.append('  const u64 save = __builtin_amdgcn_read_exec();')
    lines.append('  for (int it = 0; it < ITER; ++it) {')
    for r in (3, 2, 1, 0):
        bf = [(v, v + (1 << r)) for v in range(16) if not v & (1 << r)]
        for g in range(0, 8, W):
            lines.append(asm_stmt(bf[g:g + W], ['w[%d]' % (g + k) for k in range(W)], W, core))
    lines.append('  }')
    lines.append('  u64 *o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;')
    lines.append('  ' + ' '.join('o[%d] = x%d;' % (i, i) for i in range(16)))
    lines.append('}')
    return '\n'.join(lines)


def main():
    src = open(__file__.replace('gen_ubench_bfly3.py', 'ubench_bfly2.hip')).read()
    head = src.split('// ---- variant 1')[0]
    tail = src[src.index('template <class K> static int timeit'):src.index('int main()')]
    print('// GENERATED by tools/gen_ubench_bfly3.py -- do not edit.  W-way interleaved assembly butterflies.')
    print(head)
    if MONT == 'both':
        # the shipped chain against the BASELINE-prime form of hi64(q*N), 4 butterflies per statement, single
        # launches and ~1 s of back-to-back launches (the chip at its power limit, as in the transforms)
        print(kernel(4, mont_core_chain, 'k_asm_chain_w4'))
        print(kernel(4, mont_core_prime, 'k_asm_prime_w4'))
        print(tail)
        print('''template <class K> static int sustained(const char *name, K kernel, u64 *d, int blocksPerCU, size_t lds) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const Field f{0xfffffc6e80000001ull, 0x4000039180000001ull, 0ull - 0xfffffc6e80000001ull};
  const int blocks = 256 * blocksPerCU;
  for (int i = 0; i < 3000; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), lds, 0, d, f, 0x1234567ull);
  CHECK(hipEventRecord(e0));
  const int reps = 1000;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(512), lds, 0, d, f, 0x1234567ull);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double cyc = ms / reps * 1e-3 * 2.4e9 / ((double)blocksPerCU * 2 * 32 * ITER);
  printf("%-28s blocks/CU=%d lds=%3zuK sustained (3000 + 1000 launches back to back) %7.4f ms/launch  %7.2f cyc/butterfly/SIMD (2.4GHz nominal)\\n",
         name, blocksPerCU, lds >> 10, ms / reps, cyc);
  return 0;
}
int main() {
  u64 *d; CHECK(hipMalloc(&d, (size_t)256 * 4 * 512 * 16 * 8));
  std::vector<u64> ref, got;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t lds = pass == 0 ? 64 * 1024 : 0;
    const int bpc = pass == 0 ? 2 : 4;
    if (timeit("hipcc C++ (cndmask)", k_cxx, d, bpc, lds, &ref)) return 1;
#define RUN(K) if (timeit(#K, K, d, bpc, lds, &got)) return 1; { size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i]; printf("   vs C++: %zu mismatches of %zu\\n", bad, ref.size()); }
    RUN(k_asm_chain_w4) RUN(k_asm_prime_w4)
    for (int round = 0; round < 2; ++round) {
      if (sustained("k_asm_chain_w4", k_asm_chain_w4, d, bpc, lds)) return 1;
      if (sustained("k_asm_prime_w4", k_asm_prime_w4, d, bpc, lds)) return 1;
    }
  }
  return 0;
}''')
        return
    for W in (1, 2, 4):
        print(kernel(W))
        print()
    print(tail)
    print('''int main() {
  u64 *d; CHECK(hipMalloc(&d, (size_t)256 * 4 * 512 * 16 * 8));
  std::vector<u64> ref, got;
  for (int pass = 0; pass < 2; ++pass) {
    const size_t lds = pass == 0 ? 64 * 1024 : 0;
    const int bpc = pass == 0 ? 2 : 4;
    if (timeit("hipcc C++ (cndmask)", k_cxx, d, bpc, lds, &ref)) return 1;
#define RUN(K) if (timeit(#K, K, d, bpc, lds, &got)) return 1; { size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += ref[i] != got[i]; printf("   vs C++: %zu mismatches of %zu\\n", bad, ref.size()); }
    RUN(k_asm_w1) RUN(k_asm_w2) RUN(k_asm_w4)
  }
  return 0;
}''')


if __name__ == '__main__':
    main()
