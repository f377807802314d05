#!/usr/bin/env python3
"""Generator of the software-pipelined main loops of the head_dim-128 flash-attention BACKWARD (csrc/attn_bwd128_asm.inc, included by
attention128_bwd.hip): the dK/dV pass (one key per lane, query tiles streamed) and the dQ pass (one query per lane, key tiles streamed) of
FLUX.1 / Qwen-Image / Wan.  The schedule is gen_attn_bwd64.py's carried to 128 the way attention128_bwd.hip carries the kernels:

    body(h) =   B(h-1)   ||   V(h)   ||   A(h+1)          one MFMA, then the fillers of its gap, then the next MFMA ...

per 32-row half h:  A = the S and dP chains (8 k-steps each: 16 MFMAs),  V = 16 v_exp + mask + 8 v_pk_mul + 8..16 v_cvt_pk,  B = the second
products (dV^T, dK^T: 16 MFMAs; dQ^T: 8).  These kernels run ONE wave per SIMD (512 registers: 128 / 64 accumulator registers), so nothing
but the wave's own instruction order can put VALU work under MFMAs: the round-4 kernels issue both halves' chains back to back and overlap
half of the exp / pack arithmetic; here every MFMA gap of a body carries fillers of another half.  Same MFMAs, operand order and accumulation
order per output element as the round-4 kernels, same masks where they can act (the last tile: queries >= S / keys >= the sample's key count
get P = 0; the loop's other tiles run a copy of the body without the 32 mask instructions per half): BIT-IDENTICAL
(tests/test_gpu_flux_backward.py compares the two).

Registers (fixed).  S / dP of consecutive halves alternate between X and Y:
    both passes  v[16:47] X (s | dp)   v[48:79] Y   v[80:95] packed P | dZ (dQ pass: dZ only)   v[96:159] A fragments (first operand k-steps
                 0..7 | second operand k-steps 0..7)   v[224:228] derived read addresses   s[36:67] the half's 16 lane masks
    dK/dV pass   v[160:223] transposed fragments (d half 0 | d half 1: [dO db0 | dO db1 | Q db0 | Q db1] x (k-step 0 | k-step 1))
                 a[0:63] dV^T blocks 0..3, a[64:127] dK^T blocks 0..3, a[128:159] / a[160:191] this lane's key / value row fragments
    dQ pass      v[160:191] transposed K fragments, v[192:223] -L | -Delta splats (C operands)
                 a[0:63] dQ^T blocks 0..3, a[64:95] / a[96:127] this lane's q~ / dO row fragments
LDS: a [64][128] operand tile is two [64][64] sub-tiles (8 KiB each, 128-byte rows, the head_dim-64 swizzle); stage = first operand (16 KiB) |
second operand (16 KiB) [| -L | -Delta (512 B)]; 4-slot ring, three tiles ahead, ONE `vmcnt` (the loop always issues a tile's loads; past the
end it re-loads the last tile into a slot nobody reads).

usage: python gen_attn_bwd128.py > attn_bwd128_asm.inc
"""

X, Y, PZ, AF, TR = 16, 48, 80, 96, 160
NLD = 192
DR = 224                      # derived addresses: v224..226 = r0 ^ 32 kk (kk = 1..3), v227 / v228 = a0 / a1 ^ 64
MK = 36                       # s[36:67]: 16 lane masks of the half V works on
S_B0, S_B1, S_B2 = 70, 72, 74
S_NT, S_CNT, S_LT, S_LDSL, S_STA, S_STT, S_D, S_M0, S_EX, S_W1K, S_WNL, S_INC, S_T = 76, 77, 78, 79, 80, 81, 82, 83, 84, 86, 87, 88, 89
S_B0H, S_B1H = 90, 92         # the same bases + 8192 bytes (rows + 32)
SUB, TILE = 8192, 16384


def vr(b, n=1):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=1):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


def sr(b):
    return f"s[{b}:{b + 1}]"


class Pass:
    pass


def raddr(kk):
    return "%[r0]" if kk == 0 else vr(DR + kk - 1)


def taddr(i):
    return f"%[a{i}]" if i < 2 else vr(DR + 3 + i - 2)


def masks(qb):
    """{gap: [v_cmp]}: lane mask of element r = (this lane's remaining valid rows) > 32 qb + 16 (r >> 3) + (r & 7)"""
    return {r: [f"v_cmp_lt_i32_e64 {sr(MK + 2 * r)}, {32 * qb + 16 * (r >> 3) + (r & 7)}, %[vrem]"] for r in range(16)}


def dkv():
    P = Pass()
    P.name, P.ST, P.nloads, P.nacc, P.kv = "DKV", 2 * TILE + 512, 9, 128, 128

    def mf_A(n):
        out = []
        for j in range(8):
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n, 16)}, {vr(AF + 4 * j, 4)}, {ar(P.kv + 4 * j, 4)}, {vr(n, 16)}")
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n + 16, 16)}, {vr(AF + 32 + 4 * j, 4)}, {ar(P.kv + 32 + 4 * j, 4)}, {vr(n + 16, 16)}")
        return out

    def mf_B():
        """k-step 0 of both d halves first (they read pf0 / zf0), then k-step 1: the packed registers of the NEXT half may be written behind m7 / m15"""
        out = []
        for hs in range(2):
            pf, zf = vr(PZ + 4 * hs, 4), vr(PZ + 8 + 4 * hs, 4)
            for dh in range(2):
                t = TR + 32 * dh + 16 * hs
                dv0, dv1, dk0, dk1 = ar(32 * dh, 16), ar(32 * dh + 16, 16), ar(64 + 32 * dh, 16), ar(64 + 32 * dh + 16, 16)
                out += [f"v_mfma_f32_32x32x16_bf16 {dv0}, {vr(t, 4)}, {pf}, {dv0}", f"v_mfma_f32_32x32x16_bf16 {dk0}, {vr(t + 8, 4)}, {zf}, {dk0}",
                        f"v_mfma_f32_32x32x16_bf16 {dv1}, {vr(t + 4, 4)}, {pf}, {dv1}", f"v_mfma_f32_32x32x16_bf16 {dk1}, {vr(t + 12, 4)}, {zf}, {dk1}"]
        return out

    def rd_A(n, qb):
        out = []
        for i, o in enumerate((0, 16, 64, 80)):
            out.append(f"ds_read_b128 {vr(n + 4 * i, 4)}, %[la] offset:{o + 128 * qb}")
        for i, o in enumerate((0, 16, 64, 80)):
            out.append(f"ds_read_b128 {vr(n + 16 + 4 * i, 4)}, %[la] offset:{256 + o + 128 * qb}")
        for j in range(8):
            off = (j >> 2) * SUB + 4096 * qb
            out.append(f"ds_read_b128 {vr(AF + 4 * j, 4)}, {raddr(j & 3)} offset:{off}")
            out.append(f"ds_read_b128 {vr(AF + 32 + 4 * j, 4)}, {raddr(j & 3)} offset:{TILE + off}")
        return out

    def rd_T(qb):
        out = []
        for hs in range(2):
            for dh in range(2):
                for (blk, off) in ((0, TILE + dh * SUB), (8, dh * SUB)):       # dO sub-tile, Q sub-tile of this d half
                    for i in range(4):
                        out.append(f"ds_read_b64_tr_b16 {vr(TR + 32 * dh + 16 * hs + blk + 2 * i, 2)}, {taddr(i)} offset:{off + 2048 * hs + 4096 * qb}")
        return out

    def valu(c, qb, masked=True):
        g = {}
        def put(gap, ins):
            g.setdefault(gap, []).append(ins)
        if masked:
            for r, lst in masks(qb).items():
                put(r, lst[0])
        for r in range(16):
            put(r + 1, f"v_exp_f32 {vr(c + r)}, {vr(c + r)}")
            if masked:
                put(r + 3, f"v_cndmask_b32_e64 {vr(c + r)}, 0, {vr(c + r)}, {sr(MK + 2 * r)}")
        for j in range(8):
            put(2 * j + 6, f"v_pk_mul_f32 {vr(c + 16 + 2 * j, 2)}, {vr(c + 16 + 2 * j, 2)}, {vr(c + 2 * j, 2)}")
            lo = 9 if j < 4 else 17                      # behind the MFMAs of B that read the old packed values (m0..m7 / m8..m15)
            put(max(2 * j + 6, lo), f"v_cvt_pk_bf16_f32 {vr(PZ + j)}, {vr(c + 2 * j)}, {vr(c + 2 * j + 1)}")
            put(max(2 * j + 7, lo), f"v_cvt_pk_bf16_f32 {vr(PZ + 8 + j)}, {vr(c + 16 + 2 * j)}, {vr(c + 16 + 2 * j + 1)}")
        return g

    P.mf_A, P.mf_B, P.rd_A, P.rd_T, P.valu = mf_A, mf_B, rd_A, rd_T, valu
    P.nB, P.nA = 16, 16
    P.rdA_gaps = [g for g in range(12) for _ in range(2)]            # 24 reads, two per gap: C operands first, fragment k-step 7 last
    P.rdT_gaps = [16 + i // 3 for i in range(32)]                    # 32 reads from the gap behind B's last MFMA (16..26)
    P.a_addrs, P.t_addrs = ["la", "r0"], ["a0", "a1"]

    def stage():
        out = []
        first, prev = True, 0
        for (lo, hi) in ((S_B0, S_B0H), (S_B1, S_B1H)):
            # sub-tile 0 rows w.., rows w + 32..; sub-tile 1 (d 64..127: + 128 bytes).  The instruction offset of an LDS-DMA load is added to the
            # global AND to the LDS address: M0 carries the destination minus it
            for (base, off) in ((lo, 0), (hi, 0), (lo, 128), (hi, 128)):
                out += [f"s_add_i32 m0, s{S_W1K}, s{S_LDSL}" if first else f"s_add_i32 m0, m0, {4096 - off + prev}", "s_nop 0",
                        f"global_load_lds_dwordx4 %[g0], s[{base}:{base + 1}]" + (f" offset:{off}" if off else "")]
                first, prev = False, off
        out += [f"s_add_i32 m0, s{S_WNL}, s{S_LDSL}", f"s_mov_b64 s[{S_EX}:{S_EX + 1}], exec", "s_mov_b64 exec, 0xff",
                f"global_load_lds_dwordx4 %[g2], s[{S_B2}:{S_B2 + 1}]", f"s_mov_b64 exec, s[{S_EX}:{S_EX + 1}]"]
        return out
    P.stage = stage
    P.adv = [(S_B0, 14), (S_B1, 14), (S_B2, 9), (S_B0H, 14), (S_B1H, 14)]
    return P


def dq():
    P = Pass()
    P.name, P.ST, P.nloads, P.nacc, P.kv = "DQ", 2 * TILE, 8, 64, 64

    def mf_A(n):
        out = []
        for j in range(8):
            cs = vr(NLD, 16) if j == 0 else vr(n, 16)
            cd = vr(NLD + 16, 16) if j == 0 else vr(n + 16, 16)
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n, 16)}, {vr(AF + 4 * j, 4)}, {ar(P.kv + 4 * j, 4)}, {cs}")
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n + 16, 16)}, {vr(AF + 32 + 4 * j, 4)}, {ar(P.kv + 32 + 4 * j, 4)}, {cd}")
        return out

    def mf_B():
        out = []
        for hs in range(2):
            zf = vr(PZ + 4 * hs, 4)
            for dh in range(2):
                t = TR + 16 * dh + 8 * hs
                out += [f"v_mfma_f32_32x32x16_bf16 {ar(32 * dh, 16)}, {vr(t, 4)}, {zf}, {ar(32 * dh, 16)}",
                        f"v_mfma_f32_32x32x16_bf16 {ar(32 * dh + 16, 16)}, {vr(t + 4, 4)}, {zf}, {ar(32 * dh + 16, 16)}"]
        return out

    def rd_A(n, kb):
        out = []
        for j in range(8):
            off = (j >> 2) * SUB + 4096 * kb
            out.append(f"ds_read_b128 {vr(AF + 4 * j, 4)}, {raddr(j & 3)} offset:{off}")
            out.append(f"ds_read_b128 {vr(AF + 32 + 4 * j, 4)}, {raddr(j & 3)} offset:{TILE + off}")
        return out

    def rd_T(kb):
        out = []
        for hs in range(2):
            for dh in range(2):
                for i in range(4):
                    out.append(f"ds_read_b64_tr_b16 {vr(TR + 16 * dh + 8 * hs + 2 * i, 2)}, {taddr(i)} offset:{dh * SUB + 2048 * hs + 4096 * kb}")
        return out

    def valu(c, kb, masked=True):
        g = {}
        def put(gap, ins):
            g.setdefault(gap, []).append(ins)
        if masked:
            for r, lst in masks(kb).items():
                put(r, lst[0])
        for r in range(16):
            put(r + 1, f"v_exp_f32 {vr(c + r)}, {vr(c + r)}")
            if masked:
                put(r + 3, f"v_cndmask_b32_e64 {vr(c + r)}, 0, {vr(c + r)}, {sr(MK + 2 * r)}")
        for j in range(8):
            put(2 * j + 6, f"v_pk_mul_f32 {vr(c + 16 + 2 * j, 2)}, {vr(c + 16 + 2 * j, 2)}, {vr(c + 2 * j, 2)}")
            put(max(2 * j + 7, 5 if j < 4 else 9), f"v_cvt_pk_bf16_f32 {vr(PZ + j)}, {vr(c + 16 + 2 * j)}, {vr(c + 16 + 2 * j + 1)}")
        return g

    P.mf_A, P.mf_B, P.rd_A, P.rd_T, P.valu = mf_A, mf_B, rd_A, rd_T, valu
    P.nB, P.nA = 8, 16
    P.rdA_gaps = [0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5]        # (two MFMAs of flight before the wait in front of m8)
    P.rdT_gaps = [8 + i // 2 for i in range(16)]
    P.a_addrs, P.t_addrs = ["r0"], ["a0", "a1"]

    def stage():
        out = []
        first, prev = True, 0
        for (lo, hi) in ((S_B0, S_B0H), (S_B1, S_B1H)):
            for (base, off) in ((lo, 0), (hi, 0), (lo, 128), (hi, 128)):       # (M0 = destination - instruction offset: see the dK/dV pass)
                out += [f"s_add_i32 m0, s{S_W1K}, s{S_LDSL}" if first else f"s_add_i32 m0, m0, {4096 - off + prev}", "s_nop 0",
                        f"global_load_lds_dwordx4 %[g0], s[{base}:{base + 1}]" + (f" offset:{off}" if off else "")]
                first, prev = False, off
        return out
    P.stage = stage
    P.adv = [(S_B0, 14), (S_B1, 14), (S_B0H, 14), (S_B1H, 14)]
    return P


def derive_A():
    return [f"v_xor_b32 {vr(DR + k - 1)}, {32 * k}, %[r0]" for k in (1, 2, 3)]


def derive_T():
    return [f"v_xor_b32 {vr(DR + 3)}, 64, %[a0]", f"v_xor_b32 {vr(DR + 4)}, 64, %[a1]"]


def ring_step(P, idx_reg, addrs, derive, extra=()):
    out = [f"s_add_i32 s{idx_reg}, s{idx_reg}, 1", f"s_mov_b32 s{S_D}, {P.ST}", f"s_cmp_eq_u32 s{idx_reg}, 4",
           f"s_cselect_b32 s{S_D}, {-3 * P.ST}, s{S_D}", f"s_cselect_b32 s{idx_reg}, 0, s{idx_reg}"]
    out += [f"v_add_u32 %[{a}], s{S_D}, %[{a}]" for a in addrs]
    return out + derive + list(extra)


def sync(P):
    out = [f"s_waitcnt vmcnt({P.nloads})", "s_barrier"] + P.stage()
    out += [f"s_cmp_lt_u32 s{S_LT}, s{S_NT}", f"s_cselect_b32 s{S_INC}, 1, 0", f"s_add_u32 s{S_LT}, s{S_LT}, s{S_INC}"]     # S_NT holds nt - 1
    for (b, sh) in P.adv:
        out += [f"s_lshl_b32 s{S_T}, s{S_INC}, {sh}", f"s_add_u32 s{b}, s{b}, s{S_T}", f"s_addc_u32 s{b + 1}, s{b + 1}, 0"]
    out += [f"s_add_i32 s{S_LDSL}, s{S_LDSL}, {P.ST}", f"s_cmp_eq_u32 s{S_LDSL}, {4 * P.ST}", f"s_cselect_b32 s{S_LDSL}, 0, s{S_LDSL}"]
    return out


def body(P, cur, nxt, has_B, has_A, qb_cur, qb_A, qb_T, pre=(), masked=True):
    """B(h-1) || V(h) on buffer cur (half qb_cur of its tile: the mask constants) || A(h+1) into nxt (reads: half qb_A of the A-side tile);
    transposed reads of half h (qb_T = qb_cur)."""
    fill = {}
    def put(gap, ins):
        fill.setdefault(gap, []).append(ins)
    if has_A:
        for gp, ins in zip(P.rdA_gaps, P.rd_A(nxt, qb_A)):
            put(gp, ins)
    for gp, lst in sorted(P.valu(cur, qb_cur, masked).items()):
        for ins in lst:
            put(gp, ins)
    for gp, ins in zip(P.rdT_gaps, P.rd_T(qb_T)):
        put(gp, ins)
    out = list(pre)
    out.append("s_waitcnt lgkmcnt(0)")                    # the transposed fragments of B(h-1) (read during the previous body)
    mfB, mfA = P.mf_B(), P.mf_A(nxt)
    ngap = P.nB + P.nA
    for g in range(ngap):
        is_B = g < P.nB
        if g == P.nB:
            out.append("s_waitcnt lgkmcnt(0)")            # A(h+1)'s fragments and C operands (read in this body's first gaps)
        if (has_B if is_B else has_A):
            out.append(mfB[g] if is_B else mfA[g - P.nB])
        out += fill.get(g, [])
    for g in sorted(k for k in fill if k >= ngap):
        out += fill[g]
    return out


def main_loop(P):
    L = []
    L.append(f"s_mov_b32 s{S_M0}, m0")
    L += [f"s_mov_b64 s[{S_B0}:{S_B0 + 1}], %[b0]", f"s_mov_b64 s[{S_B1}:{S_B1 + 1}], %[b1]"]
    if P.name == "DKV":
        L.append(f"s_mov_b64 s[{S_B2}:{S_B2 + 1}], %[b2]")
    for (lo, hi) in ((S_B0, S_B0H), (S_B1, S_B1H)):
        L += [f"s_add_u32 s{hi}, s{lo}, 8192", f"s_addc_u32 s{hi + 1}, s{lo + 1}, 0"]
    L += [f"s_sub_u32 s{S_NT}, %[nt], 1", f"s_mov_b32 s{S_CNT}, s{S_NT}", f"s_min_u32 s{S_LT}, s{S_NT}, 3", f"s_mov_b32 s{S_LDSL}, {3 * P.ST}",
          f"s_mov_b32 s{S_STA}, 0", f"s_mov_b32 s{S_STT}, 0", f"s_lshl_b32 s{S_W1K}, %[wv], 10"]
    if P.name == "DKV":
        L += [f"s_lshl_b32 s{S_WNL}, %[wv], 7", f"s_add_u32 s{S_WNL}, s{S_WNL}, {2 * TILE}"]
    for i in range(P.nacc):
        L.append(f"v_accvgpr_write_b32 {ar(i)}, 0")
    # this lane's rows (MFMA B operands of the first products: 2 x 8 fragments) into AGPRs through the fragment registers
    for i in range(8):
        L.append(f"global_load_dwordx4 {vr(AF + 4 * i, 4)}, %[grow], %[p0] offset:{32 * i}")
        L.append(f"global_load_dwordx4 {vr(AF + 32 + 4 * i, 4)}, %[grow], %[p1] offset:{32 * i}")
    L.append("s_waitcnt vmcnt(0)")                        # (also tiles 0..2, staged by the shell: the barrier below publishes tile 0)
    for i in range(64):
        L.append(f"v_accvgpr_write_b32 {ar(P.kv + i)}, {vr(AF + i)}")
    if P.name == "DQ":
        for i in range(16):
            L += [f"v_mov_b32 {vr(NLD + i)}, %[nl]", f"v_mov_b32 {vr(NLD + 16 + i)}, %[nd]"]
    L += derive_A() + derive_T()
    L.append("s_barrier")
    # prologue: A(0) alone, then body(0) without B
    L += P.rd_A(X, 0)
    L.append("s_waitcnt lgkmcnt(0)")
    L += P.mf_A(X)
    L += ["s_nop 15", "s_nop 15"]
    L += body(P, X, Y, False, True, 0, 1, 0)
    # nt - 1 iterations: sync(t + 1) | body(2t + 1) on Y (A-side addresses -> tile t + 1) | body(2t + 2) on X (transposed side, masks -> tile t + 1)
    # The masks (32 VALU instructions per half) only matter in the LAST tile -- rows beyond S / keys beyond the sample's key count -- (a lane whose
    # own key is masked in the dK/dV pass is zeroed by the shell at the end): nt - 2 unmasked iterations, then one whose second body is masked.
    L.append(f"s_cmp_eq_u32 s{S_CNT}, 0")
    L.append(f"s_cbranch_scc1 L_{P.name}128_tail%=")
    L.append(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    L.append(f"s_cmp_eq_u32 s{S_CNT}, 0")
    L.append(f"s_cbranch_scc1 L_{P.name}128_last%=")
    L.append(f"L_{P.name}128_loop%=:")
    L += body(P, Y, X, True, True, 1, 0, 1, pre=sync(P) + ring_step(P, S_STA, P.a_addrs, derive_A()), masked=False)
    L += body(P, X, Y, True, True, 0, 1, 0, pre=ring_step(P, S_STT, P.t_addrs, derive_T(), extra=["v_subrev_u32 %[vrem], 64, %[vrem]"]), masked=False)
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", f"s_cbranch_scc1 L_{P.name}128_loop%="]
    L.append(f"L_{P.name}128_last%=:")
    L += body(P, Y, X, True, True, 1, 0, 1, pre=sync(P) + ring_step(P, S_STA, P.a_addrs, derive_A()), masked=False)
    L += body(P, X, Y, True, True, 0, 1, 0, pre=ring_step(P, S_STT, P.t_addrs, derive_T(), extra=["v_subrev_u32 %[vrem], 64, %[vrem]"]))
    L.append(f"L_{P.name}128_tail%=:")
    L += body(P, Y, X, True, False, 1, 0, 1)
    L += ["s_waitcnt lgkmcnt(0)", "s_nop 7"]
    L += P.mf_B()
    L += [f"s_mov_b32 m0, s{S_M0}", "s_waitcnt vmcnt(0)", "s_nop 15", "s_nop 15"]
    return L


def emit(name, lines):
    print(f"#define {name} \\")
    for ln in lines:
        print(f'    "{ln}\\n" \\')
    print('    ""')
    print()


def check(P):
    """structural checks on a steady-state body (see gen_attn_bwd64.py)"""
    import re
    b = body(P, Y, X, True, True, 1, 0, 1)
    assert sum(1 for ins in b if ins.startswith("v_mfma")) == P.nA + P.nB
    npk = 8 if P.name == "DQ" else 16
    for reg in range(PZ, PZ + npk):
        w = [i for i, ins in enumerate(b) if ins.startswith(f"v_cvt_pk_bf16_f32 v{reg},")]
        assert len(w) == 1, (P.name, reg, w)
        lo = PZ + 4 * ((reg - PZ) // 4)
        readers = [i for i, ins in enumerate(b) if ins.startswith("v_mfma") and f", v[{lo}:{lo + 3}], a[" in ins]
        assert len(readers) == 4 and all(r < w[0] for r in readers), (P.name, reg, readers, w)
    for r in range(16):
        e = [i for i, ins in enumerate(b) if ins == f"v_exp_f32 v{Y + r}, v{Y + r}"]
        k = [i for i, ins in enumerate(b) if ins.startswith(f"v_cmp_lt_i32_e64 s[{MK + 2 * r}:")]
        c = [i for i, ins in enumerate(b) if ins.startswith(f"v_cndmask_b32_e64 v{Y + r},")]
        assert len(e) == 1 and len(k) == 1 and len(c) == 1 and k[0] + 2 < c[0] and e[0] + 1 < c[0], (P.name, r, e, k, c)
        pair = f"v[{Y + (r & ~1)}:{Y + (r & ~1) + 1}]"
        users = [i for i, ins in enumerate(b) if (ins.startswith("v_pk_mul") and ins.endswith(pair)) or
                 (ins.startswith("v_cvt_pk") and re.search(rf", v{Y + r}(,|$)", ins))]
        assert len(users) == (1 if P.name == "DQ" else 2) and all(u > c[0] for u in users), (P.name, r, c, users)
    first_exp = min(i for i, ins in enumerate(b) if ins.startswith("v_exp"))
    assert sum(1 for ins in b[:first_exp] if ins.startswith("v_mfma")) >= 2
    # every transposed-fragment register is written once per body, behind the last MFMA of B
    lastB = max(i for i, ins in enumerate(b) if ins.startswith("v_mfma_f32_32x32x16_bf16 a["))
    tr_w = [i for i, ins in enumerate(b) if ins.startswith("ds_read_b64_tr_b16")]
    assert len(tr_w) == (16 if P.name == "DQ" else 32) and min(tr_w) > lastB, (P.name, lastB, min(tr_w))


def main():
    print("// GENERATED by gen_attn_bwd128.py -- do not edit.  Software-pipelined main loops of the head_dim-128 attention backward (attention128_bwd.hip);")
    print("// see the generator's docstring.")
    for P in (dkv(), dq()):
        check(P)
        emit(f"ABWD128_{P.name}_ASM", main_loop(P))
    sregs = [f'"s{i}"' for i in range(MK, MK + 32)] + [f'"s{i}"' for i in range(70, 94)] + ['"scc"', '"vcc"', '"memory"']
    print("#define ABWD128_DKV_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(16, 229)] + [f'"a{i}"' for i in range(0, 192)] + sregs))
    print("#define ABWD128_DQ_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(16, 229)] + [f'"a{i}"' for i in range(0, 128)] + sregs))
    print()
    for base in range(0, 128, 16):
        rd = " ".join(f"v_accvgpr_read_b32 %{i}, a{base + i}\\n" for i in range(16))
        outs = ", ".join(f'"=v"(t_[{i}])' for i in range(16))
        print(f"#define ABWD128_READ_ACC_{base}(d) {{ float t_[16]; asm volatile(\"{rd}\" : {outs}); _Pragma(\"unroll\") for (int i_ = 0; i_ < 16; ++i_) d[i_] = t_[i_]; }}")


if __name__ == "__main__":
    main()
