#!/usr/bin/env python3
"""Generator of the TWO-ROW-BLOCK software-pipelined main loops of the head_dim-64 flash-attention BACKWARD (csrc/attn_bwd64x2_asm.inc,
included by attention_bwd.hip).

gen_attn_bwd64.py's loops give a wave 32 keys (queries) and run two waves per SIMD; they are bound by the wave's own instruction issue
(~135 instructions per 16 MFMAs, VALU and LDS costs add).  Here a wave owns 64 keys (dK/dV pass) or 64 queries (dQ pass) = two
32-row blocks b = 0, 1, ONE wave per SIMD with the 512-register budget: the streamed operand's row fragments, its transposed fragments and the
-L | -Delta reads of a half tile feed BOTH blocks' MFMAs, so a body is 32 (24) MFMAs for 1.2 x the instructions of a 16 (12)-MFMA body.  Same
schedule, same arithmetic:

    body(h) =   B(h-1)   ||   V(h)   ||   A(h+1)          one MFMA, then the fillers of its gap, then the next MFMA ...

Every output row sees the same MFMAs in the same order as in the round-3 kernels and in gen_attn_bwd64.py's: BIT-IDENTICAL.  No tail masks (the
zero-padding contract of attention_bwd.hip).

Registers (fixed).  S / dP of consecutive halves alternate between X and Y; block b's share of a buffer is + 32 b (s | dp):
    dK/dV pass   v[16:79] X   v[80:143] Y   v[144:175] packed (block b: P at + 16 b, dZ at + 16 b + 8)   v[176:207] -L | -Delta (C operands of
                 both blocks' chains)   v[208:239] row fragments (Q kk0..3 | dO kk0..3)   v[240:244] derived read addresses
                 a[0:127] accumulators (block b: dV^T db0, db1, dK^T db0, db1 at 64 b + 0 / 16 / 32 / 48)   a[128:191] the lane's key / value rows
                 (block b: K at 128 + 32 b, V at 144 + 32 b)   a[192:223] transposed fragments [dO db0 | dO db1 | Q db0 | Q db1] x k-step
    dQ pass      v[16:79] X   v[80:143] Y   v[144:159] packed dZ (block b at + 8 b)   v[160:223] -L | -Delta splats (block b at + 32 b)
                 v[224:226] derived read addresses
                 a[0:63] dQ^T (block b: db0, db1 at 32 b + 0 / 16)   a[64:127] the lane's q~ / dO rows (block b: q~ at 64 + 32 b, dO at 80 + 32 b)
                 a[128:143] transposed K fragments   a[144:175] row fragments (K kk0..3 | V kk0..3; ds_read_b128 writes AGPRs directly)
LDS / tile ring: gen_attn_bwd64.py's (4 slots, three tiles ahead, one vmcnt; the loop always issues a tile's loads).

usage: python gen_attn_bwd64x2.py > attn_bwd64x2_asm.inc
"""

X, Y = 16, 80
# scalar registers (clobbered)
S_B0, S_B1, S_B2 = 70, 72, 74
S_NT, S_CNT, S_LT, S_LDSL, S_STA, S_STT, S_D, S_M0, S_EX, S_W1K, S_WNL, S_INC, S_T = 76, 77, 78, 79, 80, 81, 82, 83, 84, 86, 87, 88, 89
S_B0H, S_B1H = 90, 92
TILE = 8192


def vr(b, n=1):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=1):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


class Pass:
    pass


def dkv():
    P = Pass()
    P.name, P.ST, P.nloads, P.nacc = "DKV2", 2 * TILE + 512, 5, 128
    PZ, LD, AF, DR, KV, TRA = 144, 176, 208, 240, 128, 192
    P.DR = DR

    def raddr(kk):
        return "%[r0]" if kk == 0 else vr(DR + kk - 1)

    def taddr(i):
        return f"%[a{i}]" if i < 2 else vr(DR + 3 + i - 2)

    def mf_A(n):
        """per k-step: S and dP of block 0, then of block 1; the first k-step takes -L / -Delta as its C operand (vdst != src2)"""
        out = []
        for kk in range(4):
            for b in range(2):
                s, d = n + 32 * b, n + 32 * b + 16
                cs = vr(LD, 16) if kk == 0 else vr(s, 16)
                cd = vr(LD + 16, 16) if kk == 0 else vr(d, 16)
                out.append(f"v_mfma_f32_32x32x16_bf16 {vr(s, 16)}, {vr(AF + 4 * kk, 4)}, {ar(KV + 32 * b + 4 * kk, 4)}, {cs}")
                out.append(f"v_mfma_f32_32x32x16_bf16 {vr(d, 16)}, {vr(AF + 16 + 4 * kk, 4)}, {ar(KV + 32 * b + 16 + 4 * kk, 4)}, {cd}")
        return out

    def mf_B():
        """k-step 0 of both blocks first (they read the packed registers' first halves), then k-step 1"""
        out = []
        for hs in range(2):
            t = TRA + 16 * hs
            for b in range(2):
                pf, zf = vr(PZ + 16 * b + 4 * hs, 4), vr(PZ + 16 * b + 8 + 4 * hs, 4)
                dv0, dv1, dk0, dk1 = ar(64 * b, 16), ar(64 * b + 16, 16), ar(64 * b + 32, 16), ar(64 * b + 48, 16)
                out += [f"v_mfma_f32_32x32x16_bf16 {dv0}, {ar(t, 4)}, {pf}, {dv0}", f"v_mfma_f32_32x32x16_bf16 {dk0}, {ar(t + 8, 4)}, {zf}, {dk0}",
                        f"v_mfma_f32_32x32x16_bf16 {dv1}, {ar(t + 4, 4)}, {pf}, {dv1}", f"v_mfma_f32_32x32x16_bf16 {dk1}, {ar(t + 12, 4)}, {zf}, {dk1}"]
        return out

    def rd_A(n, qb):
        out = []
        for i, o in enumerate((0, 16, 64, 80)):
            out.append(f"ds_read_b128 {vr(LD + 4 * i, 4)}, %[la] offset:{o + 128 * qb}")
        for i, o in enumerate((0, 16, 64, 80)):
            out.append(f"ds_read_b128 {vr(LD + 16 + 4 * i, 4)}, %[la] offset:{256 + o + 128 * qb}")
        for kk in range(4):
            out.append(f"ds_read_b128 {vr(AF + 4 * kk, 4)}, {raddr(kk)} offset:{4096 * qb}")
            out.append(f"ds_read_b128 {vr(AF + 16 + 4 * kk, 4)}, {raddr(kk)} offset:{TILE + 4096 * qb}")
        return out

    def rd_T(qb):
        out = []
        for hs in range(2):
            for (blk, off) in ((0, TILE), (8, 0)):       # dO tile at + 8192, Q tile at + 0; k-step hs: rows + 16 = + 2048 bytes
                for i in range(4):
                    out.append(f"ds_read_b64_tr_b16 {ar(TRA + 16 * hs + blk + 2 * i, 2)}, {taddr(i)} offset:{off + 2048 * hs + 4096 * qb}")
        return out

    def valu(c, qb):
        g = {}
        def put(gap, ins):
            g.setdefault(gap, []).append(ins)
        for b in range(2):
            s, d, pz = c + 32 * b, c + 32 * b + 16, PZ + 16 * b
            e0 = 1 + 8 * b                               # block 0's chains end four MFMAs before block 1's: its exps start at gap 1, block 1's at gap 9
            for r in range(16):
                put(e0 + r, f"v_exp_f32 {vr(s + r)}, {vr(s + r)}")
            for j in range(8):
                m = e0 + 2 * j + 3
                put(m, f"v_pk_mul_f32 {vr(d + 2 * j, 2)}, {vr(d + 2 * j, 2)}, {vr(s + 2 * j, 2)}")
                lo = 9 if j < 4 else 17                  # behind the MFMAs of B that read the old packed values (m0..m7 / m8..m15)
                put(max(m, lo), f"v_cvt_pk_bf16_f32 {vr(pz + j)}, {vr(s + 2 * j)}, {vr(s + 2 * j + 1)}")
                put(max(m + 1, lo), f"v_cvt_pk_bf16_f32 {vr(pz + 8 + j)}, {vr(d + 2 * j)}, {vr(d + 2 * j + 1)}")
        return g

    P.mf_A, P.mf_B, P.rd_A, P.rd_T, P.valu = mf_A, mf_B, rd_A, rd_T, valu
    P.nB, P.nA = 16, 16
    P.rdA_gaps = [g for g in range(8) for _ in range(2)]             # 16 reads: C operands first, fragment k-step 3 last
    P.rdT_gaps = [16 + i // 2 for i in range(16)]                    # from the gap behind B's last MFMA
    P.a_addrs, P.t_addrs = ["la", "r0"], ["a0", "a1"]
    P.derive_A = [f"v_xor_b32 {vr(DR + k - 1)}, {32 * k}, %[r0]" for k in (1, 2, 3)]
    P.derive_T = [f"v_xor_b32 {vr(DR + 3)}, 64, %[a0]", f"v_xor_b32 {vr(DR + 4)}, 64, %[a1]"]
    P.PZ, P.npk = PZ, 32

    def stage():
        return [f"s_add_i32 m0, s{S_W1K}, s{S_LDSL}", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0}:{S_B0 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0H}:{S_B0H + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1}:{S_B1 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1H}:{S_B1H + 1}]",
                f"s_add_i32 m0, s{S_WNL}, s{S_LDSL}", f"s_mov_b64 s[{S_EX}:{S_EX + 1}], exec", "s_mov_b64 exec, 0xff",
                f"global_load_lds_dwordx4 %[g2], s[{S_B2}:{S_B2 + 1}]", f"s_mov_b64 exec, s[{S_EX}:{S_EX + 1}]"]
    P.stage = stage
    P.adv = [(S_B0, 13), (S_B1, 13), (S_B2, 9), (S_B0H, 13), (S_B1H, 13)]
    # the lane's rows: block b's K fragments kk0..3 at a[KV + 32 b ..], V at a[KV + 32 b + 16 ..]; loaded through the fragment registers
    P.row_loads = [(X + 16 * which + 4 * kk, "grow" + str(b), "p" + str(which), 32 * kk, KV + 32 * b + 16 * which + 4 * kk)
                   for b in range(2) for which in range(2) for kk in range(4)]
    return P


def dq():
    P = Pass()
    P.name, P.ST, P.nloads, P.nacc = "DQ2", 2 * TILE, 4, 64
    PZ, NLD, AFA, DR, KV, TRA = 144, 160, 144, 224, 64, 128          # (AFA: the K | V row fragments live in AGPRs a[144:175]: ds_read_b128 writes them directly)
    P.DR = DR

    def raddr(kk):
        return "%[r0]" if kk == 0 else vr(DR + kk - 1)

    def taddr(i):
        return f"%[a{i}]" if i < 2 else f"%[a{i}]"        # (four address operands: this pass has operand slots to spare and no free VGPR block)

    def mf_A(n):
        out = []
        for kk in range(4):
            for b in range(2):
                s, d = n + 32 * b, n + 32 * b + 16
                cs = vr(NLD + 32 * b, 16) if kk == 0 else vr(s, 16)
                cd = vr(NLD + 32 * b + 16, 16) if kk == 0 else vr(d, 16)
                out.append(f"v_mfma_f32_32x32x16_bf16 {vr(s, 16)}, {ar(AFA + 4 * kk, 4)}, {ar(KV + 32 * b + 4 * kk, 4)}, {cs}")
                out.append(f"v_mfma_f32_32x32x16_bf16 {vr(d, 16)}, {ar(AFA + 16 + 4 * kk, 4)}, {ar(KV + 32 * b + 16 + 4 * kk, 4)}, {cd}")
        return out

    def mf_B():
        out = []
        for hs in range(2):
            t = TRA + 8 * hs
            for b in range(2):
                zf = vr(PZ + 8 * b + 4 * hs, 4)
                out += [f"v_mfma_f32_32x32x16_bf16 {ar(32 * b, 16)}, {ar(t, 4)}, {zf}, {ar(32 * b, 16)}",
                        f"v_mfma_f32_32x32x16_bf16 {ar(32 * b + 16, 16)}, {ar(t + 4, 4)}, {zf}, {ar(32 * b + 16, 16)}"]
        return out

    def rd_A(n, kb):
        out = []
        for kk in range(4):
            out.append(f"ds_read_b128 {ar(AFA + 4 * kk, 4)}, {raddr(kk)} offset:{4096 * kb}")
            out.append(f"ds_read_b128 {ar(AFA + 16 + 4 * kk, 4)}, {raddr(kk)} offset:{TILE + 4096 * kb}")
        return out

    def rd_T(kb):
        out = []
        for hs in range(2):
            for i in range(4):
                out.append(f"ds_read_b64_tr_b16 {ar(TRA + 8 * hs + 2 * i, 2)}, {taddr(i)} offset:{2048 * hs + 4096 * kb}")
        return out

    def valu(c, kb):
        g = {}
        def put(gap, ins):
            g.setdefault(gap, []).append(ins)
        for b in range(2):
            s, d, pz = c + 32 * b, c + 32 * b + 16, PZ + 8 * b
            e0 = 1 + 6 * b
            for r in range(16):
                put(e0 + r, f"v_exp_f32 {vr(s + r)}, {vr(s + r)}")
            for j in range(8):
                m = e0 + 2 * j + 3
                put(m, f"v_pk_mul_f32 {vr(d + 2 * j, 2)}, {vr(d + 2 * j, 2)}, {vr(s + 2 * j, 2)}")
                put(max(m + 1, 5 if j < 4 else 9), f"v_cvt_pk_bf16_f32 {vr(pz + j)}, {vr(d + 2 * j)}, {vr(d + 2 * j + 1)}")
        return g

    P.mf_A, P.mf_B, P.rd_A, P.rd_T, P.valu = mf_A, mf_B, rd_A, rd_T, valu
    P.nB, P.nA = 8, 16
    P.rdA_gaps = [0, 0, 1, 1, 2, 2, 3, 3]
    P.rdT_gaps = [8 + i // 2 for i in range(8)]
    P.a_addrs, P.t_addrs = ["r0"], ["a0", "a1", "a2", "a3"]
    P.derive_A = [f"v_xor_b32 {vr(DR + k - 1)}, {32 * k}, %[r0]" for k in (1, 2, 3)]
    P.derive_T = []
    P.PZ, P.npk = PZ, 16

    def stage():
        return [f"s_add_i32 m0, s{S_W1K}, s{S_LDSL}", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0}:{S_B0 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0H}:{S_B0H + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1}:{S_B1 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1H}:{S_B1H + 1}]"]
    P.stage = stage
    P.adv = [(S_B0, 13), (S_B1, 13), (S_B0H, 13), (S_B1H, 13)]
    P.row_loads = [(X + 16 * which + 4 * kk, "grow" + str(b), "p" + str(which), 32 * kk, KV + 32 * b + 16 * which + 4 * kk)
                   for b in range(2) for which in range(2) for kk in range(4)]
    P.NLD = NLD
    return P


def ring_step(P, idx_reg, addrs, derive):
    out = [f"s_add_i32 s{idx_reg}, s{idx_reg}, 1", f"s_mov_b32 s{S_D}, {P.ST}", f"s_cmp_eq_u32 s{idx_reg}, 4",
           f"s_cselect_b32 s{S_D}, {-3 * P.ST}, s{S_D}", f"s_cselect_b32 s{idx_reg}, 0, s{idx_reg}"]
    out += [f"v_add_u32 %[{a}], s{S_D}, %[{a}]" for a in addrs]
    return out + list(derive)


def sync(P):
    out = [f"s_waitcnt vmcnt({P.nloads})", "s_barrier"] + P.stage()
    out += [f"s_cmp_lt_u32 s{S_LT}, s{S_NT}", f"s_cselect_b32 s{S_INC}, 1, 0", f"s_add_u32 s{S_LT}, s{S_LT}, s{S_INC}"]     # S_NT holds nt - 1
    for (b, sh) in P.adv:
        out += [f"s_lshl_b32 s{S_T}, s{S_INC}, {sh}", f"s_add_u32 s{b}, s{b}, s{S_T}", f"s_addc_u32 s{b + 1}, s{b + 1}, 0"]
    out += [f"s_add_i32 s{S_LDSL}, s{S_LDSL}, {P.ST}", f"s_cmp_eq_u32 s{S_LDSL}, {4 * P.ST}", f"s_cselect_b32 s{S_LDSL}, 0, s{S_LDSL}"]
    return out


def body(P, cur, nxt, has_B, has_A, qb_A, qb_T, pre=()):
    fill = {}
    def put(gap, ins):
        fill.setdefault(gap, []).append(ins)
    if has_A:
        for gp, ins in zip(P.rdA_gaps, P.rd_A(nxt, qb_A)):
            put(gp, ins)
    for gp, lst in sorted(P.valu(cur, 0).items()):
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
    if P.name == "DKV2":
        L.append(f"s_mov_b64 s[{S_B2}:{S_B2 + 1}], %[b2]")
    for (lo, hi) in ((S_B0, S_B0H), (S_B1, S_B1H)):
        L += [f"s_add_u32 s{hi}, s{lo}, 4096", f"s_addc_u32 s{hi + 1}, s{lo + 1}, 0"]
    L += [f"s_sub_u32 s{S_NT}, %[nt], 1", f"s_mov_b32 s{S_CNT}, s{S_NT}", f"s_min_u32 s{S_LT}, s{S_NT}, 3", f"s_mov_b32 s{S_LDSL}, {3 * P.ST}",
          f"s_mov_b32 s{S_STA}, 0", f"s_mov_b32 s{S_STT}, 0", f"s_lshl_b32 s{S_W1K}, %[wv], 10"]
    if P.name == "DKV2":
        L += [f"s_lshl_b32 s{S_WNL}, %[wv], 7", f"s_add_u32 s{S_WNL}, s{S_WNL}, {2 * TILE}"]
    for i in range(P.nacc):
        L.append(f"v_accvgpr_write_b32 {ar(i)}, 0")
    # this lane's rows (MFMA B operands of the first products) into AGPRs: block 0 through the fragment registers, then block 1
    for b in range(2):
        for (vreg, goff, base, off, areg) in P.row_loads:
            if goff.endswith(str(b)):
                L.append(f"global_load_dwordx4 {vr(vreg, 4)}, %[{goff}], %[{base}] offset:{off}")
        L.append("s_waitcnt vmcnt(0)")                    # (also tiles 0..2, staged by the shell)
        for (vreg, goff, base, off, areg) in P.row_loads:
            if goff.endswith(str(b)):
                for e in range(4):
                    L.append(f"v_accvgpr_write_b32 {ar(areg + e)}, {vr(vreg + e)}")
    if P.name == "DQ2":
        for b in range(2):
            for i in range(16):
                L += [f"v_mov_b32 {vr(P.NLD + 32 * b + i)}, %[nl{b}]", f"v_mov_b32 {vr(P.NLD + 32 * b + 16 + i)}, %[nd{b}]"]
    L += P.derive_A + P.derive_T
    L.append("s_barrier")
    # prologue: A(0) alone, then body(0) without B
    L += P.rd_A(X, 0)
    L.append("s_waitcnt lgkmcnt(0)")
    L += P.mf_A(X)
    L += ["s_nop 15", "s_nop 15"]
    L += body(P, X, Y, False, True, 1, 0)
    L.append(f"s_cmp_eq_u32 s{S_CNT}, 0")
    L.append(f"s_cbranch_scc1 L_{P.name}_tail%=")
    L.append(f"L_{P.name}_loop%=:")
    L += body(P, Y, X, True, True, 0, 1, pre=sync(P) + ring_step(P, S_STA, P.a_addrs, P.derive_A))
    L += body(P, X, Y, True, True, 1, 0, pre=ring_step(P, S_STT, P.t_addrs, P.derive_T))
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", f"s_cbranch_scc1 L_{P.name}_loop%="]
    L.append(f"L_{P.name}_tail%=:")
    L += body(P, Y, X, True, False, 0, 1)
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
    import re
    b = body(P, Y, X, True, True, 0, 1)
    assert sum(1 for ins in b if ins.startswith("v_mfma")) == P.nA + P.nB
    for reg in range(P.PZ, P.PZ + P.npk):
        w = [i for i, ins in enumerate(b) if ins.startswith(f"v_cvt_pk_bf16_f32 v{reg},")]
        assert len(w) == 1, (P.name, reg, w)
        lo = P.PZ + 4 * ((reg - P.PZ) // 4)
        readers = [i for i, ins in enumerate(b) if ins.startswith("v_mfma") and f", v[{lo}:{lo + 3}], a[" in ins]
        assert len(readers) == 2 and all(r < w[0] for r in readers), (P.name, reg, readers, w)
    for blk in range(2):
        for r in range(16):
            reg = Y + 32 * blk + r
            e = [i for i, ins in enumerate(b) if ins == f"v_exp_f32 v{reg}, v{reg}"]
            assert len(e) == 1
            pair = f"v[{reg & ~1}:{(reg & ~1) + 1}]"
            users = [i for i, ins in enumerate(b) if (ins.startswith("v_pk_mul") and ins.endswith(pair)) or
                     (ins.startswith("v_cvt_pk") and re.search(rf", v{reg}(,|$)", ins))]
            assert len(users) == (1 if P.name == "DQ2" else 2) and all(u > e[0] + 1 for u in users), (P.name, reg, e, users)
    first_exp = {blk: min(i for i, ins in enumerate(b) if ins.startswith("v_exp_f32 v") and Y + 32 * blk <= int(ins.split()[1][1:-1]) < Y + 32 * blk + 16) for blk in range(2)}
    n_mf = lambda upto: sum(1 for ins in b[:upto] if ins.startswith("v_mfma"))
    assert n_mf(first_exp[0]) >= 2 and n_mf(first_exp[1]) >= (8 if P.name == "DKV2" else 6), first_exp
    lastB = max(i for i, ins in enumerate(b) if ins.startswith("v_mfma_f32_32x32x16_bf16 a["))
    tr_w = [i for i, ins in enumerate(b) if ins.startswith("ds_read_b64_tr_b16")]
    assert len(tr_w) == (8 if P.name == "DQ2" else 16) and min(tr_w) > lastB, (P.name, lastB, min(tr_w))


def main():
    print("// GENERATED by gen_attn_bwd64x2.py -- do not edit.  Two-row-block software-pipelined main loops of the head_dim-64 attention backward")
    print("// (attention_bwd.hip); see the generator's docstring.")
    for P in (dkv(), dq()):
        check(P)
        emit(f"ABWD64X2_{P.name}_ASM", main_loop(P))
    sregs = [f'"s{i}"' for i in range(70, 94)] + ['"scc"', '"memory"']
    print("#define ABWD64X2_DKV2_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(16, 245)] + [f'"a{i}"' for i in range(0, 224)] + sregs))
    print("#define ABWD64X2_DQ2_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(16, 227)] + [f'"a{i}"' for i in range(0, 176)] + sregs))
    print()
    for base in range(0, 128, 16):
        rd = " ".join(f"v_accvgpr_read_b32 %{i}, a{base + i}\\n" for i in range(16))
        outs = ", ".join(f'"=v"(t_[{i}])' for i in range(16))
        print(f"#define ABWD64X2_READ_ACC_{base}(d) {{ float t_[16]; asm volatile(\"{rd}\" : {outs}); _Pragma(\"unroll\") for (int i_ = 0; i_ < 16; ++i_) d[i_] = t_[i_]; }}")


if __name__ == "__main__":
    main()
