#!/usr/bin/env python3
"""Generator of the software-pipelined main loops of the head_dim-64 flash-attention BACKWARD (csrc/attn_bwd64_asm.inc, included by
attention_bwd.hip): the dK/dV pass (one key per lane, query tiles streamed) and the dQ pass (one query per lane, key tiles streamed).

Why: per 32-row half tile a wave runs  A: 8 MFMAs (S and dP chains)  ->  V: 16 v_exp + 8 v_pk_mul + 8..16 v_cvt_pk  ->  B: 8 (4) MFMAs
(second products).  Written as C++ (rounds 3-6) these are three serial phases per wave -- a wave issues in order, so while it sits in front of
an MFMA waiting for the matrix pipe it cannot issue the VALU work that follows -- and with two waves per SIMD the passes ran at 0.43 MFMA-busy
(profiles/r06f_pmc_summary_train_mid1.txt).  Here the phases of THREE consecutive halves are interleaved in program order:

    body(h) =   B(h-1)   ||   V(h)   ||   A(h+1)          one MFMA, then the fillers of its gap, then the next MFMA ...

so every MFMA gap carries VALU / LDS work of another half and the matrix pipe is fed back to back by ONE wave (the second wave of the SIMD
fills what is left).  Same MFMAs, same operand order and the same accumulation order per output element as the C++ kernels: the results are
BIT-IDENTICAL (tests/test_gpu_backward.py compares the two).

Registers (fixed; the C++ shell sees clobber lists only).  S/dP of consecutive halves alternate between buffers X and Y:
    dK/dV pass   v[16:47] X (s | dp)   v[48:79] Y   v[80:95] packed P | dZ   v[96:127] A fragments (Q kk0..3 | dO kk0..3)
                 v[128:159] transposed fragments   a[64:95] this lane's key / value rows (MFMA B operands)
                 a[0:63] accumulators dV^T db0 | dK^T db0 | dV^T db1 | dK^T db1 (read out by ABWD64_READ_ACC_n behind the loop)
    dQ pass      v[16:47] X   v[48:79] Y   v[80:87] packed dZ   v[96:127] A fragments (K | V)   v[128:143] transposed K fragments
                 v[144:175] -L | -Delta splats (C operands)   a[32:63] this lane's q~ / dO rows   a[0:31] accumulators dQ^T db0 | db1
(asm operands: at most 30 per statement, a "+" operand counting twice -- hence fixed accumulators and the in-asm address arithmetic.)
Tiles arrive by global_load_lds_dwordx4 into a 4-slot ring, three tiles ahead; the loop ALWAYS issues a tile's loads (past the end it re-loads
the last tile into a slot nobody reads) so that `s_waitcnt vmcnt(N)` has one N.  No tail masks: padded query / key rows must be ZERO in
q, k, v and dO (they are: zero-initialised workspaces that only rows < S are ever written to; attention_bwd.hip states the contract) -- a
zero row contributes exactly 0 to every second product.

usage: python gen_attn_bwd64.py > attn_bwd64_asm.inc
"""

X, Y, PZ, AF, TR = 16, 48, 80, 96, 128
NLD = 144                     # dQ pass: -L splat v[144:159], -Delta splat v[160:175]
# scalar registers (clobbered)
S_B0, S_B1, S_B2 = 70, 72, 74          # 64-bit source bases of the streamed operands (next tile to load)
S_NT, S_CNT, S_LT, S_LDSL, S_STA, S_STT, S_D, S_M0, S_EX, S_W1K, S_WNL, S_INC, S_T = 76, 77, 78, 79, 80, 81, 82, 83, 84, 86, 87, 88, 89
S_B0H, S_B1H = 90, 92                  # the same bases + 4096 bytes (rows +32: this wave's second 8-row group; the instruction offset field ends at 4095)


def vr(b, n=1):
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def ar(b, n=1):
    return f"a[{b}:{b + n - 1}]" if n > 1 else f"a{b}"


class Pass:
    pass


ABL = set()       # ablation builds (timing only, results garbage): "novalu", "nolds", "nosync"


def dkv():
    P = Pass()
    P.name, P.ST, P.nloads = "DKV", 16896, 5
    P.nacc = 64
    P.kv = 64                                            # a[64:79] key row fragments kk0..3, a[80:95] value row fragments

    def mf_A(n):                                          # S and dP chains of the half whose buffer is n: 8 MFMAs
        out = []
        for kk in range(4):
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n, 16)}, {vr(AF + 4 * kk, 4)}, {ar(P.kv + 4 * kk, 4)}, {vr(n, 16)}")
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n + 16, 16)}, {vr(AF + 16 + 4 * kk, 4)}, {ar(P.kv + 16 + 4 * kk, 4)}, {vr(n + 16, 16)}")
        return out

    def mf_B():                                           # dV^T += dO^T . P, dK^T += Q^T . dZ: [dO db0 | dO db1 | Q db0 | Q db1] x 4 regs per k-step
        out = []
        for hs in range(2):
            t = TR + 16 * hs
            pf, zf = vr(PZ + 4 * hs, 4), vr(PZ + 8 + 4 * hs, 4)
            dv0, dk0, dv1, dk1 = ar(0, 16), ar(16, 16), ar(32, 16), ar(48, 16)
            out += [f"v_mfma_f32_32x32x16_bf16 {dv0}, {vr(t, 4)}, {pf}, {dv0}", f"v_mfma_f32_32x32x16_bf16 {dk0}, {vr(t + 8, 4)}, {zf}, {dk0}",
                    f"v_mfma_f32_32x32x16_bf16 {dv1}, {vr(t + 4, 4)}, {pf}, {dv1}", f"v_mfma_f32_32x32x16_bf16 {dk1}, {vr(t + 12, 4)}, {zf}, {dk1}"]
        return out

    def rd_A(n, qb):                                      # C operands (-L | -Delta of the half's 32 queries) straight into buffer n, then the fragments
        out = []
        for i, o in enumerate((0, 16, 64, 80)):
            out.append(f"ds_read_b128 {vr(n + 4 * i, 4)}, %[la] offset:{o + 128 * qb}")
        for i, o in enumerate((0, 16, 64, 80)):
            out.append(f"ds_read_b128 {vr(n + 16 + 4 * i, 4)}, %[la] offset:{256 + o + 128 * qb}")
        # fragment kk sits at 16-byte chunk (2 kk + lg) ^ swizzle(row) = chunk(kk = 0) ^ 2 kk: its address is r0 ^ 32 kk (ring slots are multiples of
        # 512 bytes) -- formed in the destination's first register instead of held in three more address registers (this pass has none to spare:
        # 256 unified registers at two waves per SIMD); entries are (instructions of one gap slot)
        for kk in range(4):
            for dst, off in ((AF + 4 * kk, 4096 * qb), (AF + 16 + 4 * kk, 8192 + 4096 * qb)):
                if kk == 0:
                    out.append(f"ds_read_b128 {vr(dst, 4)}, %[r0] offset:{off}")
                else:
                    out.append(f"v_xor_b32 {vr(dst)}, {32 * kk}, %[r0]\\nds_read_b128 {vr(dst, 4)}, {vr(dst)} offset:{off}")
        return out

    def rd_T(qb):
        out = []
        for hs in range(2):
            for (blk, off) in ((0, 8192), (8, 0)):       # dO tile at +8192, Q tile at +0; k-step hs: rows +16 = +2048 bytes
                for i in range(4):                       # pieces (db, jj) = (0,0) (0,1) (1,0) (1,1): db = 1 is chunk ^ 4 = address ^ 64
                    dst = TR + 16 * hs + blk + 2 * i
                    o = off + 2048 * hs + 4096 * qb
                    if i < 2:
                        out.append(f"ds_read_b64_tr_b16 {vr(dst, 2)}, %[a{i}] offset:{o}")
                    else:
                        out.append(f"v_xor_b32 {vr(dst)}, 64, %[a{i - 2}]\\nds_read_b64_tr_b16 {vr(dst, 2)}, {vr(dst)} offset:{o}")
        return out

    def valu(c):
        """{gap: [instr]} of V on buffer c: exps from gap 1 (the S chain's last MFMA is three MFMAs back), the packed outputs behind the
        MFMAs of B that read their old values (pf0 / zf0: m0..m3, pf1 / zf1: m4..m7)."""
        g = {}
        def put(gap, ins):
            g.setdefault(gap, []).append(ins)
        E = lambda r: f"v_exp_f32 {vr(c + r)}, {vr(c + r)}"
        M = lambda j: f"v_pk_mul_f32 {vr(c + 16 + 2 * j, 2)}, {vr(c + 16 + 2 * j, 2)}, {vr(c + 2 * j, 2)}"
        CP = lambda j: f"v_cvt_pk_bf16_f32 {vr(PZ + j)}, {vr(c + 2 * j)}, {vr(c + 2 * j + 1)}"
        CZ = lambda j: f"v_cvt_pk_bf16_f32 {vr(PZ + 8 + j)}, {vr(c + 16 + 2 * j)}, {vr(c + 16 + 2 * j + 1)}"
        for r in range(8):
            put(1 + r // 2, E(r))
        for r in range(8, 16):
            put(5 + (r - 8), E(r))
        for j, gp in enumerate((5, 6, 7, 8, 7, 9, 11, 13)):
            put(gp, M(j))
        for j, gp in enumerate((5, 6, 7, 8, 8, 9, 11, 13)):
            put(gp, CP(j))
        for j, gp in enumerate((6, 7, 8, 9, 8, 10, 12, 14)):
            put(gp, CZ(j))
        return g

    P.mf_A, P.mf_B, P.rd_A, P.rd_T, P.valu = mf_A, mf_B, rd_A, rd_T, valu
    P.nB, P.nA = 8, 8
    P.rdA_gaps = [0, 0, 0, 1, 1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5]      # C operands first (needed by m8 / m9), fragment kk3 last (read by m15 of the previous body)
    P.rdT_gaps = [8, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 11, 11, 12, 12]      # (from the gap behind B's last MFMA; three gaps of flight before the next body)
    P.a_addrs = ["la", "r0"]
    P.t_addrs = ["a0", "a1"]

    def stage():
        """this wave's five LDS-DMA pieces of tile S_LT into ring slot S_LDSL: Q rows [8 w, 8 w + 8) and [8 (w + 4), ..), the same of dO, and 32 of
        the tile's 128 -L | -Delta floats (first 8 lanes)"""
        return [f"s_add_i32 m0, s{S_W1K}, s{S_LDSL}", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0}:{S_B0 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0H}:{S_B0H + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1}:{S_B1 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1H}:{S_B1H + 1}]",
                f"s_add_i32 m0, s{S_WNL}, s{S_LDSL}", f"s_mov_b64 s[{S_EX}:{S_EX + 1}], exec", "s_mov_b64 exec, 0xff",
                f"global_load_lds_dwordx4 %[g2], s[{S_B2}:{S_B2 + 1}]", f"s_mov_b64 exec, s[{S_EX}:{S_EX + 1}]"]
    P.stage = stage
    P.adv = [(S_B0, 13), (S_B1, 13), (S_B2, 9), (S_B0H, 13), (S_B1H, 13)]      # bytes per tile = 1 << shift: 64 rows x 128 B; 128 floats
    return P


def dq():
    P = Pass()
    P.name, P.ST, P.nloads = "DQ", 16384, 4
    P.nacc = 32
    P.kv = 32                                            # a[32:47] q~ row fragments, a[48:63] dO row fragments

    def mf_A(n):
        out = []
        for kk in range(4):
            cs = vr(NLD, 16) if kk == 0 else vr(n, 16)
            cd = vr(NLD + 16, 16) if kk == 0 else vr(n + 16, 16)
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n, 16)}, {vr(AF + 4 * kk, 4)}, {ar(P.kv + 4 * kk, 4)}, {cs}")
            out.append(f"v_mfma_f32_32x32x16_bf16 {vr(n + 16, 16)}, {vr(AF + 16 + 4 * kk, 4)}, {ar(P.kv + 16 + 4 * kk, 4)}, {cd}")
        return out

    def mf_B():                                           # dQ^T += K^T . dZ^T: [K db0 | K db1] x 4 regs per k-step
        out = []
        for hs in range(2):
            t = TR + 8 * hs
            zf = vr(PZ + 4 * hs, 4)
            out += [f"v_mfma_f32_32x32x16_bf16 {ar(0, 16)}, {vr(t, 4)}, {zf}, {ar(0, 16)}", f"v_mfma_f32_32x32x16_bf16 {ar(16, 16)}, {vr(t + 4, 4)}, {zf}, {ar(16, 16)}"]
        return out

    def rd_A(n, kb):
        out = []
        for kk in range(4):
            out.append(f"ds_read_b128 {vr(AF + 4 * kk, 4)}, %[r{kk}] offset:{4096 * kb}")
            out.append(f"ds_read_b128 {vr(AF + 16 + 4 * kk, 4)}, %[r{kk}] offset:{8192 + 4096 * kb}")
        return out

    def rd_T(kb):
        out = []
        for hs in range(2):
            for i in range(4):
                out.append(f"ds_read_b64_tr_b16 {vr(TR + 8 * hs + 2 * i, 2)}, %[a{i}] offset:{2048 * hs + 4096 * kb}")
        return out

    def valu(c):
        g = {}
        def put(gap, ins):
            g.setdefault(gap, []).append(ins)
        E = lambda r: f"v_exp_f32 {vr(c + r)}, {vr(c + r)}"
        M = lambda j: f"v_pk_mul_f32 {vr(c + 16 + 2 * j, 2)}, {vr(c + 16 + 2 * j, 2)}, {vr(c + 2 * j, 2)}"
        CZ = lambda j: f"v_cvt_pk_bf16_f32 {vr(PZ + j)}, {vr(c + 16 + 2 * j)}, {vr(c + 16 + 2 * j + 1)}"
        for r in range(12):
            put(1 + r // 2, E(r))
        for r in range(12, 16):
            put(7 + (r - 12), E(r))
        for j, gp in enumerate((7, 7, 8, 8, 9, 9, 9, 11)):
            put(gp, M(j))
        for j, gp in enumerate((8, 8, 10, 10, 10, 10, 11, 11)):
            put(gp, CZ(j))
        return g

    P.mf_A, P.mf_B, P.rd_A, P.rd_T, P.valu = mf_A, mf_B, rd_A, rd_T, valu
    P.nB, P.nA = 4, 8
    P.rdA_gaps = [0, 0, 0, 0, 1, 1, 1, 1]
    P.rdT_gaps = [5, 5, 6, 6, 7, 7, 8, 8]
    P.a_addrs = ["r0", "r1", "r2", "r3"]
    P.t_addrs = ["a0", "a1", "a2", "a3"]

    def stage():
        return [f"s_add_i32 m0, s{S_W1K}, s{S_LDSL}", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0}:{S_B0 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B0H}:{S_B0H + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1}:{S_B1 + 1}]",
                "s_add_i32 m0, m0, 4096", "s_nop 0", f"global_load_lds_dwordx4 %[g0], s[{S_B1H}:{S_B1H + 1}]"]
    P.stage = stage
    P.adv = [(S_B0, 13), (S_B1, 13), (S_B0H, 13), (S_B1H, 13)]
    return P


def ring_step(P, idx_reg, addrs):
    """the read addresses `addrs` move to the next ring slot (slot index in idx_reg)"""
    out = [f"s_add_i32 s{idx_reg}, s{idx_reg}, 1", f"s_mov_b32 s{S_D}, {P.ST}", f"s_cmp_eq_u32 s{idx_reg}, 4",
           f"s_cselect_b32 s{S_D}, {-3 * P.ST}, s{S_D}", f"s_cselect_b32 s{idx_reg}, 0, s{idx_reg}"]
    out += [f"v_add_u32 %[{a}], s{S_D}, %[{a}]" for a in addrs]
    return out


def sync(P):
    """tile t + 1 landed for every wave; slot of tile t - 1 is free: load tile min(t + 3, nt - 1) into it; advance the load state"""
    out = ([] if "nosync" in ABL else [f"s_waitcnt vmcnt({P.nloads})", "s_barrier"] + P.stage())
    out += [f"s_cmp_lt_u32 s{S_LT}, s{S_NT}", f"s_cselect_b32 s{S_INC}, 1, 0", f"s_add_u32 s{S_LT}, s{S_LT}, s{S_INC}"]     # S_NT holds nt - 1 here
    for (b, sh) in P.adv:
        out += [f"s_lshl_b32 s{S_T}, s{S_INC}, {sh}", f"s_add_u32 s{b}, s{b}, s{S_T}", f"s_addc_u32 s{b + 1}, s{b + 1}, 0"]
    out += [f"s_add_i32 s{S_LDSL}, s{S_LDSL}, {P.ST}", f"s_cmp_eq_u32 s{S_LDSL}, {4 * P.ST}", f"s_cselect_b32 s{S_LDSL}, 0, s{S_LDSL}"]
    return out


def body(P, cur, nxt, has_B, has_A, qb_A, qb_T, pre=()):
    """B(h-1) || V(h) on buffer cur || A(h+1) into buffer nxt (its reads: half qb_A of the A-side tile); transposed reads of half h (qb_T)."""
    mf = (P.mf_B() if has_B else []) + (P.mf_A(nxt) if has_A else [])
    nB = P.nB
    fill = {}
    def put(gap, ins):
        fill.setdefault(gap, []).append(ins)
    if has_A:
        for gp, ins in zip(P.rdA_gaps, P.rd_A(nxt, qb_A)):
            put(gp, ins)
    for gp, lst in P.valu(cur).items():
        for ins in lst:
            put(gp, ins)
    for gp, ins in zip(P.rdT_gaps, P.rd_T(qb_T)):
        put(gp, ins)
    if "novalu" in ABL:
        fill = {g: [i for i in v if not i.startswith(("v_exp", "v_pk_mul", "v_cvt_pk"))] for g, v in fill.items()}
    if "nolds" in ABL:
        fill = {g: [i for i in v if "ds_read" not in i] for g, v in fill.items()}
    out = list(pre)
    out.append("s_waitcnt lgkmcnt(0)")                    # the transposed fragments of B(h-1) (read during the previous body)
    ngap = P.nB + P.nA
    for g in range(ngap):
        is_B = g < nB
        present = has_B if is_B else has_A
        if g == nB:
            out.append("s_waitcnt lgkmcnt(0)")            # A(h+1)'s fragments and C operands (read in this body's first gaps)
        if present:
            out.append((P.mf_B()[g] if is_B else P.mf_A(nxt)[g - nB]))
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
        L += [f"s_add_u32 s{hi}, s{lo}, 4096", f"s_addc_u32 s{hi + 1}, s{lo + 1}, 0"]
    L += [f"s_sub_u32 s{S_NT}, %[nt], 1",                 # nt - 1: loop count and the last tile's index
          f"s_mov_b32 s{S_CNT}, s{S_NT}", f"s_min_u32 s{S_LT}, s{S_NT}, 3", f"s_mov_b32 s{S_LDSL}, {3 * P.ST}",
          f"s_mov_b32 s{S_STA}, 0", f"s_mov_b32 s{S_STT}, 0",
          f"s_lshl_b32 s{S_W1K}, %[wv], 10"]              # LDS-DMA destinations of this wave inside a slot (the dynamic LDS base is 0: checked by the shell)
    if P.name == "DKV":
        L += [f"s_lshl_b32 s{S_WNL}, %[wv], 7", f"s_add_u32 s{S_WNL}, s{S_WNL}, 16384"]
    for i in range(P.nacc):
        L.append(f"v_accvgpr_write_b32 {ar(i)}, 0")
    # ---- this lane's rows (MFMA B operands of the first products) into AGPRs through the fragment registers
    for i in range(4):
        L.append(f"global_load_dwordx4 {vr(AF + 4 * i, 4)}, %[grow], %[p0] offset:{32 * i}")
        L.append(f"global_load_dwordx4 {vr(AF + 16 + 4 * i, 4)}, %[grow], %[p1] offset:{32 * i}")
    L.append("s_waitcnt vmcnt(0)")                        # (also tiles 0..2, staged by the shell: the barrier below publishes tile 0)
    for i in range(32):
        L.append(f"v_accvgpr_write_b32 {ar(P.kv + i)}, {vr(AF + i)}")
    if P.name == "DQ":
        for i in range(16):
            L += [f"v_mov_b32 {vr(NLD + i)}, %[nl]", f"v_mov_b32 {vr(NLD + 16 + i)}, %[nd]"]
    L.append("s_barrier")
    # ---- prologue: A(0) alone, then body(0) without B
    L += P.rd_A(X, 0)
    L.append("s_waitcnt lgkmcnt(0)")
    L += P.mf_A(X)
    L += ["s_nop 15", "s_nop 15"]
    L += body(P, X, Y, False, True, 1, 0)
    # ---- nt - 1 iterations: sync(t + 1) | body(2t + 1) on Y (A-side addresses -> tile t + 1) | body(2t + 2) on X (transposed-side -> tile t + 1)
    L.append(f"s_cmp_eq_u32 s{S_CNT}, 0")
    L.append(f"s_cbranch_scc1 L_{P.name}_tail%=")
    L.append(f"L_{P.name}_loop%=:")
    L += body(P, Y, X, True, True, 0, 1, pre=sync(P) + ring_step(P, S_STA, P.a_addrs))
    L += body(P, X, Y, True, True, 1, 0, pre=ring_step(P, S_STT, P.t_addrs))
    L += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", f"s_cbranch_scc1 L_{P.name}_loop%="]
    L.append(f"L_{P.name}_tail%=:")
    # ---- body(2 nt - 1) without A, then B(2 nt - 1)
    L += body(P, Y, X, True, False, 0, 1)
    L += ["s_waitcnt lgkmcnt(0)", "s_nop 7"]
    L += P.mf_B()
    L += [f"s_mov_b32 m0, s{S_M0}", "s_waitcnt vmcnt(0)", "s_nop 15", "s_nop 15"]
    return L


def emit(name, lines):
    print(f"#define {name} \\")
    lines = [part for ln in lines for part in ln.split("\\n")]
    for ln in lines:
        print(f'    "{ln}\\n" \\')
    print('    ""')
    print()


def check(P, lines):
    """structural checks on a steady-state body: MFMA count, no packed output written before the MFMAs of B that read its old value, every exp at
    least two instructions ahead of its first consumer (transcendental-use hazard)"""
    import re
    b = body(P, Y, X, True, True, 0, 1)
    assert sum(1 for ins in b if ins.startswith("v_mfma")) == P.nA + P.nB
    npk = 8 if P.name == "DQ" else 16
    for reg in range(PZ, PZ + npk):
        w = [i for i, ins in enumerate(b) if ins.startswith(f"v_cvt_pk_bf16_f32 v{reg},")]
        assert len(w) == 1, (P.name, reg, w)
        lo = PZ + 4 * ((reg - PZ) // 4)
        readers = [i for i, ins in enumerate(b) if ins.startswith("v_mfma") and f", v[{lo}:{lo + 3}], a[" in ins]
        assert len(readers) == 2 and all(r < w[0] for r in readers), (P.name, reg, readers, w)
    for r in range(16):
        e = [i for i, ins in enumerate(b) if ins == f"v_exp_f32 v{Y + r}, v{Y + r}"]
        assert len(e) == 1
        pair = f"v[{Y + (r & ~1)}:{Y + (r & ~1) + 1}]"
        users = [i for i, ins in enumerate(b) if (ins.startswith("v_pk_mul") and ins.endswith(pair)) or
                 (ins.startswith("v_cvt_pk") and re.search(rf", v{Y + r}(,|$)", ins))]
        need = 1 if P.name == "DQ" else 2
        assert len(users) == need and all(u > e[0] + 1 for u in users), (P.name, r, e, users)
    # the S / dP chains' results are three MFMAs old when the first exp reads them; -L|-Delta reads never target the buffer V works on
    first_exp = min(i for i, ins in enumerate(b) if ins.startswith("v_exp"))
    assert sum(1 for ins in b[:first_exp] if ins.startswith("v_mfma")) >= 2


def main():
    print("// GENERATED by gen_attn_bwd64.py -- do not edit.  Software-pipelined main loops of the head_dim-64 attention backward (attention_bwd.hip);")
    print("// see the generator's docstring.")
    for P in (dkv(), dq()):
        lines = main_loop(P)
        check(P, lines)
        emit(f"ABWD64_{P.name}_ASM", lines)
    # ablation builds of the dK/dV loop (scripts/attn_bwd_ablate.py; results are garbage): what bounds it?
    for tag, abl in (("NOVALU", {"novalu"}), ("NOLDS", {"nolds"}), ("NOSYNC", {"nosync"}), ("MFMAONLY", {"novalu", "nolds", "nosync"})):
        ABL.clear(); ABL.update(abl)
        emit(f"ABWD64_DKV_ASM_{tag}", main_loop(dkv()))
    ABL.clear()
    sregs = [f'"s{i}"' for i in range(70, 94)] + ['"scc"', '"memory"']
    print("#define ABWD64_DKV_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(16, 160)] + [f'"a{i}"' for i in range(0, 96)] + sregs))
    print("#define ABWD64_DQ_CLOBBERS " + ", ".join([f'"v{i}"' for i in range(16, 176)] + [f'"a{i}"' for i in range(0, 64)] + sregs))
    print()
    # accumulator read-out: ABWD64_READ_ACC_<base>(d) fills f32x16 d from a[base : base + 16)
    for base in (0, 16, 32, 48):
        rd = " ".join(f"v_accvgpr_read_b32 %{i}, a{base + i}\\n" for i in range(16))
        outs = ", ".join(f'"=v"(t_[{i}])' for i in range(16))
        print(f"#define ABWD64_READ_ACC_{base}(d) {{ float t_[16]; asm volatile(\"{rd}\" : {outs}); _Pragma(\"unroll\") for (int i_ = 0; i_ < 16; ++i_) d[i_] = t_[i_]; }}")


if __name__ == "__main__":
    main()
