#!/usr/bin/env python3
"""Generator of the hand-scheduled key loop of the 4-wave head_dim-128 attention (csrc/attn128_w4_asm.inc, included by attention128.hip).

Static-softmax flash attention (|score| <= 60 proven by the caller: no running max), one wave per SIMD, each wave TWO chains of 32 queries
(one query per lane, like attn128_kernel) so that the softmax of one chain runs on the VALU under the MFMAs of the other:

    per 64-key tile and wave:   S_c^T = K . Q_c^T   16 MFMAs (2 key blocks x 8 k-steps)          c = 0, 1
                                P_c   = exp2(S_c)   32 v_exp + 32 v_add (row sums) + 16 v_cvt_pk (in-place compaction: P lives in S's registers)
                                O_c^T += V^T . P_c  16 MFMAs (4 key steps x 4 d blocks)
    MFMA order across tiles:    S0(t) | PV1(t-1) | S1(t) | [sync] | PV0(t) | S0(t+1) | ...          (four phases of 16 MFMAs per tile)
    softmax0(t) sits behind the MFMAs of PV1(t-1) and S1(t), softmax1(t) behind PV0(t) and S0(t+1): 2.5 VALU per 32-cycle MFMA gap
    (MI355X_MICROARCH.md: <= 5 single-issue fillers hide in a v_mfma_f32_32x32x16 gap with one wave per SIMD).
    Every phase reads the 16 operand fragments of the NEXT phase (one ds_read_b128 per gap) into the other of two 64-VGPR fragment buffers.

Registers (fixed; the C++ shell sees clobber lists only -- see gen_gemm_w4.py for why):
    a[0:127]    O accumulators, O[c][db] at a[(c*4 + db)*16 ...]          a[128:191]  Q fragments (MFMA B operand), Q[c][kk] at a[128 + (c*8 + kk)*4 ...]
    v[32:95]    S / P, S[c][kb] at v[32 + (c*2 + kb)*16 ...]              v[96:159] fragment buffer X, v[160:223] buffer Y
LDS (dynamic base must be 0): [0, 64 KiB) V^T ring, [64, 128 KiB) K ring, 4 stages of 16 KiB each (stage t & 3).  K / V^T tiles arrive by
global_load_lds_dwordx4 (4 + 4 wave-instructions per wave and tile) issued TWO tiles ahead (phase 1 of tile t loads tile t + 2; the shell
stages tiles 0 and 1); the sync between phases 3 and 4 waits with vmcnt(8) for tile t + 1 only, so every load has ~1.75 tiles of flight; the layouts (two 64x64 K sub-tiles, 128-byte rows, 16-byte chunks XOR-swizzled) are attn128_kernel's.
The LAST tile masks keys >= S_kv: v_cmp_gt_i32 / v_cndmask per score register against (valid keys in the tile - 8 * half-wave).

Emitted macros: A128_BODY_MULTI (n_tiles >= 3: first tile, `mid` = n_tiles - 3 loop iterations, next-to-last, last), A128_BODY_TWO, A128_BODY_SINGLE,
A128_CLOBBERS, A128_WRITE_Q(c, kk, frag), A128_READ_O(c, db, dst).
usage: python gen_attn128_w4.py > attn128_w4_asm.inc
"""

S_BASE, FX, FY = 32, 96, 160
O_BASE, Q_BASE = 0, 128
# fixed scalars
S_KSRC, S_VSRC, S_MK, S_MV, S_CNT, S_M0SAVE, S_TMP = 80, 82, 84, 85, 86, 87, 88


def sreg(c, kb, r=0, n=16):
    b = S_BASE + (c * 2 + kb) * 16 + r
    return f"v[{b}:{b + n - 1}]" if n > 1 else f"v{b}"


def oreg(c, db):
    b = O_BASE + (c * 4 + db) * 16
    return f"a[{b}:{b + 15}]"


def qreg(c, kk):
    b = Q_BASE + (c * 8 + kk) * 4
    return f"a[{b}:{b + 3}]"


def frag(buf, slot):
    b = buf + slot * 4
    return f"v[{b}:{b + 3}]"


def mfma_s(c, buf):
    """S_c = K . Q_c^T: 16 MFMAs, key blocks alternate so that dependent MFMAs are two apart"""
    out = []
    for kk in range(8):
        for kb in range(2):
            cin = "0" if kk == 0 else sreg(c, kb)
            out.append(f"v_mfma_f32_32x32x16_bf16 {sreg(c, kb)}, {frag(buf, kk * 2 + kb)}, {qreg(c, kk)}, {cin}")
    return out


def mfma_pv(c, buf):
    """O_c += V^T . P_c: key step cc (16 keys) x d block db; P fragment cc = registers (cc & 1) * 4 .. + 3 of S[c][cc >> 1] (compacted)"""
    out = []
    for cc in range(4):
        pb = S_BASE + (c * 2 + (cc >> 1)) * 16 + (cc & 1) * 4
        for db in range(4):
            out.append(f"v_mfma_f32_32x32x16_bf16 {oreg(c, db)}, {frag(buf, cc * 4 + db)}, v[{pb}:{pb + 3}], {oreg(c, db)}")
    return out


def reads_k(buf):
    return [f"ds_read_b128 {frag(buf, kk * 2 + kb)}, %[vk{kk & 3}] offset:{(kk >> 2) * 8192 + kb * 4096}" for kk in range(8) for kb in range(2)]


def reads_v(buf, which):
    return [f"ds_read_b128 {frag(buf, cc * 4 + db)}, %[{which}{cc}] offset:{db * 4096}" for cc in range(4) for db in range(4)]


def softmax(c, masked):
    """VALU stream of one chain's softmax in dependency order; entries are lists of instructions that may share a gap.
    Row sums are plain f32 adds of the exponentials (two partial sums).  Measured alternative: v_dot2_f32_bf16 of the packed pairs against
    (1, 1) -- half the instructions, but 1190 instead of 1274 TFLOP/s (profiles/r03p_*): beside MFMAs a dot2 costs far more than its issue slot
    (MI355X_MICROARCH.md prices it at ~10 cycles as a filler)."""
    ops = []
    for r in range(32):
        kb, rr = r >> 4, r & 15
        reg = sreg(c, kb, rr, 1)
        grp = []
        if masked:
            kconst = 32 * kb + 16 * (rr >> 3) + (rr & 7)
            grp += [f"v_cmp_gt_i32 vcc, %[vrem], {kconst}", f"v_cndmask_b32 {reg}, %[vneg], {reg}, vcc"]
        grp.append(f"v_exp_f32 {reg}, {reg}")
        ops.append(grp)
        if r >= 1:                               # row sum of the PREVIOUS register (its exp has had a gap to complete); two partial sums
            pk, prr = (r - 1) >> 4, (r - 1) & 15
            ops.append([f"v_add_f32 %[l{c}{(r - 1) & 1}], %[l{c}{(r - 1) & 1}], {sreg(c, pk, prr, 1)}"])
        if r >= 3 and (r & 1):                   # pack the pair (r - 3, r - 2): both summed already
            i = (r - 3) >> 1
            kbp, ii = i >> 3, i & 7
            ops.append([f"v_cvt_pk_bf16_f32 {sreg(c, kbp, ii, 1)}, {sreg(c, kbp, 2 * ii, 1)}, {sreg(c, kbp, 2 * ii + 1, 1)}"])
    ops.append([f"v_add_f32 %[l{c}1], %[l{c}1], {sreg(c, 1, 15, 1)}"])
    ops.append([f"v_cvt_pk_bf16_f32 {sreg(c, 1, 7, 1)}, {sreg(c, 1, 14, 1)}, {sreg(c, 1, 15, 1)}"])
    return ops


def check_softmax():
    """the in-place compaction never overwrites a value that is still to be read"""
    for c in (0, 1):
        live = {(kb, rr): "s" for kb in range(2) for rr in range(16)}      # s: score, e: exp'd, a: added, p: packed-away, P: packed pair
        for grp in softmax(c, False):
            for ins in grp:
                f = ins.replace(",", "").split()
                num = lambda tok: int(tok[1:]) - S_BASE - c * 32
                if f[0] == "v_exp_f32":
                    n = num(f[1]); assert live[(n >> 4, n & 15)] == "s"; live[(n >> 4, n & 15)] = "e"
                elif f[0] == "v_add_f32":
                    n = num(f[3]); assert live[(n >> 4, n & 15)] == "e", (ins, live[(n >> 4, n & 15)]); live[(n >> 4, n & 15)] = "a"
                elif f[0] == "v_cvt_pk_bf16_f32":
                    d, a, b = num(f[1]), num(f[2]), num(f[3])
                    assert live[(a >> 4, a & 15)] == "a" and live[(b >> 4, b & 15)] == "a", ins
                    assert live[(d >> 4, d & 15)] == "p" or d in (a, b), ins      # the destination's old value has been consumed
                    live[(a >> 4, a & 15)] = live[(b >> 4, b & 15)] = "p"
                    live[(d >> 4, d & 15)] = "P"
        assert sorted(k for k, v in live.items() if v == "P") == [(kb, i) for kb in range(2) for i in range(8)], live


def spread(ops, gaps):
    """assign the op groups to the gaps in order, as evenly as possible; returns {gap: [instructions]}"""
    out = {g: [] for g in gaps}
    n, m = len(ops), len(gaps)
    for i, grp in enumerate(ops):
        out[gaps[i * m // n]] += grp
    return out


def loads(stage_sym):
    """4 K + 4 V^T LDS-DMA pieces of the NEXT tile (this wave's groups w, w + 4, w + 8, w + 12)"""
    out = []
    for i in range(4):
        out.append((f"s_add_i32 m0, s{S_MK}, {i * 4096}", f"global_load_lds_dwordx4 %[gk{i}], s[{S_KSRC}:{S_KSRC + 1}]"))
        out.append((f"s_add_i32 m0, s{S_MV}, {i * 4096}", f"global_load_lds_dwordx4 %[gv{i}], s[{S_VSRC}:{S_VSRC + 1}]"))
    return out


def rotate_after_loads():
    """source bases += one tile; LDS-DMA destinations -> next stage (K ring 2 stages at 64 KiB, V ring 4 stages at 0)"""
    return [f"s_add_u32 s{S_KSRC}, s{S_KSRC}, 16384", f"s_addc_u32 s{S_KSRC + 1}, s{S_KSRC + 1}, 0",
            f"s_add_u32 s{S_VSRC}, s{S_VSRC}, 128", f"s_addc_u32 s{S_VSRC + 1}, s{S_VSRC + 1}, 0",
            f"s_add_i32 s{S_MK}, s{S_MK}, 0x4000", f"s_and_b32 s{S_MK}, s{S_MK}, 0xffff", f"s_or_b32 s{S_MK}, s{S_MK}, 0x10000",
            f"s_add_i32 s{S_MV}, s{S_MV}, 0x4000", f"s_and_b32 s{S_MV}, s{S_MV}, 0xffff"]


def phase(mf, fill):
    out = []
    for i, m in enumerate(mf):
        out.append(m)
        out += fill.get(i, [])
    return out


def merge(*dicts):
    out = {}
    for d in dicts:
        for k, v in d.items():
            out.setdefault(k, [])
            out[k] += v
    return out


HAZ = "s_nop 7"          # an XDL write must be >= 12 wait states ahead of a VALU read of the same register (hipcc inserts s_nop 11 there)


ABL = set()        # ablation builds (timing only, results garbage): "novalu", "noread"
RGAPS = 16        # the 16 fragment reads of a phase are spread over its first RGAPS gaps (set per emitted variant)


def rd(reads):
    out = {}
    if "noread" in ABL:
        return out
    for i, r in enumerate(reads):
        out.setdefault(i * RGAPS // 16, []).append(r)
    return out


def tile(first, last, has_loads=None):
    """one tile; returns (instructions, pending) where `pending` = {gap: [...]} fillers for phase 1 of the NEXT tile (softmax1's second half)"""
    L = []
    if has_loads is None:
        has_loads = not last
    assert not (last and has_loads)
    sm0, sm1 = softmax(0, last), softmax(1, last)
    if "novalu" in ABL:
        sm0, sm1 = [["s_nop 0"]], [["s_nop 0"]]
    gaps16 = list(range(16))
    # ---- phase 1: S0(t) on buffer X | reads V(t-1) -> Y | loads of tile t + 1 | (softmax1(t-1) tail: emitted by the caller as `carry`)
    f1 = {}
    if not first:
        f1 = rd(reads_v(FY, "vp"))
    if has_loads and "noload" not in ABL:
        ld = {}
        for j, (m0, g) in enumerate(loads(None)):
            ld.setdefault(2 * j, []).append(m0)
            ld.setdefault(2 * j + 1, []).append(g)
        rot1 = rotate_after_loads()                  # behind the last load (gap 15); SALU only
        f1 = merge(f1, ld, {15: rot1})
    L.append(("ph1", mfma_s(0, FX), f1))
    # ---- softmax0(t) window: phases 2 + 3 (or phase 3 alone on the first tile); nothing in its first two gaps (MFMA -> VALU hazard)
    if not first:
        win = [("ph2", g) for g in range(2, 16)] + [("ph3", g) for g in range(0, 14)]
    else:
        win = [("ph3", g) for g in range(2, 15)]
    sp0 = spread(sm0, win)
    first_gap = win[0]
    sp0[first_gap] = [HAZ] + sp0[first_gap]
    # ---- phase 2: PV1(t-1) on Y | reads K(t) -> X
    if not first:
        f2 = {g: v for (p, g), v in sp0.items() if p == "ph2"}       # (no reads: buffer X still holds the K(t) fragments phase 3 needs)
        L.append(("ph2", mfma_pv(1, FY), f2))
    # ---- phase 3: S1(t) on X | reads V(t) -> Y | K read bases -> other stage (for the K(t+1) reads of phase 4)
    f3 = merge(rd(reads_v(FY, "vc")), {g: v for (p, g), v in sp0.items() if p == "ph3"})
    if not last:
        # the K read bases move to the next stage behind the tile's last K read (phase 4 of the PREVIOUS tile read K(t)): any gap of phase 3
        f3 = merge(f3, {4 + 3 * k: [f"v_add_u32 %[vk{k}], 0x4000, %[vk{k}]", f"v_and_b32 %[vk{k}], 0xffff, %[vk{k}]",
                                    f"v_or_b32 %[vk{k}], 0x10000, %[vk{k}]"] for k in range(4)})
    L.append(("ph3", mfma_s(1, FX), f3))
    if not last and "nosync" not in ABL:
        # tile t + 1 (issued one iteration ago, or by the shell) must have landed; this iteration's 8 loads (tile t + 2) stay in flight
        L.append(("sync", [f"s_waitcnt vmcnt({8 if has_loads else 0}) lgkmcnt(0)", "s_barrier"], {}))
    # ---- phase 4: PV0(t) on Y | reads K(t+1) -> X | V read bases rotate | softmax1(t) first half
    if not last:
        win1 = [("ph4", g) for g in range(2, 16)] + [("nx1", g) for g in range(0, 14)]
    else:
        win1 = [("ph4", g) for g in range(2, 16)]
    sp1 = spread(sm1, win1)
    sp1[win1[0]] = [HAZ] + sp1[win1[0]]
    f4 = {g: v for (p, g), v in sp1.items() if p == "ph4"}
    if not last:
        # V read bases: vp <- vc, vc <- next stage; phase 3 made the last read through vc, phase 1 of the next tile reads through vp
        rot4 = {3 + 3 * k: [f"v_mov_b32 %[vp{k}], %[vc{k}]", f"v_add_u32 %[vc{k}], 0x4000, %[vc{k}]", f"v_and_b32 %[vc{k}], 0xffff, %[vc{k}]"] for k in range(4)}
        f4 = merge(rd(reads_k(FX)), f4, rot4)
    L.append(("ph4", mfma_pv(0, FY), f4))
    pending = {g: v for (p, g), v in sp1.items() if p == "nx1"}
    if last:
        L.append(("ph5", ["s_nop 3"] + mfma_pv(1, FY), {}))        # V(t) fragments are still in Y; softmax1 finished inside phase 4
    return L, pending


def flatten(L, carry):
    out = []
    for name, mf, fill in L:
        if name == "sync":
            out += mf
            continue
        if name == "ph1":
            fill = merge(fill, carry)
        if name == "ph5":
            out += mf
            continue
        out.append("s_waitcnt lgkmcnt(0)")          # the fragments this phase multiplies were read during the previous one
        out += phase(mf, fill)
    return out


def prologue():
    return [f"s_mov_b32 s{S_M0SAVE}, m0", f"s_mov_b64 s[{S_KSRC}:{S_KSRC + 1}], %[ksrc]", f"s_mov_b64 s[{S_VSRC}:{S_VSRC + 1}], %[vsrc]",
            f"s_mov_b32 s{S_MK}, %[mk]", f"s_mov_b32 s{S_MV}, %[mv]", f"s_mov_b32 s{S_CNT}, %[mid]"] + \
           [f"v_accvgpr_write_b32 a{i}, 0" for i in range(128)] + ["v_mov_b32 %[l00], 0", "v_mov_b32 %[l01], 0", "v_mov_b32 %[l10], 0", "v_mov_b32 %[l11], 0"] + \
           reads_k(FX)                                  # K(0) fragments for the very first phase (exposed once per workgroup)


def epilogue():
    return [f"s_mov_b32 m0, s{S_M0SAVE}", "s_nop 15", "s_nop 15"]


def body_multi():
    """n_tiles >= 3: first tile (loads tile 2), `mid` = n_tiles - 3 loop iterations, the tile before the last (no loads left), the last tile"""
    first, pend_f = tile(True, False, True)
    mid, pend_m = tile(False, False, True)
    pen, pend_p = tile(False, False, False)
    last, _ = tile(False, True)
    assert pend_f == pend_m == pend_p
    out = prologue() + flatten(first, {})
    out += [f"s_cmp_eq_u32 s{S_CNT}, 0", "s_cbranch_scc1 L_a128_pen%=", "L_a128_loop%=:"]
    out += flatten(mid, pend_m)
    out += [f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1", f"s_cmp_lg_u32 s{S_CNT}, 0", "s_cbranch_scc1 L_a128_loop%=", "L_a128_pen%=:"]
    out += flatten(pen, pend_m)
    out += flatten(last, pend_m)
    return out + epilogue()


def body_two():
    """n_tiles == 2 (both tiles staged by the shell)"""
    first, pend = tile(True, False, False)
    last, _ = tile(False, True)
    return prologue() + flatten(first, {}) + flatten(last, pend) + epilogue()


def body_single():
    only, _ = tile(True, True)
    return prologue() + flatten(only, {}) + epilogue()


def emit(name, lines):
    print(f"#define {name} \\")
    for ln in lines:
        print(f'    "{ln}\\n" \\')
    print('    ""')
    print()


def main():
    check_softmax()
    print("// GENERATED by gen_attn128_w4.py -- do not edit.  Key loop of attn128_w4_kernel (attention128.hip); see the generator's docstring.")
    global RGAPS, ABL
    RGAPS = 16
    emit("A128_BODY_MULTI", body_multi())
    emit("A128_BODY_TWO", body_two())
    emit("A128_BODY_SINGLE", body_single())
    for name, ab in (("NOVALU", {"novalu"}), ("NOREAD", {"noread"}), ("MFMA", {"novalu", "noread"}), ("MFMA_NOSYNC", {"novalu", "noread", "nosync", "noload"})):
        ABL = ab
        emit(f"A128_BODY_MULTI_{name}", body_multi())
    ABL = set()
    cl = [f'"a{i}"' for i in range(192)] + [f'"v{i}"' for i in range(S_BASE, FY + 64)] + [f'"s{i}"' for i in range(80, 90)] + ['"vcc"', '"scc"', '"memory"']
    print("#define A128_CLOBBERS " + ", ".join(cl))
    print()
    print("#define A128_WQ_(x, r0, r1, r2, r3) asm volatile(\"v_accvgpr_write_b32 a\" #r0 \", %0\\n v_accvgpr_write_b32 a\" #r1 \", %1\\n"
          " v_accvgpr_write_b32 a\" #r2 \", %2\\n v_accvgpr_write_b32 a\" #r3 \", %3\" : : \"v\"(x[0]), \"v\"(x[1]), \"v\"(x[2]), \"v\"(x[3]))")
    for c in range(2):
        for kk in range(8):
            b = Q_BASE + (c * 8 + kk) * 4
            print(f"#define A128_WRITE_Q_{c}_{kk}(x) A128_WQ_(x, {b}, {b + 1}, {b + 2}, {b + 3})")
    print("#define A128_RO_(d, i, r) { float t_; asm volatile(\"v_accvgpr_read_b32 %0, a\" #r : \"=v\"(t_)); d[i] = t_; }")
    for c in range(2):
        for db in range(4):
            b = O_BASE + (c * 4 + db) * 16
            print(f"#define A128_READ_O_{c}_{db}(d) " + " ".join(f"A128_RO_(d, {i}, {b + i})" for i in range(16)))


if __name__ == "__main__":
    main()
