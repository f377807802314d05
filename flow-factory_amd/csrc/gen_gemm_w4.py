#!/usr/bin/env python3
"""Generator of the hand-scheduled main loop of the 4-wave bf16 GEMM (csrc/gemm_w4_asm.inc, included by gemm.hip).

Why a generator: the loop is ~700 instructions of gfx950 assembly whose ORDER is the design (which filler sits behind which MFMA); hipcc would
not keep such an order (round 2's source-level `gemm_w4` lost 25 % to the compiler's placement of waits, M0 set-up and the LDS-DMA cluster:
DESIGN.md section 12.2), and its register allocator does not terminate on an asm statement with 64 accumulator operands -- so the registers
are fixed here and the C++ shell (gemm.hip: gemm_w4_kernel) only sees clobber lists.

Tile 256 x 256 x 64, 4 waves (2 x 2), one wave per SIMD, each wave a 128 x 128 register tile:
    a[0:255]      accumulators, acc[mi][ni] (16 x 16 blocks, mi = row block, ni = column block) at a[(mi * 8 + ni) * 4 ...]
    v[FB ...]     operand fragments, two k-steps (K = 32 each) double-buffered: X[mi] (activation rows), W[ni] (weight rows), 4 VGPRs each
    v_mfma_f32_16x16x32_bf16 acc[mi][ni], W[ni], X[mi], acc[mi][ni]   (operands swapped: a lane ends up with 4 consecutive output columns)
LDS: two stages of 64 KiB (X rows [256][128 B] then W rows [256][128 B], 16-byte chunks XOR-swizzled by (row >> 1) & 7 -- the image of
gemm_pp_kernel), filled by global_load_lds_dwordx4 (wave w fills rows [64 w, 64 w + 64) of both operands: 16 wave-instructions per K-tile).

Schedule per K-tile t ("interval" = barrier(t - 1) .. barrier(t), 128 MFMAs = 2048 matrix-pipe cycles):
    H1: 64 MFMAs on k-step 1 of K-tile t - 1 | 16 ds_read_b128 of K-tile t's k-step 0 | all 16 LDS-DMA loads of K-tile t + 1 (other stage)
    H2: 64 MFMAs on k-step 0 of K-tile t     | 16 ds_read_b128 of K-tile t's k-step 1 | operand base += 128 B
    s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier
Every load has >= 64 MFMAs (1024 cycles) of flight before its wait, every fragment read >= 32; one filler per MFMA gap at most.  The first
K-tile of the NEXT output tile is prefetched by the last interval (stage 0), so an output tile's prologue never waits for HBM.

Second instance (round 6), prefix W6: tile 256 x 192 x 64, 4 waves (4 x 1), each wave a 64 x 192 register tile (MI = 4 row blocks x NI = 12 column
blocks = 192 accumulators, 48 MFMAs per k-step, 4 + 12 fragment reads per k-step, 8 + 6 LDS-DMA instructions per wave and K-tile; stage = 32 KiB
of X rows + 24 KiB of W rows).  Why: grids that the 256 x 256 tile cuts into 0.75 / 1.5 rounds of 256 CUs (8192 rows x N = 1536 / 3072: SD3.5 at
B = 2, 1024^2) are whole rounds of 256 x 192 tiles with a quarter less work on every CU.  The wave tile spans the whole tile width so that the
64-column chunks of the shared epilogues (a q/k head) never straddle two waves.

usage: python gen_gemm_w4.py > gemm_w4_asm.inc      (the Makefile does this; the .inc is committed so that a build needs no Python)
"""
import sys

FB = 120                      # first fragment VGPR; fragments occupy v[FB : FB + 128)
XW = 32768                    # W rows start here inside a stage


class Tile:
    """MI x NI 16 x 16 blocks per wave; GX / GW LDS-DMA instructions (8 rows each) per wave, operand and K-tile."""
    def __init__(self, prefix, mi, ni, gx, gw, stage, x_off, read_gaps, m0_gaps, ld_gaps, adv_gap0):
        self.prefix, self.MI, self.NI, self.GX, self.GW, self.STAGE = prefix, mi, ni, gx, gw, stage
        self.x_off = x_off                      # VGPR offset of the W fragments inside a k-step's 64 fragment registers
        self.read_gaps, self.m0_gaps, self.ld_gaps, self.adv_gap0 = read_gaps, m0_gaps, ld_gaps, adv_gap0
        assert mi * 4 <= x_off and x_off + ni * 4 <= 64 and mi * ni * 4 <= 256 and mi + ni <= 16


T = None                      # the instance being emitted
# fixed scalar registers of the loop (clobbered)
S_A, S_W, S_NA, S_NW = 80, 82, 84, 86       # 64-bit bases: this tile's X / W operand (advance 128 B per K-tile), next tile's
S_CNT, S_LDS, S_M0SAVE, S_LDSW = 88, 89, 90, 91        # S_LDSW: this wave's LDS-DMA destination inside the W region (W6 only: 6 KiB per wave)


def xf(kk, mi):
    b = FB + kk * 64 + mi * 4
    return f"v[{b}:{b + 3}]"


def wf(kk, ni):
    b = FB + kk * 64 + T.x_off + ni * 4
    return f"v[{b}:{b + 3}]"


def acc(mi, ni):
    b = (mi * T.NI + ni) * 4
    return f"a[{b}:{b + 3}]"


def mfmas(kk, zero=False, ablate_mfma=False):
    """64 MFMAs of one k-step; column-block-major inside row-block pairs so that consecutive MFMAs never share an accumulator."""
    out = []
    for mi in range(T.MI):
        for ni in range(T.NI):
            c = "0" if zero else acc(mi, ni)
            out.append(f"v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {wf(kk, ni)}, {xf(kk, mi)}, {c}")
    return out


def reads(kk, stage):
    """MI + NI ds_read_b128: the fragments of k-step kk from `stage`."""
    out = []
    # W[0..NI) are needed by the first NI MFMAs, X[0] by all of them: lead with X[0], then the W fragments, then the other X
    order = [("x", 0)] + [("w", i) for i in range(T.NI)] + [("x", i) for i in range(1, T.MI)]
    for kind, i in order:
        if kind == "x":
            out.append(f"ds_read_b128 {xf(kk, i)}, %[lx{stage}{kk}] offset:{i * 2048}")
        else:
            out.append(f"ds_read_b128 {wf(kk, i)}, %[lw{stage}{kk}] offset:{i * 2048}")
    return out


def loads(stage, nxt=False):
    """GX + GW (M0 set-up, LDS-DMA) pairs: this wave's X rows and W rows of one K-tile into `stage`."""
    sa, sw = (S_NA, S_NW) if nxt else (S_A, S_W)
    out = []
    for g in range(max(T.GX, T.GW)):
        for (reg, region, s, n, base) in (("ga", 0, sa, T.GX, S_LDS), ("gw", XW, sw, T.GW, S_LDS if T.GW == T.GX else S_LDSW)):
            if g >= n:
                continue
            off = stage * T.STAGE + (region if base == S_LDS else 0) + g * 1024
            out.append((f"s_add_i32 m0, s{base}, {off}", f"global_load_lds_dwordx4 %[{reg}{g}], s[{s}:{s + 1}]"))
    return out


def advance():
    return [f"s_add_u32 s{S_A}, s{S_A}, 128", f"s_addc_u32 s{S_A + 1}, s{S_A + 1}, 0",
            f"s_add_u32 s{S_W}, s{S_W}, 128", f"s_addc_u32 s{S_W + 1}, s{S_W + 1}, 0"]


def interleave(mf, fillers):
    """fillers: dict MFMA index -> list of instructions placed right behind that MFMA."""
    out = []
    for i, m in enumerate(mf):
        out.append(m)
        out.extend(fillers.get(i, []))
    return out


# Schedules: where the fillers sit (index = MFMA of the half they follow).
#   s0  reads behind MFMAs 0, 2, ..., 30; loads (M0 set-up, LDS-DMA) behind 4j + 1 / 4j + 3: reads and loads share the first 32 gaps
#   s1  reads as s0; loads behind 32 + 2j / 33 + 2j: no gap region holds both kinds
#   s3  reads behind MFMAs 0 .. 15 (one per gap); loads behind 16 + 3j / 17 + 3j
SCHED = {
    "s0": dict(read=[2 * j for j in range(16)], m0=[4 * j + 1 for j in range(16)], ld=[4 * j + 3 for j in range(16)]),
    "s1": dict(read=[2 * j for j in range(16)], m0=[32 + 2 * j for j in range(16)], ld=[33 + 2 * j for j in range(16)]),
    "s3": dict(read=[j for j in range(16)], m0=[16 + 3 * j for j in range(16)], ld=[17 + 3 * j for j in range(16)]),
}


def sched(abl):
    if "sched" in abl:
        return SCHED[abl["sched"]]
    return dict(read=T.read_gaps, m0=T.m0_gaps, ld=T.ld_gaps)


def half1(stage_read, stage_load, do_load, nxt, abl):
    """H1: MFMAs on k-step 1 (fragments read during the previous interval) | reads of k-step 0 | the loads of the next K-tile."""
    sc = sched(abl)
    f = {}
    if not abl.get("noread"):
        for j, r in enumerate(reads(0, stage_read)):
            f.setdefault(sc["read"][j], []).append(r)
    if do_load and not abl.get("noload"):
        for j, (m0, ld) in enumerate(loads(stage_load, nxt)):
            f.setdefault(sc["m0"][j], []).append(m0)
            f.setdefault(sc["ld"][j], []).append(ld)
    return interleave(mfmas(1), f)


def half2(stage_read, adv, abl, zero=False):
    sc = sched(abl)
    f = {}
    if not abl.get("noread"):
        for j, r in enumerate(reads(1, stage_read)):
            f.setdefault(sc["read"][j], []).append(r)
    if adv:
        for j, a in enumerate(advance()):
            f.setdefault(T.adv_gap0 + 2 * j, []).append(a)
    return interleave(mfmas(0, zero=zero), f)


def sync():
    return ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_barrier"]


def interval(t_odd, do_load=True, nxt=False, abl={}):
    """One steady-state interval of K-tile t (t odd: its data sit in stage 1, the prefetch goes to stage 0)."""
    sr = 1 if t_odd else 0
    out = half1(sr, sr ^ 1, do_load, nxt, abl)
    out.append("s_waitcnt lgkmcnt(0)")
    out += half2(sr, adv=not nxt, abl=abl)
    out += sync()
    return out


def main_loop(abl={}):
    L = []
    L.append(f"s_mov_b32 s{S_M0SAVE}, m0")
    L.append(f"s_mov_b64 s[{S_A}:{S_A + 1}], %[cA]")
    L.append(f"s_mov_b64 s[{S_W}:{S_W + 1}], %[cW]")
    L.append(f"s_mov_b64 s[{S_NA}:{S_NA + 1}], %[nA]")
    L.append(f"s_mov_b64 s[{S_NW}:{S_NW + 1}], %[nW]")
    L.append(f"s_mov_b32 s{S_CNT}, %[pairs]")
    L.append(f"s_mov_b32 s{S_LDS}, %[ldsw]")
    if T.GW != T.GX:
        L.append(f"s_mov_b32 s{S_LDSW}, %[ldsww]")
    # the bases point at K-tile 0 (in stage 0 already, landed, barrier passed); K-tile 1 is the first one to load
    L += advance()
    # ---- I_0: fragments of K-tile 0, loads of K-tile 1 -> stage 1, k-step 0 with C = 0 (lgkmcnt is a 4-bit counter: never more than
    #      16 LDS reads in flight)
    L += reads(0, 0)
    sc = sched(abl)
    f = {}
    for j, r in enumerate(reads(1, 0)):
        f.setdefault(sc["read"][j], []).append(r)
    if not abl.get("noload"):
        for j, (m0, ld) in enumerate(loads(1)):
            f.setdefault(sc["m0"][j], []).append(m0)
            f.setdefault(sc["ld"][j], []).append(ld)
    L.append("s_waitcnt lgkmcnt(0)")
    L += interleave(mfmas(0, zero=True), f)
    L += advance()                                        # (behind the last load that uses the K-tile-1 base)
    L += sync()
    # ---- pairs of intervals (t odd, t even), `pairs` = K / 128 - 1 times
    L.append(f"s_cmp_eq_u32 s{S_CNT}, 0")
    L.append(f"s_cbranch_scc1 L_{T.prefix}_tail%=")
    L.append(f"L_{T.prefix}_loop%=:")
    L += interval(True, abl=abl)
    L += interval(False, abl=abl)
    L.append(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 1")
    L.append(f"s_cmp_lg_u32 s{S_CNT}, 0")
    L.append(f"s_cbranch_scc1 L_{T.prefix}_loop%=")
    L.append(f"L_{T.prefix}_tail%=:")
    # ---- last interval (t = K / 64 - 1, odd): prefetches K-tile 0 of the NEXT output tile into stage 0
    L += interval(True, nxt=True, abl=abl)
    # ---- k-step 1 of the last K-tile
    L += mfmas(1)
    L.append(f"s_mov_b32 m0, s{S_M0SAVE}")
    # MFMA results are read by v_accvgpr_read right behind this block: the hardware does not interlock an XDL write against a VALU read of the
    # same register (the compiler would insert these wait states itself; inside an asm statement nobody does)
    L.append("s_nop 15")
    L.append("s_nop 15")
    return L


def emit(name, lines):
    print(f"#define {name} \\")
    for ln in lines:
        print(f'    "{ln}\\n" \\')
    print('    ""')
    print()


W4 = Tile("w4", 8, 8, 8, 8, 65536, 32, SCHED["s0"]["read"], SCHED["s0"]["m0"], SCHED["s0"]["ld"], 33)
# 48 MFMAs per half: 16 reads behind MFMAs 0, 2, ..., 30; the 14 (M0 set-up, LDS-DMA) pairs behind 3j + 1 / 3j + 2 (last: 41); base advance from 40
W6 = Tile("w6", 4, 12, 8, 6, 57344, 16, [2 * j for j in range(16)], [3 * j + 1 for j in range(14)], [3 * j + 2 for j in range(14)], 40)


def main():
    global T
    print("// GENERATED by gen_gemm_w4.py -- do not edit.  Main loops of gemm_w4_kernel / gemm_w6_kernel (gemm.hip); see the generator's docstring.")
    print(f"#define W4_FRAG_BASE {FB}")
    T = W4
    emit("W4_LOOP_ASM", main_loop())
    emit("W4_LOOP_ASM_NOLOAD", main_loop({"noload": True}))
    emit("W4_LOOP_ASM_NOREAD", main_loop({"noread": True}))
    emit("W4_LOOP_ASM_MFMA_ONLY", main_loop({"noread": True, "noload": True}))
    emit("W4_LOOP_ASM_S1", main_loop({"sched": "s1"}))
    emit("W4_LOOP_ASM_S3", main_loop({"sched": "s3"}))
    T = W6
    emit("W6_LOOP_ASM", main_loop())
    cl = [f'"a{i}"' for i in range(256)] + [f'"v{i}"' for i in range(FB, FB + 128)] + [f'"s{i}"' for i in range(80, 92)] + ['"scc"', '"memory"']
    print("#define W4_CLOBBERS " + ", ".join(cl))
    print()
    # accumulator read-out: W4_READ_CHUNK_q_c(a2) fills f32x4 a2[2][4] with the 32 x 64 chunk (row quarter q, column half c) of the wave tile
    print("#define W4_READ_ACC_(d, r0, r1, r2, r3) { float t0_, t1_, t2_, t3_; asm volatile(\"v_accvgpr_read_b32 %0, a\" #r0 \"\\n v_accvgpr_read_b32 %1, a\" #r1 \"\\n"
          " v_accvgpr_read_b32 %2, a\" #r2 \"\\n v_accvgpr_read_b32 %3, a\" #r3 : \"=v\"(t0_), \"=v\"(t1_), \"=v\"(t2_), \"=v\"(t3_)); d = (f32x4){t0_, t1_, t2_, t3_}; }")
    for tile, name, nq, nc in ((W4, "W4", 4, 2), (W6, "W6", 2, 3)):
        T = tile
        for q in range(nq):             # 32-row part of the wave tile
            for c in range(nc):         # 64-column part
                body = []
                for i in range(2):
                    for j in range(4):
                        b = ((q * 2 + i) * T.NI + (c * 4 + j)) * 4
                        body.append(f"W4_READ_ACC_(a2[{i}][{j}], {b}, {b + 1}, {b + 2}, {b + 3})")
                print(f"#define {name}_READ_CHUNK_{q}_{c}(a2) " + " ".join(body))


if __name__ == "__main__":
    main()
