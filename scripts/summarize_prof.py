"""Condense rocprofv3 output (kernel stats CSV + counter_collection CSVs) into small text/JSON summaries."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
# optional second argument: the sub-directory (below `out`) whose kernel-stats CSV is summarised.  Round 3's evidence script left the
# single-stream run's directory beside the two-stream one and this script read "the first match": its two tables were the same table.
stats_sub = sys.argv[2] if len(sys.argv) > 2 else None


def short(name):
    for key, tag in (("gemm_tn256_kernel", "gemm_tn<256x256,row-major wgrad>"), ("gemm_tn_kernel", "gemm_tn<128x128,row-major wgrad>"),
                     ("attn_bwd_dkv_pipe_kernel", "attn_bwd_dkv_pipe_kernel"), ("attn_bwd_dq_pipe_kernel", "attn_bwd_dq_pipe_kernel"), ("splitk_reduce_colsum_kernel", "splitk_reduce_colsum_kernel"),
                     ("splitk_reduce_kernel", "splitk_reduce_kernel"), ("colsum_partial_kernel", "colsum_partial_kernel"),
                     ("colsum_finish_kernel", "colsum_finish_kernel"), ("transpose_kernel", "transpose_kernel"), ("ln_mod_bwd_kernel", "ln_mod_bwd_kernel"),
                     ("rms_bwd_gather", "rms_bwd_gather_kernel"), ("attn_bwd_prep", "attn_bwd_prep_kernel"), ("gate_mul_kernel", "gate_mul_kernel"),
                     ("attn_bwd_dkv_tr_kernel", "attn_bwd_dkv_tr_kernel"), ("attn_bwd_dq_tr_kernel", "attn_bwd_dq_tr_kernel"), ("attn_kernel", "attn_kernel"),
                     ("gemm_mid_kernel", None), ("gemm_w6_kernel", None), ("gemm_w4_kernel", None), ("gemm_pp_kernel", None), ("gemm_kernel", None), ("ln_mod_kernel", "ln_mod_kernel"),
                     ("sde_step_kernel", "sde_step_kernel"), ("patchify", "patchify_kernel"), ("time_proj", "time_proj_kernel"),
                     ("convert_kernel", "convert_kernel"), ("pos_crop", "pos_crop_kernel"), ("gn_partial", "gn_partial_kernel"),
                     ("gn_finalize", "gn_finalize_kernel"), ("gn_apply", "gn_apply_kernel"), ("softmax_rows", "softmax_rows_kernel"),
                     ("vae_ingest", "vae_ingest_kernel"), ("conv_repack", "conv_repack_kernel")):
        if key in name:
            if tag:
                return tag
            # gemm_kernelILi256ELi256ELi2ELi4ELi5EE -> gemm<256,256,epi5>
            import re
            epi = ["bias", "bias_silu", "bias_gelu", "posadd", "addsrc_silu", "gate_res", "qk_norm", "vT", "unpatch", "bias_row",
                   "f32", "img", "dgelu", "qk_norm_rstd"]
            mm = re.search(r"gemm_mid_kernel<(\d+), (\d+), (\d+)>", name)
            if mm:
                return f"gemm_mid<{int(mm.group(1)) * 32}x{int(mm.group(2)) * 32},{epi[int(mm.group(3))]}>"
            m6 = re.search(r"gemm_w6_kernel<(\d+)>", name)
            if m6:
                return f"gemm_w6<256x192,{epi[int(m6.group(1))]}>"
            m4 = re.search(r"gemm_w4_kernel<(\d+)(?:, \d+)?>", name)
            if m4:
                return f"gemm_w4<256x256,{epi[int(m4.group(1))]}>"
            mp = re.search(r"gemm_pp_kernel<(\d+)(?:, \d+)?>", name)
            if mp:
                return f"gemm_pp<256x256,{epi[int(mp.group(1))]}>"
            m = re.search(r"gemm_kernelILi(\d+)ELi(\d+)ELi\d+ELi\d+ELi(\d+)E", name) or \
                re.search(r"gemm_kernel<(\d+), (\d+), \d+, \d+, (\d+)", name)
            conv = "conv" if ("true>" in name or "ELb1EE" in name) else "gemm"
            return f"{conv}<{m.group(1)}x{m.group(2)},{epi[int(m.group(3))]}>" if m else "gemm"
    return name[:60]


stats = sorted(glob.glob(os.path.join(out, stats_sub, "**", "*kernel_stats*.csv") if stats_sub else os.path.join(out, "**", "*kernel_stats*.csv"),
                        recursive=True))
if stats_sub and len(stats) != 1:
    print(f"expected exactly one kernel-stats CSV below {os.path.join(out, stats_sub)}, found {len(stats)}: {stats}")
    stats = stats[:1]
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    print("== rocprofv3 --kernel-trace --stats (bench.py --steps 1 --warmup 1):", os.path.basename(stats[0]))
    print(f"{'kernel':34s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:25]:
        print(f"{short(r['Name']):34s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")

summary = {}
for d in sorted(glob.glob(os.path.join(out, "prof_pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
    if not files:
        print("no counter csv in", d)
        continue
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
    print("== PMC", os.path.basename(d))
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:40]:
        per = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
        print(f"  {k:34s} launches={max(cnt[k].values()):6d} per-launch: " + ", ".join(f"{c}={v:.4g}" for c, v in per.items()))
        summary.setdefault(k, {}).update({c: v for c, v in per.items()})
        summary[k]["launches"] = max(cnt[k].values())
for k, v in summary.items():
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE"):
        v["mfma_busy"] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    if v.get("SQ_LDS_IDX_ACTIVE") and v.get("SQ_LDS_BANK_CONFLICT") is not None:
        v["lds_conflict_share_of_lds_cycles"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 4)
    if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAIT_INST_LDS") is not None:
        v["lds_issue_stall_share_of_wave_cycles"] = round(v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"], 4)
    if v.get("SQ_LDS_IDX_ACTIVE") and v.get("GRBM_GUI_ACTIVE"):
        # LDS-array busy cycles summed over the chip's 256 CUs against the kernel's clock cycles (GRBM_GUI_ACTIVE is summed over 8 XCDs)
        v["lds_array_busy"] = round(v["SQ_LDS_IDX_ACTIVE"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 256.0), 4)
    if v.get("FETCH_SIZE") is not None and v.get("WRITE_SIZE") is not None:
        v["fabric_bytes_per_launch"] = int(v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024)      # FETCH doubled per MI355X_MICROARCH.md (gfx950)
json.dump(summary, open(os.path.join(out, "pmc_per_launch.json"), "w"), indent=1)

# the dominant kernel's HBM traffic per launch, as bench.py's roofline.traffic reads it (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in
# KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is)
a = summary.get("attn_kernel", {})
if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
    import subprocess
    try:
        sha = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
    except Exception:
        sha = ""
    sha = sha or os.environ.get("MI355_COMMIT", "unknown")
    # MFMA pipe utilisation of the kernel: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs
    mfma_busy = None
    if a.get("SQ_VALU_MFMA_BUSY_CYCLES") and a.get("GRBM_GUI_ACTIVE"):
        mfma_busy = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (a["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    by_kernel = {}
    for k, v in summary.items():
        if v.get("SQ_VALU_MFMA_BUSY_CYCLES") and v.get("GRBM_GUI_ACTIVE") and ("gemm" in k or "attn" in k):
            by_kernel[k] = round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    json.dump({"round": int(os.environ.get("MI355_ROUND", "4")), "commit": sha, "mfma_busy": mfma_busy,
               "mfma_busy_formula": "SQ_VALU_MFMA_BUSY_CYCLES (summed over 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 * 1024), per launch",
               "mfma_busy_by_kernel": by_kernel, "kernel": "mi355::attn_kernel (mean over the joint S=4429 and dual S=4096 launches of a forward, forward batch 8)",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --steps 1 --warmup 0 --denoise-steps 2",
               "FETCH_SIZE_kb_per_launch": a["FETCH_SIZE"], "WRITE_SIZE_kb_per_launch": a["WRITE_SIZE"],
               "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> doubled; WRITE_SIZE as is",
               "hbm_bytes_per_launch": int(a["FETCH_SIZE"] * 1024 * 2 + a["WRITE_SIZE"] * 1024), "algorithmic_bytes_per_launch": 435400000},
              open(os.path.join(out, "pmc_attention.json"), "w"), indent=1)
