"""The hand-scheduled main loops are GENERATED text (csrc/gen_gemm_w4.py -> gemm_w4_asm.inc, csrc/gen_attn128_w4.py -> attn128_w4_asm.inc,
csrc/gen_attn_bwd64.py -> attn_bwd64_asm.inc, csrc/gen_attn_bwd128.py -> attn_bwd128_asm.inc;
the Makefile regenerates them when a generator changes).  The committed .inc files must be exactly what the committed generators emit, and
the generators' own structural checks (register budgets, the in-place P compaction of the attention softmax) must hold."""
import os
import subprocess
import sys

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flow-factory_amd", "csrc")


@pytest.mark.parametrize("gen,inc", [("gen_gemm_w4.py", "gemm_w4_asm.inc"), ("gen_attn128_w4.py", "attn128_w4_asm.inc"),
                                     ("gen_attn_bwd64.py", "attn_bwd64_asm.inc"), ("gen_attn_bwd128.py", "attn_bwd128_asm.inc"),
                                     ("gen_attn_bwd64x2.py", "attn_bwd64x2_asm.inc")])
def test_committed_inc_is_what_the_generator_emits(gen, inc):
    out = subprocess.run([sys.executable, os.path.join(CSRC, gen)], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(CSRC, inc)).read(), f"{inc} is stale: run `make -C flow-factory_amd/csrc {inc}`"


def test_generated_loops_stay_inside_their_register_budgets():
    """Fixed physical registers: the GEMM loop may touch a0-a255, v120-v247 and s80-s91 only (everything else belongs to the C++ shell);
    the attention loop a0-a191, v32-v223, s80-s89; the attention-backward loops a0-a95, v16-v175, s70-s93 (head_dim 128: a0-a191, v16-v228, s36-s93).  A stray register would silently
    corrupt compiler-owned state."""
    import re
    budgets = {"gemm_w4_asm.inc": ((0, 255), (120, 247), (80, 91)), "attn128_w4_asm.inc": ((0, 191), (32, 223), (80, 89)),
               "attn_bwd64_asm.inc": ((0, 95), (16, 175), (70, 93)), "attn_bwd128_asm.inc": ((0, 191), (16, 228), (36, 93)),
               "attn_bwd64x2_asm.inc": ((0, 223), (16, 244), (70, 93))}
    for inc, (ar, vr, sr) in budgets.items():
        text = open(os.path.join(CSRC, inc)).read()
        for kind, (lo, hi) in (("a", ar), ("v", vr), ("s", sr)):
            used = set()
            for m in re.finditer(r"(?<![A-Za-z0-9_%\[])" + kind + r"\[(\d+):(\d+)\]", text):
                used.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"(?<![A-Za-z0-9_%\[])" + kind + r"(\d+)(?![\d:\]])", text):
                used.add(int(m.group(1)))
            assert used, (inc, kind)
            assert min(used) >= lo and max(used) <= hi, (inc, kind, min(used), max(used))
