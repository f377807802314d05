"""TEST INFRASTRUCTURE: run the fp32 oracles (oracle/*.py, plain PyTorch) ON THE GPU in fp32.

The oracles are device-agnostic restatements; on the host cores one SD3.5 forward at 1024^2 costs ~25 s, a FLUX.1-dev / Qwen-Image forward
at full depth minutes -- which is why rounds 1-5 compared the headline configurations only at 4 steps / 1-2 blocks.  gfx950 has no TF32 /
xf32 path (fp32 GEMMs run `v_mfma_f32_32x32x2_f32`, exact fp32; MI355X_MICROARCH.md "Peak FP32 (matrix)"), so the same oracle on `cuda`
tensors is an fp32 checker that costs seconds.  Nothing in the product path imports this (or `oracle/`).

  * `on_gpu()`: no-grad + default device cuda (the oracles' `torch.arange` / `torch.zeros` / `torch.tensor` factories land on the GPU) +
    the MATH scaled-dot-product backend (explicit fp32 matmul / softmax / matmul: no flash / memory-efficient kernel inside the checker);
  * `F32View`: a state dict of bf16 GPU tensors that hands out fp32 copies on access (the family oracles use the weights as they come;
    a 20 B-parameter model never has its fp32 copy resident);
  * `ulp_fraction`: SURVEY.md 8(d)'s second criterion -- the fraction of elements within one storage-dtype ulp of the oracle's value.
"""
import contextlib

import torch


@contextlib.contextmanager
def on_gpu(grad: bool = False):
    from torch.nn.attention import SDPBackend, sdpa_kernel
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.set_float32_matmul_precision("highest")
    with contextlib.ExitStack() as st:
        if not grad:
            st.enter_context(torch.no_grad())
        st.enter_context(torch.device("cuda"))
        st.enter_context(sdpa_kernel(SDPBackend.MATH))
        yield


class F32View(dict):
    """name -> bf16 (or any dtype) GPU tensor; `view[name]` is its fp32 value."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()


def cuda(x):
    if isinstance(x, torch.Tensor):
        return x.cuda()
    if isinstance(x, dict):
        return {k: cuda(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(cuda(v) for v in x)
    return x


def rel(a, b):
    a, b = a.float().cuda(), b.float().cuda()
    return float((a - b).norm() / (b.norm() + 1e-12))


def ulp_fraction(got: torch.Tensor, ref: torch.Tensor, dtype: torch.dtype, n_ulp: float = 1.0) -> float:
    """Fraction of elements of `got` (values representable in `dtype`) within `n_ulp` units in the last place of `dtype` AT the oracle's
    value `ref` (fp32).  ulp(x) = 2^(floor(log2 |x|) - mantissa_bits), floored at the smallest normal's ulp."""
    mant = {torch.float16: 10, torch.bfloat16: 7, torch.float32: 23}[dtype]
    r = ref.float().cuda()
    g = got.float().cuda()
    tiny = torch.finfo(dtype).tiny
    e = torch.floor(torch.log2(r.abs().clamp_min(tiny)))
    ulp = torch.exp2(e - mant)
    return float(((g - r).abs() <= n_ulp * ulp).float().mean())


def oracle_loss_on(device, fn, mod, cfg_o, inp, *a, **kw):
    """Run a test module's `_oracle_loss_impl(mod, cfg_o, inp, ...)` (autograd through the fp32 oracle) on the host cores (`device="cpu"`) or on
    the GPU in fp32 (`"cuda"`: the inputs are moved, factories default to the GPU, MATH attention; the implementation builds its weight leaves
    on `inp["x"].device`).  Returns (log_prob on the host, {name: gradient})."""
    if device == "cpu":
        return fn(mod, cfg_o, inp, *a, **kw)
    with on_gpu(grad=True):
        lp, gr = fn(mod, cfg_o, cuda(dict(inp)), *a, **kw)
    return lp.cpu(), gr


def bf16_round(x):
    return x.to(torch.bfloat16).float()


def ref_on_gpu(fn, *args, **kw):
    """`fn(*args, **kw)` with every tensor / dict of tensors among the arguments moved to the GPU, under `on_gpu()`; the result stays there."""
    with on_gpu():
        return fn(*[cuda(a) for a in args], **{k: cuda(v) for k, v in kw.items()})


def check_in_band(name, got, oracle_fn, *args, factor=1.5, floor=1e-3, **kw):
    """The parity statement every full-width forward test makes since round 6: the engine's output `got` is no further
    from the fp32 oracle than `factor` x the BAND + `floor`, the band being the same oracle with a bf16 round-trip wherever the reference's bf16
    module materialises a tensor (`quant=bf16_round`) against its own fp32 self -- both oracle runs on the GPU in fp32.  Measured everywhere so
    far: engine = 1.0 x band.  Returns (ref, refq, rel-L2 to fp32, band)."""
    ref = ref_on_gpu(oracle_fn, *args, **kw)
    refq = ref_on_gpu(oracle_fn, *args, quant=bf16_round, **kw)
    band, r = rel(refq, ref), rel(got, ref)
    print(f"{name}: engine vs fp32 oracle {r:.3e}; bf16-emulating oracle vs fp32 (band) {band:.3e} ({r / max(band, 1e-30):.2f} x band); "
          f"engine vs bf16-emulating {rel(got, refq):.3e}")
    assert torch.isfinite(got.float()).all(), name
    assert r < factor * band + floor, (name, r, band)
    return ref, refq, r, band
