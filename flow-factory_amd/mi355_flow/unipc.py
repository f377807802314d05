"""UniPC multistep predictor-corrector for the EVALUATION-mode sampler of the Wan adapters.

The reference's `UniPCMultistepSDEScheduler.step` in evaluation mode (scheduler/unipc_multistep.py:282-285) is diffusers'
`UniPCMultistepScheduler.step`: `convert_model_output` (flow prediction -> x0), `multistep_uni_c_bh_update` (corrector, from the second
step on), `multistep_uni_p_bh_update` (predictor).  The solver body is not in the reference tree; its published algorithm is restated tensor by
tensor in oracle/unipc_ref.py (PARITY UNPINNED: nothing here can be checked against diffusers itself in this image).

With flow sigmas (alpha = 1 - sigma) and x0-prediction every update is LINEAR in the tensors it touches, with coefficients that depend on the
sigma schedule alone -- so the solver splits into

  * this module: the solver's order bookkeeping (warm-up order, `lower_order_final`, corrector on/off) and the coefficients, computed once per
    schedule on the host in float64 (`unipc_schedule`);
  * two streaming HIP kernels (csrc/sde_step.hip): `mi355_unipc_convert` (CFG combine + x0 = sample - sigma v) and `mi355_op_lincomb`
    (out = sum_i c_i t_i with torch's per-term dtype rounding), driven by `UniPCSampler`.

The update formulas (predict_x0, solver "bh1" / "bh2"; h = lambda_t - lambda_s0, lambda = log(alpha / sigma), hh = -h):
    UniP:  x_t = sigma_t / sigma_s0 x - alpha_t expm1(hh) m0 - alpha_t B(hh) sum_i rho_i (m_i - m0) / r_i
    UniC:  x_t = sigma_t / sigma_s0 x - alpha_t expm1(hh) m0 - alpha_t B(hh) [sum_i rho_i (m_i - m0) / r_i + rho_last (m_t - m0)]
with m0 the newest stored x0-prediction, m_i older ones, m_t the prediction at the point being corrected."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .engine import _ptr, _stream, dtype_code


@dataclass
class StepCoefs:
    """Coefficients of one solver step.  Tensor order -- corrector: [last_sample, m0 (newest stored x0), m1, ..., model_t (the new x0)];
    predictor: [sample (corrected), m0 (the new x0), m1 (previous x0), ...]."""
    order: int                                  # predictor order of this step
    corrector: Optional[List[float]]            # None on the first step / when disabled
    corrector_order: int
    predictor: List[float]


def _lam(sig: float) -> float:
    with np.errstate(divide="ignore"):
        return float(np.log(1.0 - sig) - np.log(sig))          # sigma = 0 -> +inf (the final step)


def _rhos(rks: Sequence[float], hh: float, order: int, solver_type: str, corrector: bool) -> Tuple[List[float], float, float]:
    """(rhos, expm1(hh), B(hh)) of a UniP (corrector=False: len order - 1) or UniC (len order) update."""
    h_phi_1 = float(np.expm1(hh))
    B_h = hh if solver_type == "bh1" else h_phi_1
    with np.errstate(invalid="ignore", divide="ignore"):
        h_phi_k = h_phi_1 / hh - 1.0 if np.isfinite(hh) else -1.0
    R, b = [], []
    factorial_i = 1
    for i in range(1, order + 1):
        R.append([rk ** (i - 1) for rk in rks])
        b.append(h_phi_k * factorial_i / B_h)
        factorial_i *= i + 1
        h_phi_k = (h_phi_k / hh if np.isfinite(hh) else 0.0) - 1.0 / factorial_i
    if corrector:
        rhos = [0.5] if order == 1 else np.linalg.solve(np.array(R), np.array(b)).tolist()
    else:
        rhos = [] if order == 1 else ([0.5] if order == 2 else np.linalg.solve(np.array(R)[:-1, :-1], np.array(b)[:-1]).tolist())
    return rhos, h_phi_1, B_h


def _round_rhos(rhos, rho_dtype):
    """diffusers: `rhos = torch.linalg.solve(R, b).to(device).to(x.dtype)` -- the solved weights take the SAMPLE's dtype (fp16 under the
    reference's default latent storage: 5e-4 relative) before they multiply anything; the closed-form 0.5 is exact in every dtype."""
    if rho_dtype is None or rho_dtype == torch.float32:
        return [float(np.float32(r)) for r in rhos]
    return torch.tensor(rhos, dtype=torch.float32).to(rho_dtype).double().tolist()


def unipc_schedule(sigmas: Sequence[float], solver_order: int = 2, solver_type: str = "bh2", lower_order_final: bool = True,
                   disable_corrector: Sequence[int] = (), rho_dtype: Optional[torch.dtype] = None) -> List[StepCoefs]:
    """Per-step coefficients for a schedule of N + 1 sigmas (the last one the final sigma, 0 for `final_sigmas_type="zero"`).
    `rho_dtype` = the dtype of the sample the solver steps (the latent storage dtype): the SOLVED rhos (corrector of order >= 2, predictor
    of order >= 3) are rounded to it as diffusers does; None keeps them in double (schedule-only checks)."""
    if solver_type not in ("bh1", "bh2"):
        raise ValueError(f"mi355_flow: UniPC solver_type {solver_type!r} (bh1 / bh2)")
    sig = [float(s) for s in sigmas]
    if solver_type == "bh1" and sig[-1] == 0.0:
        # B(hh) = hh = -inf on the last step: `-inf * 0` -- the published update is NaN there as well (the Wan pipelines ship bh2)
        raise NotImplementedError("mi355_flow: UniPC solver_type 'bh1' with a final sigma of 0 is undefined (B(h) = h = inf on the last step)")
    N = len(sig) - 1
    out: List[StepCoefs] = []
    lower_order_nums, prev_order = 0, 1
    for s in range(N):
        # ---- corrector (multistep_uni_c_bh_update): from sigma[s - 1] to sigma[s], order = the previous step's predictor order
        corr = None
        if s > 0 and (s - 1) not in disable_corrector:
            q = prev_order
            sigma_t, sigma_s0 = sig[s], sig[s - 1]
            alpha_t = 1.0 - sigma_t
            lam_t, lam_s0 = _lam(sigma_t), _lam(sigma_s0)
            h = lam_t - lam_s0
            rks = [(_lam(sig[s - (i + 1)]) - lam_s0) / h for i in range(1, q)] + [1.0]
            rhos, h_phi_1, B_h = _rhos(rks, -h, q, solver_type, corrector=True)
            if q >= 2 and rho_dtype is not None:
                rhos = _round_rhos(rhos, rho_dtype)
            c_m = [0.0] * q                                      # m0 (x0 at s-1), m1 (x0 at s-2), ...
            c_m[0] = -alpha_t * h_phi_1 + alpha_t * B_h * (sum(rhos[i - 1] / rks[i - 1] for i in range(1, q)) + rhos[-1])
            for i in range(1, q):
                c_m[i] = -alpha_t * B_h * rhos[i - 1] / rks[i - 1]
            corr = [sigma_t / sigma_s0] + c_m + [-alpha_t * B_h * rhos[-1]]
        # ---- predictor order (step(): lower_order_final, warm-up)
        this_order = min(solver_order, N - s) if lower_order_final else solver_order
        this_order = min(this_order, lower_order_nums + 1)
        # ---- predictor (multistep_uni_p_bh_update): from sigma[s] to sigma[s + 1]
        sigma_t, sigma_s0 = sig[s + 1], sig[s]
        alpha_t = 1.0 - sigma_t
        lam_t, lam_s0 = _lam(sigma_t), _lam(sigma_s0)
        h = lam_t - lam_s0
        with np.errstate(invalid="ignore"):
            rks = [(_lam(sig[s - i]) - lam_s0) / h for i in range(1, this_order)] + [1.0]
        if this_order > 1 and not np.isfinite(h):
            # lower_order_final=False (or solver_order > the steps left) with a final sigma of 0: h = inf, rks = 0, rhos / rks = 0 / 0 --
            # diffusers produces NaN latents there; refuse instead of dividing by zero
            raise NotImplementedError("mi355_flow: UniPC predictor of order > 1 onto a final sigma of 0 is undefined (h = inf); the published "
                                      "solver returns NaN there -- use lower_order_final=True (the Wan pipelines' setting)")
        rhos, h_phi_1, B_h = _rhos(rks, -h, this_order, solver_type, corrector=False)
        if this_order >= 3 and rho_dtype is not None:
            rhos = _round_rhos(rhos, rho_dtype)
        p_m = [0.0] * this_order
        p_m[0] = -alpha_t * h_phi_1 + alpha_t * B_h * sum(rhos[i - 1] / rks[i - 1] for i in range(1, this_order))
        for i in range(1, this_order):
            p_m[i] = -alpha_t * B_h * rhos[i - 1] / rks[i - 1]
        out.append(StepCoefs(order=this_order, corrector=corr, corrector_order=prev_order if corr is not None else 0,
                             predictor=[sigma_t / sigma_s0] + p_m))
        prev_order = this_order
        if lower_order_nums < solver_order:
            lower_order_nums += 1
    return out


# --------------------------------------------------------------------------------------------- the two kernels
def unipc_convert(v_text: torch.Tensor, v_uncond: Optional[torch.Tensor], guidance: float, sample: torch.Tensor, sigma: float) -> torch.Tensor:
    """x0 (fp32) = sample - round(sigma * v), v = CFG-combine(v_uncond, v_text) op by op in the prediction's dtype (mi355_unipc_convert)."""
    lib = _lib.load()
    v_text, sample = v_text.contiguous(), sample.contiguous()
    if v_uncond is not None:
        v_uncond = v_uncond.contiguous()
        if v_uncond.dtype != v_text.dtype or v_uncond.shape != v_text.shape:
            raise ValueError("mi355_flow: the two CFG branches must agree in dtype and shape")
    if v_text.numel() != sample.numel() or sample.numel() % 4:
        raise ValueError("mi355_flow: unipc_convert needs prediction and sample of the same size, a multiple of 4 elements")
    x0 = torch.empty(sample.shape, device=sample.device, dtype=torch.float32)
    _lib.check(lib.mi355_unipc_convert(_stream(), _ptr(v_text), _ptr(v_uncond), dtype_code(v_text.dtype), float(guidance), _ptr(sample),
                                       dtype_code(sample.dtype), float(sigma), _ptr(x0), sample.numel()), "unipc_convert")
    return x0


def lincomb(tensors: Sequence[torch.Tensor], coefs: Sequence[float], out_dtype: torch.dtype) -> torch.Tensor:
    """out = sum_i round_i(c_i * t_i) (mi355_op_lincomb; round_i = to t_i's dtype, fp32 accumulation in term order)."""
    lib = _lib.load()
    n_terms = len(tensors)
    if n_terms != len(coefs) or not 1 <= n_terms <= 5:
        raise ValueError("mi355_flow: lincomb takes 1..5 tensors and as many coefficients")
    ts = [t.contiguous() for t in tensors]
    n = ts[0].numel()
    if any(t.numel() != n for t in ts) or n % 4:
        raise ValueError("mi355_flow: lincomb operands must have the same size, a multiple of 4 elements")
    out = torch.empty(ts[0].shape, device=ts[0].device, dtype=out_dtype)
    ptrs = (C.c_void_p * n_terms)(*[_ptr(t) for t in ts])
    dts = (C.c_int * n_terms)(*[dtype_code(t.dtype) for t in ts])
    cs = (C.c_float * n_terms)(*[float(c) for c in coefs])
    _lib.check(lib.mi355_op_lincomb(_stream(), n_terms, ptrs, dts, cs, _ptr(out), dtype_code(out_dtype), n), "op_lincomb")
    return out


class UniPCSampler:
    """The solver's tensor state for one sampling run: stored x0-predictions (newest first, in torch's promoted dtype of sample and
    prediction) and the last (corrected) sample.
    `step(i, v_text, v_uncond, guidance, sample)` = diffusers' `UniPCMultistepScheduler.step` for step index i; returns the next sample in the
    sample's own dtype (`x_t.to(x.dtype)`)."""

    def __init__(self, sigmas: Sequence[float], solver_order: int = 2, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector: Sequence[int] = (), sample_dtype: Optional[torch.dtype] = None):
        self.sigmas = [float(s) for s in sigmas]
        self.solver_order = int(solver_order)
        self.coefs = unipc_schedule(self.sigmas, solver_order, solver_type, lower_order_final, disable_corrector, rho_dtype=sample_dtype)
        self.x0: List[torch.Tensor] = []
        self.last_sample: Optional[torch.Tensor] = None

    def step(self, i: int, v_text: torch.Tensor, v_uncond: Optional[torch.Tensor], guidance: float, sample: torch.Tensor) -> torch.Tensor:
        c = self.coefs[i]
        # x0 = sample - sigma * v takes torch's promoted dtype of (sample, v): fp32 for the reference's fp16 storage beside a bf16 network,
        # bf16 when both are bf16 (one more rounding of the fp32 difference = torch's bf16 subtraction, exactly)
        x0_new = unipc_convert(v_text, v_uncond, guidance, sample, self.sigmas[i]).to(torch.promote_types(sample.dtype, v_text.dtype))
        if c.corrector is not None and self.last_sample is not None:
            q = c.corrector_order
            sample = lincomb([self.last_sample] + self.x0[:q] + [x0_new], c.corrector, sample.dtype)
        self.x0.insert(0, x0_new)
        del self.x0[self.solver_order:]
        self.last_sample = sample
        return lincomb([sample] + self.x0[:c.order], c.predictor, sample.dtype)
