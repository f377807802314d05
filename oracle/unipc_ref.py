"""TEST INFRASTRUCTURE ONLY -- never imported by the product path.

CPU restatement (torch, tensor by tensor) of the solver the reference's Wan adapters sample with in EVALUATION mode:
`UniPCMultistepSDEScheduler.step` with `is_eval` (reference src/flow_factory/scheduler/unipc_multistep.py:282-285) returns
`super().step(noise_pred, timestep, latents)`, i.e. diffusers' `UniPCMultistepScheduler.step` -- a THIRD-PARTY dependency that is not
in /root/reference (diffusers is an un-vendored requirement of the reference; not installed in this image).

**PARITY UNPINNED.**  What follows restates the published algorithm of that class (UniPC, Zhao et al. 2023, "bh" variants, as
implemented in diffusers' `scheduling_unipc_multistep.py`: `convert_model_output`, `multistep_uni_p_bh_update`,
`multistep_uni_c_bh_update`, `step`) for the configuration the Wan pipelines ship with: `prediction_type="flow_prediction"`,
`use_flow_sigmas=True`, `predict_x0=True`, `solver_order=2`, `solver_type="bh2"`, `lower_order_final=True`, `disable_corrector=[]`,
`final_sigmas_type="zero"`.  There is no golden vector for it in the reference's tests and the class cannot be executed here; the
tests pin the product path (mi355_flow/unipc.py: the same solver as schedule-only linear coefficients; csrc/sde_step.hip kernels)
against THIS file, and this file against an independent closed-form check (tests/test_host_mirrors.py: order-1 UniP on a linear
x0 model is the exact exponential-integrator solution).

Type promotion follows torch's rules for the reference's tensors: `sample` arrives in the latent storage dtype (fp16 by default),
`model_output` in the transformer's dtype (bf16), the sigmas are 0-dim fp32 tensors (a 0-dim tensor does not promote a half tensor)."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


class UniPCRef:
    def __init__(self, sigmas: Sequence[float], solver_order: int = 2, solver_type: str = "bh2", lower_order_final: bool = True,
                 disable_corrector: Sequence[int] = ()):
        assert solver_type in ("bh1", "bh2")
        self.sigmas = torch.tensor(list(sigmas), dtype=torch.float32)          # N + 1 entries, the last one 0 (final_sigmas_type="zero")
        self.n_steps = len(sigmas) - 1
        self.solver_order, self.solver_type = solver_order, solver_type
        self.lower_order_final, self.disable_corrector = lower_order_final, list(disable_corrector)
        self.model_outputs: List[Optional[torch.Tensor]] = [None] * solver_order
        self.lower_order_nums = 0
        self.last_sample: Optional[torch.Tensor] = None
        self.step_index = 0
        self.this_order = 1

    # flow sigmas: alpha_t = 1 - sigma, sigma_t = sigma  (_sigma_to_alpha_sigma_t under use_flow_sigmas)
    @staticmethod
    def _alpha_sigma(s):
        return 1 - s, s

    def convert_model_output(self, model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        # flow_prediction + predict_x0:  x0_pred = sample - sigma_t * model_output
        sigma_t = self.sigmas[self.step_index]
        return sample - sigma_t * model_output

    def _rb(self, rks: torch.Tensor, hh: torch.Tensor, order: int):
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        factorial_i = 1
        B_h = hh if self.solver_type == "bh1" else torch.expm1(hh)
        R, b = [], []
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * factorial_i / B_h)
            factorial_i *= i + 1
            h_phi_k = h_phi_k / hh - 1 / factorial_i
        return torch.stack(R), torch.stack([torch.as_tensor(v, dtype=torch.float32) for v in b]), h_phi_1, B_h

    def uni_p(self, sample: torch.Tensor, order: int) -> torch.Tensor:
        m0, x = self.model_outputs[-1], sample
        sigma_t, sigma_s0 = self.sigmas[self.step_index + 1], self.sigmas[self.step_index]
        alpha_t, sigma_t = self._alpha_sigma(sigma_t)
        alpha_s0, sigma_s0 = self._alpha_sigma(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            si = self.step_index - i
            mi = self.model_outputs[-(i + 1)]
            alpha_si, sigma_si = self._alpha_sigma(self.sigmas[si])
            lambda_si = torch.log(alpha_si) - torch.log(sigma_si)
            rk = (lambda_si - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in rks])
        hh = -h                                                   # predict_x0
        R, b, h_phi_1, B_h = self._rb(rks, hh, order)
        if D1s:
            D1s = torch.stack(D1s, dim=1)
            rhos_p = torch.tensor([0.5], dtype=x.dtype) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1]).to(x.dtype)
        else:
            D1s = None
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        pred_res = torch.einsum("k,bkc...->bc...", rhos_p.to(D1s.dtype), D1s) if D1s is not None else 0
        x_t = x_t_ - alpha_t * B_h * pred_res
        return x_t.to(x.dtype)

    def uni_c(self, this_model_output: torch.Tensor, last_sample: torch.Tensor, this_sample: torch.Tensor, order: int) -> torch.Tensor:
        m0, x, model_t = self.model_outputs[-1], last_sample, this_model_output
        sigma_t, sigma_s0 = self.sigmas[self.step_index], self.sigmas[self.step_index - 1]
        alpha_t, sigma_t = self._alpha_sigma(sigma_t)
        alpha_s0, sigma_s0 = self._alpha_sigma(sigma_s0)
        lambda_t = torch.log(alpha_t) - torch.log(sigma_t)
        lambda_s0 = torch.log(alpha_s0) - torch.log(sigma_s0)
        h = lambda_t - lambda_s0
        rks, D1s = [], []
        for i in range(1, order):
            si = self.step_index - (i + 1)
            mi = self.model_outputs[-(i + 1)]
            alpha_si, sigma_si = self._alpha_sigma(self.sigmas[si])
            lambda_si = torch.log(alpha_si) - torch.log(sigma_si)
            rk = (lambda_si - lambda_s0) / h
            rks.append(rk)
            D1s.append((mi - m0) / rk)
        rks.append(torch.tensor(1.0))
        rks = torch.stack([torch.as_tensor(r, dtype=torch.float32) for r in rks])
        hh = -h
        R, b, h_phi_1, B_h = self._rb(rks, hh, order)
        D1s = torch.stack(D1s, dim=1) if D1s else None
        rhos_c = torch.tensor([0.5], dtype=x.dtype) if order == 1 else torch.linalg.solve(R, b).to(x.dtype)
        x_t_ = sigma_t / sigma_s0 * x - alpha_t * h_phi_1 * m0
        corr_res = torch.einsum("k,bkc...->bc...", rhos_c[:-1].to(D1s.dtype), D1s) if D1s is not None else 0
        D1_t = model_t - m0
        x_t = x_t_ - alpha_t * B_h * (corr_res + rhos_c[-1] * D1_t)
        return x_t.to(x.dtype)

    def step(self, model_output: torch.Tensor, sample: torch.Tensor) -> torch.Tensor:
        use_corrector = self.step_index > 0 and (self.step_index - 1) not in self.disable_corrector and self.last_sample is not None
        converted = self.convert_model_output(model_output, sample)
        if use_corrector:
            sample = self.uni_c(converted, self.last_sample, sample, self.this_order)
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = converted
        this_order = min(self.solver_order, self.n_steps - self.step_index) if self.lower_order_final else self.solver_order
        self.this_order = min(this_order, self.lower_order_nums + 1)
        assert self.this_order > 0
        self.last_sample = sample
        prev = self.uni_p(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev
