"""``DPMSolverMultistepScheduler`` with the constructor / ``set_timesteps`` / ``step`` surface
and attributes LiDiff touches (tools/diff_completion_pipeline.py:38-46,58-66,163;
models/models.py:65-73,141), restating diffusers==0.18.0 for
``algorithm_type='sde-dpmsolver++'`` / ``'dpmsolver++'``, ``solver_order<=2``,
``solver_type='midpoint'``, ``prediction_type='epsilon'``.

Differences from the upstream object, all host-side:
  * the step index is looked up on a host copy of ``timesteps`` -- no ``.nonzero().item()``
    device sync per step;
  * ``step(..., noise=z)`` lets a caller inject the Gaussian draw (parity tests share one z
    between device and CPU oracle); without it the draw comes from torch's RNG on the
    sample's device, as upstream.
"""
from __future__ import annotations

import math

import numpy as np
import torch


class SchedulerOutput(dict):
    @property
    def prev_sample(self):
        return self["prev_sample"]


class DPMSolverMultistepScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 solver_order=2, prediction_type="epsilon", algorithm_type="dpmsolver++",
                 solver_type="midpoint", lower_order_final=True):
        if beta_schedule != "linear":
            raise NotImplementedError("LiDiff uses beta_schedule='linear'")
        if algorithm_type not in ("dpmsolver++", "sde-dpmsolver++") or solver_type != "midpoint":
            raise NotImplementedError(algorithm_type)
        if prediction_type != "epsilon" or solver_order not in (1, 2):
            raise NotImplementedError("epsilon prediction, solver_order 1 or 2")
        self.num_train_timesteps = num_train_timesteps
        self.solver_order = solver_order
        self.algorithm_type = algorithm_type
        self.lower_order_final = lower_order_final
        self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.sigmas = ((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5
        self.init_noise_sigma = 1.0
        self.timesteps = torch.from_numpy(
            np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy())
        self._host_timesteps = [int(v) for v in self.timesteps.tolist()]
        self.num_inference_steps = None
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0

    def set_timesteps(self, num_inference_steps, device=None):
        ts = (np.linspace(0, self.num_train_timesteps - 1, num_inference_steps + 1)
              .round()[::-1][:-1].copy().astype(np.int64))
        _, first = np.unique(ts, return_index=True)
        ts = ts[np.sort(first)]
        self._host_timesteps = [int(v) for v in ts]
        self.timesteps = torch.from_numpy(ts).to(device)
        self.num_inference_steps = len(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums = 0

    def to(self, device):
        """What DiffCompletion.scheduler_to_cuda does attribute by attribute (pipeline:58-66)."""
        for name in ("timesteps", "betas", "alphas", "alphas_cumprod", "alpha_t", "sigma_t", "lambda_t", "sigmas"):
            setattr(self, name, getattr(self, name).to(device))
        return self

    # -- one-step updates.  Upstream evaluates the scalar coefficients as 0-dim fp32 tensors on
    #    the sample's device (a dozen tiny kernels per step whose exp/log differ by an ulp between
    #    CPU and GPU); here they are Python floats computed on the host in float64 from the same
    #    fp32 tables -- no launches, identical on every device, within 1e-7 relative of upstream.
    def _coeffs(self, t, t_prev):
        lam, alpha, sigma = self._host_tables()
        h = lam[t_prev] - lam[t]
        return h, alpha[t_prev], sigma[t_prev], sigma[t]

    def _host_tables(self):
        if getattr(self, "_tables", None) is None:
            self._tables = tuple(x.detach().cpu().double().tolist() for x in (self.lambda_t, self.alpha_t, self.sigma_t))
        return self._tables

    def _first_order(self, x0, t, t_prev, sample, noise):
        h, a_p, s_p, s_t = self._coeffs(t, t_prev)
        if self.algorithm_type == "dpmsolver++":
            return (s_p / s_t) * sample - (a_p * math.expm1(-h)) * x0
        return ((s_p / s_t * math.exp(-h)) * sample + (a_p * -math.expm1(-2.0 * h)) * x0
                + (s_p * math.sqrt(-math.expm1(-2.0 * h))) * noise)

    def _second_order(self, t_prev_call, t, t_prev, sample, noise):
        m0, m1 = self.model_outputs[-1], self.model_outputs[-2]
        h, a_p, s_p, s_t = self._coeffs(t, t_prev)
        lam = self._host_tables()[0]
        r0 = (lam[t] - lam[t_prev_call]) / h
        d1 = (1.0 / r0) * (m0 - m1)
        if self.algorithm_type == "dpmsolver++":
            g = a_p * math.expm1(-h)
            return (s_p / s_t) * sample - g * m0 - (0.5 * g) * d1
        g = a_p * -math.expm1(-2.0 * h)
        return ((s_p / s_t * math.exp(-h)) * sample + g * m0 + (0.5 * g) * d1
                + (s_p * math.sqrt(-math.expm1(-2.0 * h))) * noise)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, noise=None):
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps first")
        t = int(timestep) if not isinstance(timestep, torch.Tensor) else self._host_value(timestep)
        n = len(self._host_timesteps)
        step_index = self._host_timesteps.index(t) if t in self._host_timesteps else n - 1
        t_prev = 0 if step_index == n - 1 else self._host_timesteps[step_index + 1]
        lower_final = step_index == n - 1 and self.lower_order_final and n < 15
        # epsilon -> data prediction
        _, alpha, sigma = self._host_tables()
        x0 = (sample - sigma[t] * model_output) / alpha[t]
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        if self.algorithm_type == "sde-dpmsolver++" and noise is None:
            noise = torch.randn(x0.shape, generator=generator, device=x0.device, dtype=x0.dtype)
        if self.solver_order == 1 or self.lower_order_nums < 1 or lower_final:
            prev = self._first_order(x0, t, t_prev, sample, noise)
        else:
            prev = self._second_order(self._host_timesteps[step_index - 1], t, t_prev, sample, noise)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return SchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def step_plan(self, timestep):
        """The host half of step() for a caller that evaluates the update in its own kernel (ops.cfg_dpm_step, the fused step
        boundary of DiffCompletion.denoise_step): returns the update's scalars -- the same Python floats, from the same
        expressions, that _first_order / _second_order multiply the tensors with -- plus the previous data prediction.  PURE: the
        multistep bookkeeping step() does (history shift, lower_order_nums) is applied by commit(x0) once the caller's launch
        has succeeded; a launch that raises leaves the scheduler exactly as it was."""
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps first")
        if self.algorithm_type != "sde-dpmsolver++":
            raise NotImplementedError("step_plan covers sde-dpmsolver++ (the sampler LiDiff constructs)")
        t = int(timestep) if not isinstance(timestep, torch.Tensor) else self._host_value(timestep)
        n = len(self._host_timesteps)
        step_index = self._host_timesteps.index(t) if t in self._host_timesteps else n - 1
        t_prev = 0 if step_index == n - 1 else self._host_timesteps[step_index + 1]
        lower_final = step_index == n - 1 and self.lower_order_final and n < 15
        lam, alpha, sigma = self._host_tables()
        h, a_p, s_p, s_t = self._coeffs(t, t_prev)
        g = a_p * -math.expm1(-2.0 * h)
        plan = {"t": t, "sigma_t": sigma[t], "alpha_t": alpha[t], "c_sample": s_p / s_t * math.exp(-h), "c_m0": g,
                "c_noise": s_p * math.sqrt(-math.expm1(-2.0 * h)), "m_prev": None, "c_d1": 0.0, "inv_r0": 0.0}
        if not (self.solver_order == 1 or self.lower_order_nums < 1 or lower_final):
            m_prev = self.model_outputs[-1]                 # (the history has not been shifted yet: the last data prediction)
            if m_prev is None:
                raise RuntimeError("second-order step without a previous data prediction (a step was planned but never committed)")
            r0 = (lam[t] - lam[self._host_timesteps[step_index - 1]]) / h
            plan.update(m_prev=m_prev, c_d1=0.5 * g, inv_r0=1.0 / r0)
        return plan

    def commit(self, x0):
        """The bookkeeping of the step that step_plan() described, with its data prediction: what step() does around the update."""
        for i in range(self.solver_order - 1):
            self.model_outputs[i] = self.model_outputs[i + 1]
        self.model_outputs[-1] = x0
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1

    @staticmethod
    def _host_value(timestep: torch.Tensor) -> int:
        """A tensor timestep is accepted as upstream does (one device read if it lives on the
        GPU); lidiff_amd's own loops pass Python ints from ``host_timesteps`` and never sync."""
        return int(timestep.reshape(-1)[0].item())

    @property
    def host_timesteps(self):
        return list(self._host_timesteps)
