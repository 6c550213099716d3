"""Fused optimiser step for the librvit training path.

Replaces `clip_grad_norm_(model.parameters(), max_norm)` + `torch.optim.AdamW.step()` of the reference's
loop (train_vit.py:879-883,942-946) by two launches over flat buffers: the training engine already writes every
gradient into ONE flat fp32 buffer; FusedAdamW moves the parameters into a flat buffer of the same layout
(each `nn.Parameter` becomes a view of it; conv weights become channels-last views, which is also the layout
the kernels pack from) and keeps `exp_avg` / `exp_avg_sq` flat too.  Under data parallelism it runs after
the bucketed all-reduce has been waited for (the gradients handed to autograd are already the averages).

It is a `torch.optim.Optimizer`, so `LambdaLR` (the reference's warm-up + cosine schedule,
train_vit.py:885-893) and `zero_grad()` work unchanged; `param_groups[0]["lr"]` is read at every step.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, model, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None):
        params = list(model.parameters())
        if not params or any(not p.is_cuda for p in params):
            raise RuntimeError("FusedAdamW: move the model to a ROCm device first (no CPU path exists)")
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("FusedAdamW: fp32 master parameters expected")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW: one parameter group (the reference uses one, train_vit.py:879-883)")
        self.model = model
        self.max_grad_norm = max_grad_norm
        eng = model.train_engine
        self._eng = eng
        total = eng.layout.total
        dev = params[0].device
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        views = eng.grads_in_param_order(flat)  # same layout and view shapes as the gradients
        with torch.no_grad():
            for (name, p), v in zip(model.named_parameters(), views):
                if v.shape != p.shape:
                    raise RuntimeError(f"FusedAdamW: layout mismatch for {name}")
                v.copy_(p.detach())
                p.data = v                      # the parameter now lives in the flat buffer
        self.flat_params = flat
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self._norm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        # torch.optim.AdamW keeps one step count PER PARAMETER, created at the parameter's first gradient: after
        # unfreeze_backbone() (train_vit.py:918-923) the stem's bias correction starts at step 1 while the tail is at
        # step N+1.  Two counters cover that (the only frozen set supported is the whole stem = the tail of the flat
        # layout): `_step` for [0, stem_start), `_step_stem` for [stem_start, total), the latter advanced only on steps
        # where the stem is active.  While they are equal the whole buffer is one launch.
        self._step = 0
        self._step_stem = 0
        m_views, v_views = eng.grads_in_param_order(self.exp_avg), eng.grads_in_param_order(self.exp_avg_sq)
        self._step_t = torch.tensor(0.0)
        self._step_stem_t = torch.tensor(0.0)
        stem = set() if model.cnn_backbone is None else {id(p) for p in model.cnn_backbone.parameters()}
        self._stem_ids = stem
        for p, mv, vv in zip(model.parameters(), m_views, v_views):
            self.state[p] = {"step": self._step_stem_t if id(p) in stem else self._step_t, "exp_avg": mv, "exp_avg_sq": vv}
        named = list(model.named_parameters())
        self._probe = [(p, eng.layout.offsets[n]) for n, p in (named[0], named[-1])]
        # bf16 operand copies for the training engine, written by the AdamW kernel from now on
        from .trainer import WeightMirror
        self.mirror = WeightMirror(eng, flat)
        eng.mirror = self.mirror
        model.invalidate_packs()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("FusedAdamW: closures are not supported")
        # torch optimisers skip parameters without a gradient (freeze_backbone, train_vit.py:310-318).  The flat
        # kernels update a contiguous range; the stem's parameters are the TAIL of the flat layout, so a frozen
        # cnn_backbone simply shortens the range (no weight decay, no moment update on it: as torch).  Any other
        # frozen subset is refused rather than silently stepped.
        n_active = self._eng.layout.total
        frozen = {id(p) for p in self.param_groups[0]["params"] if not p.requires_grad}
        if frozen:
            cb = self.model.cnn_backbone
            if cb is None or frozen != {id(p) for p in cb.parameters()}:
                raise RuntimeError("FusedAdamW: the only frozen set supported is the whole cnn_backbone "
                                   "(freeze_backbone, train_vit.py:310-318); use a torch optimiser for other subsets")
            n_active = self._eng.stem_start()
        flat_g = getattr(self._eng, "last_flat", None)
        first = self._probe[0][0]
        if flat_g is None or first.grad is None:
            raise RuntimeError("FusedAdamW.step: no gradients; call loss.backward() on a train()-mode forward first")
        # the gradients must be the views the librvit backward handed to autograd (not accumulated copies):
        # checked on the first and last parameter only, to keep the per-step host work to a few microseconds
        for p, (off, _, _) in self._probe:
            if p.data_ptr() != self.flat_params.data_ptr() + 4 * off:
                raise RuntimeError("FusedAdamW.step: the model's parameters no longer live in the optimiser's flat "
                                   "buffer (model.to()/half() after constructing the optimiser?); build a new one")
            if p.grad is None or p.grad.data_ptr() != flat_g.data_ptr() + 4 * off:
                raise RuntimeError("FusedAdamW.step: parameter gradients are not views of the engine's flat buffer "
                                   "(gradient accumulation over several backward passes is not supported)")
        grp = self.param_groups[0]
        total = self._eng.layout.total
        stem_active = bool(self._stem_ids) and n_active == total
        self._step += 1
        if stem_active:
            self._step_stem += 1
        if stem_active and self._step_stem != self._step:
            ranges = [(slice(0, self._eng.stem_start()), self._step), (slice(self._eng.stem_start(), total), self._step_stem)]
        else:
            ranges = [(slice(0, n_active), self._step)]
        norm = None
        with torch.cuda.device(self.flat_params.device):
            if self.max_grad_norm is not None:       # ONE norm over everything that is stepped (clip_grad_norm_)
                self._norm_sq.zero_()
                norm = ops.sumsq(flat_g[:n_active], self._norm_sq)
            b1, b2 = grp["betas"]
            for rng, step in ranges:
                ops.adamw_step(self.flat_params[rng], flat_g[rng], self.exp_avg[rng], self.exp_avg_sq[rng], lr=grp["lr"],
                               beta1=b1, beta2=b2, eps=grp["eps"], weight_decay=grp["weight_decay"], step=step,
                               grad_norm_sq=norm, max_grad_norm=float(self.max_grad_norm or 0.0),
                               params_bf16=self.mirror.bf16[rng])
            self.mirror.refresh(full=False)      # K-major copies for the backward GEMMs: one batched-transpose launch
        self.model.invalidate_packs()        # inference engine: re-pack at the next eval forward
        self._step_t += 1     # one tensor shared by the state entries of the tail's parameters, one by the stem's
        if stem_active:
            self._step_stem_t += 1
        return None

    def grad_norm(self) -> torch.Tensor:
        """Global gradient norm of the last step (before clipping), as clip_grad_norm_ returns it."""
        return self._norm_sq.sqrt().squeeze(0)

    def load_state_dict(self, state_dict):
        # keep exp_avg / exp_avg_sq as views of the flat buffers: copy the loaded values in
        mine = [self.state[p] for p in self.param_groups[0]["params"]]
        super().load_state_dict(state_dict)
        for p, old in zip(self.param_groups[0]["params"], mine):
            new = self.state[p]
            old["exp_avg"].copy_(new["exp_avg"])
            old["exp_avg_sq"].copy_(new["exp_avg_sq"])
            t = self._step_stem_t if id(p) in self._stem_ids else self._step_t
            t.fill_(float(new["step"]))
            old["step"] = t
            self.state[p] = old
        self._step = int(float(self._step_t))
        self._step_stem = int(float(self._step_stem_t))
