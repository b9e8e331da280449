"""HipAdam — torch.optim.Adam's arithmetic as ONE hand-written launch over a list of tensors (csrc/tn_optim.hip, tn_adam_step).

The reference's method config asks for ``AdamOptimizerConfig(lr=1e-2, eps=1e-15)`` per parameter group
[REF thermo_nerf/thermal_nerf/config_thermal_nerf.py:31-44]; nerfstudio's ``Optimizers.optimizer_step_all`` steps them one after
the other.  As ``torch.optim.Adam(fused=True)`` that is, per group and step, a ``_foreach_add`` on the step counters, the
multi-tensor kernel, and ~0.1 ms of Python (grouping, state initialisation, dispatch) — and every group's launch sits on the one
stream, behind the field's table-gradient scatter.  Here:

* the small tensors of a group (MLP layers, embeddings, pose adjustments) are one ``tn_adam_step`` launch on the calling stream;
* a tensor named in ``deferred`` (the field's 64 MB hash table) is its own launch on the training step's SECOND stream, behind
  the scatter that produced its gradient (training.hash_encode_bwd(defer=True)); the calling stream does not wait — the next
  training forward joins right before its field launch (``_hip.join_pending``), so the next step's ray gather, camera optimizer,
  proposal pass and field_prepare overlap the scatter and this launch;
* state layout and ``state_dict`` are torch.optim.Adam's (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter), so a checkpoint
  written by either loads into the other.

Same update rule as torch's (L2 weight decay, bias corrections in float64 on the host), rounding may differ in the last bit.
No CPU path: parameters must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Iterable, Optional

import torch

from . import _hip


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 deferred: Optional[Iterable[torch.nn.Parameter]] = None) -> None:
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._deferred = {id(p) for p in (deferred or ())}
        self._lists: dict = {}  # cached ctypes descriptor arrays
        self._fixed: dict = {}  # (array id, position) -> the step-invariant fields written there

    def __setstate__(self, state) -> None:  # (torch's Optimizer pickles defaults / state / param_groups only)
        super().__setstate__(state)
        self.__dict__.setdefault("_deferred", set())
        self._lists, self._fixed = {}, {}

    def _slot(self, group, p: torch.nn.Parameter, arr, k: int) -> None:
        """fill descriptor ``k`` of the launch list ``arr`` for this step.  The fields that do not change from step to step (parameter
        and moment addresses, size, betas, eps, weight decay) are written once per (list, position): the arrays are cached per
        optimizer and reused while the same parameters receive gradients — a step then sets three fields per tensor"""
        g = p.grad
        if g.is_sparse:
            raise RuntimeError("HipAdam does not support sparse gradients")
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("HipAdam updates contiguous fp32 parameters on a ROCm device (no CPU path exists)")
        if not g.is_contiguous():
            g = p.grad = g.contiguous()
        st = self.state[p]
        if not st:
            st["step"] = 0.0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        elif torch.is_tensor(st["step"]):  # a state written by torch.optim.Adam
            st["step"] = float(st["step"])
        st["step"] += 1.0
        t = st["step"]
        b1, b2 = group["betas"]
        d = arr[k]
        key = (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), b1, b2, group["eps"], group["weight_decay"])
        fixed = self._fixed.get((id(arr), k))
        if fixed != key:
            d.param, d.exp_avg, d.exp_avg_sq = key[0], key[1], key[2]
            d.n = p.numel()
            d.one_minus_beta1, d.beta2, d.one_minus_beta2 = 1.0 - b1, b2, 1.0 - b2
            d.eps, d.weight_decay = group["eps"], group["weight_decay"]
            self._fixed[(id(arr), k)] = key
        d.grad = g.data_ptr()
        d.step_size = group["lr"] / (1.0 - b1 ** t)
        d.bias_correction2_sqrt = math.sqrt(1.0 - b2 ** t)

    def _list(self, n: int, tag):
        """a cached descriptor array of n entries (one per distinct (tag, n): the small tensors of a step, each deferred tensor)"""
        hit = self._lists.get((tag, n))
        if hit is None:
            hit = self._lists[(tag, n)] = (_hip.tn_adam_tensor * n)()
        return hit

    @staticmethod
    def _launch(arr, count: int, stream: int) -> None:
        lib = _hip.load()
        for k in range(0, count, _hip.ADAM_MAX_TENSORS):
            n = min(_hip.ADAM_MAX_TENSORS, count - k)
            part = C.cast(C.byref(arr, k * C.sizeof(_hip.tn_adam_tensor)), C.POINTER(_hip.tn_adam_tensor))
            _hip.check(lib.tn_adam_step(part, n, stream), "tn_adam_step")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        small = [(group, p) for group in self.param_groups for p in group["params"] if p.grad is not None and id(p) not in self._deferred]
        if small:
            # (the list is keyed by WHICH parameters have gradients this step: frozen-proposal and update steps alternate)
            arr = self._list(len(small), tuple(id(p) for _, p in small))
            for k, (group, p) in enumerate(small):
                self._slot(group, p, arr, k)
            self._launch(arr, len(small), _hip.current_stream())
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None and id(p) in self._deferred:
                    arr = self._list(1, ("deferred", id(p)))
                    self._slot(group, p, arr, 0)
                    self._step_deferred(p, arr)
        return loss

    def _step_deferred(self, p: torch.nn.Parameter, arr) -> None:
        """the table's launch on the step's second stream: behind the bucketed half of its gradient's scatter (queued there) and the
        atomic half (third stream: waited for), or — scatter already joined — behind the calling stream's work so far"""
        from .training import _step_streams

        main, second, third = _step_streams(p.device)
        pend = _hip.pending(p.device)
        if pend is None or all(s.cuda_stream != second.cuda_stream for s in pend["streams"]):
            second.wait_stream(main)  # the scatter was joined by the backward: the gradient is the calling stream's work
        if pend is not None:
            for s in pend["streams"]:
                if s.cuda_stream != second.cuda_stream:
                    second.wait_stream(s)
        self._launch(arr, 1, second.cuda_stream)
        st = self.state[p]
        _hip.defer(p.device, [second], [p.grad, st["exp_avg"], st["exp_avg_sq"]])

    def state_dict(self):
        _hip.join_pending()
        sd = super().state_dict()
        # torch.optim.Adam keeps `step` as a tensor: written that way (copies: the live state keeps its float)
        sd["state"] = {k: {**st, "step": torch.tensor(float(st["step"]), dtype=torch.float32)} if "step" in st else dict(st)
                       for k, st in sd["state"].items()}
        return sd
