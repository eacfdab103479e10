"""``Adam`` / ``AdamW`` with the ``torch.optim`` interface on one streaming HIP kernel (``csrc/optimizer.hip``).

The reference trains with ``torch.optim.Adam`` (``train_IEDB_wFT.py:69-74``) and ``torch.optim.AdamW``
(``train_Cancer_wFT.py:76-92``).  These classes keep the constructor arguments, ``param_groups`` (so the reference's
learning-rate schedulers work unchanged), ``zero_grad`` and ``state_dict`` layout of a torch optimizer; ``step()`` is
one launch over a device-resident chunk table per parameter group instead of torch's multi-tensor loop.  Step count
and learning rate live in device memory: the step can be captured in a HIP graph; call :meth:`refresh` (the engine
does) after a scheduler changed ``lr`` so that the device copy follows.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ["Adam", "AdamW"]

CHUNK = 16384


class Adam(torch.optim.Optimizer):
    decoupled_weight_decay = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not used by the reference and not implemented")
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._groups = {}       # id(group) -> dict(table, key, state, hyper, hyper_host)
        # every gradient is multiplied by this inside the update kernel: 1 / world size when the data-parallel reducer
        # hands over the SUM of the ranks' gradients (saves its own pass over the 25 MB bucket)
        self.grad_scale = 1.0

    # ---- helpers -------------------------------------------------------------
    def _hyper_host(self, group):
        return (float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]), float(group["eps"]),
                float(group["weight_decay"]), 1.0 if self.decoupled_weight_decay else 0.0, float(self.grad_scale), 0.0)

    def _group_state(self, group, params):
        dev = params[0].device
        gs = self._groups.get(id(group))
        if gs is None:
            gs = dict(tables={}, hyper_host=None, captured={}, pending=[], spares={}, prepared=False, updated=set(),
                      state=torch.zeros(3, dtype=torch.float32, device=dev), hyper=torch.zeros(8, dtype=torch.float32, device=dev))
            self._groups[id(group)] = gs
        return gs

    def refresh(self):
        """copy changed hyper-parameters (learning-rate schedulers) to their device copies and upload the chunk tables of
        freshly captured steps; call before replaying a captured step (the engine does); not capturable"""
        for group in self.param_groups:
            gs = self._groups.get(id(group))
            if gs is not None:
                self._flush_group(gs)
                self.refresh_group(group, gs)

    flush_tables = refresh

    @staticmethod
    def _flush_group(gs):
        while gs["pending"]:
            table, host = gs["pending"].pop()
            table.copy_(host)

    def _rows(self, params):
        rows = []
        for p in params:
            st = self.state[p]
            for off in range(0, p.numel(), CHUNK):
                cnt = min(CHUNK, p.numel() - off)
                rows.append((p.data_ptr() + 4 * off, p.grad.data_ptr() + 4 * off, st["exp_avg"].data_ptr() + 4 * off,
                             st["exp_avg_sq"].data_ptr() + 4 * off, cnt))
        return torch.from_numpy(np.asarray(rows, dtype=np.int64))

    def _table(self, group, params, capturing):
        """(group state, device chunk table) for the update of ``params`` (a subset of the group's parameters, each with a
        gradient); tables are cached per parameter / gradient address set"""
        _lib.require_device(*params)
        for p in params:
            if p.dtype != torch.float32 or not p.is_contiguous() or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                raise ValueError("Adam (HIP) needs contiguous fp32 parameters and gradients")
            st = self.state[p]
            if not st:
                if capturing:
                    raise RuntimeError("optimizer state must exist before a HIP-graph capture: run one eager step first")
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        gs = self._group_state(group, params)
        # the rows hold the moment buffers' addresses too: a table is only valid for the state tensors it was built from
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                     p.numel()) for p in params)
        if capturing:
            # gradients produced inside a capture live at fixed addresses of the graph's memory pool.  The step being
            # captured gets a chunk table of its OWN, kept alive with the optimizer: eager steps before or after never
            # rebind or overwrite it.  It comes from the spares the last eager step left (memory allocated while
            # capturing belongs to the graph's pool and may alias a buffer the replayed graph writes earlier in the
            # step).  Its rows are uploaded by flush_tables() once the capture has ended -- the kernel only reads
            # them at replay time, so the graph needs no upload node.
            ent = gs["captured"].get(key)
            if ent is None:
                host = self._rows(params)
                pool = gs["spares"].get(tuple(host.shape), [])
                if not pool:
                    raise RuntimeError("the chunk table of a captured step is allocated by an eager step with the same "
                                       "parameters (and the same split): run one eager step before the HIP-graph capture")
                ent = pool.pop()
                gs["captured"][key] = ent
                gs["pending"].append((ent, host))
            return gs, ent
        self._flush_group(gs)
        table = gs["tables"].get(key)
        if table is None:
            if len(gs["tables"]) > 8:
                gs["tables"].clear()
            table = gs["tables"][key] = self._rows(params).to(params[0].device)
        pool = gs["spares"].setdefault(tuple(table.shape), [])
        while len(pool) < 3:
            pool.append(torch.empty_like(table))
        return gs, table

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        capturing = torch.cuda.is_current_stream_capturing()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            gs, table = self._table(group, params, capturing)
            if not capturing:
                self.refresh_group(group, gs)
            _lib.check(lib.is_adam_step(_lib.ptr(table), int(table.shape[0]), _lib.ptr(gs["state"]), _lib.ptr(gs["hyper"]),
                                        _lib.stream_ptr()), "is_adam_step")
        return loss

    @torch.no_grad()
    def step_subset(self, params, first):
        """The update of :meth:`step` for the parameters in ``params`` only; ``first`` = this is the step's first part (the step
        count advances and the bias corrections are formed once per step, by the first part; every other part of the same step
        passes False).  The data-parallel engine updates the first gradient bucket's parameters while the second bucket's
        all-reduce is still on the wire.  Same arithmetic as one :meth:`step` over all of them."""
        lib = _lib.load()
        capturing = torch.cuda.is_current_stream_capturing()
        ids = set(id(p) for p in params)
        for group in self.param_groups:
            with_grad = [p for p in group["params"] if p.grad is not None]
            if not with_grad:
                continue
            sub = [p for p in with_grad if id(p) in ids]
            gs = self._group_state(group, with_grad)
            table = self._table(group, sub, capturing)[1] if sub else None
            if not capturing:
                self.refresh_group(group, gs)
            if first:
                _lib.check(lib.is_adam_prepare(_lib.ptr(gs["state"]), _lib.ptr(gs["hyper"]), _lib.stream_ptr()), "is_adam_prepare")
                gs["prepared"] = True
                gs["updated"] = set()
            elif not gs["prepared"] or any(id(p) in gs["updated"] for p in sub):
                # a later part without a first part would silently reuse the PREVIOUS step's step size and bias corrections
                raise RuntimeError("step_subset(first=False) without a step_subset(first=True) of the same step in front of it "
                                   "(none yet, or these parameters were already updated since the last one): every step's "
                                   "first part advances the step count and forms the bias corrections")
            gs["updated"].update(id(p) for p in sub)
            if table is not None:
                _lib.check(lib.is_adam_apply(_lib.ptr(table), int(table.shape[0]), _lib.ptr(gs["state"]), _lib.ptr(gs["hyper"]),
                                             _lib.stream_ptr()), "is_adam_apply")

    # ---- checkpointing ---------------------------------------------------------
    def state_dict(self):
        """torch's layout; every parameter's state also carries ``step`` (torch.optim.Adam's key) read from the device-side counter
        of its group, so that a checkpoint resumes with the right bias corrections -- here or under ``torch.optim.Adam``"""
        for group in self.param_groups:
            gs = self._groups.get(id(group))
            if gs is None:
                continue
            step = torch.tensor(float(gs["state"][0].item()))
            for p in group["params"]:
                if self.state.get(p):
                    self.state[p]["step"] = step.clone()
        return super().state_dict()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """values are copied INTO the existing moment buffers where they exist (chunk tables and captured steps hold their
        addresses); otherwise the loaded tensors are adopted and every cached table of the group is dropped.  The device-side
        step counter is restored from the ``step`` entries."""
        old = {p: dict(st) for p, st in self.state.items()}
        old_groups = {i: self._groups.get(id(g)) for i, g in enumerate(self.param_groups)}
        super().load_state_dict(state_dict)
        self._groups = {id(g): old_groups[i] for i, g in enumerate(self.param_groups) if old_groups.get(i) is not None}
        for group in self.param_groups:
            gs = self._groups.get(id(group))
            adopted, step = False, None
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                if "step" in st:
                    step = float(st.pop("step"))
                for k in ("exp_avg", "exp_avg_sq"):
                    new = st[k].to(device=p.device, dtype=torch.float32).contiguous()
                    have = old.get(p, {}).get(k)
                    if have is not None and have.shape == new.shape:
                        have.copy_(new)
                        st[k] = have
                    else:
                        # (torch's load_state_dict hands tensors that already sit on the right device through WITHOUT a copy:
                        #  adopted as they are, the checkpoint's own tensors would be updated in place by the next step)
                        st[k] = new.clone()
                        adopted = True
            if gs is None and step is not None:
                with_state = [p for p in group["params"] if self.state.get(p)]
                gs = self._group_state(group, with_state) if with_state else None
            if gs is not None:
                if adopted:
                    if gs["captured"]:
                        raise RuntimeError("load_state_dict would replace moment buffers a captured step holds the addresses of")
                    gs["tables"].clear()
                if step is not None:
                    gs["state"][0] = step
                gs["prepared"] = False

    def refresh_group(self, group, gs):
        hh = self._hyper_host(group)
        if gs["hyper_host"] != hh:
            gs["hyper_host"] = hh
            gs["hyper"].copy_(torch.tensor(hh, dtype=torch.float32))


class AdamW(Adam):
    decoupled_weight_decay = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad)
