"""Fused clip_grad_norm_ + Adam (+ data-parallel gradient all-reduce) on flat fp32 buffers.

Replaces train.py:72-75 (`nn.utils.clip_grad_norm_(max_norm=5)`; `optim.Adam(lr, betas).step()`) and,
when torch.distributed is initialised, DDP's bucketed gradient all-reduce plus the four
`reduce_value` scalar all-reduces of train.py:92-96 -- as ONE RCCL all-reduce of
[flat gradients | loss scalars] followed by ONE fused kernel sequence (mmif_clip_adam_step).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import dist as D
from . import engine as E
from ._lib import check, lib
from .tensor import stream_ptr

N_TAIL = E.GRAD_TAIL  # scalar slots appended to the flat gradient buffer (losses ride along in the all-reduce)


def _scalar_vector(scalars):
    """a list of 0-dim tensors, or already one 1-d tensor (core.loss.FusionLoss.values: no stack kernel)"""
    if torch.is_tensor(scalars):
        return scalars.detach().float()
    return torch.stack([s.detach().float().reshape(()) for s in scalars])


class FusedClipAdam(torch.optim.Optimizer):
    """Adam(lr, betas, eps, weight_decay=0) with optional global-norm clipping, bias-corrected exactly as
    torch.optim.Adam; parameters are re-pointed to one flat fp32 buffer on first use."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, max_norm=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, max_norm=max_norm)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise ValueError("FusedClipAdam supports a single parameter group")
        self._flat_p = self._m = self._v = self._ws = self._stage = None
        self._last_flat = None
        self._steps = 0
        self.grad_norm = None  # device float[1]: pre-clip global L2 norm of the last step
        self.reduced_scalars = None

    def _params(self):
        return [p for p in self.param_groups[0]["params"] if p.requires_grad]

    def add_param_group(self, param_group):
        if len(getattr(self, "param_groups", ())) >= 1:
            raise ValueError("FusedClipAdam supports a single parameter group (the moments live in one flat buffer)")
        super().add_param_group(param_group)

    # ---- checkpointing: the moments live in flat private buffers, so expose them in torch.optim.Adam's own layout
    # ({index: {'step', 'exp_avg', 'exp_avg_sq'}}): a resumed run continues the moments and the bias correction, and the
    # file interchanges with a torch.optim.Adam over the same parameters.
    def state_dict(self):
        sd = super().state_dict()
        state = {}
        if self._m is not None and self._steps > 0:
            off = 0
            for i, p in enumerate(self.param_groups[0]["params"]):
                if not p.requires_grad:
                    continue
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self._steps)),
                            "exp_avg": self._m[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self._v[off:off + n].view(p.shape).clone()}
                off += n
        sd["state"] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        state = state_dict.get("state", {})
        ps = self._params()
        self._steps = 0
        if not state:
            if self._m is not None:
                self._m.zero_()
                self._v.zero_()
            return
        self._flatten_params(ps)
        self._m.zero_()
        self._v.zero_()
        off = 0
        for i, p in enumerate(self.param_groups[0]["params"]):
            if not p.requires_grad:
                continue
            n = p.numel()
            st = state.get(i, state.get(str(i)))
            if st is not None:
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError(f"FusedClipAdam.load_state_dict: moment shape {tuple(st['exp_avg'].shape)} != parameter shape {tuple(p.shape)}")
                self._m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self._steps = max(self._steps, int(float(st["step"])))
            off += n

    def _flatten_params(self, ps):
        dev = ps[0].device
        total = sum(p.numel() for p in ps)
        ok = self._flat_p is not None and self._flat_p.device == dev and self._flat_p.numel() == total
        if ok:
            off = 0
            for p in ps:
                if p.data_ptr() != self._flat_p.data_ptr() + 4 * off:
                    ok = False
                    break
                off += p.numel()
        if ok:
            return
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        off = 0
        for p in ps:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
        old_m, old_v = self._m, self._v
        self._flat_p = flat
        self._m = torch.zeros_like(flat) if old_m is None or old_m.numel() != total else old_m.to(dev)
        self._v = torch.zeros_like(flat) if old_v is None or old_v.numel() != total else old_v.to(dev)
        self._ws = torch.empty(lib.mmif_clip_adam_workspace(total) // 4 + 1, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)

    def _flat_grads(self, ps):
        """The engine's flat gradient buffer when every .grad is a view of it (zero-copy), else a staging copy."""
        total = sum(p.numel() for p in ps)
        if all(p.grad is None for p in ps):
            raise RuntimeError("FusedClipAdam.step(): no parameter has a gradient (call backward() first)")
        g0 = ps[0].grad
        flat = E.FLAT_BUFFERS.get(g0.data_ptr()) if g0 is not None else None
        if flat is not None and flat.numel() == total + N_TAIL:
            off, ok = 0, True
            for p in ps:
                if p.grad is None or p.grad.data_ptr() != flat.data_ptr() + 4 * off or not p.grad.is_contiguous():
                    ok = False
                    break
                off += p.numel()
            if ok:
                return flat
        if self._stage is None or self._stage.numel() != total + N_TAIL or self._stage.device != ps[0].device:
            self._stage = torch.empty(total + N_TAIL, dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            n = p.numel()
            if p.grad is None:
                # a parameter the forward never used (PMGI's transfer1[1], Res2ConvBlock's inherited dwconv): torch.optim.Adam
                # skips it; a zero gradient does the same here (moments stay zero => zero update, no effect on the clip norm)
                self._stage[off:off + n].zero_()
            else:
                self._stage[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        return self._stage

    @torch.no_grad()
    def stage_scalars(self, scalars):
        """Call BEFORE backward() with the loss values of this step (up to 8 0-dim device tensors): they are parked in the tail slots of
        mmif/dist.py; the engine copies them into the tail slots of its flat gradient buffer so that they travel in the EARLY gradient
        all-reduce; step(scalars=the same list) then only reads them back.  A no-op (step() writes them itself) until the early path
        is armed, i.e. on the first step."""
        if scalars is None or len(scalars) == 0 or len(scalars) > N_TAIL or not D.early_reduce_armed():
            return False
        D.stage_tail(_scalar_vector(scalars))
        return True

    @torch.no_grad()
    def prepare(self):
        """Re-point the parameters to the flat buffer NOW (normally done by the first step): anything that records parameter
        addresses -- a hipGraph capture of the forward/backward (mmif.graph) -- must see the final storage."""
        self._flatten_params(self._params())

    @torch.no_grad()
    def step(self, closure=None, scalars=None):
        """scalars: optional list of up to 8 0-dim device tensors (loss values), or ONE 1-d device tensor of them (core.loss.FusionLoss
        `.values`), that are summed across
        ranks in the same all-reduce as the gradients; their rank-mean is left in `reduced_scalars`."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps = self._params()
        self._flatten_params(ps)
        flat_g = self._flat_grads(ps)
        self._last_flat = flat_g if E.FLAT_BUFFERS.get(flat_g.data_ptr()) is flat_g else None
        total = self._flat_p.numel()
        in_group = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if in_group else 1
        # (lo, hi): that range holds the result of the LATEST backward's early all-reduce (mmif/dist.py); anything else that is still
        # pending belongs to a backward this step does not consume (skipped step, another buffer): speculative copies, dropped
        early = D.take_early(flat_g) if (in_group and self._last_flat is not None) else None
        if in_group:
            D.drain_early()
        D.stage_tail(None)     # scalars parked for a backward whose early reduce never ran must not ride in a later one
        k = len(scalars) if scalars is not None else 0
        tail_done = early is not None and early[1] >= total + k and k > 0
        if k and not in_group:    # one rank: nothing to reduce, the caller's values ARE the result (no copy into the tail and back out)
            self.reduced_scalars = _scalar_vector(scalars)
        elif k and not tail_done:   # straight into the tail of the flat buffer (one small launch)
            if torch.is_tensor(scalars):
                flat_g[total:total + k].copy_(scalars.detach())
            else:
                torch.stack([s.detach().float().reshape(()) for s in scalars], out=flat_g[total:total + k])
        if in_group:
            if early is None:
                dist.all_reduce(flat_g)  # ONE collective: gradients + loss scalars (SUM); mean taken below
            else:
                lo, hi = early
                if lo > 0:
                    dist.all_reduce(flat_g[0:lo])
                if hi < total:
                    dist.all_reduce(flat_g[hi:total])
                if k and not tail_done:
                    dist.all_reduce(flat_g[total:total + N_TAIL])
            D.arm_early_reduce(self._last_flat is not None)
        if k and in_group:
            tail = flat_g[total:total + k]
            self.reduced_scalars = tail / world if world > 1 else tail.clone()
        g = self.param_groups[0]
        self._steps += 1
        p = lambda t: C.c_void_p(t.data_ptr())
        check(lib.mmif_clip_adam_step(p(self._flat_p), p(flat_g), p(self._m), p(self._v), total, g["lr"], g["betas"][0],
                                      g["betas"][1], g["eps"], self._steps, float(g["max_norm"] or 0.0), 1.0 / world,
                                      p(self.grad_norm), p(self._ws), self._ws.numel() * 4, stream_ptr()), "clip_adam_step")
        E.WEIGHTS_EPOCH[0] += 1  # packed bf16 weight images are stale now
        return loss
