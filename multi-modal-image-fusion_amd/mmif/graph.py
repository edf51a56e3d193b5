"""hipGraph capture of the launch-bound part of a train step.

A PFNet step is ~110 kernel launches; at batch 32 x 256x256 the GPU needs 5 ms for them and the host is ahead, but at the
reference's CPU-sized configuration (batch 4, 64x64: 0.2 ms of kernels) the step is pure launch latency.  `GraphedStep`
captures forward + the three losses + backward of ONE fixed batch shape into a graph (torch.cuda.CUDAGraph = hipGraph on
ROCm) and replays it per step: same kernels, same order, same results, one launch.  The gradient all-reduce (RCCL) and the
fused clip+Adam stay outside the graph -- their arguments (step count, learning rate) change every step.

What makes the engine capturable: every launch goes to torch's *current* stream (tensor.stream_ptr), buffers are leased once
per shape and reused, nothing synchronises with the host inside a step, and the weight re-pack after an optimiser step is itself
a launch (one, mmif_pack_weights_multi) that the graph simply contains.
"""
import torch

from . import dist as D
from . import engine as E


class GraphedStep:
    def __init__(self, model, loss_fn, optimizer, img1, img2, warmup=2):
        """loss_fn(img1, img2, imgf) -> tuple of 0-dim tensors, the first one is the total that gets .backward()."""
        self.model, self.loss_fn, self.opt = model, loss_fn, optimizer
        self.img1, self.img2 = img1.clone(), img2.clone()          # static inputs: replay reads these addresses
        if hasattr(optimizer, "prepare"):
            optimizer.prepare()                                    # FusedClipAdam moves the parameters into one flat buffer: before capture
        # data parallel: the warm-up backwards below have no optimizer step and the captured backward must not contain a collective:
        # no early gradient all-reduce from here to the end of the capture (the next eager optimizer step re-arms it)
        self.params = [p for p in model.parameters() if p.requires_grad]
        D.arm_early_reduce(False)
        D.drain_early()
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):                              # eager steps on a side stream: lazy init, buffer leases, allocator
            for _ in range(max(1, warmup)):
                optimizer.zero_grad(set_to_none=True)
                outs = loss_fn(self.img1, self.img2, self._forward())
                self._backward(outs[0])
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # the bf16 operand images are re-packed by the first forward after an optimiser step (host-side staleness check): mark
        # them stale now so that the captured forward CONTAINS the re-pack launch -- a replay never runs the host-side check
        E.WEIGHTS_EPOCH[0] += 1
        with torch.cuda.graph(self.graph):
            self.imgf = self._forward()
            self.outs = tuple(loss_fn(self.img1, self.img2, self.imgf))
            self._backward(self.outs[0])
        # the gradient tensors the captured backward writes; an eager step in between (ragged last batch, another shape)
        # re-points p.grad elsewhere, so every replay hands these back to the optimiser
        D.drain_early()
        self.grads = [p.grad for p in self.params]

    def _forward(self):
        E.FRESH_LEAVES[0] = self._leaves = []
        try:
            return self.model(self.img1, self.img2)
        finally:
            E.FRESH_LEAVES[0] = None

    def _backward(self, total):
        """total.backward() onto fresh leaves.  A parameter's AccumulateGrad node runs on the stream that was current when it was created
        and lives as long as any autograd graph that reaches the parameter -- a loss tensor of an earlier eager step that the caller
        still holds is enough.  Such a node runs on the DEFAULT stream, which then joins the capture through the autograd engine's event
        waits, and hipStreamEndCapture of this ROCm crashes on a null stream among the capture's parallel streams (SIGSEGV in
        hip::Stream::EndCapture; round 4: bench.py --model NestFuse --graph).  The engine models therefore hang their autograd node off
        stand-in leaves while E.FRESH_LEAVES is set (mmif/engine.py: ModelEngine.run): their accumulators are born on the current
        stream and adopt the engine's gradient views without a copy, and the views are handed to the parameters here.  (Layer-wise
        models, core/block.py, use their parameters directly: for those the old rule of torch's graph capture holds -- no autograd graph
        of an earlier default-stream step may be alive.)"""
        from core.loss import unit_gradient
        total.backward(unit_gradient(total))      # (= total.backward() without autograd's ones_like fill; core/loss.py)
        for p, q in self._leaves:
            p.grad = q.grad
        self._leaves.clear()

    def matches(self, img1, img2):
        return img1.shape == self.img1.shape and img2.shape == self.img2.shape and img1.dtype == self.img1.dtype

    def __call__(self, img1, img2, step_optimizer=True):
        """One train step on (img1, img2): replay + (all-reduce, clip, Adam).  Returns the loss tensors (static: overwritten by the
        next call) -- after the optimiser step their rank-mean is in optimizer.reduced_scalars."""
        if img1.data_ptr() != self.img1.data_ptr():
            self.img1.copy_(img1, non_blocking=True)
        if img2.data_ptr() != self.img2.data_ptr():
            self.img2.copy_(img2, non_blocking=True)
        self.graph.replay()
        for p, g in zip(self.params, self.grads):
            p.grad = g
        if step_optimizer:
            self.opt.step(scalars=list(self.outs))
        return self.outs
