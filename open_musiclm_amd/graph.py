"""HIP-graph capture of one training micro-step (forward + backward).

A coarse-small micro-step is ~700 kernel launches and ~600 buffer allocations issued from Python (about 8 ms of host time
for 37 ms of GPU work on the MI355X box; the early "launch-bound" diagnosis turned out to be a 49 ms stall inside the optimizer
step).  Shapes in training are static (fixed crops, data.py), so the whole micro-step -- id preparation, forgetful mask, weight
re-pack, embedding gather, trunk, heads, loss and the hand-written backward, all accumulating into the optimizer's
flat gradient buffer -- is captured ONCE into a HIP graph through torch's stream capture and replayed with a single
hipGraphLaunch.  Everything that varies per step lives in device memory: token ids (static input buffers),
torch's graph-safe Philox state (forgetful mask) and the dropout salt (engine.dropout_salt).
The gradient exchange and the fused optimizer stay outside the graph (3 launches + 1 collective)."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch


class GraphedForwardBackward:
    def __init__(self, fn: Callable[..., torch.Tensor], loss_scale: float = 1.0, warmup_iters: int = 2,
                 enabled: bool = True, instances: int = 2):
        """fn(**inputs) -> scalar loss (with autograd graph).  loss_scale multiplies the loss before backward
        (1 / grad_accum_every)."""
        self.fn, self.loss_scale, self.warmup_iters, self.enabled = fn, loss_scale, warmup_iters, enabled
        # Two captured instances are replayed alternately, so the launch of step k+1 never has to wait for the exec of
        # step k (a graph exec cannot be launched again while its previous launch is still running).
        self.instances = max(1, instances)
        self.graphs = []
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_in: Dict[str, torch.Tensor] = {}
        self.static_ins = []
        self.static_losses = []
        self._next = 0
        self.static_loss: Optional[torch.Tensor] = None
        self.key = None
        self.capture_error: Optional[str] = None

    def _eager(self, inputs):
        loss = self.fn(**inputs)
        (loss * self.loss_scale).backward()
        return loss.detach()

    def prepare(self, inputs: Dict[str, torch.Tensor], after_warmup: Optional[Callable[[], None]] = None):
        """Warm up eagerly (executes real steps: the caller must discard their gradients in `after_warmup`),
        then capture.  Falls back to eager launches if capture is not possible (still the HIP path)."""
        self.key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(inputs.items()))
        self.static_in = {k: v.clone() for k, v in inputs.items()}
        if not self.enabled:
            return
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(self.warmup_iters):
                self._eager(self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        try:
            for _ in range(self.instances):
                sin = {k: v.clone() for k, v in inputs.items()}
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    loss = self._eager(sin)
                self.graphs.append(g)
                self.static_ins.append(sin)
                self.static_losses.append(loss)
            self.graph = self.graphs[0]
        except Exception as e:                       # pragma: no cover - depends on runtime capture support
            self.graph, self.graphs, self.capture_error = None, [], f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
        if after_warmup is not None:
            after_warmup()
        # Everything alive now (model, optimizer state, captured graphs) is long-lived: move it out of the collector's
        # young generations so a full collection (measured: a ~65 ms pause every other step) does not rescan it.
        import gc
        gc.collect()
        gc.freeze()

    def __call__(self, **inputs) -> torch.Tensor:
        key = tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(inputs.items()))
        if self.graph is None or key != self.key:
            return self._eager(inputs)
        i = self._next
        self._next = (i + 1) % len(self.graphs)
        for k, v in inputs.items():
            self.static_ins[i][k].copy_(v, non_blocking=True)
        self.graphs[i].replay()
        return self.static_losses[i]
