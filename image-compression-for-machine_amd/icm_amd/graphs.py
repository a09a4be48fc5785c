"""hipGraph replay of the eval-mode forward pass (inference serving).

A forward pass of these codecs is 300-400 kernel launches for the cnn model and several thousand for the stf variants;
at small batch sizes each kernel runs for a few microseconds and the step is bound by the host issuing launches through
Python.  ``GraphedForward`` captures the whole launch sequence of one ``model(x)`` call in eval mode into a HIP graph
(``torch.cuda.CUDAGraph``: every ``icm_*`` entry point enqueues on the stream it is handed, never allocates or
synchronises, so the sequence is capturable as is) and replays it for later inputs of the same shape: one host call per
forward.  The graph reads the model's eval-mode packed weight copies (``CompressionModel._pack_cache``); when a parameter
changes (training between evaluations, ``load_state_dict``) that cache is rebuilt and the graph is captured again on the
next call, so replays never see stale weights.

    fwd = GraphedForward(net)            # net.eval()
    out = fwd(x)                         # first call per input shape: warm-up + capture; then replay

The returned tensors are the graph's static output buffers: they are overwritten by the next call with the same input
shape (clone them to keep them).  Training steps are not captured: they draw fresh noise and bump Adam's step count
on the host every iteration."""
from __future__ import annotations

from typing import Dict, Tuple

import torch


class GraphedForward:
    def __init__(self, model: torch.nn.Module, warmup: int = 2):
        self.model = model
        self.warmup = max(1, int(warmup))
        self._graphs: Dict[Tuple, tuple] = {}

    def _key(self, x: torch.Tensor):
        return (tuple(x.shape), x.device.index)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> dict:
        if self.model.training:
            raise RuntimeError("GraphedForward replays the eval-mode forward: call model.eval() first")
        if not x.is_cuda or x.dtype != torch.float32:
            raise ValueError("GraphedForward expects a float32 device tensor")
        k = self._key(x)
        self.model._pack_cache()                      # refreshes the packed-weight cache if a parameter changed
        sig = getattr(self.model, "_pack_sig", None)
        ent = self._graphs.get(k)
        if ent is not None and ent[3] != sig:         # captured against packed copies that no longer exist
            ent = None
        if ent is None:
            static_x = x.detach().clone().contiguous()
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # warm-up off the capture: kernel attributes, allocator pools
                for _ in range(self.warmup):
                    self.model(static_x)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.model(static_x)
            ent = (g, static_x, out, getattr(self.model, "_pack_sig", None))
            self._graphs[k] = ent
        g, static_x, out, _ = ent
        static_x.copy_(x)
        g.replay()
        return out

    def clear(self):
        self._graphs.clear()
