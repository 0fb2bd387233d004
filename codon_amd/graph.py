"""HIP-graph replay of the inference forward for launch-bound sizes.

At the reference's own use case (one image per call, test.py:125) the ~95 kernel launches of a forward cost more
than the kernels at small sizes (config 0: 1x128x128).  The forward is a fixed launch sequence with no host
synchronisation and no data-dependent control flow, so it is captured ONCE into a hipGraph (torch.cuda.CUDAGraph:
capture stream, graph-private memory pool) and replayed per call: one launch instead of ~95."""
from __future__ import annotations

import torch


class GraphedCODON:
    def __init__(self, model, example_x: torch.Tensor, example_y: torch.Tensor, warmup: int = 2):
        if not example_x.is_cuda:
            raise RuntimeError("GraphedCODON needs HIP tensors")
        self.model = model.eval()
        self.x = example_x.detach().clone().float().contiguous()
        self.y = example_y.detach().clone().float().contiguous()
        side = torch.cuda.Stream(device=self.x.device)
        side.wait_stream(torch.cuda.current_stream(self.x.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                      # packs the weights, warms the allocator
                self.model._forward_impl(self.x, self.y, None)
        torch.cuda.current_stream(self.x.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self.model._forward_impl(self.x, self.y, None)

    def __call__(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        if x.shape != self.x.shape:
            raise RuntimeError(f"GraphedCODON was captured for {tuple(self.x.shape)}, got {tuple(x.shape)}")
        self.x.copy_(x)
        self.y.copy_(y)
        self.graph.replay()
        return self.out.clone()
