"""hipGraph-captured generator step for the interactive (batch-1) configuration.

The interactive path of the reference renders ONE patch per websocket request
(``forger/ui/util.py:175-195`` -> ``PaintingHelper.render_stroke``), where ~100 tiny launches per patch
make it launch-latency bound.  Here the whole step -- mapping, styles, noise, 15 fused conv
launches (the >= 128x128 layers on the split-f16 kernels, the small ones on the fp32 split-K kernels), ToRGB +
compositing fused into the last one -- is about 20 launches, captured once into a hipGraph (static shapes;
z / ws, geometry features, positions and user colors are graph inputs that are overwritten in place)
and replayed with a single ``hipGraphLaunch``.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from .networks import Generator


class GraphedTriadRender:
    """Capture ``Generator.render_triad`` for a fixed batch size; call it like a function."""

    def __init__(self, G: Generator, batch: int = 1, render_mode: str = "clear", use_ws: bool = False,
                 use_positions: bool = True, want_f32: bool = False, warmup: int = 3, plan_slot: Optional[int] = None):
        cfg = G.cfg
        dev = G.synthesis.get_last_block().conv1.weight.device
        assert dev.type == "cuda"
        self.G, self.batch, self.use_ws = G, batch, use_ws
        self.z = torch.zeros([batch, cfg.z_dim], dtype=torch.float32, device=dev)
        self.ws = torch.zeros([batch, cfg.num_ws, cfg.w_dim], dtype=torch.float32, device=dev)
        self.geom = [torch.zeros([batch, c, r, r], dtype=torch.float32, device=dev)
                     for c, r in zip(cfg.geom_feature_channels, cfg.geom_feature_resolutions)]
        self.positions = torch.zeros([batch, 2], dtype=torch.int64, device=dev) if use_positions else None
        self.user_colors = torch.full([batch, 3, 3], float("nan"), dtype=torch.float32, device=dev)
        self._kw = dict(geom_feature=self.geom, positions=self.positions, render_mode=render_mode,
                        user_colors=self.user_colors, want_u8=True, want_f32=want_f32)
        if plan_slot is not None:
            # own workspace: graphs that replay concurrently on different streams must not share one (slot 0 is the
            # default of eager calls; the sub-batch streams of an eager split call use 1 .. sub_streams)
            self._kw["_plan_slot"] = plan_slot
        # warm up on a side stream (creates the plan / zero page / packed weights outside the capture)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._run()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out_u8, self.out_f32, self.out_dbg = self._run()

    def _run(self):
        if self.use_ws:
            return self.G.render_triad(ws=self.ws, **self._kw)
        return self.G.render_triad(z=self.z, **self._kw)

    def set_inputs(self, z=None, ws=None, geom_feature: Optional[List[torch.Tensor]] = None, positions=None,
                   user_colors=None) -> None:
        if z is not None:
            self.z.copy_(z)
        if ws is not None:
            self.ws.copy_(ws)
        if geom_feature is not None:
            for dst, src in zip(self.geom, geom_feature):
                dst.copy_(src)
        if positions is not None and self.positions is not None:
            self.positions.copy_(positions)
        if user_colors is not None:
            self.user_colors.copy_(user_colors)

    def replay(self):
        self.graph.replay()
        return self.out_u8, self.out_f32, self.out_dbg

    def __call__(self, **inputs):
        self.set_inputs(**inputs)
        return self.replay()
