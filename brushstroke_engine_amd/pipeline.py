"""Throughput mode of the generator: independent batches software-pipelined over two HIP streams.

A batch-32 step is ~20 launches.  The first ones -- mapping network, all layers' styles, the small layers' noise and the
five <= 16 x 16 convolutions -- hold 2 % of the step's FLOPs but a tenth of its time: they are latency-bound launches of a
few hundred small workgroups that leave most of the 256 CUs idle.  Batches of a throughput job (the tiles of a canvas, the
patches of ``bench.py``'s steps) are independent, so this HEAD of batch k+1 can run on a second stream while the big
convolutions of batch k -- the TAIL -- own the chip on the first: the generator's split entry (``_stop_after`` /
``_resume``, the same one the tiled-canvas schedule uses for feature blending) cuts the forward pass after block
``split_res``, the head pass leaves every layer's styles and demodulation coefficients in its workspace slot and the tail
pass of the same batch picks them up (``_reuse_styles``).  Tails run back to back on one stream; only small head
kernels ever share the chip with them, so a big kernel's duration stays what it is alone.

The reference has no counterpart: it evaluates one patch at a time on one stream (forger/ui/brush.py:731-805).
Results are bit-identical to ``Generator.render_triad`` of the same batch (tests/test_hip_generator.py).
"""
from __future__ import annotations

from typing import List, Optional

import torch

PIPE_SLOT0 = 24          # workspace slots of the pipeline (generator sub-batches: 1.., painting: 8.., graphs: 16..)


class TriadStepPipeline:
    """``submit(z, geom_feature, positions)`` enqueues one batch and returns its uint8 RGBA tiles ``[N, R, R, 4]`` -- valid
    on ``tail_stream`` (make a consumer stream wait with ``wait()`` / ``flush()``, or synchronise the device; the tiles are
    recorded on the submitting stream for the caching allocator and the pipeline keeps no reference to them).  Up to
    ``depth`` batches are in flight: the tail of batch k and the head of batch k+1."""

    def __init__(self, G, split_res: int = 16, depth: int = 2, render_mode: str = "clear"):
        dev = G.synthesis.get_last_block().conv1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("TriadStepPipeline needs the generator on a GPU")
        if split_res not in G.synthesis.block_resolutions or split_res >= G.img_resolution:
            raise RuntimeError(f"split_res {split_res} is not an inner block resolution of this generator")
        self.G, self.device, self.split_res, self.depth, self.render_mode = G, dev, int(split_res), int(depth), render_mode
        self.head_stream = torch.cuda.Stream(device=dev)
        self.tail_stream = torch.cuda.Stream(device=dev)
        self._tail_done: List[Optional[torch.cuda.Event]] = [None] * self.depth      # slot free again (its tail has finished)
        self._k = 0
        self._pending = None          # (slot, ws, x, geom, positions, user_colors, sfactor, head event)
        self._forked = False

    # -- the two halves of one batch --
    def _head(self, slot, z, geom, positions):
        G = self.G
        ws = G.mapping(z, None)
        x = G.forward_pre_mapped(ws, geom, positions=positions, noise_mode="const", _stop_after=self.split_res,
                                 _plan_slot=PIPE_SLOT0 + slot)
        return ws, x

    def _tail(self, slot, ws, x, geom, positions, user_colors, sfactor):
        u8, _, _ = self.G.render_triad(ws=ws, geom_feature=geom, positions=positions, render_mode=self.render_mode,
                                       user_colors=user_colors, sfactor=sfactor, _resume=(self.split_res, x),
                                       _plan_slot=PIPE_SLOT0 + slot, _reuse_styles=True)
        return u8

    def _eligible(self, n: int) -> bool:
        """The tail may reuse the head's workspace only if every resumed layer computes its noise itself (large split-f16
        kernels); otherwise ``submit`` falls back to the unsplit call on the tail stream."""
        syn = self.G.synthesis
        if not (syn.noise_in_kernel and syn.conv_mode in ("h3", "f8", "f16")):
            return False
        syn._n, syn._h3_batch_ok = n, n >= syn.h3_min_batch
        syn._ensure_packed()
        return all((syn._h3_up2_eligible(sp) if sp.up == 2 else syn._h3_eligible(sp))
                   for sp in syn.cfg.layers if sp.block_res > self.split_res)

    def submit(self, z, geom_feature, positions, user_colors=None, sfactor=None) -> torch.Tensor:
        if positions is None:
            raise RuntimeError("TriadStepPipeline renders positioned patches (shifted noise); use Generator.render_triad otherwise")
        cur = torch.cuda.current_stream(self.device)
        if not self._forked:                       # inputs were produced on the caller's stream
            self.head_stream.wait_stream(cur)
            self.tail_stream.wait_stream(cur)
            self._forked = True
        geom = list(geom_feature)
        n = z.shape[0]
        if not self._eligible(n):
            # unsplit pass on the tail stream, in workspace slot 0 of the pipeline: like a tail it may start only once the slot's
            # previous tail is through (stream order) AND it must keep the next HEAD of that slot -- which rewrites the slot's
            # styles on the head stream -- waiting until it has finished (mixed batch sizes: eligible and ineligible submits)
            with torch.cuda.stream(self.tail_stream):
                u8, _, _ = self.G.render_triad(z=z, geom_feature=geom, positions=positions, render_mode=self.render_mode,
                                               user_colors=user_colors, sfactor=sfactor, _plan_slot=PIPE_SLOT0)
                done = torch.cuda.Event()
                done.record(self.tail_stream)
            self._tail_done[0] = done
            u8.record_stream(cur)
            return u8
        slot = self._k % self.depth
        self._k += 1
        # head of this batch: on the head stream, once the slot's previous tail has finished with the workspace
        if self._tail_done[slot] is not None:
            self.head_stream.wait_event(self._tail_done[slot])
        with torch.cuda.stream(self.head_stream):
            ws, x = self._head(slot, z, geom, positions)
            ev = torch.cuda.Event()
            ev.record(self.head_stream)
        for t in (ws, x):
            t.record_stream(self.tail_stream)       # allocated on the head stream, read by the tail
        # tail of this batch: behind the previous batch's tail on the tail stream
        self.tail_stream.wait_event(ev)
        with torch.cuda.stream(self.tail_stream):
            u8 = self._tail(slot, ws, x, geom, positions, user_colors, sfactor)
            done = torch.cuda.Event()
            done.record(self.tail_stream)
        self._tail_done[slot] = done
        u8.record_stream(cur)                      # allocated on the pipeline's stream, consumed on the caller's (see wait())
        return u8

    def wait(self, stream=None) -> None:
        """Make ``stream`` (default: the caller's current stream) wait for everything submitted so far.  The tiles returned by
        ``submit`` were allocated on the tail stream; ``submit`` itself records each of them on the stream it was called from
        (``Tensor.record_stream``), so the caching allocator does not hand their memory to a later step while a read or copy
        enqueued on that stream is pending, and the pipeline keeps NO reference to them (a caller that drops a tile without ever
        calling ``wait`` -- e.g. a benchmark loop that only synchronises the device -- frees it).  A consumer on yet another
        stream calls ``tile.record_stream(that_stream)`` itself."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        stream.wait_stream(self.tail_stream)
        stream.wait_stream(self.head_stream)

    def flush(self) -> None:
        self.wait()
        self._forked = False


PREFETCH_SLOT0 = 28      # workspace slots of the prefetch pipeline


class TriadPrefetchPipeline:
    """The lighter software pipeline: only what a step needs BEFORE its first layer runs ahead.

    Mapping network, all layers' styles and demodulation coefficients, the small layers' noise images and the early geometry
    packs (fp32 geometry features x the consumer's styles -> operand format: 265 MB of HBM traffic per batch of 32 at R=256)
    hold no matrix work; enqueued in front of a step they are 35 us of latency-bound launches plus a memory-bound kernel that
    costs whatever layers it runs beside ~45 us.  Here they run for step k+1 on a side stream, released by an event recorded
    right before the LAST layer of step k (64 -> 64 @ 256 with the fused ToRGB: a matrix-bound launch with HBM bandwidth to
    spare) -- the dominant up=2 launches keep the chip to themselves.  ``Synthesis.forward(_prepare_only=True)`` returns the
    handle (packed operand tensors + events; styles etc. sit in the workspace slot) that the step's own pass starts from
    (``_prepared``).  Results are bit-identical to ``Generator.render_triad`` (tests/test_hip_generator.py).

    ``submit`` returns the batch's uint8 RGBA tiles ``[N, R, R, 4]``, valid on ``main_stream`` (``wait()`` / ``flush()``)."""

    def __init__(self, G, depth: int = 2, render_mode: str = "clear", mark_layer: Optional[str] = None):
        dev = G.synthesis.get_last_block().conv1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("TriadPrefetchPipeline needs the generator on a GPU")
        self.G, self.device, self.depth, self.render_mode = G, dev, int(depth), render_mode
        self.mark_layer = mark_layer or G.synthesis.cfg.layers[-1].name
        self.prep_stream = torch.cuda.Stream(device=dev)
        self.main_stream = torch.cuda.Stream(device=dev)
        self._main_done: List[Optional[torch.cuda.Event]] = [None] * self.depth     # slot free again (its pass has finished)
        self._mark_prev: Optional[torch.cuda.Event] = None                          # previous step reached its marked layer
        self._k = 0
        self._forked = False

    def submit(self, z, geom_feature, positions, user_colors=None, sfactor=None) -> torch.Tensor:
        if positions is None:
            raise RuntimeError("TriadPrefetchPipeline renders positioned patches (shifted noise); use Generator.render_triad otherwise")
        G = self.G
        cur = torch.cuda.current_stream(self.device)
        if not self._forked:                       # inputs were produced on the caller's stream
            self.prep_stream.wait_stream(cur)
            self.main_stream.wait_stream(cur)
            self._forked = True
        geom = list(geom_feature)
        slot = self._k % self.depth
        self._k += 1
        # preparation of this step: once the slot's previous pass has finished with the workspace, and not before the previous
        # step has reached its last layer
        if self._main_done[slot] is not None:
            self.prep_stream.wait_event(self._main_done[slot])
        if self._mark_prev is not None:
            self.prep_stream.wait_event(self._mark_prev)
        with torch.cuda.stream(self.prep_stream):
            ws = G.mapping(z, None)
            handle = G.synthesis(ws, geom, noise_mode="const", _positions=positions, _plan_slot=PREFETCH_SLOT0 + slot,
                                 _prepare_only=True)
            ready = torch.cuda.Event()
            ready.record(self.prep_stream)
        for t in [ws] + [dst for dst, _ in handle["pre_h2"].values()] + [t_ for t_ in handle["keep"] if isinstance(t_, torch.Tensor)]:
            t.record_stream(self.main_stream)       # allocated on the side stream, read by the step's own pass
        self.main_stream.wait_event(ready)
        mark = torch.cuda.Event()
        with torch.cuda.stream(self.main_stream):
            u8, _, _ = G.render_triad(ws=ws, geom_feature=geom, positions=positions, render_mode=self.render_mode,
                                      user_colors=user_colors, sfactor=sfactor, _plan_slot=PREFETCH_SLOT0 + slot,
                                      _prepared=handle, _mark=(mark, self.mark_layer))
            done = torch.cuda.Event()
            done.record(self.main_stream)
        self._main_done[slot] = done
        self._mark_prev = mark
        u8.record_stream(cur)                      # allocated on the pipeline's stream, consumed on the caller's (see wait())
        return u8

    def wait(self, stream=None) -> None:
        """Make ``stream`` (default: the caller's current stream) wait for everything submitted so far (the tiles were recorded
        on the submitting stream by ``submit``; see TriadStepPipeline.wait)."""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        stream.wait_stream(self.main_stream)
        stream.wait_stream(self.prep_stream)

    def flush(self) -> None:
        self.wait()
        self._forked = False


ROUND_SLOT0 = 32         # workspace slots of the round-robin schedule (one per stream)


class RoundRobinStreams:
    """``k`` HIP streams that independent work items are dealt to in turn: item ``i`` runs on stream ``i % k`` with the generator
    workspace slot ``slot0 + i % k`` (a slot's styles / coefficient / noise buffers are rewritten by every pass, so passes that may
    overlap need their own; passes on ONE stream are ordered by it).  The kernels of the other chains fill the CUs a chain leaves idle
    in its kernel tails and in its small, latency-bound launches (a fifth of a batch-32 step at R=256).  Shared by the throughput
    schedule of whole steps (``ConcurrentTriadSteps``, ``bench.py``) and the tiled-canvas schedule (``painting.TileOps``)."""

    def __init__(self, device, k: int, slot0: int = ROUND_SLOT0):
        if k < 1:
            raise RuntimeError("RoundRobinStreams needs at least one stream")
        self.device, self.k, self.slot0 = device, int(k), int(slot0)
        self.streams = [torch.cuda.Stream(device=device) for _ in range(self.k)]
        self._forked = set()

    def slot(self, i: int) -> int:
        return self.slot0 + i % self.k

    def stream_of(self, i: int) -> "torch.cuda.Stream":
        return self.streams[i % self.k]

    def stream(self, i: int):
        """Context manager: work of item ``i`` goes to stream ``i % k``, which first (once per fork) waits for the caller's stream."""
        j = i % self.k
        if j not in self._forked:
            self.streams[j].wait_stream(torch.cuda.current_stream(self.device))
            self._forked.add(j)
        return torch.cuda.stream(self.streams[j])

    def join(self, tensors=(), stream=None) -> None:
        """``stream`` (default: the caller's current one) waits for all k streams; ``tensors`` produced there are about to be read
        here.  The next ``stream(i)`` forks from the caller's stream again."""
        main = torch.cuda.current_stream(self.device) if stream is None else stream
        for st in self.streams:
            main.wait_stream(st)
        self._forked = set()
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(main)


class ConcurrentTriadSteps:
    """The default throughput schedule of whole generator steps: up to ``streams`` independent batches in flight, dealt round-robin to
    that many HIP streams, each with its own workspace slot (``RoundRobinStreams``).  Every batch is the plain
    ``Generator.render_triad`` pass, so results are bit-identical to the serial loop (tests/test_hip_generator.py).  The reference's
    counterpart is the batched generate loop of ``forger/metrics/util.py:280-292`` (one batch after the other on one stream).

    ``submit`` returns the batch's uint8 RGBA tiles ``[N, R, R, 4]``, valid on the stream they were rendered on (``last_stream``);
    ``wait()`` makes the caller's stream wait for everything submitted.  ``streams=0`` picks 1 or 3 streams by a probe
    (``choose``): concurrent chains gain 4-9 % on most boxes and lost 14 % on one (three chains of 156 KB-LDS workgroups evicting each
    other at kernel boundaries), so the schedule measures instead of assuming."""

    def __init__(self, G, streams: int = 3, render_mode: str = "clear", slot0: int = ROUND_SLOT0):
        dev = G.synthesis.get_last_block().conv1.weight.device
        if dev.type != "cuda":
            raise RuntimeError("ConcurrentTriadSteps needs the generator on a GPU")
        self.G, self.device, self.render_mode, self.slot0 = G, dev, render_mode, slot0
        self.probe = None
        self._rr = {}
        self._set_streams(max(1, int(streams)) if streams else 3)
        self._auto = not streams
        self._i = 0
        self.last_stream = None

    def _set_streams(self, k: int) -> None:
        if k not in self._rr:
            self._rr[k] = RoundRobinStreams(self.device, k, self.slot0)
        self.rr = self._rr[k]
        self.streams = k

    def choose(self, z, geom_feature, positions, rounds: int = 4, steps: int = 6) -> int:
        """1 or 3 streams for batches like this one: ``rounds`` interleaved pairs of ``steps`` steps each, the MEDIAN of each; three if
        that is at least 1 % faster HERE, else one.  (Until the end of round 6: best of each, three unless 3 % slower -- on one box of the
        pool three streams ran bimodal, 17 300 or 19 800 patches/s from run to run against a steady 18 000 on one, and the best-of probe
        kept choosing them.)  ~50 steps' worth of time; the figures stay in ``probe``."""
        import time
        times = {1: [], 3: []}

        def run(k, nsteps):
            self._set_streams(k)
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(nsteps):
                self.submit(z, geom_feature, positions)
            self.wait()
            torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / nsteps * 1e3
        run(1, 2); run(3, 3)                              # workspaces of the slots, code objects
        for _ in range(rounds):
            times[1].append(run(1, steps))
            times[3].append(run(3, steps))
        best = {k: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2]) for k, v in times.items()}
        pick = 3 if best[3] <= 0.99 * best[1] else 1
        self._set_streams(pick)
        self.probe = {"ms_per_step": {str(k): round(v, 4) for k, v in best.items()}, "chosen": pick, "batch": int(z.shape[0])}
        return pick

    def submit(self, z=None, geom_feature=None, positions=None, ws=None, user_colors=None, sfactor=None) -> torch.Tensor:
        if self._auto and self.probe is None:
            self._auto = False                              # (choose() itself submits)
            if z is not None and ws is None:
                self.choose(z, geom_feature, positions)
        i, self._i = self._i, self._i + 1
        cur = torch.cuda.current_stream(self.device)
        with self.rr.stream(i):
            u8, _, _ = self.G.render_triad(z=z, ws=ws, geom_feature=geom_feature, positions=positions, render_mode=self.render_mode,
                                           user_colors=user_colors, sfactor=sfactor, _plan_slot=self.rr.slot(i))
        self.last_stream = self.rr.stream_of(i)
        u8.record_stream(cur)             # allocated on a side stream, consumed on the caller's (after wait()); no reference is kept
        return u8

    def wait(self, stream=None) -> None:
        """Make ``stream`` (default: the caller's current one) wait for every step submitted so far."""
        self.rr.join(stream=stream)
