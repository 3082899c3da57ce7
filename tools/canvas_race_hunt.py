"""Developer tool: hunt for the wrong pixels that ~1 tiled canvas in 7 showed right after set_conv_mode('f32') (first seen as a
one-off failure of test_tiled_canvas_matches_reference[f32-0]).  Knobs (environment): NB_SWITCH=0 no mode switch, NB_PREPACK /
NB_PREPLAN=1..4 weights / workspaces created ahead, NB_UPLOAD, NB_NOFAST, NB_NONOISE, NB_KEEPALL (no temporary freed), NB_EAGER_ENC,
NB_FORCE_LAZY, NB_NANFILL.  What they showed: not the allocator, not packing, not uninitialised memory -- timing.  The cause
(tools/gen_race_hunt.py: only `img` / RGBA of the standalone ToRGB kernel differ) is the packed-fp32 op_sel hazard described at
NB_NO_PACKED_F32 in csrc/nb_common.h; with those kernels compiled without packed fp32 ops this tool finds nothing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import load_golden
from brushstroke_engine_amd import config as cfgmod, weights as wmod, encoder as encmod, painting
from brushstroke_engine_amd.networks import Generator
g = load_golden("engine_r128.npz")
cfg = cfgmod.style1_config(128); sd = wmod.random_state_dict(cfg, seed=0); esd = encmod.random_encoder_state_dict(5)
z = np.random.RandomState(594).randn(1, cfg.z_dim)
G = Generator(cfg, sd, conv_mode="f32").to("cuda"); enc = encmod.HipGeometryEncoder(esd)
ops = painting.TileOps(G, enc)
from brushstroke_engine_amd import networks as _nw
if os.environ.get("NB_UPLOAD") == "pinned":
    def _up(self, descs):
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        return torch.from_numpy(raw).pin_memory().to(self.device, non_blocking=True)
    _nw._Plan._upload = _up
if os.environ.get("NB_UPLOAD") == "main":
    def _up(self, descs):
        raw = np.frombuffer(bytes(descs), dtype=np.uint8).copy()
        cur = torch.cuda.current_stream()
        with torch.cuda.stream(torch.cuda.default_stream()):
            t = torch.from_numpy(raw).to(self.device)
        t.record_stream(cur)
        return t
    _nw._Plan._upload = _up
if os.environ.get('NB_NOFAST'): G.synthesis._styles_fast = False
if os.environ.get('NB_NONOISE'):
    with torch.no_grad():
        for nme, prm in G.named_parameters():
            if nme.endswith('noise_strength'): prm.zero_()
    G._invalidate()
KEEP = []
if os.environ.get('NB_KEEPALL'):
    _empty, _el, _zeros = torch.empty, torch.empty_like, torch.zeros
    def keep(fn):
        def w(*a, **k):
            t = fn(*a, **k); KEEP.append(t); return t
        return w
    torch.empty, torch.empty_like, torch.zeros = keep(_empty), keep(_el), keep(_zeros)
    _contig = torch.Tensor.contiguous
    def contig(self, *a, **k):
        t = _contig(self, *a, **k); KEEP.append(t); return t
    torch.Tensor.contiguous = contig
    _to = torch.Tensor.to
    def to_(self, *a, **k):
        t = _to(self, *a, **k); KEEP.append(t); return t
    torch.Tensor.to = to_
if os.environ.get('NB_EAGER_ENC'): painting.TileOps.lazy_geometry = False
if os.environ.get('NB_FORCE_LAZY'):
    painting.TileOps.encode = lambda self, geom: self.encoder.lazy(geom)
    painting.TileOps.prepare = lambda self, n, slots: None
if os.environ.get('NB_NANFILL'):
    _e0, _el0 = torch.empty, torch.empty_like
    def _fill(t):
        if t.is_cuda and t.numel():
            if t.dtype in (torch.float32, torch.float16): t.fill_(float('nan'))
            elif t.dtype == torch.uint8: t.fill_(77)
        return t
    torch.empty = lambda *a, **k: _fill(_e0(*a, **k))
    torch.empty_like = lambda *a, **k: _fill(_el0(*a, **k))
ref = None
for it in range(80):
    KEEP.clear() if False else None
    if os.environ.get("NB_SWITCH", "1") == "1": G.set_conv_mode("f32")
    if os.environ.get("NB_PREPACK"):
        G.synthesis._n, G.synthesis._h3_batch_ok = 4, False
        G.synthesis._ensure_packed(); torch.cuda.synchronize()
    if os.environ.get("NB_PREPLAN") == "1":
        for sl in (8, 9): G.synthesis._get_plan(4, torch.device("cuda", 0), sl)
        torch.cuda.synchronize()
    if os.environ.get("NB_PREPLAN") == "2":           # on the side streams, like the schedule does, but finished before painting
        for k_, sl in enumerate((8, 9)):
            with ops.stream(k_):
                G.synthesis._get_plan(4, torch.device("cuda", 0), sl)
        ops.join_streams(); torch.cuda.synchronize()
    if os.environ.get("NB_PREPLAN") == "3":           # only slot 9 ahead
        G.synthesis._get_plan(4, torch.device("cuda", 0), 9); torch.cuda.synchronize()
    if os.environ.get("NB_PREPLAN") == "4":           # only slot 8 ahead
        G.synthesis._get_plan(4, torch.device("cuda", 0), 8); torch.cuda.synchronize()
    helper = painting.PaintingHelper(ops, batch=4); helper.set_feature_blending(0)
    opts = painting.GanBrushOptions(); opts.set_style(torch.from_numpy(z), 594)
    out, full, crops, padded = helper.paint_image(g["geom"], opts, crop_margin=int(g["crop_margin"]), return_full=True)
    torch.cuda.synchronize(); KEEP.clear()
    if ref is None: ref = full.copy(); continue
    d = np.abs(full.astype(np.int32) - ref.astype(np.int32))
    if d.max() > 0:
        idx = np.argwhere(d > 0)
        ys, xs, cs = idx[:, 0], idx[:, 1], idx[:, 2]
        print("it", it, "n", len(idx), "rows", sorted(set(ys.tolist()))[:6], "x range", xs.min(), xs.max(), "x mod 4", sorted(set((xs % 4).tolist())), "ch", sorted(set(cs.tolist())),
              "got", full[ys[0], xs[0]].tolist(), "want", ref[ys[0], xs[0]].tolist(), "crops", [tuple(c) for c in np.asarray(crops)[:4].tolist()])
print("done")
