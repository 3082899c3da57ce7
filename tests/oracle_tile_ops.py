"""Oracle-backed stand-in for ``brushstroke_engine_amd.painting.TileOps`` (TEST ONLY, CPU): lets the host logic and
the (sharded) three-phase schedule of ``PaintingHelper`` run on CPU against the reference-generated canvases.  Every
device operation is restated with plain torch; the canvas replay is the reference's sequential tile loop."""
import numpy as np
import torch

from oracle import neube_oracle as no
from oracle import painting_oracle as po


class OracleTileOps:
    def __init__(self, cfg, sd, esd, preproc_type=None):
        self.G = no.OracleGenerator(cfg, sd)
        self.cfg, self.esd, self.preproc = cfg, esd, preproc_type
        self.device = torch.device("cpu")
        self.patch_width = cfg.img_resolution

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def geom_tiles(self, geom, tile_yx):
        r = self.patch_width
        out = torch.empty([tile_yx.shape[0], 1, r, r], dtype=torch.float32)
        for i, (y, x) in enumerate(tile_yx.tolist()):
            patch = 255 - geom[y:y + r, x:x + r]                      # uint8, paint_image_main.py:162
            out[i, 0] = 1 - patch.to(torch.float32) / 255.0           # brush.py:679
        return out

    def encode(self, geom):
        return po.encoder_encode(self.esd, geom, self.preproc)

    def map_style(self, z=None, ws=None):
        return ws.to(torch.float32) if ws is not None else self.G.mapping(z, None)

    def head(self, ws, geom_feats, positions, stop_res, slot=0):
        _, dbg = self.G.forward_pre_mapped(ws, geom_feats, positions=positions, return_features=[stop_res])
        return dbg[f"features{stop_res}_preblend"]

    def tail(self, ws, feats, geom_feats, positions, resume_res, render_mode, user_colors, sfactor=None, slot=0):
        one = torch.ones([1, 1, resume_res, resume_res])
        _, dbg = self.G.forward_pre_mapped(ws, geom_feats, positions=positions, return_debug_data=True,
                                           blended_features={resume_res: {"features": feats, "alpha": one}})
        rgba = no.triad_composite(dbg["uvs"], dbg["colors"], render_mode, user_colors, sfactor)
        return no.rgba_to_uint8(rgba).permute(0, 2, 3, 1).contiguous()

    def full(self, ws, geom_feats, positions, render_mode, user_colors, sfactor=None, slot=0):
        _, dbg = self.G.forward_pre_mapped(ws, geom_feats, positions=positions, return_debug_data=True)
        rgba = no.triad_composite(dbg["uvs"], dbg["colors"], render_mode, user_colors, sfactor)
        return no.rgba_to_uint8(rgba).permute(0, 2, 3, 1).contiguous()

    def background_weight(self, ws, geom_feats):
        _, dbg = self.G.forward_pre_mapped(ws, geom_feats, return_debug_data=True)
        return dbg["uvs"][:, 2:3]

    def new_feature_canvas(self, c, hc, wc):
        return torch.zeros([1, c, hc, wc]), torch.zeros([hc, wc], dtype=torch.uint8)

    def replay(self, tiles, tile_yx, alpha0, crop, canvas, mask, cell_off, cell_tiles, box=None):
        return sequential_replay(tiles, tile_yx, alpha0, crop, canvas, mask)

    def replay_pieces(self, pieces, hw, alpha0, crop, canvas, mask, cell_off, cell_pieces, box):
        return sequential_replay_pieces(pieces, hw, alpha0, crop, canvas, mask)

    def paste(self, canvas_u8, tiles_u8, dst_yx, crop, cell_off, cell_tiles):
        r = tiles_u8.shape[1]
        for t, (y, x) in enumerate(dst_yx.tolist()):
            canvas_u8[y + crop:y + r - crop, x + crop:x + r - crop] = tiles_u8[t, crop:r - crop, crop:r - crop]


def sequential_replay(tiles, tile_yx, alpha0, crop, canvas, mask):
    """brush.py:190-227 + stitching.py:24-25 + brush.py:82-92, tile after tile (in place); returns the new mask."""
    mask = mask.clone().bool()
    hw = tiles.shape[-1]
    hc, wc = mask.shape
    for t, (y, x) in enumerate(tile_yx.tolist()):
        if y < 0 or x < 0 or y + hw > hc or x + hw > wc:
            continue
        m = mask[y:y + hw, x:x + hw]
        upd = (alpha0 > 0.99) | (m & (alpha0 > 0))
        a = alpha0.clone()
        a[~m] = 1
        a = 1 - a
        if crop > 0:
            upd[:crop, :] = False
            upd[-crop:, :] = False
            upd[:, :crop] = False
            upd[:, -crop:] = False
        f = canvas[0, :, y:y + hw, x:x + hw]
        tiles[t] = a * f + (1 - a) * tiles[t]
        u = upd[None].expand(tiles.shape[1], -1, -1)
        f[u] = tiles[t][u]
        m[upd] = True
    return mask.to(torch.uint8)


def sequential_replay_pieces(pieces, hw, alpha0, crop, canvas, mask):
    """The same loop over rectangular PIECES of tiles (view [C,h,w], cy, cx, ly0, lx0) in paint order -- what a rank
    of the halo-exchange schedule replays: received strips of earlier foreign tiles, then its own full tiles."""
    mask = mask.clone().bool()
    inner = torch.zeros([hw, hw], dtype=torch.bool)
    inner[crop:hw - crop, crop:hw - crop] = True
    for v, cy, cx, ly0, lx0 in pieces:
        h, w = v.shape[1:]
        a0 = alpha0[ly0:ly0 + h, lx0:lx0 + w]
        m = mask[cy:cy + h, cx:cx + w]
        upd = ((a0 > 0.99) | (m & (a0 > 0))) & inner[ly0:ly0 + h, lx0:lx0 + w]
        a = a0.clone()
        a[~m] = 1
        a = 1 - a
        f = canvas[0, :, cy:cy + h, cx:cx + w]
        v.copy_(a * f + (1 - a) * v)
        u = upd[None].expand(v.shape[0], -1, -1)
        f[u] = v[u]
        m[upd] = True
    return mask.to(torch.uint8)
