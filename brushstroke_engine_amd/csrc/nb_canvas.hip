// Canvas-side kernels of the painting engine for gfx950 (SURVEY 8 rows e/f2): geometry tile extraction,
// the feature-canvas blending replay, and the RGBA tile paste.  All three are pointwise and HBM-bound; the
// point of writing them as kernels is the SCHEDULE: the reference walks the tiles of a canvas one by one
// (forger/viz/paint_image_main.py:157-177) because tile t blends against features that tiles < t left on the
// feature canvas (forger/ui/brush.py:190-227, 239-242).  That dependency is pointwise per canvas pixel, so here
// one launch replays the whole tile sequence: every thread owns one feature-canvas pixel (x CG channels), keeps
// the canvas value and mask bit in registers and walks the tiles that cover it in the reference's order.
#include "nb_common.h"

// ------------------------------------------------------------------------------------------------
// geometry tiles: out[t,0,y,x] = 1 - (255 - g[ty+y, tx+x]) / 255
// (paint_image_main.py:162 `255 - geom[...]`, brush.py:679 `1 - patch/255.0`)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void geom_tiles_kernel(const uint8_t* __restrict__ g, int gh, int gw,
                                                         const int* __restrict__ tile_yx, int r,
                                                         float* __restrict__ out) {
    const int t = blockIdx.y;
    const int ty = tile_yx[2 * t], tx = tile_yx[2 * t + 1];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < r * r; i += gridDim.x * 256) {
        const int y = i / r, x = i - y * r;
        const int gy = ty + y, gx = tx + x;
        const int v = (gy >= 0 && gy < gh && gx >= 0 && gx < gw) ? g[(size_t)gy * gw + gx] : 255;
        out[(size_t)t * r * r + i] = 1.f - (float)(255 - v) / 255.0f;
    }
}

extern "C" int nb_geom_tiles_f32(const uint8_t* geom, int gh, int gw, const int32_t* tile_yx, int t, int r, float* out,
                                 void* stream) {
    NB_REQUIRE(geom && tile_yx && out, "geom_tiles: null pointer");
    NB_REQUIRE(gh >= 1 && gw >= 1 && t >= 1 && r >= 1, "geom_tiles: bad sizes");
    int gx = nb_cdiv(r * r, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(geom_tiles_kernel, dim3(gx, t), dim3(256), 0, (hipStream_t)stream, geom, gh, gw, tile_yx, r, out);
    NB_CHECK_LAUNCH("geom_tiles");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// feature-canvas replay
// ------------------------------------------------------------------------------------------------
struct ReplayParams {
    float* tiles;            // [T, C, hw, hw]  in: block output before blending; out: blended features
    const int* tile_yx;      // [T, 2] tile origin on the feature canvas
    const float* alpha0;     // [hw, hw] generate_dirty_area_alpha template
    float* canvas;           // [C, hc, wc]
    const uint8_t* mask_in;  // [hc, wc] which canvas pixels hold saved features before this launch
    uint8_t* mask_out;       // [hc, wc] ... and after it (a different buffer: blocks of other channel groups still read mask_in)
    const int* cell_off;     // [ncells + 1]
    const int* cell_tiles;   // tile indices per cell, ascending = the reference's paint order
    int c, hw, hc, wc, crop;
    int cell_x0, cell_y0, cells_per_row;       // launch covers cells [cell_y0 + blockIdx.y][cell_x0 + blockIdx.x]
};

template <int CG>
__global__ __launch_bounds__(256) void canvas_replay_kernel(ReplayParams p) {
    const int bx = blockIdx.x + p.cell_x0, by = blockIdx.y + p.cell_y0;
    const int cx = bx * NB_CELL_W + (threadIdx.x & (NB_CELL_W - 1));
    const int cy = by * NB_CELL_H + (threadIdx.x / NB_CELL_W);
    const int c0 = blockIdx.z * CG;
    const int cell = by * p.cells_per_row + bx;
    if (cx >= p.wc || cy >= p.hc) return;
    const size_t cplane = (size_t)p.hc * p.wc;
    const size_t cpix = (size_t)cy * p.wc + cx;
    const int k0 = p.cell_off[cell], k1 = p.cell_off[cell + 1];
    if (k0 == k1) {
        // no tile touches this cell (an interactive stroke replays ONE tile on a large canvas): the canvas stays as it
        // is, only the mask moves to the other buffer
        if (blockIdx.z == 0) p.mask_out[cpix] = p.mask_in[cpix];
        return;
    }
    float cv[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) cv[c] = (c0 + c < p.c) ? p.canvas[(size_t)(c0 + c) * cplane + cpix] : 0.f;
    bool m = p.mask_in[cpix] != 0;
    const size_t tplane = (size_t)p.hw * p.hw;
    for (int k = k0; k < k1; ++k) {
        const int t = p.cell_tiles[k];
        const int ly = cy - p.tile_yx[2 * t], lx = cx - p.tile_yx[2 * t + 1];
        if (ly < 0 || ly >= p.hw || lx < 0 || lx >= p.hw) continue;
        const float a0 = p.alpha0[ly * p.hw + lx];
        // brush.py:209, 214, 221-225: what this tile writes back to the canvas
        bool upd = (a0 > 0.99f) || (m && a0 > 0.f);
        if (ly < p.crop || ly >= p.hw - p.crop || lx < p.crop || lx >= p.hw - p.crop) upd = false;
        // brush.py:215-216: alpha = 1 where nothing is saved, then inverted; stitching.py:24-25 blend
        const float a = 1.f - (m ? a0 : 1.f);
        const float na = 1.f - a;
        float* tp = p.tiles + ((size_t)t * p.c + c0) * tplane + (size_t)ly * p.hw + lx;
        // (where nothing was painted before -- m false: a = 0, na = 1 -- the blend returns the tile's own value: nothing to write back,
        //  and nothing to read unless the tile updates the canvas here.  Most pixels of a fresh canvas: ~3 GB less traffic per 400 tiles)
        if (!m && !upd) continue;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            if (c0 + c < p.c) {
                const float f = a * cv[c] + na * tp[(size_t)c * tplane];
                if (m) tp[(size_t)c * tplane] = f;
                if (upd) cv[c] = f;
            }
        }
        m = m || upd;
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
        if (c0 + c < p.c) p.canvas[(size_t)(c0 + c) * cplane + cpix] = cv[c];
    if (blockIdx.z == 0) p.mask_out[cpix] = m ? 1 : 0;
}

static int nb_canvas_replay_impl(float* tiles, int t, int c, int hw, const int32_t* tile_yx, const float* alpha0, int crop,
                                 float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc, const int32_t* cell_off,
                                 const int32_t* cell_tiles, int cell_x0, int cell_y0, int cells_x, int cells_y, void* stream) {
    NB_REQUIRE(tiles && tile_yx && alpha0 && canvas && mask_in && mask_out && cell_off && cell_tiles,
               "canvas_replay: null pointer");
    NB_REQUIRE(mask_in != mask_out, "canvas_replay: mask_in and mask_out must be different buffers");
    NB_REQUIRE(t >= 1 && c >= 1 && hw >= 1 && hc >= 1 && wc >= 1 && crop >= 0 && 2 * crop <= hw, "canvas_replay: bad sizes");
    const int cpr = nb_cdiv(wc, NB_CELL_W), cpc = nb_cdiv(hc, NB_CELL_H);
    NB_REQUIRE(cell_x0 >= 0 && cell_y0 >= 0 && cells_x >= 1 && cells_y >= 1 && cell_x0 + cells_x <= cpr && cell_y0 + cells_y <= cpc,
               "canvas_replay: cell box outside the canvas");
    ReplayParams p{tiles, tile_yx, alpha0, canvas, mask_in, mask_out, cell_off, cell_tiles, c, hw, hc, wc, crop, cell_x0, cell_y0, cpr};
    constexpr int CG = 8;
    dim3 grid(cells_x, cells_y, nb_cdiv(c, CG));
    NB_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "canvas_replay: canvas too large");
    hipLaunchKernelGGL(canvas_replay_kernel<CG>, grid, dim3(256), 0, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("canvas_replay");
    return NB_OK;
}

extern "C" int nb_canvas_replay_f32(float* tiles, int t, int c, int hw, const int32_t* tile_yx, const float* alpha0,
                                    int crop, float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc,
                                    const int32_t* cell_off,
                                    const int32_t* cell_tiles, void* stream) {
    return nb_canvas_replay_impl(tiles, t, c, hw, tile_yx, alpha0, crop, canvas, mask_in, mask_out, hc, wc, cell_off, cell_tiles,
                                 0, 0, nb_cdiv(wc, NB_CELL_W), nb_cdiv(hc, NB_CELL_H), stream);
}

extern "C" int nb_canvas_replay_box_f32(float* tiles, int t, int c, int hw, const int32_t* tile_yx, const float* alpha0,
                                        int crop, float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc,
                                        const int32_t* cell_off, const int32_t* cell_tiles, int cell_x0, int cell_y0,
                                        int cells_x, int cells_y, void* stream) {
    return nb_canvas_replay_impl(tiles, t, c, hw, tile_yx, alpha0, crop, canvas, mask_in, mask_out, hc, wc, cell_off, cell_tiles,
                                 cell_x0, cell_y0, cells_x, cells_y, stream);
}

// ------------------------------------------------------------------------------------------------
// feature-canvas replay over tile PIECES (multi-GPU halo exchange): a rank replays its own tiles (full
// pieces) together with the strips of earlier foreign tiles that overlap them (received from the neighbouring
// ranks, compact buffers).  A piece is a rectangle of ONE tile; pieces of the same tile are disjoint, so a canvas
// pixel meets every tile at most once, and ascending piece order = paint order.  Same arithmetic per (pixel, tile)
// as canvas_replay_kernel above.
// ------------------------------------------------------------------------------------------------
struct ReplayPiecesParams {
    const NbTilePiece* pieces;
    const float* alpha0;     // [hw, hw]
    float* canvas;
    const uint8_t* mask_in;
    uint8_t* mask_out;
    const int* cell_off;
    const int* cell_pieces;
    int c, hw, hc, wc, crop;
    int cell_x0, cell_y0, cells_per_row;
};

template <int CG>
__global__ __launch_bounds__(256) void canvas_replay_pieces_kernel(ReplayPiecesParams p) {
    const int bx = blockIdx.x + p.cell_x0, by = blockIdx.y + p.cell_y0;
    const int cx = bx * NB_CELL_W + (threadIdx.x & (NB_CELL_W - 1));
    const int cy = by * NB_CELL_H + (threadIdx.x / NB_CELL_W);
    const int c0 = blockIdx.z * CG;
    const int cell = by * p.cells_per_row + bx;
    if (cx >= p.wc || cy >= p.hc) return;
    const size_t cplane = (size_t)p.hc * p.wc;
    const size_t cpix = (size_t)cy * p.wc + cx;
    const int k0 = p.cell_off[cell], k1 = p.cell_off[cell + 1];
    if (k0 == k1) {
        if (blockIdx.z == 0) p.mask_out[cpix] = p.mask_in[cpix];
        return;
    }
    float cv[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) cv[c] = (c0 + c < p.c) ? p.canvas[(size_t)(c0 + c) * cplane + cpix] : 0.f;
    bool m = p.mask_in[cpix] != 0;
    for (int k = k0; k < k1; ++k) {
        const NbTilePiece pc = p.pieces[p.cell_pieces[k]];
        const int py = cy - pc.cy, px = cx - pc.cx;
        if (py < 0 || py >= pc.h || px < 0 || px >= pc.w) continue;
        const int ly = py + pc.ly0, lx = px + pc.lx0;                  // position inside the piece's tile
        const float a0 = p.alpha0[ly * p.hw + lx];
        bool upd = (a0 > 0.99f) || (m && a0 > 0.f);
        if (ly < p.crop || ly >= p.hw - p.crop || lx < p.crop || lx >= p.hw - p.crop) upd = false;
        const float a = 1.f - (m ? a0 : 1.f);
        const float na = 1.f - a;
        float* tp = (float*)pc.data + (size_t)c0 * pc.cstride + (size_t)py * pc.rstride + px;
        if (!m && !upd) continue;                // (as in canvas_replay_kernel: the blend is the identity where nothing was painted before)
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            if (c0 + c < p.c) {
                const float f = a * cv[c] + na * tp[(size_t)c * pc.cstride];
                if (m) tp[(size_t)c * pc.cstride] = f;
                if (upd) cv[c] = f;
            }
        }
        m = m || upd;
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
        if (c0 + c < p.c) p.canvas[(size_t)(c0 + c) * cplane + cpix] = cv[c];
    if (blockIdx.z == 0) p.mask_out[cpix] = m ? 1 : 0;
}

extern "C" int nb_canvas_replay_pieces_f32(const NbTilePiece* pieces, int n, int c, int hw, const float* alpha0, int crop,
                                           float* canvas, const uint8_t* mask_in, uint8_t* mask_out, int hc, int wc,
                                           const int32_t* cell_off, const int32_t* cell_pieces, int cell_x0, int cell_y0,
                                           int cells_x, int cells_y, void* stream) {
    NB_REQUIRE(pieces && alpha0 && canvas && mask_in && mask_out && cell_off && cell_pieces, "canvas_replay_pieces: null pointer");
    NB_REQUIRE(mask_in != mask_out, "canvas_replay_pieces: mask_in and mask_out must be different buffers");
    NB_REQUIRE(n >= 1 && c >= 1 && hw >= 1 && hc >= 1 && wc >= 1 && crop >= 0 && 2 * crop <= hw, "canvas_replay_pieces: bad sizes");
    const int cpr = nb_cdiv(wc, NB_CELL_W), cpc = nb_cdiv(hc, NB_CELL_H);
    NB_REQUIRE(cell_x0 >= 0 && cell_y0 >= 0 && cells_x >= 1 && cells_y >= 1 && cell_x0 + cells_x <= cpr && cell_y0 + cells_y <= cpc,
               "canvas_replay_pieces: cell box outside the canvas");
    ReplayPiecesParams p{pieces, alpha0, canvas, mask_in, mask_out, cell_off, cell_pieces, c, hw, hc, wc, crop, cell_x0, cell_y0, cpr};
    constexpr int CG = 8;
    dim3 grid(cells_x, cells_y, nb_cdiv(c, CG));
    NB_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "canvas_replay_pieces: canvas too large");
    hipLaunchKernelGGL(canvas_replay_pieces_kernel<CG>, grid, dim3(256), 0, (hipStream_t)stream, p);
    NB_CHECK_LAUNCH("canvas_replay_pieces");
    return NB_OK;
}

// ------------------------------------------------------------------------------------------------
// RGBA tile paste (paint_image_main.py:173-177 with the server-side crop of brush.py:369-374): the interior
// [crop, r-crop)^2 of tile t lands at dst_yx[t] + crop; where interiors overlap the LAST tile wins.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void paste_tiles_kernel(const uint32_t* __restrict__ tiles, int r,
                                                          const int* __restrict__ dst_yx, int crop,
                                                          uint32_t* __restrict__ canvas, int h, int w,
                                                          const int* __restrict__ cell_off,
                                                          const int* __restrict__ cell_tiles) {
    const int cx = blockIdx.x * NB_CELL_W + (threadIdx.x & (NB_CELL_W - 1));
    const int cy = blockIdx.y * NB_CELL_H + (threadIdx.x / NB_CELL_W);
    const int cell = blockIdx.y * gridDim.x + blockIdx.x;
    if (cx >= w || cy >= h) return;
    for (int k = cell_off[cell + 1] - 1; k >= cell_off[cell]; --k) {
        const int t = cell_tiles[k];
        const int ly = cy - dst_yx[2 * t], lx = cx - dst_yx[2 * t + 1];
        if (ly >= crop && ly < r - crop && lx >= crop && lx < r - crop) {
            canvas[(size_t)cy * w + cx] = tiles[((size_t)t * r + ly) * r + lx];
            return;
        }
    }
}

extern "C" int nb_paste_tiles_u8(const uint8_t* tiles, int t, int r, const int32_t* dst_yx, int crop, uint8_t* canvas,
                                 int h, int w, const int32_t* cell_off, const int32_t* cell_tiles, void* stream) {
    NB_REQUIRE(tiles && dst_yx && canvas && cell_off && cell_tiles, "paste_tiles: null pointer");
    NB_REQUIRE(t >= 1 && r >= 1 && h >= 1 && w >= 1 && crop >= 0 && 2 * crop < r, "paste_tiles: bad sizes");
    dim3 grid(nb_cdiv(w, NB_CELL_W), nb_cdiv(h, NB_CELL_H));
    NB_REQUIRE(grid.y <= 65535, "paste_tiles: canvas too large");
    hipLaunchKernelGGL(paste_tiles_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t*)tiles, r, dst_yx,
                       crop, (uint32_t*)canvas, h, w, cell_off, cell_tiles);
    NB_CHECK_LAUNCH("paste_tiles");
    return NB_OK;
}
