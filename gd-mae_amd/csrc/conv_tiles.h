// Tile-compact storage of the decoder's conv_out map (csrc/conv_tiles.hip).
//
// The 3x3 conv output Y of the generative decoder is only evaluated on the 8x8-site tiles whose one-site halo touches
// an active site of some source stage ("active tiles"); every other site holds one of 9 per-channel constants that
// depend only on which taps of the site fall inside the map (its border class).  Y is therefore stored as
//   Yc   (n_act * 64, C)  rows of the active tiles, tile-major, site (ty, tx) of a tile at row ty * 8 + tx
//   slot (B * TH * TW)    tile -> index into Yc / -1,  TH = ceil(H / 8), TW = ceil(W / 8)
//   ybg  (9, C)           the class constants, class = 3 * cy + cx, c = 0 first row/column, 2 last, 1 otherwise
// A null `slot` means "Y is the plain dense (B*H*W, C) map" (fp32 parity mode keeps the dense dataflow).
#pragma once
#include "common.h"

#define GD_TILE 8
#define GD_TILE_SITES 64

struct GdTiles {
  const int* slot;
  const void* ybg;
  int TH, TW;
};

__device__ inline int gd_border_class(int y, int x, int H, int W) {
  const int cy = y == 0 ? 0 : (y == H - 1 ? 2 : 1);
  const int cx = x == 0 ? 0 : (x == W - 1 ? 2 : 1);
  return cy * 3 + cx;
}

// element offset (in units of C-element rows) is resolved to a row pointer; ES = element size in bytes
template <int ES>
__device__ inline const char* gd_y_row(const void* Y, const GdTiles& T, int b, int y, int x, int H, int W, int C) {
  if (T.slot == nullptr) return (const char*)Y + ((long long)(b * H + y) * W + x) * C * ES;
  const int s = T.slot[(b * T.TH + (y >> 3)) * T.TW + (x >> 3)];
  if (s < 0) return (const char*)T.ybg + (long long)gd_border_class(y, x, H, W) * C * ES;
  return (const char*)Y + ((long long)s * GD_TILE_SITES + (y & 7) * GD_TILE + (x & 7)) * C * ES;
}
