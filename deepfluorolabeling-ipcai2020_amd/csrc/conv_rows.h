// Row-tiled 3x3 convolution kernels (conv_rows.hip), dispatched by dfl_conv2d.
#pragma once
#include "conv_epilogue.h"

namespace dfl {
// tiles (pixels x output channels); the values are what dfl_conv_config reports
constexpr int ROWS_192x32 = 6, ROWS_96x64 = 7, ROWS_96x32 = 8;
constexpr int conv_rows_bm(int tile) { return tile == ROWS_192x32 ? 192 : 96; }
int conv_rows_tile(const ConvK& k);              // 0 = not eligible, else one of the ROWS_* tiles
int conv_rows_launch(const ConvK& k, hipStream_t s);
int conv_rows_splits(const ConvK& k);            // K slices these kernels want for the layer (0 = not their layer)
int conv_rows_set_min_tiles(int n);              // returns the previous threshold
}  // namespace dfl
