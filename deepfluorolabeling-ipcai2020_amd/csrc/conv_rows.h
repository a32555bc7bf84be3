// Row-tiled 3x3 convolution kernels (conv_rows.hip), dispatched by dfl_conv2d.
#pragma once
#include "conv_epilogue.h"

namespace dfl {
constexpr int CFG_ROWS192 = 6, CFG_ROWS96 = 7;   // dfl_conv_config values (192 x 32 and 96 x 64 tiles)
int conv_rows_tile(const ConvK& k);              // 0 = not eligible, else the tile's pixel count (192 / 96)
int conv_rows_launch(const ConvK& k, hipStream_t s);
int conv_rows_splits(const ConvK& k);            // K slices these kernels want for the layer (0 = not their layer)
int conv_rows_set_min_tiles(int n);              // returns the previous threshold
}  // namespace dfl
