// Direct kernels for layers with K = taps*Cin <= DIRECT_MAX_K (csrc/direct_small.hip); dispatched from dfl_conv2d and
// dfl_conv2d_wgrad.
#pragma once
#include "common.h"

namespace dfl {

constexpr int DIRECT_MAX_K = 12;
constexpr int CFG_DIRECT = 5;   // value dfl_conv_config / dfl_wgrad_config report for these kernels

bool direct_conv_ok(const dfl_conv_args* a);
bool direct_conv_rows_usable(const dfl_conv_args* a);   // the 1-channel 3x3 row form will run
int direct_conv_blocks(const dfl_conv_args* a);   // workgroups = rows of stat_partials
int direct_conv_launch(const dfl_conv_args* a, hipStream_t s);

bool direct_wgrad_ok(const dfl_wgrad_args* a);
int direct_wgrad_splits(const dfl_wgrad_args* a);
int direct_wgrad_launch(const dfl_wgrad_args* a, hipStream_t s);

}  // namespace dfl
