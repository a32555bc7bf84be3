// Patch-resident bf16 convolution and weight-gradient kernels (convp_bf16.hip, wgradp_bf16.hip), dispatched by
// dfl_conv2d / dfl_conv2d_wgrad when the argument block says its tensors are bf16.
#pragma once
#include "common.h"

namespace dfl {

struct ConvP {
  dfl_conv_args a;
  int Mtot, Cout;               // GEMM rows (gather-grid pixels), channels per output pixel
  int Hg, Wg;                   // gather grid per image (= Hout x Wout; Hin x Win for the 2x2-scatter form)
  int PH, PW, IPP;              // patch: PH x PW grid pixels of IPP images (IPP > 1 only for whole images)
  int npy, npx, npatch;         // patches per image along y / x, patches in all
  int IH, IW;                   // input pixels a patch needs per image, halo included
  int CK, nblk, blk_per_slice;  // input channels resident in LDS per block, blocks over Cin, blocks per K slice
  int splits, T;                // K slices (grid.z), taps
  int pix_stride, upp_shift;    // bytes per staged pixel (2 CK + 16), log2(CK / 8)
  int lds_bytes, tile;          // staged image size, index into the tile configuration table
  int ntiles, grid, xcd_mode;   // column tiles; workgroups launched (linear grid); workgroup -> (patch, tile, slice) map
  uint32_t x_bytes, w_bytes;
  uint32_t x2_bytes, xo_bytes;  // extent of the second input tensor (dfl_conv_args.x_mode) and of x_out
  uint32_t mPP, mPW;            // ceil(2^32 / (PH * PW)), ceil(2^32 / PW): divisions of patch row indices by multiply-high
  int tab_off, pad1;            // LDS offset of the live-BatchNorm tables (set at launch)
  // latency form (convs_bf16.hip; tile == CONVS_TILE): 32 x 32 tiles over (pixels, columns), waves of a workgroup per tile (log2),
  // 16-channel chunks per tap (log2), k-steps of the layer and per wave
  int s_mt, s_nt, s_ksplit_shift, s_cpk_shift, s_ksteps, s_kper;
  // unrolled 3x3 form (convq_bf16.hip; tile >= CONVQ_TILE): ceil(2^32 / d) for d = npatch, ntiles, patches per image, npx
  // (narrow form, convn_bf16.hip, tile >= CONVN_TILE: qm_perimg, qm_npx; q_ngroups = patches per XCD)
  uint32_t qm_npatch, qm_ntiles, qm_perimg, qm_npx;
  int q_ngroups, q_stride;      // persistent form: workgroups per (column tile, K slice); qm_npatch is then the magic of q_ngroups
                                // (narrow form: patches per XCD; its persistent form: workgroups per XCD = the stride of a workgroup's patches)
};
constexpr int CONVS_TILE = 39;  // value of ConvP.tile for the latency form (= number of convp tile configurations)
constexpr int CONVQ_TILE = 40;  // ... and for the unrolled 3x3 form of convq_bf16.hip: 40 ... 48 = its nine wave layouts (kQ there), 49 ... 57 = the same, persistent
constexpr int CONVQ_LAYOUTS = 9;
constexpr int CONVN_TILE = 58;  // ... and for the narrow 3x3 form of convn_bf16.hip (32 / 64 output columns): 58 ... 63 = its six layouts (kN there)
constexpr int CONVN_LAYOUTS = 6;
constexpr int CONVN_PERS_TILE = 64;  // 64, 65: the persistent form of its layouts 4 and 5 (three rows per wave), 32 -> 32 layers

// Chooses the geometry for these arguments.  force_splits: 0 = free choice, else the K-slice count to plan for.
int convp_plan(const dfl_conv_args* a, ConvP* p, int force_splits);
int convp_launch(const ConvP& p, hipStream_t s);
int convp_finish_rows(const ConvP& p);
int convp_candidates(const dfl_conv_args* a, int32_t* out, int max);
int convp_force(const int32_t* g);
int convp_tune_add(const int32_t* key, const int32_t* g);

// Unrolled 3x3 form for the deep levels (convq_bf16.hip): 8 x 12 patches, 128 columns, 64 / 128 resident channels
bool convq_shape_ok(const dfl_conv_args& a);
int convq_launch(const ConvP& p, int mode, int pers, hipStream_t s);     // pers: the persistent form (a workgroup walks q_ngroups-strided patches)
size_t convq_lds_bytes(int ck, int mode, int blk_per_slice, int pers);      // LDS a workgroup asks for
int convq_threads(int mode);
bool convq_pers_ok(int mode, int ck, int x_mode);                 // is the persistent form built for this layout / operand?
bool convq_ck_ok(int mode, int ck);                               // is configuration `mode` instantiated for ck resident channels?

// Narrow 3x3 form for the shallow levels (convn_bf16.hip): 32 / 64 output columns, epilogue on the accumulator registers
bool convn_shape_ok(const dfl_conv_args& a);
bool convn_layout_ok(int layout, int ntot, int cin);              // is the layout built for this column count / one or several channel blocks?
void convn_patch(int layout, int* ph, int* pw);                   // the layout's patch (rows, pixels per row)
size_t convn_lds_bytes(int layout, int cin, int ntot, int pers);
bool convn_pers_ok(int layout, const dfl_conv_args& a);            // is the persistent form built for this layout / layer?
int convn_launch(const ConvP& p, int layout, int pers, hipStream_t s);

// Latency form for the small problems of a batch-1 inference forward (convs_bf16.hip)
bool convs_eligible(const dfl_conv_args& a, const ConvP& p);
void convs_plan(const dfl_conv_args& a, ConvP* p, int force_splits);
int convs_launch(const ConvP& p, hipStream_t s);
bool convs_first_ok(const dfl_conv_args* a);                              // the 1-channel 3x3 first layer
int convs_first_launch(const dfl_conv_args* a, hipStream_t s);
int convs_pair_ok(const dfl_conv_args* a, const dfl_conv_args* b);     // dfl_conv_pair_ok
int convs_pair_launch(const dfl_conv_args* a, const dfl_conv_args* b, hipStream_t s);

// ... and for fp32 tensors, math modes 0 (fp32 matrix instructions) and 1 (bf16x3): convs_f32.hip
bool convs32_eligible(const dfl_conv_args* a);
int convs32_suggest_splits(const dfl_conv_args* a);                      // 0: not this form
int convs32_launch(const dfl_conv_args* a, hipStream_t s, int* splits_out);
int convs32_pair_ok(const dfl_conv_args* a, const dfl_conv_args* b);
int convs32_pair_launch(const dfl_conv_args* a, const dfl_conv_args* b, hipStream_t s);

struct WgP;
int wgradp_suggest_splits(const dfl_wgrad_args* a);
int wgradp_launch(const dfl_wgrad_args* a, hipStream_t s);
int wgradp_config(const dfl_wgrad_args* a);

}  // namespace dfl
