/* liboai_hip.so -- C ABI of the MI355X (gfx950) hot path of OAI_analysis_2.
 *
 * The reference (uncbiag/OAI_analysis_2 @ 2024_10_08) has no FFI of its own: its seam is
 * Python duck typing.  The entry points below are what a binding for that seam
 * calls; each one names the reference interface it stands in for.  INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; oai_last_error() gives the text
 *     (thread-local).  Nothing aborts, nothing throws across the boundary.
 *   - every pointer named *_dev is device memory owned by the caller (e.g. a PyTorch-ROCm
 *     tensor's data_ptr()); host pointers are named *_host.  No torch types cross the boundary.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Launch functions
 *     enqueue work and return; they never synchronise, allocate or free (graph-capturable).
 *   - volumes are [z][y][x] (numpy / torch order), fp32 unless stated.  Vector fields and
 *     coordinate maps are channel-first [3][D][H][W], channel c along tensor axis c (z,y,x),
 *     in ICON's normalised [0,1] units (index / (n-1)).
 *   - a handle is not thread-safe; distinct handles are.
 */
#ifndef OAI_HIP_H
#define OAI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OAI_OK 0
#define OAI_ERR_ARG 1
#define OAI_ERR_HIP 2
#define OAI_ERR_WORKSPACE 3

int oai_version(void);
const char* oai_last_error(void);
/* Fills name (<= cap bytes) with the device's gcnArchName; returns the CU count or -1. */
int oai_device_info(char* name, int cap);

/* ------------------------------------------------------------------------------------------
 * Registration warp family.  Replaces torch.nn.functional.grid_sample / avg_pool3d /
 * interpolate as reached from oai_analysis/registration.py:25 (icon_registration.itk_wrapper.
 * register_pair -> network_wrappers / mermaidlite.compute_warped_image_multiNC), and the ITK
 * resample of test/test_all.py:42-52 / oai_analysis/dask_processing.py:95-111.
 * ---------------------------------------------------------------------------------------- */

/* out[c][D][H][W] = trilinear sample of src[c][d][h][w] at coords (ICON [0,1] units; border
 * clamp, align_corners=True).  coords_dev == NULL means the identity map of [D][H][W].
 * = mermaidlite.compute_warped_image_multiNC(src, coords, spacing, 1). */
int oai_grid_sample3d(const float* src_dev, int C, int d, int h, int w,
                      const float* coords_dev, int D, int H, int W, float* out_dev, void* stream);

/* Process-wide tuning options of the warp kernels (bit-preserving; the library does not read the environment):
 *   "brick" 0|1 (0)  oai_grid_sample3d / oai_compose (C = 1 or 3) through sample_brick_kernel: a block owns a 16 x 8 x 4 output brick and, when the
 *                    bounding box of the brick's corners fits 24 KB, stages that box of the source in LDS by LDS-DMA and takes the corners
 *                    from there (the gather form is bound by the CU's L1 tag pipeline: profiles/r02_registration.md); bricks whose box does
 *                    not fit gather from memory as before.  Same arithmetic in the same order: bit-identical outputs.  Replaces the same
 *                    reference op, mermaidlite.compute_warped_image_multiNC as reached from oai_analysis/registration.py:25. */
int oai_warp_set_option(const char* name, int value);

/* out = coords + sample(disp, coords): FunctionFromVectorField's transform(coords).
 * coords_dev == NULL means the identity map (the "sampled" path at another resolution);
 * if additionally (d,h,w)==(D,H,W) and shortcut != 0, out = identity + disp exactly
 * (the package's isIdentity shortcut, no interpolation). */
int oai_compose(const float* disp_dev, int d, int h, int w, const float* coords_dev,
                int D, int H, int W, int shortcut, float* out_dev, void* stream);

/* F.avg_pool3d(x, 2, ceil_mode=True) on [C][D][H][W] -> [C][ceil(D/2)][ceil(H/2)][ceil(W/2)]. */
int oai_avgpool2_3d(const float* in_dev, int C, int D, int H, int W, float* out_dev, void* stream);

/* F.interpolate(x, size=(D,H,W), mode="trilinear", align_corners=False) on [C][d][h][w]. */
int oai_resize_trilinear(const float* in_dev, int C, int d, int h, int w,
                         float* out_dev, int D, int H, int W, void* stream);

/* itk_wrapper.create_itk_transform's vector image: disp[z][y][x][3] (float64, components x,y,z,
 * network-voxel units) = reverse_components((phi - identity) * (shape - 1)). */
int oai_phi_to_itk_displacement(const float* phi_dev, int D, int H, int W, double* disp_dev, void* stream);

/* A whole compose chain of icon_registration's TwoStepRegistration / DownsampleRegistration closures per output voxel, without
 * materialising the intermediate maps (SURVEY.md K15 "fuse chains", K18):
 *     c = identity(D,H,W) [+ start_dev]        start_dev [3][D][H][W] may be NULL (the package's isIdentity shortcut when given)
 *     c = c + sample(fields[i], c)             i = 0 .. n_fields-1 (n_fields <= OAI_WARP_CHAIN_MAX_FIELDS), fields[i] is
 *                                              [3][fd][fh][fw] with (fd,fh,fw) = field_dims_zyx[3i..3i+2] -- any resolution; a step
 *                                              tree of N FunctionFromVectorFields flattens to N links (oai_icon_create)
 *     out = image_dev ? sample(image_dev [id][ih][iw], c)  ->  out_dev [D][H][W]
 *                     : c                                   ->  out_dev [3][D][H][W]
 * `sample` is oai_grid_sample3d's (grid_sample bilinear / border / align_corners=True on [0,1] coordinates).  Bit-identical to
 * the sequence of oai_compose / oai_grid_sample3d calls it replaces.  `fields` and `field_dims_zyx` are HOST arrays. */
#define OAI_WARP_CHAIN_MAX_FIELDS 8
int oai_warp_chain(const float* start_dev, int D, int H, int W, int n_fields, const float* const* fields,
                   const int* field_dims_zyx, const float* image_dev, int id, int ih, int iw, float* out_dev, void* stream);

/* Geometry of one image for the resample: index_xyz -> physical = A*idx + b (row-major 3x3 + 3). */
typedef struct oai_affine { double A[9]; double b[3]; } oai_affine;

/* warped[zB][yB][xB] = prob_A(phi_AB(p)): ITK ResampleImageFilter(prob, transform=phi_AB,
 * LinearInterpolateImageFunction, reference grid = image_B, default pixel 0), with phi_AB the
 * CompositeTransform [to_network_space, DisplacementFieldTransform, from_network_space].
 *   b_index_to_net : B index  -> network index space   (T_B^-1 o index_to_physical_B)
 *   net_to_a_index : network index space -> A continuous index (physical_to_index_A o T_A)
 *   disp_dev       : float64 [Dn][Hn][Wn][3] from oai_phi_to_itk_displacement. */
int oai_resample_through_disp(const float* prob_dev, int nzA, int nyA, int nxA,
                              const double* disp_dev, int Dn, int Hn, int Wn,
                              const oai_affine* b_index_to_net, const oai_affine* net_to_a_index,
                              float* out_dev, int nzB, int nyB, int nxB, void* stream);

/* The same resample for n_maps (1..4) probability maps probs_dev[n_maps][nzA][nyA][nxA] at once, straight from the network's
 * dense map phi_dev[3][Dn][Hn][Wn] (fp32, [0,1] units): create_itk_transform's displacement (fp32 arithmetic, widened to
 * double -- the value oai_phi_to_itk_displacement stores) is rebuilt at the 8 corners in registers, so neither the 71 MB
 * fp64 field nor a second pass over phi exists.  Results are bit-identical to oai_phi_to_itk_displacement +
 * oai_resample_through_disp per map.  Replaces the two deform_probmap calls of test/test_all.py:54-58 /
 * dask_processing.py:95-111.  out_dev[n_maps][nzB][nyB][nxB].  nxA >= 2 (the two x corners of a row are one 8-byte gather). */
int oai_resample_maps_through_phi(const float* probs_dev, int n_maps, int nzA, int nyA, int nxA,
                                  const float* phi_dev, int Dn, int Hn, int Wn,
                                  const oai_affine* b_index_to_net, const oai_affine* net_to_a_index,
                                  float* out_dev, int nzB, int nyB, int nxB, void* stream);

/* ------------------------------------------------------------------------------------------
 * Intensity windowing, the step just before the hot path: image_normalize(image, lo, hi, omin, omax) of
 * oai_analysis/dask_processing.py:10-26 (called with 0.1, 99.9, 0, 1 at :75 and :177) =
 * np.percentile window (exact order statistics, numpy float32 interpolation) + itk.IntensityWindowingImageFilter.
 * window_out_dev (optional, 2 floats on the device) receives (window_min, window_max).
 * ---------------------------------------------------------------------------------------- */
size_t oai_image_normalize_workspace_bytes(void);
int oai_image_normalize(const float* in_dev, size_t n, float pct_lo, float pct_hi, float out_min, float out_max,
                        float* out_dev, float* window_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Segmentation.  Replaces `self.model(temp_input.to(self.device)).cpu()` and the code around
 * it in Segmenter3DInPatchClassWise.segment (oai_analysis/segmentation/segmenter.py:100-131):
 * Partition.__call__ (image_transforms.py:395-455), UNet.forward (networks.py:109-149),
 * torch.sigmoid / >0.5 (segmenter.py:121-124) and Partition.assemble (image_transforms.py:457-519).
 * ---------------------------------------------------------------------------------------- */

typedef struct oai_unet oai_unet;

/* One conv block in the reference's own tensor layouts (host memory, fp32):
 *   kind 0: Conv3d k3 s1 p1           weight [Cout][Cin][3][3][3]
 *   kind 1: ConvTranspose3d k3 s1 p1  weight [Cin][Cout][3][3][3]
 *   kind 2: ConvTranspose3d k2 s2     weight [Cin][Cout][2][2][2]
 *   kind 3: Conv3d k1                 weight [Cout][Cin]
 * bias / bn_* may be NULL (bias=False / BN=False, networks.py:39). */
typedef struct oai_layer_params {
    int kind, cin, cout;
    const float* weight_host;
    const float* bias_host;
    const float* bn_gamma_host;
    const float* bn_beta_host;
    const float* bn_mean_host;
    const float* bn_var_host;
} oai_layer_params;

#define OAI_UNET_NUM_LAYERS 18 /* ec0..ec7, dc9, dc8, dc7, dc6, dc5, dc4, dc3, dc2, dc1, dc0 */

/* Ingests a reference state_dict (strict, like utils.initialize_model, utils.py:29): packs
 * the weights for the kernels (transposed-conv flip, K-permuted MFMA panels) and uploads them. */
int oai_unet_create(const oai_layer_params layers_host[OAI_UNET_NUM_LAYERS], float bn_eps, oai_unet** out);
void oai_unet_destroy(oai_unet* h);

/* Arithmetic of the 3x3x3 conv layers and the k2s2 up-convs (ec0, the head and everything outside the U-Net are always fp32).  A handle
 * starts in OAI_PREC_F32; the Python surface (Segmenter3DInPatchClassWise, bench.py) selects OAI_PREC_FP16X3 by default and falls back to
 * OAI_PREC_F32 per volume when the range flag is raised:
 *   OAI_PREC_F32     v_mfma_f32_32x32x2_f32: exact fp32 products; the accumulation is TWO-LEVEL (round 5): one fresh accumulator set per
 *                    8-channel chunk (108 MFMAs), folded into the running sum by one fp32 add -- 0.8 x the reference's own distance from its
 *                    float64 run on four reference networks (one running sum over K = 27 Cin: 2.9-3.5 x).  Activations fp32 in memory.
 *   OAI_PREC_BF16X6  every fp32 operand split into 3 bf16 terms, 6 bf16 MFMA passes per product: fp32-grade
 *                    results (dropped terms are O(2^-24)) at 2.7x the fp32 MFMA rate.  Activations fp32 in memory, split while staging.
 *   OAI_PREC_BF16X3  2 terms, 3 passes: ~2^-17 relative error per product, 5.3x the fp32 MFMA rate.  Activations fp32 in memory.
 *   OAI_PREC_FP16X3  (below) activations live in memory AS their two fp16 terms ("format S", 4 bytes per element like fp32: unet_sres.h),
 *                    scaled per layer by a power of two (oai_unet_calibrate_step); fp32 in and out of every entry point.
 * Collectives: there is no oai_comm_* / oai_zslab_* entry point (SURVEY.md 8b lists them as candidates) BY DESIGN -- the multi-GPU
 * exchange steps (volume broadcast, all_gather of kept-centre blocks, z-slab gather, the 76-byte range state) are torch.distributed
 * calls over RCCL on device tensors in oai_analysis_2_amd/parallel.py: host orchestration stays in Python (BASELINE.json north_star),
 * and this library only provides what those steps need on either side (oai_segment_tiles' tile ranges, oai_stitch_blocks_ranged,
 * oai_resample_maps_through_phi's z ranges, oai_unet_range_state_snapshot). */
#define OAI_PREC_F32 0
#define OAI_PREC_BF16X3 1
#define OAI_PREC_BF16X6 2
#define OAI_PREC_FP16X3 3 /* 2 fp16 terms (22 mantissa bits), 3 fp16 MFMA passes: fp32-grade results at the BF16X3 rate; needs
                             activations below 65504 in magnitude (weights are range-scaled per output channel, exactly) */
int oai_unet_set_precision(oai_unet* h, int mode);
/* OAI_PREC_FP16X3 only: *out = the range flag of the work queued since the last reset; non-zero means the results of that
 * run are invalid and it must be repeated with OAI_PREC_F32 (what Segmenter3DInPatchClassWise / VolumePipeline do):
 *   bit 0  an activation was beyond fp16's range (|x| * 2^e > 65504);
 *   bit 1  some layer's largest stored activation was below 8.0 (and not zero): its low fp16 terms are subnormal, the
 *          arithmetic is no longer fp32 grade.  Either the handle was never calibrated or this input is > 128 x quieter
 *          than the calibration input (oai_unet_calibrate_step).
 * A network replaces nothing in the reference here: segmenter.py:52-62 loads an arbitrary checkpoint whose per-layer
 * activation scale is free, and networks.py:109-149 runs it in fp32, which has no such window.
 * The read (and the reset) are ordered on `stream` -- pass the stream the segment calls were queued on -- and the call
 * blocks until that stream has drained. */
int oai_unet_range_flag(oai_unet* h, int reset, int* out, void* stream);
/* Asynchronous form for pipelined callers (cohort.py): evaluates the flag word into dst_dev[0] and clears flag and census,
 * queued on `stream` behind the segment calls of ONE volume, so the flag is attributed to that volume; the caller reads
 * dst_dev with its own D2H copy of the results.  No synchronisation. */
int oai_unet_range_flag_snapshot(oai_unet* h, int* dst_dev, void* stream);
/* The same snapshot as RAW STATE for a volume whose tiles were computed by several ranks (pipeline.run_sharded): state_dev[0] = the
 * overflow bit, state_dev[1 + k] = the bits of layer k's largest stored activation (a non-negative float: ordered like its integer
 * bits), so that an elementwise MAX all-reduce over the ranks gives the state of the WHOLE volume -- a rank that holds only quiet
 * background tiles must not raise the LOW bit on its own subset.  Clears flag and census like the snapshot.
 * oai_unet_range_flag_from_state evaluates a (reduced) state into the two-bit flag word, flag_dev[0]. */
#define OAI_UNET_RANGE_STATE_WORDS (1 + OAI_UNET_NUM_LAYERS)
int oai_unet_range_state_snapshot(oai_unet* h, int* state_dev, void* stream);
int oai_unet_range_flag_from_state(const int* state_dev, int* flag_dev, void* stream);
/* Per-layer activation exponents of OAI_PREC_FP16X3.  Layer k (order of OAI_UNET_NUM_LAYERS) stores its output as
 * x * 2^e[k] in fp16 term pairs; powers of two fold exactly into the epilogue affine of the producer, the epilogue scale of
 * the consumer and -- for the skip inputs of dc8 / dc5 / dc2 -- the weight panel, so results change only where fp16's
 * exponent range had been costing bits.  e[17] (dc0: logits) is always 0.
 *   oai_unet_census          max |stored activation| per layer since the last reset (0 = nothing stored); blocks on `stream`.
 *   oai_unet_calibrate_step  call after one oai_segment_tiles / oai_unet_forward_tiles pass over representative input:
 *                            moves every layer whose maximum is outside [2^9, 2^12) to [2^10, 2^11), resets census and flag,
 *                            *more = 1 if anything moved (run another pass and call again; a layer downstream of an overflow
 *                            is only right after its inputs are), 0 = calibrated.  Two passes for a healthy network.
 *   oai_unet_set_act_exponents / get  explicit form (reproducible runs, all ranks of a tile-sharded volume).  Set waits for
 *                            the device (hipDeviceSynchronize) and rewrites the handle's epilogue arrays in place. */
int oai_unet_census(oai_unet* h, float max_out[OAI_UNET_NUM_LAYERS], int reset, void* stream);
int oai_unet_calibrate_step(oai_unet* h, void* stream, int* more);
int oai_unet_get_act_exponents(const oai_unet* h, int e_out[OAI_UNET_NUM_LAYERS], int* calibrated);
int oai_unet_set_act_exponents(oai_unet* h, const int e[OAI_UNET_NUM_LAYERS]);
/* Tuning options of the OAI_PREC_FP16X3 path (bit-preserving -- same k order, identical maps -- except "winograd" / "winograd_layers" / "m16" / "m16_layers"), by name
 * (the library does not read the environment):
 *   "sres" 0|1 (1)      activations resident as fp16 term pairs (unet_sres.h) / fp32-resident split kernels
 *   "sres_mrep" 2|4 (4) z slices per workgroup of the split-resident conv kernel
 *   "sres_ring" 0|1 (0) six-slot z-plane ring staging (implies sres_mrep 2)
 *   "xcd_group" n (32)  logical blocks dealt to one XCD at a time; 0 = plain launch order
 *   "fuse_first" 0|1 (1) ec0 (networks.py:43) computed inside ec1's halo staging instead of as its own launch (when ec1 is one main-shape launch)
 *   "b_lds" 0|1 (0)     conv weight fragments through a three-slot LDS ring shared by the four waves of a workgroup
 *   "wide" 0|1|2 (1)    layers with Cout % 128 == 0 run conv3_igemm_sres2: one 8-wave workgroup per CU computes 128 couts of a
 *                       block from ONE double-buffered halo box (unet_sres2.h); 1 = launches of >= 1024 workgroups, 2 = always
 *   "shared_enc" 0|1 (1) oai_segment_tiles computes ec0 -> ec1 once over the padded volume + a 2-voxel shell per tile (needs the workspace
 *                       of oai_segment_workspace_bytes; geometries it does not fit fall back to per-tile computation)
 *   "winograd" 0..63 (19) NOT bit-preserving (same precision class, other rounding points; probabilities within ~2e-6 of the direct
 *                       form's): the plain k3 layers (no fused ec0 / pool / head) run conv3_wino_sres (unet_wino.h), the x axis in Winograd
 *                       F(2,3) form = 2/3 of the MFMAs.  bit 0: layers with Cout % 128 == 0, bit 1: layers with one block of 64 couts
 *                       and >= 8 input chunks (dc2); 0 = the direct kernels everywhere; bits 2, 3: A/B of alternative kernel forms; bit 4 (round 4): the
 *                       two-group form's taps on v_mfma_f32_16x16x32_f16 with K = a pair of taps (same cycles per FLOP, the shape the chip clocks
 *                       ~13 % higher at the power wall; another summation order, same gates); bit 5: the same for the 64-cout layer, all its launch shapes
 *                       (measured within noise of bit 4 alone: not in the default).  A value depends on the parity of its voxel's x only -- not on blocks, strips
 *                       or batching.  "winograd_layers" (mask, all): bit k = layer k may take it (A/B of single layers)
 *   "m16" 0|1 (1)       NOT bit-preserving (round 5): the direct kernel conv3_igemm_sres -- ec1 with the fused ec0, ec2, dc1 with the fused head, and any
 *                       layer "winograd" leaves to it -- runs its taps on v_mfma_f32_16x16x32_f16 with K = a PAIR of taps (27 = 13 pairs + 1) for every
 *                       layer with Cout % 128 != 0 (a layer that may take the bit-identical 128-cout form conv3_igemm_sres2 keeps 32x32x16, so "wide"
 *                       stays bit-preserving); every launch shape of the kernel has the variant: one summation order per layer.  Same cycles per FLOP,
 *                       +12-14 % clock at the power wall: ec1 16.8 -> 14.6, ec2 4.9 -> 3.9, dc1 8.5 -> 7.2 ms per 160 tiles.  "m16_layers" (mask, all):
 *                       bit k = layer k may take it (A/B of single layers)
 *   "persistent" 0|1 (0) bit-preserving (round 5): the 64-cout Winograd layer (dc2) with ONE persistent workgroup per CU that pulls blocks from per-XCD counters
 *                       (a small plan kernel in front of every launch) while its staging waves run one block ahead.  Built for VERDICT r4 #1 (b) / (d); measured
 *                       +-0 ... +0.8 % per pass (profiles/r05_persistent.md): not the default
 *   "winograd_f32" 0|1 (1) NOT bit-preserving (round 6): every k3 layer of OAI_PREC_F32 (ec1 ... dc1; ec1 / ec3 / ec5 with their MaxPool3d fused where the launch is a
 *                       whole tile) runs conv3_wino_f32 (unet_wino_f32.h): the x axis in Winograd F(2,3) form, exact fp32 products, the direct kernel's two-level
 *                       accumulation = 2/3 of the fp32 MFMAs (pass 590 -> 445 ms), and 0.66-0.70 x the reference's own fp32 distance from its float64 run where the
 *                       direct form (0) sits at 0.80-0.83 (profiles/r06_wino_f32.md).  A voxel's bits depend on the parity of its x, not on batches, launch boxes or
 *                       strips.  "winograd_layers" applies
 *   "first_blocks" 1..4096 (24) bit-preserving (round 6): workgroups per tile of the ec0 kernel (networks.py:43 where it is not fused into ec1's staging: the 3-voxel
 *                       shell of every tile, the fp16x3 path without fuse_first); each walks the tile's voxel pairs with a grid stride, the next pair's 36 inputs
 *                       gathered under the current pair's FMAs (2.17 -> 1.38 ms per 160 tiles)
 *   "up_nbw" 0..64 (0)  bit-preserving (round 6): column blocks of 256 a workgroup of the k2s2 up-conv kernel (networks.py:56,59,62) walks one after the other over its
 *                       128 voxels -- the voxel table, the A-row plan and the workgroup launch are paid once per walk; 0 = as many as keep >= 16 workgroups per slot of
 *                       the chip, 1 = one column block per workgroup (rounds 1-5), n = at most n
 *   "dead_stores" 0|1 (1) the encoder does not write the part of a skip tensor that the trimmed decoder never reads
 *   "census" 0|1 (1)    the kernels record per-layer activation maxima (activation exponents, LOW bit of the range flag)
 *   "calibrated" 0      forget that the activation exponents were calibrated (they keep their values): oai_unet_get_act_exponents reports 0 until
 *                       oai_unet_set_act_exponents or a settled oai_unet_calibrate_step.  For a caller that found its calibration unfit for the data
 *                       (three flagged volumes in a row under a sidecar file): ONE place holds "calibrated?" -- this handle.  Only 0 is accepted
 * Unknown names and out-of-range values return OAI_ERR_ARG. */
int oai_unet_set_option(oai_unet* h, const char* name, int value);

/* Bytes of device scratch oai_unet_forward_* needs for `batch` tiles of (td,th,tw). */
size_t oai_unet_workspace_bytes(const oai_unet* h, int td, int th, int tw, int batch);
/* ... and what oai_segment_tiles would like for a (D,H,W) volume: the same plus the max-pooled ec1 over the reflect-padded volume (1 GB at
 * 384x384x160) when the geometry allows the shared encoder pass -- ec0 -> ec1 (networks.py:109-113) computed once per volume instead of once
 * per overlapping tile (image_transforms.py:407-434: tiles overlap 2 x 1.33 x 1.33), bit-identical maps.  With only
 * oai_unet_workspace_bytes the call computes every tile on its own. */
size_t oai_segment_workspace_bytes(const oai_unet* h, int D, int H, int W, const int tile_zyx[3], const int overlap_zyx[3], int batch);

/* B3 seam: logits[B][n_classes][td][th][tw] = UNet(tiles[B][1][td][th][tw]) (NCDHW like
 * networks.py:109-149), zero conv padding at the tile border, no trimming. */
int oai_unet_forward_tiles(oai_unet* h, const float* tiles_dev, float* logits_dev, int B,
                           int td, int th, int tw, void* workspace_dev, size_t workspace_bytes,
                           void* stream);

/* Fused a3-a7 of SURVEY.md 8a for tiles [tile_begin, tile_end) of the reference's z-major tile
 * order: reflect-pad addressing of vol[D][H][W] (no tiles materialised) -> UNet on each tile ->
 * 1x1x1 head -> sigmoid (out_mode 0) or sigmoid>0.5 as 0/1 (out_mode 1) or raw logits (2) ->
 * kept centre blocks blocks_dev[tile - tile_begin][n_classes][ez][ey][ex] (e = tile - 2*overlap).
 * Only what the stitched result can depend on is computed (bit-identical dead-output trim): per layer the
 * box its consumers need, and per tile only the part of the kept centre that
 * Partition.assemble keeps -- crop_zyx (may be NULL) is the frame oai_stitch_blocks will zero, and the part
 * beyond the image is trimmed.  Block voxels outside that part are left unwritten. */
int oai_segment_tiles(oai_unet* h, const float* vol_dev, int D, int H, int W,
                      const int tile_zyx[3], const int overlap_zyx[3], const int crop_zyx[3],
                      int tile_begin, int tile_end,
                      int out_mode, float* blocks_dev, int batch,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* Partition.assemble (non-vote branch): scatter centre blocks of ALL tiles into
 * maps[n_classes][D][H][W], trim to the image, zero the outer frame of crop_zyx voxels.  A crop with a zero component
 * gives an all-zero map, as the reference's `[c:-c]` slicing does (image_transforms.py:509-513). */
int oai_stitch_blocks(const float* blocks_dev, int n_classes, int D, int H, int W,
                      const int tile_zyx[3], const int overlap_zyx[3], const int crop_zyx[3],
                      float* maps_dev, void* stream);

/* The same assemble, reading the blocks where an all_gather of per-rank tile ranges left them (the reference's
 * counterpart is the Dask gather of per-task results, dask_processing.py:170-189): rank r computed the tiles
 * [bounds_host[r], bounds_host[r + 1]) and its blocks sit in slots [r * slot_stride, r * slot_stride + its count) of blocks_dev --
 * ragged ranges are padded to slot_stride blocks per rank by the collective (all_gather_into_tensor needs equal pieces), and this
 * entry reads through the table instead of a compacting copy of 189 MB per volume.  bounds_host: n_ranges + 1 ascending tile
 * indices, [0] = 0, [n_ranges] = the tile count; n_ranges <= 64. */
int oai_stitch_blocks_ranged(const float* blocks_dev, int n_classes, int D, int H, int W,
                             const int tile_zyx[3], const int overlap_zyx[3], const int crop_zyx[3],
                             const int* bounds_host, int n_ranges, int slot_stride,
                             float* maps_dev, void* stream);

/* Partition.__call__ (image_transforms.py:395-455) as a standalone gather: tiles [tile_begin, tile_end) of the reference's
 * z-major order, tiles_dev[t - tile_begin][tz][ty][tx] = reflect-padded volume (pad lo = overlap; numpy.pad 'reflect').  For
 * callers that use Partition directly; oai_segment_tiles never materialises tiles. */
int oai_partition_tiles(const float* vol_dev, int D, int H, int W, const int tile_zyx[3], const int overlap_zyx[3],
                        int tile_begin, int tile_end, float* tiles_dev, void* stream);

/* Partition.assemble(is_vote=True) (image_transforms.py:466-484): every tile votes with all its voxels (overlaps included);
 * out_dev[D][H][W] (uint8) = index of the label plane with the most votes (lowest index on a tie, np.argmax).  tile_labels_dev
 * [n_tiles][tz][ty][tx] int32 must hold values in [0, n_labels) -- the reference indexes its vote array with the label VALUE,
 * so its labels are 0..L-1 too. */
int oai_assemble_vote(const int* tile_labels_dev, int n_labels, int D, int H, int W, const int tile_zyx[3], const int overlap_zyx[3],
                      unsigned char* out_dev, void* stream);

/* Roofline instrumentation (bench.py): when enabled, every launch of the dominant kernel (the 3x3x3
 * implicit-GEMM conv) is bracketed by hipEvents on the launch stream.  oai_unet_profile_read waits for the
 * recorded events and returns their summed duration and launch count, then clears them. */
int oai_unet_profile(oai_unet* h, int enable);
int oai_unet_profile_read(oai_unet* h, double* conv3_ms, long long* conv3_launches);

/* Algorithmic FLOPs (2*MACs) of one tile, full or with the dead-output trim. */
double oai_unet_tile_flops(const oai_unet* h, int td, int th, int tw, const int overlap_zyx[3], int trimmed);
/* Algorithmic FLOPs of segmenting a whole D x H x W volume as oai_segment_tiles does it (per-tile boxes included);
 * conv3_only restricts the sum to the layers of the 3x3x3 implicit-GEMM kernel. */
double oai_unet_volume_flops(const oai_unet* h, int D, int H, int W, const int tile_zyx[3], const int overlap_zyx[3],
                             const int crop_zyx[3], int trimmed, int conv3_only);
/* FLOPs of every tile of a volume as oai_segment_tiles computes it (border tiles cost less: trimmed kept centres): the weights
 * for splitting ONE volume's tiles over ranks (the reference's z-major order; oai_analysis_2_amd/parallel.py). */
int oai_unet_tile_costs(const oai_unet* h, int D, int H, int W, const int tile_zyx[3], const int overlap_zyx[3],
                        const int crop_zyx[3], double* costs_host, int n_tiles);
/* Same as oai_unet_tile_flops, restricted to the layers the 3x3x3 implicit-GEMM kernel runs (ec1-ec7, dc8, dc7, dc5, dc4, dc2, dc1). */
double oai_unet_tile_flops_conv3(const oai_unet* h, int td, int th, int tw, const int overlap_zyx[3], int trimmed);

/* ------------------------------------------------------------------------------------------
 * ICON registration network.  Replaces icon_registration.pretrained_models.
 * OAI_knees_gradICON_model (registration.py:20) + the network part of register_pair (:25).
 * ---------------------------------------------------------------------------------------- */

typedef struct oai_icon oai_icon;

/* One tallUNet2 = UNet2(5, [[2,16,32,64,256,512],[16,32,64,128,256]], 3) in the package's layouts
 * (host, fp32): downConvs[d].weight [Co][Ci][3][3][3], upConvs[d].weight [Ci][Co][4][4][4],
 * batchNorms[d].{weight,bias,running_mean,running_var}, lastConv.weight [3][18][3][3][3]. */
typedef struct oai_icon_unet_params {
    const float* down_w[5]; const float* down_b[5];
    const float* up_w[5];   const float* up_b[5];
    const float* bn_gamma[5]; const float* bn_beta[5]; const float* bn_mean[5]; const float* bn_var[5];
    const float* last_w; const float* last_b;
} oai_icon_unet_params;

/* The registration network is a TREE of the package's wrapper modules around n_nets tallUNet2s -- whatever the checkpoint behind
 * OAI_knees_gradICON_model (registration.py:20) holds; the nesting of `netPhi` / `netPsi` / `net` in its state_dict keys IS this
 * tree (oai_analysis_2_amd/registration.py:parse_icon_tree):
 *   OAI_ICON_FFVF   network_wrappers.FunctionFromVectorField(net = tallUNet2 number a): d = net(A, B) on A's grid;
 *                   transform(x) = x + d when x is the tagged identity map of d's own shape (the isIdentity shortcut), else
 *                   x + sample(d, x)
 *   OAI_ICON_DOWN   DownsampleRegistration(net = node a): the child sees avg_pool3d(A, 2, ceil_mode=True), avg_pool3d(B, ...);
 *                   its transform works in the same [0,1] coordinates
 *   OAI_ICON_TWO    TwoStepRegistration(netPhi = node a, netPsi = node b): phi = netPhi(A, B);
 *                   psi = netPsi(A warped by phi(identity map of A's grid), B); transform(x) = phi(psi(x))
 * nodes[root] is regis_net.  Every node is used exactly once; a child's index differs from its parent's.  A net index may not
 * repeat.  Limits: n_nets, chain length <= OAI_WARP_CHAIN_MAX_FIELDS; every grid a U-Net runs on needs each axis >= 17.
 * Examples (u_k = FFVF(net k)):  SURVEY Appendix A's three-step  TWO(DOWN(TWO(u0,u1)), u2);  "the definition of our final 4 step
 * registration network"  TWO(TWO(DOWN(TWO(u0,u1)), u2), u3);  the gradICON multi-resolution form  TWO(DOWN(TWO(DOWN(u0), u1)), u2). */
enum { OAI_ICON_FFVF = 0, OAI_ICON_DOWN = 1, OAI_ICON_TWO = 2 };
typedef struct oai_icon_node { int kind; int a; int b; } oai_icon_node;

/* BatchNorm3d behind every up-conv (networks.UNet2.batchNorms of the package the reference calls at registration.py:20): pass all four
 * bn_* arrays of a level, or NULL for all four = no normalisation at that level.  icon_registration 1.1.2 is not vendored in the
 * reference tree and one recollection of it has the batchNorms[depth] call commented out in UNet2.forward (the parameters are in the
 * state_dict either way): the caller decides, nothing is assumed silently. */
int oai_icon_create(const oai_icon_unet_params* nets_host, int n_nets, const oai_icon_node* nodes_host, int n_nodes, int root,
                    int D, int H, int W, oai_icon** out);
/* What the tree flattens to: n_nets, the number of launched U-Net passes per direction at each halving level (levels_host[k] =
 * U-Nets on the grid halved k times, k < 8), and the length of the final compose chain. */
int oai_icon_describe(const oai_icon* h, int* n_nets, int* levels_host, int* chain_len);
void oai_icon_destroy(oai_icon* h);
size_t oai_icon_workspace_bytes(const oai_icon* h);

/* phi_AB(identity)[3][D][H][W] for network-resolution images A, B [D][H][W] (one direction of
 * GradientICON.forward followed by model.phi_AB(model.identity_map)). */
int oai_icon_forward(oai_icon* h, const float* A_dev, const float* B_dev, float* phi_dev,
                     void* workspace_dev, size_t workspace_bytes, void* stream);

/* oai_icon_forward replays its dependent launches (~70 for three U-Nets) as ONE hipGraph (captured on the first call per workspace, on an internal
 * stream; inputs and result have fixed homes inside the workspace, copied in / out around the replay).  oai_icon_set_graph(h, 0)
 * runs the same launches directly.  oai_icon_graph_info: *captured = 1 graph in use, 0 not captured yet, -1 capture failed on this
 * runtime (direct launches are used: same kernels, same results); counts of replays / direct runs. */
int oai_icon_set_graph(oai_icon* h, int enable);
/* Restatement switches of the un-vendored package (defaults = SURVEY Appendix A):
 *   "pad_front" 0|1 (1)  networks.pad_or_crop zero-pads the residual's missing channels in front (1) or behind (0) of the existing ones
 *                        (down path: avg_pool3d(x) has fewer channels than the conv's output). */
int oai_icon_set_option(oai_icon* h, const char* name, int value);
int oai_icon_graph_info(const oai_icon* h, int* captured, long long* replays, long long* direct_runs);

/* One tallUNet2 forward on its own (unit-test seam): out[3][D][H][W] = net number `which` (a, b). */
int oai_icon_unet_forward(oai_icon* h, int which, const float* a_dev, const float* b_dev, int D, int H, int W,
                          float* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Iso-surface, smoothing, thickness distance: the step just after the hot path,
 * oai_analysis/mesh_processing.py:298-340, 381-395.
 *   get_mesh():      skimage.measure.marching_cubes(img, level=0.5, spacing, step_size=1)   -> oai_mc_count + oai_mc_emit
 *                    vtkSmoothPolyDataFilter(num_iterations)                                -> oai_mesh_smooth
 *   get_distance():  vtkDistancePolyDataFilter(SignedDistanceOff, second distance on)       -> oai_mesh_point_distance (x2)
 * Conventions (oracle/mesh.py): volume [z][y][x]; inside = value > iso; one vertex per sign-changing grid edge, linear
 * interpolation, (x, y, z) * spacing, ordered by owning voxel then axis; triangles wind outward (inside -> outside), ordered by
 * cell then case-table order; the case table is face-consistent (watertight surface).
 * ---------------------------------------------------------------------------------------- */
/* The 256 x 16 case table (edge ids, -1 terminated) the kernels use, for cross-checks against the oracle's generator. */
int oai_mc_table(signed char* out_256x16_host);
size_t oai_mc_workspace_bytes(int D, int H, int W);
/* Pass 1: classify + prefix sums into the workspace; returns the vertex and triangle counts (synchronises the stream). */
int oai_mc_count(const float* vol_dev, int D, int H, int W, float iso, void* workspace_dev, size_t workspace_bytes,
                 long long* n_verts_host, long long* n_tris_host, void* stream);
/* Pass 2 (same volume, iso and workspace): verts float32 [n_verts][3], faces int32 [n_tris][3]. */
int oai_mc_emit(const float* vol_dev, int D, int H, int W, float iso, const float spacing_xyz_host[3], const void* workspace_dev,
                float* verts_dev, int* faces_dev, void* stream);
/* x <- x + relaxation * (mean of edge neighbours - x), `iterations` Jacobi sweeps over the CSR edge graph
 * (offsets [n_verts+1], neighbours); tmp and out are [n_verts][3] and distinct from the input. */
int oai_mesh_smooth(const float* verts_in_dev, long long n_verts, const int* offsets_dev, const int* neighbours_dev,
                    int iterations, float relaxation, float* tmp_dev, float* verts_out_dev, void* stream);
/* dist[i] = unsigned distance from points[i] to the closest point of the triangle mesh (verts, faces). */
int oai_mesh_point_distance(const float* points_dev, long long n_points, const float* verts_dev, const int* faces_dev,
                            long long n_tris, float* dist_dev, void* stream);
/* Same result with a uniform-grid broad phase (cells of `cell_size` from `grid_lo`, `grid_dims` cells per axis, covering the mesh;
 * cell_size must be >= the longest triangle edge so that a triangle touches at most 8 cells).  Synchronises once. */
size_t oai_mesh_grid_workspace_bytes(const int grid_dims_xyz[3], long long n_tris);
int oai_mesh_point_distance_grid(const float* points_dev, long long n_points, const float* verts_dev, const int* faces_dev,
                                 long long n_tris, const float grid_lo_xyz_host[3], float cell_size, const int grid_dims_xyz_host[3],
                                 void* workspace_dev, size_t workspace_bytes, float* dist_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OAI_HIP_H */
