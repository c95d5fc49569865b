/*
 * vistracker.h -- C ABI of libvistracker_hip.so: the MI355X (gfx950) implementation of the VisTracker
 * per-frame SMPL-H / object fitting hot path.
 *
 * The reference has no FFI: its "operator API" for this path is the set of duck-typed Python callables the
 * fitters invoke.  Each entry point below names the reference callable it replaces
 * (paths relative to the reference tree).  The Python shims in vistracker_amd/ bind these with ctypes and
 * re-expose the reference's call signatures (smpl(), get_landmarks(), model.query()/get_preds(),
 * SilLossROI(...), chamfer_distance(...)).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a contiguous row-major array (float = fp32, int = int32)
 *     unless the parameter is documented "host".
 *   - the library never allocates outputs or scratch: the caller passes them (sizes documented per call);
 *     handles own only immutable constants uploaded at creation.
 *   - every call enqueues on the caller's HIP stream (`stream` = hipStream_t cast to void*) and returns
 *     without synchronising; no global mutable state besides the per-thread error string.
 *   - return value: 0 = ok, negative = error (see vt_last_error()).
 */
#ifndef VISTRACKER_H
#define VISTRACKER_H

#ifdef __cplusplus
extern "C" {
#endif

#define VT_OK 0
#define VT_ERR_ARG -1
#define VT_ERR_HIP -2
#define VT_ERR_BUSY -3     /* the resource is already held by the same thread (vt_stream_set_skip_flag: another stop flag registered for the stream) */

#define VT_SMPL_V 6890
#define VT_SMPL_J 52
#define VT_SMPL_NB 10
#define VT_SMPL_NP 459
#define VT_FEAT 611
#define VT_HID 128

const char *vt_last_error(void);
int vt_version(void);

/* ---------------------------------------------------------------------------------------------------
 * SMPL-H layer.  Replaces SMPL_Layer.forward (lib_smpl/smplpytorch/smplpytorch/pytorch/smpl_layer.py:73-176),
 * th_posemap_axisang / batch_rodrigues (tensutils.py:6-19, rodrigues_layer.py:38-52) and its autograd backward.
 * ------------------------------------------------------------------------------------------------- */
typedef struct vt_smplh vt_smplh;

/* HOST pointers: v_template (V,3), shapedirs (V,3,NB), posedirs (V,3,NP), J_regressor (J,V) dense,
 * weights (V,J), parents (J) [parents[0] ignored].  smpl_layer.py:46-71. */
int vt_smplh_create(vt_smplh **out, const float *v_template, const float *shapedirs, const float *posedirs,
                    const float *J_regressor, const float *weights, const int *parents, void *stream);
void vt_smplh_destroy(vt_smplh *h);

/* floats of per-call workspace `ws` for batch B (kept by the caller between forward and backward) */
long vt_smplh_workspace_floats(int B);

/* pose (B,156) betas (B,10) trans (B,3) -> verts (B,6890,3), jtr (B,52,3), v_posed (B,6890,3)  */
int vt_smplh_forward(const vt_smplh *h, const float *pose, const float *betas, const float *trans, int B,
                     float *verts, float *jtr, float *v_posed, float *ws, void *stream);

/* dverts (B,6890,3), djtr (B,52,3) or NULL -> dpose (B,156), dbetas (B,10), dtrans (B,3)  (overwritten).
 * `ws` and `v_posed` must be the buffers the matching forward filled.  scratch: vt_smplh_bwd_scratch_floats(B). */
long vt_smplh_bwd_scratch_floats(int B);
int vt_smplh_backward(const vt_smplh *h, const float *pose, const float *betas, int B, const float *dverts,
                      const float *djtr, const float *v_posed, const float *ws, float *scratch,
                      float *dpose, float *dbetas, float *dtrans, void *stream);

/* batch_rodrigues alone: aa (n,3) -> R (n,9); and its VJP */
int vt_rodrigues_forward(const float *aa, int n, float *R, void *stream);
int vt_rodrigues_backward(const float *aa, int n, const float *dR, float *daa, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Sparse landmark regressors.  Replaces load_regressors + batch_sparse_dense_matmul
 * (lib_smpl/body_landmark.py:16-28, lib_smpl/torch_functions.py:52-76).
 * ------------------------------------------------------------------------------------------------- */
typedef struct vt_landmarks vt_landmarks;
/* HOST CSR of the (K x V) regressor */
int vt_landmarks_create(vt_landmarks **out, const int *indptr, const int *indices, const float *data, int K, int V,
                        void *stream);
void vt_landmarks_destroy(vt_landmarks *h);
/* verts (B,V,3) -> out (B,K,3) */
int vt_landmarks_forward(const vt_landmarks *h, const float *verts, int B, float *out, void *stream);
/* dout (B,K,3) -> dverts (B,V,3) += (accumulate != 0) or = (accumulate == 0; rows without entries zeroed) */
int vt_landmarks_backward(const vt_landmarks *h, const float *dout, int B, float *dverts, int accumulate, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Mahalanobis priors.  Replaces th_Mahalanobis.__call__ (lib_smpl/th_smpl_prior.py:30-38) and
 * HandPrior.__call__ (lib_smpl/th_hand_prior.py:57-72):  value[b] = | (x[b, off:off+n] - mean) @ prec |^2.
 * dx (B,stride) += gscale * d value[b] / dx  when dx != NULL.   mean (n), prec (n,n) device, n <= 64.
 * ------------------------------------------------------------------------------------------------- */
int vt_mahalanobis(const float *x, int B, int stride, int off, int n, const float *mean, const float *prec,
                   float *value, float *dx, float gscale, void *stream);

/* Loss-term accumulators: every `double *term` below is a DEVICE fp64 scalar the kernels add to with atomics
 * (fp64 keeps the sum order-independent to ~1e-16, so step losses and the early-stop decision are reproducible).
 * Zero them (vt_fill_f64) before the step. */
int vt_fill_f64(double *p, long n, double value, void *stream);
/* *term += scale * sum(value[0..n)) */
int vt_sum_to_term(const float *value, int n, float scale, double *term, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * SIF-Net point query.  Replaces CHORETriplane.query + CHORETriplaneVisibility.decode + get_preds
 * (model/chore_triplane.py:97-164,207-251; model/chore_tri_vis.py:17-50; model/camera.py:45-90;
 * model/geometry.py:4-14) and the autograd backward to the query coordinates.
 * Heads (bit index in head masks): 0 df(2) 1 pca(9) 2 parts(14) 3 centers(3) 4 vis(1).
 * ------------------------------------------------------------------------------------------------- */
typedef struct vt_sifnet vt_sifnet;
/* HOST weights: w[h*4+l] is (out,in) row-major (torch Conv1d.weight[:, :, 0]), b[h*4+l] is (out);
 * cam (host, 5 floats) = {fx_px, fy_px, cx_px, cy_px, crop_size} (camera.py:26-41). */
int vt_sifnet_create(vt_sifnet **out, const float *const *w, const float *const *b, const float *cam, void *stream);
void vt_sifnet_destroy(vt_sifnet *h);

/* Feature maps are consumed channel-last.  Converts one NCHW map (B,C,H,W) to NHWC (B,H,W,C). */
int vt_nchw_to_nhwc(const float *src, int B, int C, int H, int W, float *dst, void *stream);

/* maps (host array of 8 device pointers, NHWC) in the order im_feat(256) tmpx(64) tri_tmpx{right,back,top}(32)
 * tri_feat{right,back,top}(64); res (host, 8 ints) = H (=W) of each map. */
typedef struct {
    const float *maps[8];
    int res[8];
    /* optional: hoisted layer-1 projection of im_feat, (B, res[0], res[0], proj_cols) floats written by vt_query_build_projection for THESE
     * maps and THIS network (NULL = not available).  Layer 1 of the decoders (Conv1d(611,128), chore.py:113-126) is linear in the features
     * and the features are bilinear blends of texels (geometry.py:4-14), so its im_feat part can be applied to the texels once per batch
     * instead of to every query point at every optimisation step; the fused objective / projection-step entry points then blend rows of
     * this array instead of gathering the 256 im_feat channels and multiplying them by W1.  Results agree to fp32 round-off. */
    const float *proj;
    int proj_cols;
    /* operand-range level of the split-f16 decoders for the calls that pass THESE maps: level k runs the last hidden layer at the operand scale
     * 2^(6 - 4 k) (representable |activation| < 1023, 16 368, 262 016 for k = 0, 1, 2; the earlier layers have at least that range -- the decoders
     * of model/chore.py:113-126 have no such bound); beyond it a call yields NaN, loudly.  Same weights, exact power-of-two rescaling of biases /
     * features / outputs, so a checkpoint with large activations costs one repeated launch at the next level, not the 5x slower fp32 route.
     * proj_level = the level `proj` was built for (vt_query_build_projection uses act_level; a projection of another level is ignored).
     * Zero-initialised = level 0. */
    int act_level;
    int proj_level;
    /* non-zero: these calls run on the strict-fp32 kernels whatever the handle's precision (per batch, so concurrent fits through one handle do
     * not see each other's switch) */
    int force_fp32;
} vt_maps;

/* pts (B,N,3), crop_center (B,2), body_center (B,3).  Outputs may be NULL (head skipped); layout as the
 * reference: df (B,2,N), pca (B,9,N), parts (B,14,N), centers (B,3,N), vis (B,1,N). */
int vt_query_forward(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center,
                     const float *body_center, int B, int N,
                     float *df, float *pca, float *parts, float *centers, float *vis, void *stream);
/* upstream gradients in the same layouts (NULL = zero) -> dpts (B,N,3) (overwritten) */
int vt_query_backward(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center,
                      const float *body_center, int B, int N,
                      const float *d_df, const float *d_pca, const float *d_parts, const float *d_centers,
                      const float *d_vis, float *dpts, void *stream);

/* floats of the projection array for a batch of B frames, and its construction (one fp32 GEMM over all im_feat texels; heads df | parts) */
long vt_query_projection_floats(const vt_maps *maps, int B);
int vt_query_build_projection(const vt_sifnet *h, const vt_maps *maps, int B, float *proj, void *stream);

/* Fused objective kernels of the two fit loops (no autograd tape, one launch per step):
 *  human:  loss += w_dfh * mean_{B,N} clamp(df[:,0], max=.1) + w_part * mean_B sum_N CE(parts, labels)
 *          (recon_fit_base.py:640-647, recon_fit_behave.py:486); labels (N) int32
 *  object: loss += w_obj * mean_B( mean_N clamp(df[:,1], max=.8) * occ[b] )   (recon_fit_trivis_full.py:155-162)
 * dpts (B,N,3) is overwritten with the weighted gradient; terms[0..1] (device fp64) receive the UNWEIGHTED term
 * values (df_h, part) or (object, -) accumulated with atomics -- zero them before the call. */
/* order (N) int32 or NULL: a permutation of the point indices -- workgroup slot n processes point order[n] (inputs read, labels looked
 * up and dpts written at the ORIGINAL index, so results do not depend on it); a locality-preserving order (e.g. the Morton order of the
 * template vertices) makes the 64 points of a workgroup gather from neighbouring texels. */
int vt_query_human_loss(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center,
                        const float *body_center, int B, int N, const int *labels, const int *order, float w_dfh, float w_part,
                        float *dpts, double *terms, void *stream);
int vt_query_object_loss(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center,
                         const float *body_center, int B, int N, const float *occ, float w_obj,
                         float *dpts, double *terms, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * 3x3 / stride 1 / pad 1 bias-free convolution of the HGFilter encoders (model/HGFilters.py:56-203, model/net_util.py:346-396 ConvBlock) as a
 * split-f16 implicit GEMM (fp32 accumulate, same operand format as the point query).  weight (Cout, Cin, 3, 3) fp32 host, Cout in {32, 64, 128} (32: the 64-channel kernel with zero rows, half its waves idle in the MFMA phase),
 * Cin a multiple of 32.  in (B,H,W,Cin) NHWC fp32 device, H % 8 == 0, W % 16 == 0; the result goes to channels [out_coff, out_coff + Cout) of an
 * NHWC tensor with out_cstride channels (a ConvBlock's concatenation can be written in place).  A value beyond the operand range yields NaN.
 * ------------------------------------------------------------------------------------------------- */
typedef struct vt_conv3x3 vt_conv3x3;
int vt_conv3x3_create(vt_conv3x3 **out, const float *weight, int cout, int cin, void *stream);
void vt_conv3x3_destroy(vt_conv3x3 *h);
int vt_conv3x3_forward(const vt_conv3x3 *h, const float *in, int B, int H, int W, float *out, int out_cstride, int out_coff, void *stream);
/* The pre-activated form of the ConvBlock (GroupNorm -> ReLU -> conv, model/net_util.py:374-388) in one pass: the input is channels
 * [in_coff, in_coff + Cin) of an NHWC tensor with in_cstride channels; with gn_stats != NULL (the (B, groups) {mean, rstd} pairs of
 * vt_groupnorm_stats) max((x - mean) rstd gamma + beta, 0) is applied while the operand planes are staged (zero padding pads the rectified value). */
int vt_conv3x3_forward_gn(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                          const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff, void *stream);
/* The same with the GroupNorm partial sums of the OUTPUT as a by-product (per 8 x 16 pixel tile, vt_conv3x3_tiles(H, W) of them, at
 * stats_ws + B * stats_groups doubles): vt_groupnorm_finalize(stats_ws, tiles, B, H * W, Cout, stats_groups, eps) turns them into the {mean, rstd}
 * pairs at the start of stats_ws -- the statistics the next convolution of a ConvBlock needs, without a pass over the tensor.
 * stats_ws >= B stats_groups + tiles B Cout 2 doubles (NULL: no statistics). */
int vt_conv3x3_forward_gn_stats(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                                const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff, double *stats_ws,
                                int stats_groups, void *stream);
/* The general form (one ConvBlock step): out (or NULL) <- convolution, fin (or NULL) <- convolution + res, the block's result for these channels
 * (model/net_util.py:390-394: the residual add without a pass of its own). */
int vt_conv3x3_forward_block(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                             const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff,
                             const float *res, int res_cstride, int res_coff, float *fin, int fin_cstride, int fin_coff,
                             double *stats_ws, int stats_groups, void *stream);
int vt_conv3x3_tiles(int H, int W);
/* The 7 x 7 / stride 2 / pad 3 convolution at the head of an encoder (model/HGFilters.py:118-130: self.conv1 = nn.Conv2d(in_ch, 64, kernel_size=7, stride=2,
 * padding=3), applied at HGFilters.py:166 `x = F.relu(self.bn1(self.conv1(x)), True)`): plain fp32 multiply-adds like the reference's, weights as scalar
 * operands (csrc/stem.hip).  weight (Cout, Cin, 7, 7) fp32 host, bias (Cout) fp32 host or NULL, Cout in {32, 64} (triplane / image encoder), Cin in 1..8.
 * forward: the input is channels [in_coff, in_coff + Cin) of an NHWC tensor (B, H, W, in_cstride), the result (B, ceil(H / 2), ceil(W / 2), Cout) goes to channels [out_coff, out_coff + Cout)
 * of an NHWC tensor with out_cstride channels (both multiples of 4).  Replaces the encoder's last MIOpen call (round 6). */
typedef struct vt_stem7x7 vt_stem7x7;
int vt_stem7x7_create(vt_stem7x7 **out, const float *weight, const float *bias, int cout, int cin, void *stream);
void vt_stem7x7_destroy(vt_stem7x7 *h);
int vt_stem7x7_forward(const vt_stem7x7 *h, const float *in, int in_cstride, int in_coff, int B, int H, int W, float *out, int out_cstride, int out_coff, void *stream);
/* The 1 x 1 convolutions of the same encoders (conv_last / l / bl / al of a stack, the ConvBlock's down-sampling projection: model/HGFilters.py:150-203,
 * model/net_util.py:364-372) on the same split-f16 kernel (one tap, the 8 x 16 tile as the patch).  weight (Cout, Cin) fp32 host, bias (Cout) or NULL,
 * Cout in {64, 128, 256} (256 runs as two launches of 128), Cin in {32, 64, 128, 256}.  forward: out <- conv(in) + bias [+ res]; gn_stats != NULL applies
 * max((x - mean) rstd gamma + beta, 0) to the input while it is staged (the GroupNorm -> ReLU in front of l / bl / the projection); stats_ws != NULL
 * leaves the GroupNorm partial sums of conv(in) + bias behind like vt_conv3x3_forward_gn_stats (finalize with C = Cout). */
typedef struct vt_conv1x1 vt_conv1x1;
int vt_conv1x1_create(vt_conv1x1 **out, const float *weight, const float *bias, int cout, int cin, void *stream);
void vt_conv1x1_destroy(vt_conv1x1 *h);
int vt_conv1x1_forward(const vt_conv1x1 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma, const float *beta, int groups,
                       int B, int H, int W, float *out, int out_cstride, int out_coff, const float *res, int res_cstride, int res_coff,
                       double *stats_ws, int stats_groups, void *stream);
int vt_groupnorm_finalize(double *ws, int nblk, int B, int HW, int C, int groups, float eps, void *stream);
/* vt_conv3x3_forward_block + the GroupNorm partial sums of `fin` (convolution + residual: the ConvBlock's result, model/net_util.py:390-394) for the
 * ConvBlock that reads it next: every launch of a block writes the partials of its channel slice [fin_coff, fin_coff + Cout) into one block of
 * fin_cstride channels per 8 x 16 pixel tile at fin_stats_ws + B * fin_stats_groups doubles; after the block's last launch
 * vt_groupnorm_finalize(fin_stats_ws, vt_conv3x3_tiles(H, W), B, H * W, fin_cstride, fin_stats_groups, eps) yields what vt_groupnorm_stats(fin) would. */
int vt_conv3x3_forward_block_stats(const vt_conv3x3 *h, const float *in, int in_cstride, int in_coff, const float *gn_stats, const float *gamma,
                                   const float *beta, int groups, int B, int H, int W, float *out, int out_cstride, int out_coff,
                                   const float *res, int res_cstride, int res_coff, float *fin, int fin_cstride, int fin_coff,
                                   double *stats_ws, int stats_groups, double *fin_stats_ws, int fin_stats_groups, void *stream);
/* 2 x 2 average pooling of an NHWC tensor (F.avg_pool2d(x, 2, stride=2): model/HGFilters.py:33,131-136), x (B, H, W, C) -> out (B, H/2, W/2, C), and
 * -- stats_ws != NULL -- the GroupNorm partial sums of the OUTPUT: vt_sweep_blocks(H/2 * W/2) blocks at stats_ws + B * stats_groups doubles, for
 * vt_groupnorm_finalize(stats_ws, vt_sweep_blocks(..), B, H/2 * W/2, C, stats_groups, eps) in place of a statistics pass by the consumer. */
int vt_avgpool2x2_stats(const float *x, int B, int H, int W, int C, float *out, double *stats_ws, int stats_groups, void *stream);
int vt_sweep_blocks(int HW);
/* GroupNorm statistics of a channel slice: ws >= vt_groupnorm_workspace_doubles(B, HW, C, groups) doubles; the (B, groups) {mean, rstd} float pairs
 * are written at the START of ws (pass `(const float *)ws` as gn_stats) */
int vt_groupnorm_stats(const float *x, int cstride, int coff, int B, int HW, int C, int groups, float eps, double *ws, void *stream);

/* Arithmetic of the decoder GEMMs behind every vt_query_* call of a handle:
 *   VT_PRECISION_SPLIT_F16 (default): 22-bit split-f16 operands on the f16 MFMA, fp32 accumulate (5.3x the f32-input MFMA rate; forward within
 *       ~1e-6 of the fp32 reference); activations must satisfy |x| < 1023 * 16^(vt_maps::act_level), beyond that the result is NaN;
 *   VT_PRECISION_FP32: exact fp32 products on the f32-input MFMA (the reference's nn.Conv1d arithmetic, model/chore.py:113-126), any magnitude,
 *       ~1/5 of the speed; the hoisted projection of the maps is ignored.
 * The fused fit loops of the host layer switch a handle to VT_PRECISION_FP32 and re-run the batch when the split route produced a non-finite
 * loss.  Per handle, may be changed between calls. */
#define VT_PRECISION_SPLIT_F16 0
#define VT_PRECISION_FP32 1
int vt_sifnet_set_precision(vt_sifnet *h, int mode);
int vt_sifnet_get_precision(const vt_sifnet *h);

/* Kernel selection for vt_query_human_loss when the maps carry the hoisted projection.  The product library holds ONE kernel (256: the two-workgroups-per-CU
 * kernel all query entry points use) and refuses anything else with VT_ERR_ARG.  The experiments build (csrc/experiments, `make experiments`, never shipped)
 * additionally accepts 512 (one 512-thread workgroup per 64-point tile, thin waves) and 128 (producer / consumer waves, 128 points per workgroup): measured
 * 52 % / 22 % slower, kept for A/B measurements and as independently written cross-checks of the same arithmetic.  Process-wide. */
int vt_query_set_human_kernel(int threads);

/* One projection step of the SIF-Net surface-point generator, fused.  Replaces one iteration of
 * Generator.approx_surface (recon/gen/generator.py:72-103): query -> target = clamp(df[:, df_idx], max = threshold) ->
 * autograd of sum(target) to the samples -> samples - F.normalize(grad, dim=2) * target.  df_idx 0 = human, 1 = object.
 * pts_out (B,N,3) may alias pts (every point is read and written by exactly one workgroup); df_target (B,N) or NULL receives
 * the clamped distance at the INPUT positions (what the caller thresholds with filter_val). */
int vt_query_project_step(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center,
                          const float *body_center, int B, int N, int df_idx, float threshold, float *pts_out,
                          float *df_target, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused heads / tails of one Adam step of the two fit loops (the vt_objfit_step / vt_smplfit_step of SURVEY.md 8(b)): the arithmetic of the
 * single-purpose entry points below, element for element and in the same order, in 4 launches per object-stage step (head, query, stencils,
 * tail) and one tail per SMPL-stage step.  `terms` (device fp64), `w` (host, per-term weights), `state`, `stop_flag`, `history`, `slot`, `tol`,
 * `armed` as in vt_loss_reduce_and_stop; `ticket`: one zero-initialised device int per fit loop; the workgroup that finishes last closes the step
 * (loss, history, the reference's stop rule recon_fit_trivis_full.py:372 / recon_fit_behave.py:447) and zeroes terms[0, nzero).
 *   vt_objstep_head   R = project_so3(M0 + 1e-4 noise) (recon_fit_base.py:179-199,462-469); X = (X0 R + t) s for the surface points and, if X_verts
 *                     != NULL, the template vertices (recon_fit_base.py:455-459); terms[0, nzero) = 0
 *   vt_temporal_loss2 vt_accel_loss then vt_velocity_loss of (B, D) in one pass; init_zero: dv is written, not accumulated
 *   vt_objstep_tail   rigid VJP over the vertex set (dX_verts != NULL: phase 'sil') and the points, the translation regulariser
 *                     mean (t - t_init)^2 (t_init != NULL; recon_fit_trivis_full.py:227), SO(3) VJP, Adam (torch.optim.Adam) on pR (B,9) / pT (B,3)
 *                     (NULL: that group is not optimised), step end
 *   svd_ws            (B, 22) floats or NULL: vt_objstep_head leaves the SVD of M0 + 1e-4 noise there (U, V, s, det sign) and the tail of the SAME step takes
 *                     it from there instead of decomposing the matrix a second time (the same numbers; ~10 us of one thread per step)
 *   vt_smplstep_tail  body-pose prior th_Mahalanobis on pose[:, 3:66] (th_smpl_prior.py:30-38; gscale_prior = w / B), pinit term mean_B sum
 *                     (pose[:, 3:72] - pose_init)^2 (recon_fit_behave.py:500-503), Adam on up to three column slices (p, stride, g, stride, m, v,
 *                     columns, lr; p == NULL: unused), step end
 * ------------------------------------------------------------------------------------------------- */
/* SMPL-stage step forms: vt_kpts_step = vt_landmarks_forward + vt_kpts_loss + vt_landmarks_backward in one launch (J (B,K,3) or NULL receives the joints; dverts
 * is written -- accumulate = 0 -- or accumulated); vt_query_human_step = vt_query_human_loss whose gradient is ADDED to dpts (accumulate != 0: after vt_kpts_step)
 * and, with term_accel != NULL, the vertex acceleration stencil vt_accel_loss(pts, B, 3 N, NULL, w_accel, term_accel, dpts) in the same launch.  Same values as the
 * separate launches (float additions in the same order).  Split-f16 route only. */
int vt_kpts_step(const vt_landmarks *h, const float *verts, const float *kpts, const float *crop_center, int B, int mode, const float *cam, float net_size,
                 float gscale, double *term, float *J, float *dverts, int accumulate, void *stream);
int vt_query_human_step(const vt_sifnet *h, const vt_maps *maps, const float *pts, const float *crop_center, const float *body_center,
                        int B, int N, const int *labels, const int *order, float w_dfh, float w_part, int accumulate, float w_accel,
                        double *term_accel, float *dpts, double *terms, void *stream);
int vt_objstep_head(const float *M0, const float *noise, const float *t, const float *s, int B, const float *X0_points, int N, float *X_points,
                    const float *X0_verts, int NV, float *X_verts, float *R, double *terms, int nzero, float *svd_ws, void *stream);
int vt_temporal_loss2(const float *v, int B, int D, float gscale_accel, double *term_accel, float gscale_velocity, double *term_velocity, float *dv,
                      int init_zero, void *stream);
int vt_objstep_tail(const float *X0_verts, int NV, const float *dX_verts, const float *X0_points, int N, const float *dX_points, const float *s, int B,
                    const float *M0, const float *noise, const float *t, const float *t_init, float w_trans, double *term_trans,
                    float *dR, float *dt, float *dM,
                    float *pR, float *mR, float *vR, float lrR, float *pT, float *mT, float *vT, float lrT, int adam_step, float beta1, float beta2, float eps,
                    double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero,
                    float *svd_ws, void *stream);
/* vt_objstep_tail with the stencils of vt_temporal_loss2(X_points, B, 3 N, ...) evaluated inside it (the same float additions in the same order: bit-identical
 * parameters, one launch less per step); init_zero as there (dX_points is then not read).  For the phases in which nothing else adds to dX_points between the
 * stencils and the tail ('object only', 'sil'). */
int vt_objstep_tail_temporal(const float *X_points, float gscale_accel, double *term_accel, float gscale_velocity, double *term_velocity, int init_zero,
                             const float *X0_verts, int NV, const float *dX_verts, const float *X0_points, int N, const float *dX_points, const float *s, int B,
                             const float *M0, const float *noise, const float *t, const float *t_init, float w_trans, double *term_trans,
                             float *dR, float *dt, float *dM,
                             float *pR, float *mR, float *vR, float lrR, float *pT, float *mT, float *vT, float lrT, int adam_step, float beta1, float beta2, float eps,
                             double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero,
                             float *svd_ws, void *stream);
int vt_smplstep_tail(float *pose, const float *pose_init, float *dpose, int B, const float *prior_mean, const float *prior_prec, float gscale_prior,
                     double *term_prior, float w_pinit, double *term_pinit,
                     float *p0, int ps0, const float *g0, int gs0, float *m0, float *v0, int n0, float lr0,
                     float *p1, int ps1, const float *g1, int gs1, float *m1, float *v1, int n1, float lr1,
                     float *p2, int ps2, const float *g2, int gs2, float *m2, float *v2, int n2, float lr2,
                     int adam_step, float beta1, float beta2, float eps,
                     double *terms, const float *w, int nterms, float tol, int armed, float *state, int *stop_flag, float *history, int slot, int *ticket, int nzero,
                     void *stream);

/* ---------------------------------------------------------------------------------------------------
 * SO(3) projection.  Replaces ReconFitterBase.project_so3 / decopose_axis (recon/recon_fit_base.py:179-199,
 * 462-469): R = U diag(1,1,det(U V^T)) V^T of M; noise (B,3,3) or NULL is the U[0,1) sample, M = M0 + 1e-4*noise.
 * ------------------------------------------------------------------------------------------------- */
int vt_so3_project_forward(const float *M0, const float *noise, int B, float *R, void *stream);
int vt_so3_project_backward(const float *M0, const float *noise, int B, const float *dR, float *dM, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Rigid transform.  Replaces transform_obj_verts (recon/recon_fit_base.py:455-459): X = (X0 @ R + t) * s.
 * X0 (N,3) shared by all frames if shared_x0 != 0 else (B,N,3).
 * ------------------------------------------------------------------------------------------------- */
int vt_rigid_forward(const float *X0, int shared_x0, const float *R, const float *t, const float *s, int B, int N,
                     float *X, void *stream);
/* dX (B,N,3) -> dR (B,3,3), dt (B,3)  (+= if accumulate) */
int vt_rigid_backward(const float *X0, int shared_x0, const float *s, int B, int N, const float *dX,
                      float *dR, float *dt, int accumulate, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Temporal stencils (recon/recon_fit_trivis_full.py:170-177,379-391; preprocess/fit_SMPLH_30fps.py:189-200).
 *   accel:    mse(v[1:-1]-v[:-2], v[2:]-v[1:-1]) (optionally weighted per element by elem_w (D))
 *   velocity: mse(v[1:], v[:-1])
 * v (B,D).  *term (device scalar) += value;  dv (B,D) += gscale * gradient (dv may be NULL).
 * ------------------------------------------------------------------------------------------------- */
int vt_accel_loss(const float *v, int B, int D, const float *elem_w, float gscale, double *term, float *dv, void *stream);
/* same on the first D columns of a row-strided matrix (v, dv have `stride` floats per frame): pose accelerations */
int vt_accel_loss_strided(const float *v, int B, int D, int stride, const float *elem_w, float gscale, double *term,
                          float *dv, void *stream);
int vt_velocity_loss(const float *v, int B, int D, float gscale, double *term, float *dv, void *stream);

/* 2D keypoint terms.  mode 0: BaseFitter.compute_loss 'kpts' (preprocess/fit_SMPLH_kpts.py:280-310):
 *   mean_{B,25,2}((proj - k_xy)^2 * conf), full-image pinhole.
 * mode 1: projection_loss 'j2d' (recon/recon_fit_base.py:781-802): mean_{B,25}(sum_xy(proj - k)^2 * conf) with the
 *   crop-space projection scaled by net_size / crop_size; crop_center (B,2).
 * J (B,K,3), kpts (B,K,3);  *term += value;  dJ (B,K,3) = gscale * gradient (overwritten). cam (host,5). */
int vt_kpts_loss(const float *J, const float *kpts, const float *crop_center, int B, int K, int mode,
                 const float *cam, float net_size, float gscale, double *term, float *dJ, void *stream);

/* sum_i w_i (a_i - b_i)^2 / denom  over n elements laid out with row strides (rows x cols):
 * pinit terms (fit_SMPLH_30fps.py:178; recon_fit_behave.py:493-495) and 'trans' (recon_fit_trivis_full.py:223). */
int vt_sqdiff_loss(const float *a, int a_stride, const float *b, int b_stride, int rows, int cols, float denom,
                   float gscale, double *term, float *da, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Human / object interpenetration ("collide", weight 3^2 / (1 + decay) in phase 'joint').  Replaces RegistrationBase.smpl_obj_collision /
 * compute_collision_loss (recon/recon_fit_base.py:97-100,736-765; recon_fit_trivis_full.py:260-264): mesh_intersection's BVH(max_collisions=8)
 * + DistanceFieldPenetrationLoss(sigma=0.5, point2plane=False) -- un-vendored, PARITY UNPINNED, restated from Tzionas et al. 2016.
 * smpl_verts (B,NVs,3), smpl_faces (NFs,3) int32, obj_verts (B,NVo,3) = the TRANSFORMED template, obj_faces (NFo,3) int32, all device.
 * *term += mean over frames of the penetration of the human-object triangle pairs (first max_collisions colliding SMPL faces per object face);
 * d_obj_t (B,3) += gscale * d value / d(object translation)  (the parameter phase 'joint' optimises; NULL to skip);
 * pairs_per_frame (B) int32 device, optional.  workspace: vt_collision_workspace_bytes(B, NFs) bytes, 8-byte aligned.
 * ------------------------------------------------------------------------------------------------- */
long vt_collision_workspace_bytes(int B, int n_smpl_faces);
int vt_collision_loss(const float *smpl_verts, int n_smpl_verts, const int *smpl_faces, int n_smpl_faces, const float *obj_verts, int n_obj_verts,
                      const int *obj_faces, int n_obj_faces, int B, float sigma, int max_collisions, float gscale, double *term,
                      float *d_obj_t, int *pairs_per_frame, void *workspace, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Ragged contact Chamfer.  Replaces pytorch3d.loss.chamfer_distance(Pointclouds, Pointclouds)[0]
 * (recon/recon_fit_trivis_full.py:454-457; pytorch3d defaults, see DESIGN.md "unpinned").
 * x (nx_total,3), y (ny_total,3), offx/offy (P+1) int32 device.  *term += value.
 * dx, dy (+=, may be NULL) receive gscale * gradient.
 * ------------------------------------------------------------------------------------------------- */
int vt_chamfer_ragged(const float *x, const int *offx, const float *y, const int *offy, int P, float gscale,
                      double *term, float *dx, float *dy, void *stream);
/* The same term and gradients with every pair split over several workgroups (one workgroup per pair makes the largest contact sets the launch's critical
 * path): total_x / total_y = offx[P] / offy[P] (rows of x / y), ws = vt_chamfer_ws_bytes(total_x, total_y, P) bytes of device scratch.  Gradients are still
 * accumulated without float atomics (run-to-run reproducible); the far-side sums associate differently from vt_chamfer_ragged (last-bit differences). */
long vt_chamfer_ws_bytes(long total_x, long total_y, int P);
int vt_chamfer_ragged_ws(const float *x, const int *offx, long total_x, const float *y, const int *offy, long total_y, int P, float gscale,
                         double *term, float *dx, float *dy, void *ws, void *stream);

/* The same with the y clouds taken THROUGH AN INDEX LIST out of one big array (the contact term of phase 'joint', recon_fit_trivis_full.py:393-457: the
 * object-side contact points are rows idx_y[r] of the transformed surface samples (B * N, 3) of the batch): y row r = y_base[idx_y[r]], and the gradient of
 * row r is ADDED to dy_base[idx_y[r]] -- the additions of "index_select, vt_chamfer_ragged_ws on the compact list, index_add_" in their order (bit-identical
 * results), without the three gather / zero / scatter launches around it.  idx_y must not repeat a row (a surface point belongs to one (frame, part) pair).
 * run_plan = 0 reuses the work-item plan a previous call left in ws for the SAME offx / offy (the contact set is computed once per batch). */
int vt_chamfer_ragged_idx(const float *x, const int *offx, long total_x, const float *y_base, const int *idx_y, const int *offy, long total_y, int P,
                          float gscale, double *term, float *dy_base, void *ws, int run_plan, void *stream);

/* Evaluation Chamfer, one direction (recon/eval/chamfer_distance.py:10-52, sklearn NearestNeighbors(k=1, metric='l2')): for each of
 * the nq points of cloud pair p the Euclidean distance to the nearest of the ns points of `search`; query (P,nq,3), search (P,ns,3),
 * dist (P,nq).  The bidirectional Chamfer of the reference is mean(dist(x->y)) + mean(dist(y->x)). */
int vt_nn_distance(const float *query, int nq, const float *search, int ns, int P, float *dist, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Silhouette.  Replaces neural_renderer.Renderer(K, R=I, t=0, orig_size=1, anti_aliasing=False)(verts, faces,
 * mode='silhouettes') and its backward as used by SilLossROI.forward (recon/obj_pose_roi.py:77-94,183-207).
 * verts (B,NV,3) camera space, faces (NF,3) int32 shared, K (B,9), image (B,size,size) row 0 = top.
 * face_index (B,size,size) int32 (face id in [0,2*NF) of the doubled fill_back list, -1 = background) and the workspace
 * `ws` (vt_sil_workspace_floats(B,NV,NF,size) floats: depth keys, sweep masks, projected vertices, face corners, pixel boxes,
 * visibility flags) are
 * written by the forward and read again by the backward.
 * ------------------------------------------------------------------------------------------------- */
long vt_sil_workspace_floats(int B, int NV, int NF, int size);
int vt_sil_forward(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int size,
                   float *image, int *face_index, float *ws, void *stream);
/* d_image (B,size,size) -> dverts (B,NV,3) (overwritten). eps = NMR's 1e-4. */
int vt_sil_backward(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int size,
                    const int *face_index, const float *d_image, float eps, float *ws, float *dverts, void *stream);
/* SilLossROI's per-batch set-up on the device (recon/obj_pose_roi.py:39-75 __init__; :111-121 to_original_bbox; :123-155 compute_K_roi; :157-181 cvt_masks;
 * recon/bbox.py:26-48 make_bbox_square; recon/opt_utils.py:148-153 mask2bbox).  mask_h / mask_o (B,H,W) person / object masks of the network input in [0,1];
 * crop_centers (B,2) full-image pixels; expansion 0.3; out = render size (256); crop_size 1200, net_size 512; cam_norm = {fx, fy, cx, cy} / image_width (host
 * doubles).  Writes image_ref, keep_mask (B,out,out) and K (B,9) = normalised intrinsics of the square ROI; boxes_ws: 4 B doubles of scratch.  An empty
 * object mask gives the reference's degenerate box and all-zero crops (image_ref = 0, keep_mask = 1).  The ROIAlign underneath (detectron2
 * BitMasks.crop_and_resize: aligned, adaptive sampling ratio, >= 0.5) is third-party: PARITY UNPINNED beyond the restatement in silhouette.py. */
int vt_sil_setup(const float *mask_h, const float *mask_o, int B, int H, int W, const float *crop_centers, double expansion, int out, double crop_size,
                 double net_size, const double *cam_norm, double image_width, float *image_ref, float *keep_mask, float *K, double *boxes_ws, void *stream);

/* fused occlusion-aware mask term (obj_pose_roi.py:191-198; recon_fit_trivis_full.py:164-168):
 * per[b] = sum_px (keep*sil - ref)^2 ; *term += mean_b(per[b]*occ[b]); d_image = gscale * d/d sil */
int vt_sil_mask_loss(const float *image, const float *keep, const float *ref, const float *occ, int B, int size,
                     float gscale, double *term, float *per_frame, float *d_image, void *stream);

/* The silhouette term of ONE Adam step of phase 'sil' (recon_fit_trivis_full.py:164-168, 329-375; SilLossROI.forward, obj_pose_roi.py:183-207, and its backward):
 * vt_sil_forward + vt_sil_mask_loss + vt_sil_backward as 5 launches instead of 10 -- the same arithmetic, operation for operation (owner map, d_image, vertex
 * gradients bit-identical; *term += mean_b(sum_px (keep sil - ref)^2 occ[b]) accumulated in 2^-34 fixed point, frame order).  face_index, d_image: (B,size,size)
 * outputs; dverts (B,NV,3); ws: vt_sil_workspace_floats(B,NV,NF,size) floats. */
int vt_sil_step(const float *verts, int B, int NV, const int *faces, int NF, const float *K, int size, const float *keep, const float *ref,
                const float *occ, float gscale, float eps, double *term, int *face_index, float *d_image, float *ws, float *dverts, void *stream);

/* Triplane masks of the SMPL mesh.  Replaces TriplaneNrRenderer.render_3views
 * (render/render_triplane_nr.py:86-139): the mesh centred on `center` (B,3) (the SMPL centre, body25 joint 8) is rendered
 * orthographically from the right (x' = z, y' = -y, depth -x + 10), the back (-x, -y, -z + 10) and the top (x, z, y + 10);
 * mask = a face covers the pixel with 0.1 < depth < 100.  masks (B,3,size,size) in {0,1}, row 0 = top; face_index (B,3,size,size)
 * int32 scratch output (front face id or -1); ws >= vt_sil_workspace_floats(3 B, NV, NF, size) floats.  PARITY UNPINNED for the
 * rasterisation rule (neural_renderer, see vt_sil_forward); the view transforms are pinned (tests/golden/triplane_views.npz). */
int vt_triplane_render(const float *verts, const float *center, int B, int NV, const int *faces, int NF, int size,
                       float *masks, int *face_index, float *ws, void *stream);

/* Hourglass up-path of the image encoder: out = skip + bicubic x2 upsampling of low, NHWC fp32.  Replaces
 * `up1 + F.interpolate(low3, scale_factor=2, mode='bicubic', align_corners=True)` (model/HGFilters.py:45-47): cubic convolution with
 * A = -0.75, source coordinate dst * (h-1)/(2h-1), taps clamped to the border.  low (B,h,w,C), skip (B,2h,2w,C) or NULL, out (B,2h,2w,C);
 * C must be a multiple of 4.  (The convolutions and group norms of the encoder run on MIOpen through PyTorch.) */
int vt_upsample2x_bicubic_add(const float *low, const float *skip, int B, int h, int w, int C, float *out, void *stream);
/* the same arithmetic + the GroupNorm partial sums of the output (vt_sweep_blocks(4 h w) blocks, see vt_avgpool2x2_stats) */
int vt_upsample2x_bicubic_add_stats(const float *low, const float *skip, int B, int h, int w, int C, float *out, double *stats_ws, int stats_groups,
                                    void *stream);

/* GroupNorm (+ ReLU) of an NHWC fp32 tensor, the `bnK -> F.relu` prologue of every pre-activated convolution of the encoder
 * (model/net_util.py:374-388, model/HGFilters.py:176,192-193): y = [relu]((x - mean_g) / sqrt(var_g + eps) * gamma + beta) with the
 * biased variance over (H, W, C/groups), exactly torch.nn.functional.group_norm.  x, y (B,HW,C); gamma, beta (C); ws >= vt_groupnorm_workspace_doubles(B, HW, C, groups) doubles
 * ((B, groups) float pairs {mean, rstd} first, then per-block fp64 partial sums: no atomics, deterministic). */
long vt_groupnorm_workspace_doubles(int B, int HW, int C, int groups);
int vt_groupnorm_nhwc(const float *x, const float *gamma, const float *beta, int B, int HW, int C, int groups, float eps,
                      int relu, double *ws, float *y, void *stream);

/* ---------------------------------------------------------------------------------------------------
 * Adam.  Replaces torch.optim.Adam(...).step() (defaults betas=(.9,.999), eps=1e-8) on one parameter tensor.
 * `stop_flag` (device int, may be NULL): when *stop_flag != 0 the update is skipped (device-side early stop).
 * ------------------------------------------------------------------------------------------------- */
int vt_adam_step(float *p, const float *g, float *m, float *v, long n, int step, float lr, float beta1, float beta2,
                 float eps, const int *stop_flag, void *stream);

/* the same update on the first `cols` columns of row-strided p/g (m, v are (rows, cols) contiguous): lets one pose
 * tensor (B,156) be optimised on its [0:66) slice, i.e. global_pose + body_pose of SMPLPyTorchWrapperBatchSplitParams */
int vt_adam_step_2d(float *p, long p_stride, const float *g, long g_stride, float *m, float *v, int rows, int cols,
                    int step, float lr, float beta1, float beta2, float eps, const int *stop_flag, void *stream);

/* Early-stop rule of the fit loops evaluated on the device (recon_fit_behave.py:447-455,
 * recon_fit_trivis_full.py:372-373, fit_SMPLH_kpts.py:161):
 *   loss = sum_k w[k]*terms[k];  if (armed && |prev-loss|/prev < prev*tol) *stop_flag = 1;  prev = loss;
 * terms (nterms) device, w (host, nterms), state (device, 2 floats: prev_loss, last_loss), history (device) or NULL
 * receives loss at index `slot`. */
int vt_loss_reduce_and_stop(const double *terms, const float *w, int nterms, float tol, int armed, float *state,
                            int *stop_flag, float *history, int slot, void *stream);

/* small utilities */
int vt_fill(float *p, long n, float value, void *stream);

/* Device-side early stop.  The fits evaluate their stop rules on the device (the reference breaks out of its inner loop on the host:
 * recon_fit_behave.py:447, recon_fit_trivis_full.py:372, fit_SMPLH_kpts.py:161) and the host reads the flag once per outer iteration of 10 steps; with
 * `flag` registered for `stream` the query and SMPL-H launches queued behind the stopping step return at once when *flag != 0 (the Adam / loss-history
 * launches ignore those steps already).  flag = NULL removes the registration; the owner must remove it before the flag's memory is released.
 * The registration belongs to the CALLING HOST THREAD: only launches issued by that thread on `stream` see the flag (two fits driven by two threads
 * through one stream do not interfere); registering a second, different flag for the same stream from the same thread is refused with VT_ERR_BUSY. */
int vt_stream_set_skip_flag(void *stream, const int *flag);

/* ---- bookkeeping of one round of the surface-point generator (recon/gen/generator.py:149-212, Generator.gen_pc_batch) ----------------------------------
 * vt_gen_round_compact: surface (B,S,3) projected samples, df_target (B,S) clamped distance of the last query, pre (B,S,3) positions of that query (or NULL),
 *   active (B) bytes.  A sample is kept when df_target < filter_val and z > zmin and its frame is active (generator.py:160-166).  order (B,S) int32 receives the
 *   kept sample indices in sample order (first cnt[b] entries: torch.argsort(~mask, stable=True)'s prefix); kept_pre (B,S,3) (or NULL) the positions `pre` at
 *   them; with write != 0 the kept points are appended to buf_points (B,cap+1,3) at row fill[b] (rows >= cap dropped: the reference cuts every frame to the common
 *   count anyway); cnt (B), fill_out (B) = min(fill + cnt, cap) (= fill without write) as int64.
 * vt_gen_scatter_heads: pred (B,C,kmax) predictions at the kept points -> buf (B,cap+1,C) rows fill_old[b] + k, k < cnt[b].
 * vt_gen_resample: the next round's M samples per frame (generator.py:190-210): samples[order[floor(u * cnt)]] + near_scale * pert where cnt > 1, else a
 *   restart init[floor(u * S0)] + 0.5 * pert; u (B,M) uniform, pert (B,M,3) normal (drawn by the caller's seeded generator). */
int vt_gen_round_compact(const float *surface, const float *df_target, const float *pre, const unsigned char *active, int B, int S, float filter_val,
                         float zmin, const long long *fill, int cap, int write, float *buf_points, int *order, float *kept_pre, long long *cnt,
                         long long *fill_out, void *stream);
int vt_gen_scatter_heads(const float *pred, int B, int C, int kmax, const long long *fill_old, const long long *cnt, int cap, float *buf, void *stream);
int vt_gen_resample(const float *samples, const int *order, const long long *cnt, const float *init, int B, int S, int S0, const float *u,
                    const float *pert, int M, float near_scale, float *out, void *stream);

/* ---- box calibration (measurement infrastructure of bench.py; no counterpart in the reference, which times whole processes: README.md:55) ------------------
 * Two fixed micro-kernels exercising the resources the dominant kernel of the fit is limited by: out[0] = dense f16 MFMA TFLOP/s (v_mfma_f32_16x16x32_f16, two
 * workgroups of 256 threads per CU, non-trivial operands), out[1] = shader clock sustained during it (MHz: s_memtime against the 100 MHz s_memrealtime),
 * out[2] = L2 -> register delivery of lane-linear 16-byte loads (TB/s), out[3], out[4] = their durations in ms.  `work`: vt_calibrate_workspace_bytes() bytes
 * of device memory.  Synchronises `stream`; ~25 ms. */
/* clock probe of the fused-objective query kernels (measurement only): while `counters` (3 device-side 64-bit words, zeroed by the caller) is set, every 1024th
 * workgroup of every vt_query_* launch adds its life time in shader clocks, in 100 MHz ticks, and 1 -- counters[0] / counters[1] x 100 = the shader clock in MHz the
 * chip sustained DURING those launches.  NULL switches it off (the default). */
int vt_query_set_clock_probe(unsigned long long *counters);
long vt_calibrate_workspace_bytes(void);
int vt_calibrate(void *work, double *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VISTRACKER_H */
