/*
 * diffus_hip.h -- C ABI of libdiffus_hip.so, the MI355X (gfx950) implementation
 * of the DiffUS `UltrasoundRenderer.plot_beam_frame` hot path.
 *
 * The reference (gduguey/DiffUS) is pure Python and has no FFI of its own; the
 * boundary it offers is the Python call surface used by its notebooks
 * (reference src/renderer.py:19, :201-217, :275; src/cone.py:242).  This header
 * is the C-ABI drop-in underneath that surface: each entry point names the
 * reference function(s) it replaces.  A reference maintainer binds it with
 * ctypes (INTEGRATION.md shows the stub); diffus_amd/renderer.py is such a
 * binding, mirroring the reference class.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch tensor storage),
 *     except where stated; tensors are dense row-major;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     all work is enqueued on it, nothing synchronises, nothing allocates, so
 *     every call is legal inside hipGraph stream capture;
 *   - return value: 0 on success, a negative DIFFUS_E* code otherwise; nothing
 *     throws; diffus_strerror() explains a code;
 *   - volume layout: (d0,d1,d2) float32, point coordinate c indexes dim c
 *     (reference src/renderer.py:750-758), dim 2 contiguous;
 *   - a *pose* is one probe position: `src` (3) + `dirs` (R,3); P poses are
 *     batched as src (P,3), dirs (P,R,3);  S = num_samples; `start` is the
 *     resolved integer crop (reference :237-240), 0 <= start <= S-1;
 *     N1 = S - start samples per ray are produced.
 */
#ifndef DIFFUS_HIP_H
#define DIFFUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIFFUS_ABI_VERSION 8 /* 2: 40-float PAIRED records + one-pass step workspace (round 2); 3: round-3 entry points; 4: strided y / gy in diffus_mlp_fwd / _bwd; 5: winner raster kept between diffus_splat_fwd / _bwd; 6: diffus_convert_volume_box; 7: DIFFUS_FANS_PLANAR / DIFFUS_BWD_REPAIR_FRAME flag bits, a per-ray flag array in the workspace (diffus_workspace_bytes grows); 8: diffus_fan_pose_fwd / _bwd */

/* error codes */
#define DIFFUS_OK            0
#define DIFFUS_EINVAL      (-1) /* bad argument (null pointer, non-positive size, bad enum) */
#define DIFFUS_EUNSUPPORTED (-2) /* shape outside what the kernels cover (S - start > MAX_SAMPLES * MAX_SEGMENTS, ...) */
#define DIFFUS_ELAUNCH     (-3) /* HIP reported a launch error (hipGetLastError) */
#define DIFFUS_EWORKSPACE  (-4) /* workspace too small, see diffus_workspace_bytes */

/* element type of `src` / `dirs`: the reference evaluates source + k*dir with
 * torch's promotion rules, so f64 inputs change which roundings happen
 * (reference src/renderer.py:119-124, :751; oracle/diffus_oracle.c orc_point) */
#define DIFFUS_F32 0
#define DIFFUS_F64 1
#define DIFFUS_I64 2 /* diffus_splat_axes only: the int64 index planes plot_beam_frame returns */

/* sampler */
#define DIFFUS_NEAREST   0 /* reference custom_nearest_sampler, src/renderer.py:741-759 */
#define DIFFUS_TRILINEAR 1 /* grid_sample(bilinear, border, align_corners=True) semantics;
                              the only mode with a pose gradient */

/* volume layout (argument `layout` of the calls below; applies to `vol` and, in
 * the backward, to `gvol`) */
#define DIFFUS_CANONICAL 0 /* the caller's (d0,d1,d2) row-major tensor */
#define DIFFUS_BRICKED   1 /* 4x4x2-voxel bricks of 32 floats (one 128-B line), row-major over
                              (ceil(d0/4), ceil(d1/4), ceil(d2/2)); made by diffus_brick_volume */
#define DIFFUS_PAIRED    2 /* volume only: one 160-B record per (4x4 column block, depth z): 4 rows of dim 0 x 5
                              columns of dim 1 (the fifth repeats the next block's first, clamped at the edge) x
                              the pair (v[z], v[min(z+1,d2-1)]), so that a trilinear sample is two 16-byte loads;
                              made by diffus_pair_volume; 2.5x the memory; the gradient that goes with it (gvol)
                              is DIFFUS_BRICKED */
#define DIFFUS_GRAD_BRICKED 0x10 /* backward entry points only, OR'ed into `layout`: `gvol` / `gvol_touched` are the
                              DIFFUS_BRICKED scratch and its flags whatever the layout of `vol` -- a DIFFUS_CANONICAL
                              volume (one the caller updates in place, e.g. a slice per training step) then gets the
                              sparse, never-memset hand-back (diffus_gradbuf_flush) too.  ALSO the more accurate gradient: bricked
                              scratch tiles accumulate planar and tilted fans in doubles (per-voxel bound 4e-6 of the voxel's mass),
                              a canonical `gvol` without this flag goes through a 3-D tile in 32-bit fixed point with one scale per
                              patch of 32 rays x 32 steps (<= 2^-20 of the patch's sum of |contributions| per add: seen 7e-3 of
                              max |gvol| on 0.04-voxel steps whose contributions cancel).  The Python mirror sets it by default. */

#define DIFFUS_FANS_PLANAR 0x20 /* backward entry points only, OR'ed into `layout`: a HINT that no ray of the call moves along
                              dim 2 (every fan of the reference: src/cone.py:258 writes a zero dim-2 component).  The volume
                              scatter then runs its launch for planar fans alone (24 KiB tile, 6 blocks per CU).  Without the
                              hint it runs the launch that also carries the slab path for fans that leave the slice
                              (`directions` may be anything: src/renderer.py:119-124, :201-217) -- 36 KiB tile, 4 blocks per
                              CU, ~10 % slower on planar fans.  A wrong hint costs time, never correctness: oblique patches of
                              a hinted call take the general 3-D tile */

/* one wavefront marches one ray; a lane owns ceil(N1/64) consecutive samples, at most 16, so one launch
 * covers 1024 cropped samples.  Longer rays (N1 = S - start up to MAX_SAMPLES * MAX_SEGMENTS) are processed
 * as segments of 1024 samples, one launch each, chained through per-ray carries in the workspace (the running
 * transfer-matrix product forward, the adjoint matrix backward). */
#define DIFFUS_MAX_SAMPLES 1024
#define DIFFUS_MAX_SEGMENTS 64

typedef void *diffus_stream_t;

int diffus_abi_version(void);
const char *diffus_strerror(int code);

/* Scratch the calls below need for a given problem (bytes; 256-B aligned device
 * memory supplied by the caller, reusable across calls on one stream). */
size_t diffus_workspace_bytes(int P, int R, int S, int start);
/* Byte offset, inside that workspace, of zbar (P,R,S-start) float32 = dL/d(impedance sample): what DIFFUS_BWD_SCAN
 * leaves for DIFFUS_BWD_SCATTER (see diffus_render_bwd).  A caller that runs the two stages as separate calls may read
 * it (per-sample impedance gradients) or add its own terms to it in between. */
size_t diffus_workspace_zbar_offset(int P, int R, int S, int start);

/*
 * Data layout in HBM.  The reference indexes a dense (d0,d1,d2) tensor with dim 2
 * contiguous (src/renderer.py:750-758) while every demo fan lies in a plane of
 * constant dim 2 (src/cone.py:258) -- the worst case for that layout.  The
 * kernels therefore also accept a bricked copy (DIFFUS_BRICKED).  These two calls
 * convert; both are coalesced streaming passes over the volume.
 *   diffus_paired_floats   number of floats of the paired buffer (about 2.5*d0*d1*d2)
 *   diffus_pair_volume     canonical -> paired
 *   diffus_bricked_floats  number of floats of the bricked buffer (>= d0*d1*d2)
 *   diffus_brick_volume    canonical -> bricked (padding voxels are written as 0)
 *   diffus_unbrick_volume  bricked -> canonical; accumulate != 0 adds instead of
 *                          storing (used to fold a bricked gradient into a
 *                          canonical one)
 *   diffus_convert_volume_box   canonical -> `layout` (DIFFUS_BRICKED or DIFFUS_PAIRED) for every record / brick that
 *                          holds a voxel of the box [x0,x1) x [y0,y1) x [z0,z1) -- what a caller runs after it
 *                          changed only that part of the canonical volume (the reference's training loop rewrites ONE
 *                          slice per step: notebooks/[DEMO] Train MRI to Impedance MLP - GPU.ipynb cell 16), instead of
 *                          a pass over the whole volume.  Whole blocks of the conversion kernels are re-converted, i.e.
 *                          somewhat more than the box; an empty box is a no-op.
 */
/*
 * Sparse gradient hand-back.  diffus_gradbuf_flush visits only the bricks whose flag is set:
 * adds (mode ACCUMULATE) or stores them into the canonical (d0,d1,d2) tensor `vol`, zeroes them
 * in `bricked` and clears the flags -- so a buffer pair that starts all-zero is all-zero again
 * after the flush, and neither a 64 MiB memset nor a dense conversion is needed per step.
 * With mode STORE only touched voxels are written: zero `vol` first if it must be dense.
 * Mode PERSISTENT is for a gradient tensor the caller keeps across steps (all-zero before the first one,
 * written by nothing but this call, always with the same `touched` array): bricks stored by the previous
 * call and not touched since are zeroed in `vol`, so `vol` always equals the dense gradient of the latest
 * step and is never memset.  (`touched` then holds 2 for the bricks `vol` currently has values in.)
 * Mode DENSE writes EVERY voxel of `vol`: the touched bricks' values and zeros everywhere else -- a fresh dense
 * gradient in one launch (the drop-in autograd path: no torch.zeros + flush, i.e. one launch and one dispatch fewer).
 * `bricked` must be 16-byte aligned (DIFFUS_EINVAL otherwise; any hipMalloc / torch allocation is); `vol` may sit at
 * any float offset (8-byte aligned and d2 even: a z pair moves as one word).
 */
#define DIFFUS_FLUSH_STORE      0
#define DIFFUS_FLUSH_ACCUMULATE 1
#define DIFFUS_FLUSH_PERSISTENT 2
#define DIFFUS_FLUSH_DENSE      3
size_t diffus_brick_count(int d0, int d1, int d2);
int diffus_gradbuf_flush(float *bricked, int *touched, int d0, int d1, int d2, float *vol,
                         int mode, diffus_stream_t stream);

size_t diffus_paired_floats(int d0, int d1, int d2);
int diffus_pair_volume(const float *vol, int d0, int d1, int d2, float *paired,
                       diffus_stream_t stream);
size_t diffus_bricked_floats(int d0, int d1, int d2);
int diffus_brick_volume(const float *vol, int d0, int d1, int d2, float *bricked,
                        diffus_stream_t stream);
int diffus_unbrick_volume(const float *bricked, int d0, int d1, int d2, float *vol,
                          int accumulate, diffus_stream_t stream);
int diffus_convert_volume_box(const float *vol, int d0, int d1, int d2, int layout, float *converted,
                              int x0, int x1, int y0, int y1, int z0, int z1, diffus_stream_t stream);

/*
 * Forward: replaces UltrasoundRenderer.plot_beam_frame with artifacts=False
 * (reference src/renderer.py:201-275) and everything under it: trace_ray
 * (:90-180), custom_nearest_sampler (:741-759), compute_reflection_coeff
 * (:27-33), the start crop + median (:241-244), compute_echo_traces /
 * propagate_full_rays_batched / prop_single_ray (:367-457, evaluated as an O(N)
 * running product of 2x2 transfer matrices instead of N+1 dense solves) and the
 * attenuation (:256-259) -- for P poses at once.
 *
 *   frame  out (P,R,N1) float32   processed_output of the reference
 *   idx    out, nullable: (3,P,R,N1) int64 -- the x,y,z index planes the
 *          reference returns, already cropped to [:, start:]
 */
int diffus_render_fwd(const float *vol, int d0, int d1, int d2, int layout,
                      const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype,
                      int P, int R, int S, int start, float alpha, int sampler,
                      float *frame, int64_t *idx,
                      void *workspace, size_t workspace_bytes,
                      diffus_stream_t stream);

/*
 * Backward of diffus_render_fwd: what torch autograd does through the
 * reference's sub-functions custom_nearest_sampler (reference
 * src/renderer.py:741-759) -> compute_reflection_coeff (:27-33) ->
 * compute_echo_traces (:439-457) -> attenuation (:256-259), see SURVEY.md §3.2 /
 * App. A.4, without storing any dense system: given gframe = dL/dframe (P,R,N1) it recomputes the forward
 * per ray and produces any of
 *   gvol   nullable, float32 in the layout that goes with vol's ((d0,d1,d2) canonical for a
 *          canonical vol, diffus_bricked_floats() bricked for a bricked or paired vol),
 *          ACCUMULATED with float atomics
 *          (caller zeroes it; shared by all poses)
 *   gvol_touched  nullable, only with a bricked gvol: diffus_brick_count() ints; the
 *          entry of every brick this call adds into is set to 1 (plain stores).
 *          Together with diffus_gradbuf_flush it makes the gradient SPARSE: keep
 *          gvol and gvol_touched all-zero between steps, flush after each backward.
 *   gsrc   nullable (P,3) float32, overwritten   (trilinear only, else zeros)
 *   gdirs  nullable (P,R,3) float32, overwritten (trilinear only, else zeros)
 * With start > 0 the median written into column 0 (reference :243-244) routes
 * its gradient to the ray that supplied the median, like torch.median.
 *
 * `stages` selects which launches run: DIFFUS_BWD_SCAN (the per-ray adjoint
 * scan: pose gradients, and d L/d impedance per sample into the workspace),
 * DIFFUS_BWD_SCATTER (workspace -> gvol through LDS tiles), or DIFFUS_BWD_ALL.
 * Running them as two calls with the same workspace equals one ALL call; it
 * exists so the scatter can be timed / overlapped on its own.
 * DIFFUS_BWD_KEEP_MEDIAN (start > 0 only, OR-ed into `stages`): the workspace still holds the per-pose median
 * (reference src/renderer.py:243) that diffus_render_fwd left there for the SAME problem -- nothing else has used the
 * workspace since -- so the backward does not recompute it.  A start > 0 backward is then the same two launches as a
 * start = 0 one: the median's gradient is routed to the ray that supplied it, and d/dsource reduced, by extra
 * blocks of the scatter launch.
 */
#define DIFFUS_BWD_SCAN    1
#define DIFFUS_BWD_SCATTER 2
#define DIFFUS_BWD_ALL     3
#define DIFFUS_BWD_KEEP_MEDIAN 4
#define DIFFUS_BWD_REPAIR_FRAME 8 /* diffus_render_step_mse only, OR'ed into `stages`: rays whose echo series is ILL-CONDITIONED
                                     (|echo| > 1 somewhere: a ray grazing the skull, echo = b/d with d nearly cancelled -- 63 rays of
                                     8192 at BASELINE config 3) get their frame row and loss term evaluated again in float64, from
                                     float32 samples taken with the reference's own lerp sequence (reference src/renderer.py:33,
                                     :407-457; golden G19).  Every float32 evaluation -- the reference's dense LU included -- carries
                                     (condition number) x eps of noise there; diffus_render_fwd and diffus_echo_traces do this repair
                                     inside their kernels.  Here it costs a launch of the per-pose epilogue after the scatter
                                     (~15 us per step), so it is opt-in: the one-pass step's frame is a by-product of a training
                                     step, and the gradients carry the same condition number whatever the arithmetic */
int diffus_render_bwd(const float *vol, int d0, int d1, int d2, int layout,
                      const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype,
                      int P, int R, int S, int start, float alpha, int sampler,
                      const float *gframe,
                      float *gvol, int *gvol_touched, float *gsrc, float *gdirs,
                      int stages,
                      void *workspace, size_t workspace_bytes,
                      diffus_stream_t stream);

/*
 * The same backward for the loss  L_p = loss_scale * sum over the frame of pose p of (frame - target)^2  (target NULL:
 * zeros, i.e. the frame's energy), fused: `frame` is what diffus_render_fwd produced, dL/dframe = 2 loss_scale
 * (frame - target) is formed on the fly while the row is loaded, and `loss` (nullable, (P) float32, overwritten) receives
 * L_p, summed over rays in a fixed order by the per-pose blocks that end the call.  It replaces a separate loss kernel
 * (one more pass over the frame, a (P,R,N1) gradient buffer, one more launch) in loops whose loss has this form -- the
 * sum of squares bench.py and the examples use, an MSE against a target frame.  Everything else as diffus_render_bwd.
 */
int diffus_render_bwd_mse(const float *vol, int d0, int d1, int d2, int layout,
                          const void *src, int src_dtype,
                          const void *dirs, int dirs_dtype,
                          int P, int R, int S, int start, float alpha, int sampler,
                          const float *frame, const float *target, float loss_scale, float *loss,
                          float *gvol, int *gvol_touched, float *gsrc, float *gdirs,
                          int stages,
                          void *workspace, size_t workspace_bytes,
                          diffus_stream_t stream);

/*
 * Forward, that loss and its backward in ONE pass: the backward recomputes the forward per ray anyway (nothing is
 * saved between the two), so for the fused loss above the frame itself can come out of the adjoint-scan kernel -- the
 * arithmetic of diffus_render_fwd, equal to it up to the rounding of the scan's association (chunk length, two waves
 * per ray) -- and the separate forward launch, a second gather of every sample, disappears.  `frame` (nullable, (P,R,N1) float32) receives plot_beam_frame's
 * processed_output (reference src/renderer.py:201-275, artifacts off), `loss` (nullable, (P)) the per-pose loss, the
 * gradients and `stages` as in diffus_render_bwd (the frame and the loss belong to the SCAN stage).  Rays longer than
 * one launch (N1 > 1024) run as diffus_render_fwd + diffus_render_bwd_mse inside the call (`frame` then required).
 */
int diffus_render_step_mse(const float *vol, int d0, int d1, int d2, int layout,
                           const void *src, int src_dtype,
                           const void *dirs, int dirs_dtype,
                           int P, int R, int S, int start, float alpha, int sampler,
                           const float *target, float loss_scale,
                           float *frame, float *loss,
                           float *gvol, int *gvol_touched, float *gsrc, float *gdirs, int stages,
                           void *workspace, size_t workspace_bytes,
                           diffus_stream_t stream);

/*
 * Stage 1 alone: replaces UltrasoundRenderer.trace_ray / simulate_rays
 * (reference src/renderer.py:90-180, :35-71) = custom_nearest_sampler
 * (:741-759) + compute_reflection_coeff (:27-33).  Any of the outputs may be
 * NULL.  No crop: all S samples.
 *   imp   (P,R,S)   float32  sampled impedances (trace_ray's ray_values)
 *   refl  (P,R,S-1) float32  reflection coefficients (simulate_rays' R)
 *   idx   (3,P,R,S) int64    rounded, clamped voxel indices x,y,z
 */
int diffus_trace_rays(const float *vol, int d0, int d1, int d2, int layout,
                      const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype,
                      int P, int R, int S, int sampler,
                      float *imp, float *refl, int64_t *idx,
                      diffus_stream_t stream);

/*
 * Stage 2 alone: replaces compute_echo_traces' first return value (reference
 * src/renderer.py:439-457, i.e. propagate_full_rays_batched :412-436 and
 * prop_single_ray :367-410): refl (B,N) -> echo (B,N+1), echo[:,0] = 0.
 * Any N: rows longer than 1024 samples are walked in pieces by the same wave.
 */
int diffus_echo_traces(const float *refl, int B, int N, float *echo,
                       diffus_stream_t stream);

/*
 * Backward of stage 2: what torch autograd computes through the reference's compute_echo_traces (the N+1
 * LinalgSolveBackward nodes behind src/renderer.py:407, the stack/cumsum/diff of :434-435,454): refl (B,N),
 * gecho (B,N+1) = dL/d echo -> grefl (B,N) = dL/d refl, by the O(N) adjoint sweep, float64
 * inside.  An echo that nan_to_num (:408) turned into the constant 0 passes no gradient.  `workspace`: device
 * scratch of diffus_echo_bwd_workspace_bytes(B, N) bytes (the forward's running products).
 */
size_t diffus_echo_bwd_workspace_bytes(int B, int N);
int diffus_echo_traces_bwd(const float *refl, int B, int N, const float *gecho, float *grefl,
                           void *workspace, size_t workspace_bytes, diffus_stream_t stream);

/*
 * The remaining module functions `from src.renderer import *` gives the notebooks:
 *   diffus_prop_single_ray  replaces prop_single_ray (reference src/renderer.py:367-410): refl (B,N), float32 or
 *                           float64 like the caller's tensor -> w (B, 2N+2) = [g0,d0,...,gN,dN], the full solution of
 *                           the dense system, in closed form (O(N) per ray); a ray with a non-finite coefficient gives
 *                           all zeros, as linalg.solve + nan_to_num (:407-408) do.
 *   diffus_propagate_rays   replaces propagate_full_rays_batched (:412-436): refl (B,N) -> (B,N+1), the surface
 *                           return d0 per truncation depth, cumulated along the depth (:435).
 *   diffus_sample_points    replaces custom_nearest_sampler (:741-759) for ARBITRARY points (n,3) float32 (the
 *                           reference casts to float32 at :751): values (n) and, nullable, idx (3,n) int64.
 */
int diffus_prop_single_ray(const void *refl, int dtype, int B, int N, void *w, diffus_stream_t stream);
int diffus_propagate_rays(const float *refl, int B, int N, float *d0_cum, diffus_stream_t stream);
int diffus_sample_points(const float *vol, int d0, int d1, int d2, int layout, const float *points, long n, int sampler,
                         float *values, int64_t *idx, diffus_stream_t stream);

/*
 * The pulse stage of compute_gaussian_pulse (reference src/renderer.py:459-479, the F.conv1d at :477): every row of
 * `in` (B,N) correlated with `kernel` (L taps, no flip) with `pad` zeros on both sides -> out (B, N + 2*pad - L + 1).
 */
int diffus_rows_conv1d(const float *in, int B, int N, const float *kernel, int L, int pad, float *out,
                       diffus_stream_t stream);

/*
 * Scan conversion (the step after the path): replaces
 * differentiable_splat (reference src/renderer.py:694-737) for P frames at once.
 *   c0, c1  (P,n) float32: the two plotted coordinates of every sample (the
 *           caller picks the two highest-variance axes like :702-710 and casts)
 *   val     (P,n) float32 intensities
 *   cols    0, or the row length when the n samples of a pose form a (n/cols, cols)
 *           grid of rays x steps (a performance hint only: results are identical)
 *   out     (P,W,H) float32: per pose the (W,H) image the reference returns
 * Pixel = clamp(round_half_even(coord), 0, size-1) (:717-718); of the samples
 * landing on one pixel the LAST in flattened order wins (index_put without
 * accumulation, :721-722); weight is 1 where any landed; both planes are blurred
 * with the normalised Gaussian of size int(6 sigma)|1 (zero padding, :725-734) and
 * divided with eps 1e-8 (:735); the result is transposed (:737).
 * diffus_splat_bwd: gout (P,W,H) -> gval (P,n), every sample receiving the
 * gradient of its pixel, exactly as torch autograd does through the reference.
 * sigma <= 8 (kernel half-width <= 24).
 */
/*
 * The loss of the reference's training notebook (notebooks/[DEMO] Train MRI to Impedance MLP - GPU.ipynb cell 16,
 * `UltrasoundSynthesisModel.loss`), which attaches to differentiable_splat's image:
 *     synth = (img - img.min()) / (img.max() - img.min() + 1e-8)        [normalise != 0]
 *     loss  = 1 - ssim(synth, ref)
 * with the SSIM of Wang et al. (IEEE TIP 2004) as piq.ssim(x, y, data_range=1.0) evaluates it: a win x win Gaussian
 * window of `sigma` ('valid' correlation), constants (k1, k2), mean over the map -- piq's defaults are win 11, sigma
 * 1.5, k1 0.01, k2 0.03.  img, ref: (H,W) float32; loss: device scalar.  diffus_ssim_loss_bwd: gloss (device scalar,
 * nullable = 1) -> gimg (H,W) = dL/dimg, including what img.min() / img.max() hand back to the pixels that attain them
 * (shared equally among ties, like torch).  reuse_stats != 0: `workspace` is the forward's, untouched since, and `img` is
 * unchanged -- the minimum / maximum pass is skipped.  win odd, <= 15.
 */
size_t diffus_ssim_workspace_bytes(int H, int W, int win);
int diffus_ssim_loss_fwd(const float *img, const float *ref, int H, int W, int normalise, int win, float sigma, float k1,
                         float k2, float *loss, void *workspace, size_t workspace_bytes, diffus_stream_t stream);
int diffus_ssim_loss_bwd(const float *img, const float *ref, int H, int W, int normalise, int win, float sigma, float k1,
                         float k2, const float *gloss, float *gimg, int reuse_stats, void *workspace,
                         size_t workspace_bytes, diffus_stream_t stream);

/*
 * rotate_around_apex (reference src/renderer.py:655-692): x, z (n float32 each) -> x_rot, z_rot: every point
 * (x - shift, z) -- shift is the reference's hard-wired 128 -- turned by the angle between (0, 1) and `median`, then
 * moved by `apex`.  apex, median: DEVICE pointers to two floats each, so that the call needs nothing from the host.
 */
int diffus_rotate_around_apex(const float *x, const float *z, long n, const float *apex, const float *median, float shift,
                              float *x_rot, float *z_rot, diffus_stream_t stream);

/*
 * The probe-pose parameterisation, both ways.  Forward: P fans of n_rays unit directions each,
 *     dirs[p][i] = R(rotvec[p]) (cos(median[p] + a_i), sin(median[p] + a_i), 0),  a_i = opening * linspace(-1/2, 1/2, n_rays)[i]
 * -- without a rotation vector (rotvec = NULL) the fan of generate_cone_directions((cos m, sin m), opening, n_rays)
 * (reference src/cone.py:242-258: in the (0,1) plane, third component an exact 0) that cone_us_to_mri_world (:187-209)
 * places; with one, that fan turned about its apex by the rotation vector (axis x angle, radians; Rodrigues), i.e. rolled /
 * pitched out of the slice: the six-degree-of-freedom pose a registration optimises.  All pointers DEVICE float32:
 * median (P), opening (P when opening_stride = 1, one shared value when 0), rotvec (P,3) or NULL, dirs (P, n_rays, 3).
 * Backward: gdirs (P, n_rays, 3) = dL/ddirs (what diffus_render_bwd returns in gdirs) -> g_median (P), g_opening (P: per
 * pose also when the opening angle is shared -- the caller sums), g_rotvec (P,3); each output nullable.  Evaluated in float64.
 */
int diffus_fan_pose_fwd(const float *median, const float *opening, int opening_stride, const float *rotvec, int n_poses,
                        int n_rays, float *dirs, diffus_stream_t stream);
int diffus_fan_pose_bwd(const float *median, const float *opening, int opening_stride, const float *rotvec, const float *gdirs,
                        int n_poses, int n_rays, float *g_median, float *g_opening, float *g_rotvec, diffus_stream_t stream);

/*
 * The axis choice of differentiable_splat (reference src/renderer.py:702-710), on the device: x, y, z are the three
 * coordinate planes of the n samples (each float32, float64 or int64: *_dtype = DIFFUS_F32 / F64 / I64); axes[0..1]
 * (device int32) get the two axes of largest variance, largest first, ties in axis order -- the variance of the values
 * cast to float32, like the reference's `c.float().var()`, accumulated in float64 --, and c0 / c1 (nullable pair,
 * n float32 each) those two planes cast to float32, ready for diffus_splat_fwd.  No host round trip (the reference
 * syncs three times for `.item()`): the chain render -> splat -> loss is capturable.
 */
int diffus_splat_axes(const void *x, int x_dtype, const void *y, int y_dtype, const void *z, int z_dtype, long n, int *axes,
                      float *c0, float *c1, diffus_stream_t stream);

size_t diffus_splat_workspace_bytes(int P, int H, int W);
/* winner_keep (nullable, P*H*W int32): where the forward builds its winner raster (sample index per pixel, -1 = none) when
 * the caller wants it kept; handed to diffus_splat_bwd as winner_kept the backward needs no scatter of its own (and, for
 * kernel half-widths up to 8, three launches instead of six). */
int diffus_splat_fwd(const float *c0, const float *c1, const float *val, int P, long n, int cols,
                     int H, int W, float sigma, float *out, int *winner_keep,
                     void *workspace, size_t workspace_bytes, diffus_stream_t stream);
int diffus_splat_bwd(const float *c0, const float *c1, int P, long n,
                     int H, int W, float sigma, const float *gout, float *gval, const int *winner_kept,
                     void *workspace, size_t workspace_bytes, diffus_stream_t stream);

/*
 * B-mode artifact chain: replaces, for P frames (P,R,N) at once, the
 * artifacts=True branch of plot_beam_frame (reference src/renderer.py:264-273) =
 * add_speckle_arcs_np (:545-583) -> add_depth_dependent_lateral_blur_np (:585-601) ->
 * sharpen_np (:535-543), all in float64 like the NumPy/SciPy original.
 *   radial_noise (P,N), local_noise (P,R,N): nullable float64 multiplicative factors.  When
 *   NULL they are drawn as N(1, std*(1+depth^p)) from Philox4x32-10 keyed by `seed` (the
 *   reference uses the unseeded NumPy global RNG and is not reproducible).
 *   out (P,R,N) float64.
 */
size_t diffus_artifacts_workspace_bytes(int P, int R, int N);
int diffus_artifacts(const float *frame, int P, int R, int N,
                     double std_radial, double std_local, double max_sigma, double alpha,
                     const double *radial_noise, const double *local_noise, uint64_t seed,
                     double *out, void *workspace, size_t workspace_bytes, diffus_stream_t stream);

/*
 * Utility, not a reference function: the energy loss the benchmarks and examples
 * optimise.  loss[p] = sum(frame[p,:]^2) over the n floats of pose p, and (if
 * gframe != NULL) gframe = d loss / d frame = 2 * frame, in one streaming pass, one launch.
 * The workspace (>= 512 * P bytes) holds arrival counters: zero-fill it once before its first use; every
 * call leaves it zero-filled where it matters.  Deterministic (fixed-order tree sum of 16 partials per pose).
 */
int diffus_loss_sumsq(const float *frame, int P, long n, float *loss, float *gframe,
                      void *workspace, size_t workspace_bytes /* >= 512 * P bytes, zeroed once */,
                      diffus_stream_t stream);

/* ------------------------------------------------------------------------
 * MRI -> acoustic impedance, the producer of `vol` in the training loops.
 *
 * The MLP is ImpedanceEstimator (reference src/impedance.py:6-17): Linear(1,32) ReLU Linear(32,32) ReLU
 * Linear(32,1).  `params` = the 1153 floats of its state_dict in order: model.0.weight (32), model.0.bias (32),
 * model.2.weight (32x32, [out][in]), model.2.bias (32), model.4.weight (32), model.4.bias (1).
 */
#define DIFFUS_MLP_HIDDEN 32
#define DIFFUS_MLP_PARAMS 1153

/*
 * y[i * y_stride] = (mask == NULL || mask[i]) ? out_scale * mlp((x[i] - in_shift) / in_div) : fill        i < n
 * (y_stride in elements, 1 = contiguous: a stride lets the result land where it is used -- e.g. one slice of the
 * volume, `Z_vol[:, :, k] = model(x) * 1e6` of the reference's training notebook, cell 16 -- without a copy launch).
 * Replaces ImpedanceEstimator.forward (:16-17; in_shift 0, in_div 1, out_scale 1, mask NULL) and, with the
 * z-score constants, the 1e6 scale and fill = 400, the body of compute_impedance_volume (:45-53) in one pass;
 * the hidden layer runs on the f32 matrix cores, activations stay in registers.
 */
int diffus_mlp_fwd(const float *x, const unsigned char *mask, size_t n, const float *params,
                   float in_shift, float in_div, float out_scale, float fill, float *y, size_t y_stride,
                   diffus_stream_t stream);

/*
 * Backward of the above for upstream gy (n floats, gy_stride elements apart: the matching slice of d/dvolume is read
 * in place): gparams (1153 floats, overwritten) = d/dparams of sum(y * gy); gx (nullable, n floats) = d/dx.
 * Deterministic (fixed-order reductions).
 */
size_t diffus_mlp_workspace_bytes(void);
int diffus_mlp_bwd(const float *x, const unsigned char *mask, size_t n, const float *params,
                   float in_shift, float in_div, float out_scale, const float *gy, size_t gy_stride,
                   float *gparams, float *gx, void *workspace, size_t workspace_bytes,
                   diffus_stream_t stream);

/*
 * create_brain_mask (reference src/utils.py:12-21): mask = vol > threshold, then `iterations` binary dilations
 * and `iterations` binary erosions with SciPy's default structure (6-neighbourhood) and border value 0.
 * mask: d0*d1*d2 bytes of 0/1.
 */
size_t diffus_brain_mask_workspace_bytes(int d0, int d1, int d2);
int diffus_brain_mask(const float *vol, int d0, int d1, int d2, float threshold, int iterations,
                      unsigned char *mask, void *workspace, size_t workspace_bytes, diffus_stream_t stream);

/*
 * The statistics of zscore_normalize (reference src/utils.py:34-36) over the voxels with mask[i] != 0 (all when
 * mask == NULL): out[0] = mean, out[1] = unbiased standard deviation, out[2] = count (device doubles).
 */
size_t diffus_masked_stats_workspace_bytes(void);
int diffus_masked_stats(const float *vol, const unsigned char *mask, size_t n, double *out,
                        void *workspace, size_t workspace_bytes, diffus_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFUS_HIP_H */
