/*
 * pn2_abi.h -- C ABI of libpn2_hip.so: the MI355X (gfx950) drop-in for the
 * PointNet++ SA/FP custom ops of isl-org/Open3D-PointNet2-Semantic3D.
 *
 * Every entry point replaces one reference launcher / CPU op; dimension and
 * pointer order are kept identical to the reference launcher it replaces so the
 * mapping is auditable (reference file:line cited per function).  Differences
 * from the reference launchers, all deliberate:
 *   - extern "C", returns int (0 = ok, <0 = PN2_E* argument error, >0 = hipError_t
 *     from the launch) instead of void-with-no-error-check;
 *   - explicit stream (a hipStream_t passed as void*; NULL = the HIP null stream)
 *     instead of the legacy default stream (tf_grouping.cu:141 etc.);
 *   - gradient outputs are zero-filled by the callee with a stream-ordered
 *     memset (the reference does it in the TF op glue: tf_sampling.cpp:236,
 *     tf_grouping.cpp:271, tf_interpolate.cpp:477).
 *
 * Conventions: all tensors dense row-major contiguous DEVICE memory; float =
 * IEEE fp32; indices int32.  The caller owns every buffer (outputs and scratch);
 * the library never allocates, never synchronises, keeps no pointers and has no mutable global state (the kernel-
 * selection knobs of the A/B scripts exist only in a -DPN2_TUNING_HOOKS build, csrc/pn2_common.h).
 * Re-entrant; safe from several host threads on different streams.
 */
#ifndef PN2_ABI_H_
#define PN2_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PN2_ABI_VERSION 1

/* error codes (negative); positive return values are hipError_t */
#define PN2_OK 0
#define PN2_EINVAL (-1)  /* non-positive dimension / bad attribute            */
#define PN2_ENULL (-2)   /* required pointer is NULL                          */
#define PN2_ERANGE (-3)  /* dimension exceeds what the kernel supports        */
#define PN2_EUNSUP (-4)  /* unsupported configuration (e.g. channel widths)   */

/* squared-distance arithmetic, see DESIGN.md "Arithmetic modes".  The reference
 * CUDA build leaves nvcc's --fmad=true default on (tf_ops/CMakeLists.txt:5,13). */
#define PN2_ARITH_STRICT 0  /* (dx*dx + dy*dy) + dz*dz, each op rounded        */
#define PN2_ARITH_FMA 1     /* fma(dz,dz, fma(dx,dx, dy*dy))  (default)        */
#define PN2_ARITH_FMA_ALT 2 /* fma(dz,dz, fma(dy,dy, dx*dx))                   */

int pn2_abi_version(void);
const char *pn2_build_info(void);     /* "gfx950 ..." static string */
const char *pn2_strerror(int code);   /* static string for PN2_E* / hipError_t */

/* ---- sampling (replaces tf_ops/tf_sampling.cu) --------------------------- */

/* farthestpointsamplingLauncher(b,n,m,inp,temp,out)  tf_sampling.cu:218-221,
 * declared tf_sampling.cpp:114.  inp (b,n,3) -> out (b,m) int32; first pick is
 * index 0; tie-break (max dist, k mod 512, k) exactly as the 512-thread
 * reference block produces.  `temp` is the reference's (32,n) float scratch: it
 * is only used when n > PN2_FPS_MAX_REG_POINTS (then it must hold
 * min(b,32)*n floats); may be NULL otherwise.  Requires m >= 1. */
#define PN2_FPS_MAX_REG_POINTS 16384
int pn2_farthest_point_sample(int b, int n, int m, const float *inp, float *temp,
                              int *out, int arith_mode, void *stream);

/* farthest_point_sample + gather_point fused (the layer API always chains them,
 * util/pointnet_util.py:36-37): out (b,m) and new_xyz (b,m,3) = inp[out], bit-identical to the two
 * separate calls. */
int pn2_fps_gather(int b, int n, int m, const float *inp, float *temp, int *out,
                   float *new_xyz, int arith_mode, void *stream);

/* Nested sampling: the SA levels chain farthest_point_sample on each other's output (util/pointnet_util.py:36-37 called
 * by model.py:104-113 with l1_xyz, l2_xyz, l3_xyz).  When inp is the new_xyz of an FPS run over a larger cloud, its rows
 * are that run's picks in pick order and sampling them again retraces the same picks -- idx = 0..m-1 -- unless the parent
 * run met a TIE (two points holding the maximum distance) before step m: only then can this level's tie-break
 * (tf_sampling.cu:153-170: lowest k mod 512, then k, in THIS cloud's numbering) choose differently.
 *   tie_in  (b) int32 or NULL: first tied step of the run that produced inp (0x7fffffff = none).  A cloud with
 *           tie_in[i] >= m gets idx = 0..m-1 and new_xyz = its first m rows, no sampling; any other cloud (and every cloud
 *           when tie_in is NULL) is sampled exactly as pn2_fps_gather does.  The result is bit-identical either way.
 *   tie_out (b) int32 or NULL: this level's record, to be passed as the next level's tie_in.  Ties between points of equal
 *           coordinates do not count until the distance maximum reaches 0 (their twin is then picked too).
 * new_xyz (b,m,3) may be NULL (indices only).  temp as for pn2_farthest_point_sample. */
int pn2_fps_nested(int b, int n, int m, const float *inp, float *temp, int *out, float *new_xyz,
                   const int *tie_in, int *tie_out, int arith_mode, void *stream);

/* In-place reads of a batch that keeps xyz and rgb side by side (model.py:26-29 slices point_cloud (b,n,6) into l0_xyz /
 * l0_points; the reference's TF graph materialises both slices): the *_ld entry points take the ROW STRIDE in floats of such a
 * column block (ld >= 3 resp. >= c; the pointer is the block's first element, rows are ld floats apart, cloud i starts at
 * i * n * ld) and read it where it lies -- same bits as the dense entry point on a copy.  Shapes a strided kernel does not
 * exist for return PN2_EUNSUP (the caller then makes the copy).
 * pn2_fps_nested_ld: pn2_fps_nested for n <= 16384 (the register-resident kernels need no temp). */
int pn2_fps_nested_ld(int b, int n, int m, const float *inp, int ld, int *out, float *new_xyz,
                      const int *tie_in, int *tie_out, int arith_mode, void *stream);

/* The coarse levels of the pyramid in ONE launch (extension; replaces 3 launches per level): for l = 0 .. nlev-1, with the
 * source cloud of level l = the samples of level l-1 (level 0: xyz0 (b,n0,3)):
 *   fps_idx[l] (b,npoint[l]) int32, new_xyz[l] (b,npoint[l],3)   = pn2_fps_nested   (util/pointnet_util.py:36-37)
 *   bq_idx[l] (b,npoint[l],nsample[l]) int32, bq_cnt[l] (b,npoint[l]) or NULL = pn2_query_ball_point(radius[l], nsample[l],
 *                                                                 source cloud, new_xyz[l])      (util/pointnet_util.py:39)
 *   nn_dist[l], nn_idx[l] (b,n_l,3), n_l = points of the source cloud, or NULL = pn2_three_nn(source cloud, new_xyz[l])
 *                                                                 (util/pointnet_util.py:300 of the FP level above it)
 * bit for bit.  npoint / radius / nsample are HOST arrays of nlev entries, fps_idx ... nn_idx HOST arrays of nlev device
 * pointers (nn_dist / nn_idx / bq_cnt may be NULL as a whole or per level).  tie_in (b) or NULL: tie record of the run that
 * produced xyz0 (pn2_fps_nested); tie_out (b) or NULL: the record of the last level.  One workgroup per cloud walks down the
 * levels.  PN2_EUNSUP: n0 > 1024, nlev > 4, npoint[l] > points of its source cloud, or a 3-NN table over more than 256
 * samples -- use the separate entry points. */
int pn2_coarse_geometry(int b, int n0, int nlev, const int *npoint, const float *radius, const int *nsample,
                        const float *xyz0, const int *tie_in, int *const *fps_idx, float *const *new_xyz,
                        int *const *bq_idx, int *const *bq_cnt, float *const *nn_dist, int *const *nn_idx,
                        int *tie_out, int fps_arith_mode, int bq_arith_mode, void *stream);

/* probsampleLauncher(b,n,m,inp_p,inp_r,temp,out)  tf_sampling.cu:212-216, tf_sampling.cpp:72.  inp_p (b,n)
 * non-negative weights, inp_r (b,m) uniforms in [0,1) -> out (b,m) int32: the index at which the running sum of
 * the weights reaches inp_r * total (the reference's exact fp32 summation order and branch-free binary search).
 * temp (b,n) receives the running sums (the reference's scratch tensor). */
int pn2_prob_sample(int b, int n, int m, const float *inp_p, const float *inp_r, float *temp, int *out,
                    void *stream);

/* Farthest point sampling for clouds beyond PN2_FPS_MAX_REG_POINTS (16384 < n <= 131072), same picks as
 * pn2_farthest_point_sample (bit-identical), ~6x faster than its streaming path at n = 65536: the cloud is
 * sorted along a Morton curve into 64-point buckets and a round only revisits the buckets whose bounding box the
 * new pick can reach (exact: skipped buckets provably keep their running minima).  new_xyz (b,m,3) may be NULL.
 * `workspace`: 256-byte aligned device scratch of pn2_fps_large_workspace_bytes(b, n) bytes (replaces `temp`). */
size_t pn2_fps_large_workspace_bytes(int b, int n);
int pn2_fps_large(int b, int n, int m, const float *inp, void *workspace, size_t workspace_bytes, int *out,
                  float *new_xyz, int arith_mode, void *stream);

/* gatherpointLauncher(b,n,m,inp,idx,out)  tf_sampling.cu:222-225, tf_sampling.cpp:158 */
int pn2_gather_point(int b, int n, int m, const float *inp, const int *idx,
                     float *out, void *stream);

/* scatteraddpointLauncher(b,n,m,out_g,idx,inp_g)  tf_sampling.cu:226-229,
 * tf_sampling.cpp:195.  inp_g (b,n,3) is zeroed by the callee first. */
int pn2_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx,
                          float *inp_g, void *stream);

/* ---- grouping (replaces tf_ops/tf_grouping.cu) --------------------------- */

/* queryBallPointLauncher(b,n,m,radius,nsample,xyz1,xyz2,idx,pts_cnt)
 * tf_grouping.cu:138-144, tf_grouping.cpp:72.  xyz1 (b,n,3) dataset, xyz2
 * (b,m,3) queries -> idx (b,m,nsample), pts_cnt (b,m).  First nsample hits in
 * index order with max(sqrtf(d2),1e-20f) < radius; short rows padded with the
 * first hit.  Rows with no hit: idx row = 0, pts_cnt = 0 (the reference leaves
 * the row uninitialised; unreachable when queries are dataset points). */
int pn2_query_ball_point(int b, int n, int m, float radius, int nsample,
                         const float *xyz1, const float *xyz2, int *idx,
                         int *pts_cnt, int arith_mode, void *stream);
/* pn2_query_ball_point with the cloud's rows ld1 floats apart (see pn2_fps_nested_ld); only the shapes the LDS-grid kernel
 * takes (4096 <= n <= 8192, m >= 256, nsample <= 64), else PN2_EUNSUP. */
int pn2_query_ball_point_ld(int b, int n, int m, float radius, int nsample, const float *xyz1, int ld1,
                            const float *xyz2, int *idx, int *pts_cnt, int arith_mode, void *stream);
/* The same operator on an explicitly chosen kernel -- 0 by shape (= pn2_query_ball_point), 1 wave-per-queries scan,
 * 2 lane-per-query scan, 3 LDS grid; a kernel whose preconditions do not hold falls through to the next.  Every kernel
 * returns the same bits; this door lets the parity tests hold each of them to the oracle.  Stateless. */
int pn2_query_ball_point_kernel(int b, int n, int m, float radius, int nsample, const float *xyz1,
                                const float *xyz2, int *idx, int *pts_cnt, int arith_mode, int kernel,
                                void *stream);

/* Multi-radius ball query for MSG set abstraction (util/pointnet_util.py:245-250 calls query_ball_point once per
 * radius on the same xyz / new_xyz): ONE scan of xyz1, every squared distance tested against all thresholds.
 * radii, nsamples: host arrays of nradius (1..3) entries; idx[r] (b,m,nsamples[r]) and pts_cnt[r] (b,m): host
 * arrays of device pointers.  Bit-identical to nradius separate pn2_query_ball_point calls.  PN2_EUNSUP when
 * nradius > 3 or the hit lists do not fit LDS (sum of nsamples > ~130): call pn2_query_ball_point per radius. */
int pn2_query_ball_point_multi(int b, int n, int m, int nradius, const float *radii, const int *nsamples,
                               const float *xyz1, const float *xyz2, int *const *idx, int *const *pts_cnt,
                               int arith_mode, void *stream);

/* selectionSortLauncher(b,n,m,k,dist,outi,out)  tf_grouping.cu:145-149, tf_grouping.cpp:135.
 * dist (b,m,n) -> outi (b,m,n) int32, out (b,m,n): partial selection sort of the first k positions
 * of every row including the reference's swaps (whole rows bit-identical).  n <= 19200. */
int pn2_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                       void *stream);

/* groupPointLauncher(b,n,c,m,nsample,points,idx,out)  tf_grouping.cu:150-154,
 * tf_grouping.cpp:178.  out (b,m,nsample,c). */
int pn2_group_point(int b, int n, int c, int m, int nsample, const float *points,
                    const int *idx, float *out, void *stream);

/* groupPointGradLauncher(b,n,c,m,nsample,grad_out,idx,grad_points)
 * tf_grouping.cu:155-162, tf_grouping.cpp:220.  grad_points (b,n,c) zeroed first. */
int pn2_group_point_grad(int b, int n, int c, int m, int nsample,
                         const float *grad_out, const int *idx, float *grad_points,
                         void *stream);

/* The same gradient with caller-provided scratch (see pn2_three_interpolate_grad_ws): a per-point list of the grouped
 * rows that reference it, then a gather -- no float atomics on large levels.  `workspace`: 4-byte aligned, at least
 * pn2_group_point_grad_workspace_bytes(b, n, m, nsample) bytes. */
size_t pn2_group_point_grad_workspace_bytes(int b, int n, int m, int nsample);
int pn2_group_point_grad_ws(int b, int n, int c, int m, int nsample, const float *grad_out, const int *idx,
                            float *grad_points, void *workspace, size_t workspace_bytes, void *stream);

/* ---- interpolation (replaces the CPU ops of tf_ops/tf_interpolate.cpp) --- */

/* threenn_cpu(b,n,m,xyz1,xyz2,dists,indices)  tf_interpolate.cpp:213-243.
 * xyz1 (b,n,3) unknown, xyz2 (b,m,3) known, m >= 3 -> dist (b,n,3) squared L2
 * ascending (computed in float64 like the reference's KD-tree), idx (b,n,3);
 * ties -> lowest index. */
int pn2_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                 float *dist, int *idx, void *stream);
/* pn2_three_nn with the QUERY rows ld1 floats apart (see pn2_fps_nested_ld) */
int pn2_three_nn_ld(int b, int n, int m, const float *xyz1, int ld1, const float *xyz2,
                    float *dist, int *idx, void *stream);

/* threeinterpolate_cpu(b,m,c,n,points,idx,weight,out)  tf_interpolate.cpp:307-330 */
int pn2_three_interpolate(int b, int m, int c, int n, const float *points,
                          const int *idx, const float *weight, float *out,
                          void *stream);

/* threeinterpolate_grad_cpu(b,n,c,m,grad_out,idx,weight,grad_points)
 * tf_interpolate.cpp:397-421.  grad_points (b,m,c) zeroed first. */
int pn2_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out,
                               const int *idx, const float *weight,
                               float *grad_points, void *stream);

/* The same gradient with caller-provided scratch: large levels build a per-source-point list of (query, weight)
 * with 3*n integer atomics per cloud and then GATHER each grad_points row (no float atomics: 690 -> ~60 us on the
 * b=16, n=8192, m=1024, c=128 level of the training step); small levels, c % 4 != 0 or workspace == NULL run
 * pn2_three_interpolate_grad.  `workspace`: 4-byte aligned device scratch of at least
 * pn2_three_interpolate_grad_workspace_bytes(b, n, m) bytes.  Summation order varies from run to run. */
size_t pn2_three_interpolate_grad_workspace_bytes(int b, int n, int m);
int pn2_three_interpolate_grad_ws(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                  const float *weight, float *grad_points, void *workspace,
                                  size_t workspace_bytes, void *stream);

/* The list of the two *_grad_ws calls above as an object of its own.  It depends on (idx, weight) only -- geometry, not
 * weights or activations -- so a training step builds it AHEAD (with the FPS / ball query / three_nn of the batch, beside the
 * previous step's dense work) and its backward pass only gathers:
 *   out[b, idx[b,e], :] = sum over the entries e with that idx of  w[b,e] * rows_in[b, e / div, 0:c]      (e < nent)
 *   groupPointGradLauncher      (tf_grouping.cu:155-162):   nent = m*nsample, div = 1, nsrc = n, weight_kind 0 (w = 1)
 *   threeinterpolate_grad_cpu   (tf_interpolate.cpp:397-421): nent = 3*n, div = 3, nsrc = m, weight_kind 1 (weight (b,n,3))
 *     or 2: `weight` holds three_nn's squared distances (b,n,3) and w is the inverse-distance weight of
 *     util/pointnet_util.py:300-303, formed with the float expressions of pn2_fp_interp_concat.
 * idx (b,nent) int32 in [0,nsrc).  plan: 4-byte aligned, pn2_scatter_plan_bytes(b,nent,nsrc) bytes, opaque, position
 * independent (may be copied).  apply: rows_in rows are `in_stride` floats apart (>= c: a column slice of a wider gradient
 * is read in place), c % 4 == 0, c <= 1024, out (b,nsrc,c) 16-byte aligned, overwritten (no zero fill needed).  The order
 * of a list (= the fp32 summation order) is fixed by the build. */
size_t pn2_scatter_plan_bytes(int b, int nent, int nsrc);
/* pn2_scatter_plan_build for up to 8 plans of one batch in ONE memset + three launches (a training step builds seven beside the
 * previous step's dense work): plan i = the pn2_scatter_plan_bytes(b, nent[i], nsrc[i]) bytes at buffer + offset[i] (offsets
 * ascending, 16-byte aligned, non-overlapping, inside buffer_bytes), each a plan pn2_scatter_plan_apply takes. */
int pn2_scatter_plan_build_multi(int nplans, int b, const int *nent, const int *div, const int *nsrc, const int *const *idx,
                                 const float *const *weight, const int *weight_kind, void *buffer, const size_t *offset,
                                 size_t buffer_bytes, void *stream);
int pn2_scatter_plan_build(int b, int nent, int div, int nsrc, const int *idx, const float *weight, int weight_kind,
                           void *plan, size_t plan_bytes, void *stream);
int pn2_scatter_plan_apply(int b, int nent, int div, int c, int nsrc, const float *rows_in, int in_stride,
                           const void *plan, size_t plan_bytes, float *out, void *stream);

/* n (<= 48) independent device-to-device copies in ONE launch: dst[i][0..bytes[i]) = src[i][0..bytes[i]); the three arrays
 * are host arrays read at call time; regions must not overlap.  Training-step plumbing: a batch's geometry tensors (mixed
 * int32 / float32 / byte buffers) into the static buffers the captured step reads. */
int pn2_multi_copy(int n, const void *const *srcs, void *const *dsts, const unsigned long long *bytes, void *stream);
/* pn2_multi_copy + nfill (<= 4) scalar stores of 4 or 8 bytes in the same launch: *fill_dsts[i] = low fill_bytes[i] bytes of
 * fill_vals[i] (host arrays; the values travel in the launch arguments).  A captured training step's per-step scalars (Adam's
 * bias-corrected rate, the dropout step) ride with its input copy. */
int pn2_multi_copy_fill(int n, const void *const *srcs, void *const *dsts, const unsigned long long *bytes, int nfill,
                        void *const *fill_dsts, const unsigned long long *fill_vals, const int *fill_bytes, void *stream);

/* ---- fused layer kernels (new: no reference kernel; they replace the TF
 *      sub-graphs of util/pointnet_util.py:44-54,150-170 and :300-325) ------- */

/* One dense layer of the shared MLP with the BatchNorm folded in (inference):
 *   y = relu?( x @ W + bias )           x (rows,cin)  W (cin,cout)  y (rows,cout)
 * optionally followed by a max over each consecutive group of `pool` rows
 * (pool = nsample reproduces tf.reduce_max over K, pointnet_util.py:167-170;
 * pool <= 1 = no pooling; otherwise pool must be 32 or a multiple of 32 ... see
 * DESIGN.md).  y is (rows/pool, cout) when pooling.  fp32 MFMA, exact-f32
 * products, fp32 accumulation. */
int pn2_linear(int rows, int cin, int cout, const float *x, const float *w,
               const float *bias, int relu, int pool, float *y, void *stream);

/* Weight gradient of that layer for the training path: dw (cin,cout) = x^T (cin,rows) . dy (rows,cout), overwritten.
 * (The reduction over all B*M*K rows that TF / hipBLASLt run as a tall-skinny GEMM.)  fp32 MFMA, partial tiles
 * merged with fp32 atomics: the summation order varies from run to run like the reference's atomicAdd gradients. */
int pn2_linear_wgrad(int rows, int cin, int cout, const float *x, const float *dy, float *dw, void *stream);
/* dw += x^T . dy (no zero fill: the caller owns the initial value, e.g. a gradient arena zero-filled once per step). */
int pn2_linear_wgrad_accumulate(int rows, int cin, int cout, const float *x, const float *dy, float *dw, void *stream);

/* Training-mode batch normalisation + ReLU of a dense layer's output y (rows,c), channels last
 * (util/tf_util.py:555-581 batch_norm_template -> tf.contrib.layers.batch_norm, applied by conv2d / conv1d /
 * fully_connected at tf_util.py:186-204 and followed by tf.nn.relu):
 *   mean, var = per-channel batch moments over all rows (biased var; fp64 accumulation)
 *   z = relu?( gamma * (y - mean) / sqrt(var + eps) + beta )
 *   running_mean = decay*running_mean + (1-decay)*(mean + bias),  running_var likewise with the UNBIASED batch variance
 * `bias` (nullable, (c)) is the layer bias the caller folded away: a per-channel constant in front of BN only moves
 * the mean, so it enters the moving average and nothing else.  running_mean / running_var may both be NULL.
 * pool > 1 fuses the SA layer's max over each group of `pool` consecutive rows (tf.reduce_max over the K neighbours,
 * pointnet_util.py:167-170): z is then (rows/pool, c) and `ties` (rows/pool, c) receives the number of rows that
 * attain the maximum (the gradient is shared equally among them, as tf / torch do); the (rows,c) activation is never
 * written.  pool <= 1: z is (rows,c), ties unused (NULL).
 * save_mean / save_invstd (c) are kept for pn2_bn_relu_backward.  `workspace`: 8-byte aligned device scratch of at
 * least pn2_bn_workspace_bytes(c) bytes (fp64 per-channel accumulators, contents irrelevant on entry).
 * c <= 1024; c % 4 != 0 needs c <= 256. */
size_t pn2_bn_workspace_bytes(int c);
int pn2_bn_relu_forward(long long rows, int c, const float *y, const float *gamma, const float *beta,
                        const float *bias, float eps, float decay, int relu, int pool, float *running_mean,
                        float *running_var, void *workspace, size_t workspace_bytes, float *save_mean,
                        float *save_invstd, float *z, float *ties, void *stream);
/* Its gradient (what tf.gradients derives for the ops above): with g = dz * [z > 0] (relu; behind the fused max
 * pool dz (rows/pool,c) first goes to the rows that attain zmax, divided by `ties`) and xhat = (y - mean) * invstd,
 *   dbeta = sum_r g,  dgamma = sum_r g*xhat,  dy = gamma*invstd * (g - dbeta/rows - xhat * dgamma/rows).
 * zmax / ties: the forward's z / ties when pool > 1 (else NULL).  dy (rows,c) may alias dz when pool <= 1.  The ReLU
 * mask and the pooled maxima are recomputed from y with the forward's own float expressions. */
int pn2_bn_relu_backward(long long rows, int c, const float *dz, const float *y, const float *gamma,
                         const float *beta, const float *save_mean, const float *save_invstd, int relu,
                         int pool, const float *zmax, const float *ties, void *workspace,
                         size_t workspace_bytes, float *dy, float *dgamma, float *dbeta, void *stream);
/* The two calls above with a workspace the CALLER has already zero-filled: one fill of an arena holding the scratch of
 * every layer of a training step replaces one memset per call (88 per step for the semantic.json model). */
int pn2_bn_relu_forward_ws0(long long rows, int c, const float *y, const float *gamma, const float *beta,
                            const float *bias, float eps, float decay, int relu, int pool, float *running_mean,
                            float *running_var, void *workspace, size_t workspace_bytes, float *save_mean,
                            float *save_invstd, float *z, float *ties, void *stream);
int pn2_bn_relu_backward_ws0(long long rows, int c, const float *dz, const float *y, const float *gamma,
                             const float *beta, const float *save_mean, const float *save_invstd, int relu,
                             int pool, const float *zmax, const float *ties, void *workspace,
                             size_t workspace_bytes, float *dy, float *dgamma, float *dbeta, void *stream);
/* conv2d -> batch_norm of the training path (util/tf_util.py:186-204) without the statistics pass over y:
 * pn2_linear_bn_stats computes y (rows,cout) = x (rows,cin) . w (cin,cout) (no bias, no activation; cout % 32 == 0)
 * and adds the column sums of y and y*y to `bn_workspace` (pn2_bn_workspace_bytes(cout) bytes, ZEROED by the caller)
 * from the GEMM's accumulators; pn2_bn_relu_forward_stats is pn2_bn_relu_forward for such a (y, workspace) pair. */
int pn2_linear_bn_stats(int rows, int cin, int cout, const float *x, const float *w, float *y,
                        void *bn_workspace, size_t workspace_bytes, void *stream);
int pn2_bn_relu_forward_stats(long long rows, int c, const float *y, const float *gamma, const float *beta,
                              const float *bias, float eps, float decay, int relu, int pool, float *running_mean,
                              float *running_var, void *workspace, size_t workspace_bytes, float *save_mean,
                              float *save_invstd, float *z, float *ties, void *stream);

/* Fused set-abstraction MLP (pointnet_util.py:43-54 + :150-170, inference BN
 * folded): for every (b, j) group gathers nsample neighbours by idx, builds
 * [xyz[idx]-new_xyz | points[idx]] (xyz first, pointnet_util.py:52-54), runs up
 * to 3 dense layers (+bias, ReLU) and max-pools over the neighbours without the
 * grouped tensor ever reaching HBM.
 *   xyz (b,n,3)  new_xyz (b,m,3)  points (b,n,c) or NULL (c = 0)  idx (b,m,nsample)
 *   w[l] (cin_l, widths[l]) row-major with cin_0 = 3 + c;  bias[l] (widths[l])
 *   out (b,m,widths[nlayers-1])
 * Constraints: nsample 16, 32, 64, 128 or 256 (32 is the native tile; 16 needs b*m even), 1 <= nlayers <= 3,
 * widths multiples of 32, <= 128 and a supported layer pattern (returns PN2_EUNSUP otherwise: callers fall back
 * to pn2_group_point + pn2_linear). */
int pn2_sa_mlp_max_fused(int b, int n, int m, int nsample, int c, const float *xyz,
                         const float *new_xyz, const float *points, const int *idx,
                         int nlayers, const int *widths, const float *const *w,
                         const float *const *bias, float *out, void *stream);
/* pn2_sa_mlp_max_fused with the rows of xyz / points ld_xyz / ld_points floats apart (see pn2_fps_nested_ld); a strided
 * `points` needs the un-vectorised feature path (c % 8 != 0), else PN2_EUNSUP. */
int pn2_sa_mlp_max_fused_ld(int b, int n, int m, int nsample, int c, const float *xyz, int ld_xyz,
                            const float *new_xyz, const float *points, int ld_points, const int *idx,
                            int nlayers, const int *widths, const float *const *w,
                            const float *const *bias, float *out, void *stream);

/* bf16 variant of pn2_sa_mlp_max_fused (BASELINE configs[4]: large-scene inference, nsample 64): points is
 * (b,n,c) bfloat16, weights/biases are fp32 and are rounded to bf16 (RNE) by the kernel, products exact,
 * fp32 accumulation, hidden activations bf16(relu(acc+bias)), out (b,m,widths[last]) fp32 -- the precision
 * contract is spelled out in csrc/pn2_sa_fused_bf16.hip and restated by oracle.mlp_max_bf16.
 * Constraints: nsample 32 or 64, c % 16 == 0, points 16-byte aligned, widths multiples of 32 <= 128, a
 * supported layer pattern ([128], [128,128], [64,128], [64,64,128], [128,128,128]); PN2_EUNSUP otherwise. */
int pn2_sa_mlp_max_fused_bf16(int b, int n, int m, int nsample, int c, const float *xyz,
                              const float *new_xyz, const void *points_bf16, const int *idx,
                              int nlayers, const int *widths, const float *const *w,
                              const float *const *bias, float *out, void *stream);

/* Same gather + MLP chain without the max over the neighbours: out (b,m,nsample,widths[last]),
 * ReLU applied.  Feeds a wider last layer that runs on pn2_linear with pool = nsample.
 * Supported: nsample == 32, widths [128] or [128,128]; PN2_EUNSUP otherwise. */
int pn2_sa_mlp_rows_fused(int b, int n, int m, int nsample, int c, const float *xyz,
                          const float *new_xyz, const float *points, const int *idx,
                          int nlayers, const int *widths, const float *const *w,
                          const float *const *bias, float *out, void *stream);

/* Fused feature-propagation front end (pointnet_util.py:300-311):
 *   weight = (1/max(dist,1e-10)) / sum(1/max(dist,1e-10));
 *   out[b,j,:] = [ sum_i weight_i * points2[b,idx_i,:]  |  points1[b,j,:] ]
 * i.e. three_interpolate + concat (interpolated FIRST) in one pass.
 *   dist,idx (b,n,3)  points2 (b,m,c2)  points1 (b,n,c1) or NULL -> out (b,n,out_stride)
 * out_stride >= c2+c1 (0 = c2+c1): row stride of `out`; pad columns are zero-filled so that the
 * consumer can use 16-byte loads (e.g. 131 -> 136). */
int pn2_fp_interp_concat(int b, int n, int m, int c1, int c2, const float *dist,
                         const int *idx, const float *points1, const float *points2,
                         float *out, int out_stride, void *stream);

/* Dense-row MLP chain (pointnet_util.py:312-325, inference BN folded): up to 2 layers
 *   y = relu(relu(x @ W0 + b0) @ W1 + b1)      x (rows,cin)  W_l (cin_l, widths[l])
 * with all weights resident in LDS and the hidden activation kept in registers; pool = 32 adds a
 * max over each consecutive group of 32 rows (y is (rows/32, w_last)), pool = 0 keeps all rows.
 * widths multiples of 32, <= 128; PN2_EUNSUP when the configuration does not fit (callers fall
 * back to pn2_linear). */
int pn2_mlp_chain(int rows, int cin, const float *x, int nlayers, const int *widths,
                  const float *const *w, const float *const *bias, int pool, float *y,
                  void *stream);

/* Wide dense-row MLP chain for the coarse levels (SA3 tail, SA4, FP2, FP3: pointnet_util.py:150-170 and :312-325,
 * inference BN folded): up to 3 layers with widths in {128, 256, 512},
 *   y = act(...relu(relu(x @ W0 + b0) @ W1 + b1)...)     x (rows, x_stride >= cin)   W_l (cin_l, widths[l]) row-major
 * one workgroup per 32-row tile through all layers (activations in LDS, weights streamed from L2); relu_last selects the
 * activation of the last layer; pool = 32: max over each group of 32 consecutive rows, y (rows/32, w_last); pool = 0:
 * y (rows, w_last).  w, bias, y 16-byte aligned; W0 must hold ((cin + 7) & ~7) rows (zero rows behind the cin real ones:
 * the kernel contracts in groups of 8).  PN2_EUNSUP when the configuration does not fit (callers fall back to pn2_linear
 * per layer). */
int pn2_mlp_wide(int rows, int cin, int x_stride, const float *x, int nlayers, const int *widths,
                 const float *const *w, const float *const *bias, int relu_last, int pool, float *y, void *stream);
/* The same chain behind the SA front end (pointnet_util.py:39-54: group_point, centre, concat [xyz | features]) for
 * nsample = 32: rows are gathered through idx (b,m,32) straight into the first layer's operand tile.  W0's rows in the
 * order the tile is built in: [features (c rows) | x y z (3 rows) | zero rows up to a multiple of 8] (the reference's
 * variable is [xyz | features]: the caller rotates it once).  pool != 0: max over the 32 neighbours, y (b,m,w_last);
 * else (b,m,32,w_last). */
int pn2_sa_mlp_wide(int b, int n, int m, int nsample, int c, const float *xyz, const float *new_xyz,
                    const float *points, const int *idx, int nlayers, const int *widths, const float *const *w,
                    const float *const *bias, int pool, float *y, void *stream);
/* ... and behind the FP front end (pointnet_util.py:300-311), i.e. pn2_fp_interp_concat + pn2_mlp_wide in one kernel:
 *   x[b,j,:] = [ three_interpolate(points2, idx, w(dist))[b,j,:] | points1[b,j,:] ]   (never stored)
 * dist,idx (b,n,3)  points2 (b,m,c2)  points1 (b,n,c1) or NULL; n % 32 == 0, c1 % 4 == 0, c2 % 4 == 0; W0 holds
 * ((c2 + c1 + 7) & ~7) rows, interpolated channels first  ->  y (b*n, w_last), ReLU after every layer. */
int pn2_fp_mlp_wide(int b, int n, int m, int c1, int c2, const float *dist, const int *idx, const float *points1,
                    const float *points2, int nlayers, const int *widths, const float *const *w,
                    const float *const *bias, float *y, void *stream);

/* Fused feature-propagation block (pointnet_util.py:300-325, inference BN folded):
 *   x[b,j,:] = [ three_interpolate(points2, idx, w(dist))[b,j,:] | points1[b,j,:] ]   (never stored)
 *   y = relu(relu(x @ W0 + b0) @ W1 + b1)                                            (nlayers 1 or 2)
 * i.e. pn2_fp_interp_concat + pn2_mlp_chain in one kernel: the interpolated rows go from L2 straight
 * into the MFMA operands.  dist,idx (b,n,3)  points2 (b,m,c2)  points1 (b,n,c1) or NULL
 * w[0] (>= c2+c1 rows, widths[0]), interpolated channels first  ->  y (b*n, widths[nlayers-1]).
 * Constraints: c2 % 8 == 0, widths multiples of 32 and <= 128; PN2_EUNSUP otherwise. */
int pn2_fp_mlp_fused(int b, int n, int m, int c1, int c2, const float *dist, const int *idx,
                     const float *points1, const float *points2, int nlayers, const int *widths,
                     const float *const *w, const float *const *bias, float *y, void *stream);

/* ---- post-processing next to the path ----------------------- */

/* interpolate_label_with_color_cpu(num_sparse_points, num_dense_points, sparse_points, sparse_labels,
 * dense_points, dense_labels, dense_colors, knn)  tf_interpolate.cpp:71-115 (CPU + Open3D KD-tree in the
 * reference).  sparse_points (ns,3), sparse_labels (ns), dense_points (nd,3) -> dense_labels (nd) int32,
 * dense_colors (nd,3) uint8: majority label among the min(knn, ns) nearest sparse points (exact, float64
 * squared L2, ascending, ties -> lowest index; vote and colour table of :45-47,96-113).  ns == 0 gives
 * label -1 / colour 0; labels outside [0,9) get colour 0 (both undefined behaviour in the reference).
 * `workspace`: 256-byte aligned device scratch of at least pn2_interpolate_label_workspace_bytes(ns)
 * bytes (uniform grid over the sparse points).  knn <= 16 (PN2_EUNSUP above). */
size_t pn2_interpolate_label_workspace_bytes(int num_sparse_points);
int pn2_interpolate_label_with_color(int num_sparse_points, int num_dense_points,
                                     const float *sparse_points, const int *sparse_labels,
                                     const float *dense_points, int *dense_labels,
                                     uint8_t *dense_colors, int knn, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ---- training step ----------------------------------------------------------------- */

/* Data gradient of a dense layer: dx (rows,cin) = dy (rows,cout) . W^T, W (cin,cout) row-major as the forward pass holds
 * it; any cin / cout (tf.gradients of tf.nn.conv2d, util/tf_util.py:181-186). */
int pn2_linear_dgrad(int rows, int cin, int cout, const float *dy, const float *w, float *dx, void *stream);
/* The same when dx IS the gradient reaching the batch norm (+ReLU) of the layer below (the layer whose output is this
 * layer's input, consumed by nothing else): y_below (rows,cin) = that layer's pre-normalisation output, gamma / beta /
 * save_mean / save_invstd its parameters and saved batch moments, relu its activation flag.  While the accumulator tiles of
 * dx are at hand the kernel adds sum g and sum g*xhat per channel (g = dx * [ReLU mask], xhat = (y-mean)*invstd) to that
 * layer's ZEROED batch-norm workspace (pn2_bn_workspace_bytes(cin) bytes); pn2_bn_relu_backward_stats -- pn2_bn_relu_backward
 * for such a (dz, workspace) pair, pool <= 1 -- then skips its reduction pass over (dz, y).  Same results up to fp64
 * summation order.  PN2_EUNSUP for cout <= 16 (streaming kernel): use pn2_linear_dgrad + pn2_bn_relu_backward.
 * (tf.gradients through tf_util.py:186-204: conv2d -> batch_norm -> relu.) */
int pn2_linear_dgrad_bn_grad_stats(int rows, int cin, int cout, const float *dy, const float *w, float *dx,
                                   const float *y_below, const float *gamma, const float *beta, const float *save_mean,
                                   const float *save_invstd, int relu, void *bn_workspace, size_t workspace_bytes,
                                   void *stream);
int pn2_bn_relu_backward_stats(long long rows, int c, const float *dz, const float *y, const float *gamma,
                               const float *beta, const float *save_mean, const float *save_invstd, int relu,
                               int pool, const float *zmax, const float *ties, void *workspace,
                               size_t workspace_bytes, float *dy, float *dgamma, float *dbeta, void *stream);

/* model.get_loss  model.py:152-161: weighted sparse softmax cross-entropy, reduction SUM_BY_NONZERO_WEIGHTS.
 * logits (rows,num_class) f32, labels (rows) int32 (label64 = 0) or int64 (label64 = 1), weights (rows) f32 ->
 * *loss (device f32) = sum_r w_r*ce_r / max(1, #{w_r != 0}).  lse (rows) f32 and acc (2 doubles, zeroed by the callee)
 * carry the forward's results to the backward.  num_class <= 64. */
int pn2_weighted_ce_forward(int rows, int num_class, const float *logits, const void *labels, int label64,
                            const float *weights, float *lse, double *acc, float *loss, void *stream);
/* d loss / d logits = gout * w_r / nz * (softmax_r - onehot_r); gout: device scalar (upstream gradient) or NULL (= 1). */
int pn2_weighted_ce_backward(int rows, int num_class, const float *logits, const void *labels, int label64,
                             const float *weights, const float *lse, const double *acc, const float *gout,
                             float *dlogits, void *stream);

/* tf_util.dropout  util/tf_util.py:646-665 (tf.nn.dropout): y = x / keep_prob where kept, else 0; mask (n bytes) for the
 * backward.  state: device int64[2] = {seed, step}; the draw is a pure function of (seed, step, element index), so a
 * captured graph can be replayed while the caller advances `step` in device memory. */
int pn2_dropout(long long n, const float *x, float keep_prob, const long long *state, float *y,
                unsigned char *mask, void *stream);
int pn2_dropout_grad(long long n, const float *dy, const unsigned char *mask, float keep_prob, float *dx,
                     void *stream);

/* Backward of bias + ReLU on a layer WITHOUT batch norm (util/tf_util.py:186-204 with bn=False, tf.nn.relu's ReluGrad):
 * dx[i] = z[i] > 0 ? dz[i] : 0 over n elements, z = the layer's OUTPUT; dx may alias dz. */
int pn2_relu_grad(long long n, const float *z, const float *dz, float *dx, void *stream);

/* tf.train.AdamOptimizer.apply_gradients  train.py:381-388, one launch over flat fp32 buffers of n elements:
 * m <- b1 m + (1-b1) g;  v <- b2 v + (1-b2) g^2;  p <- p - lr_t * m / (sqrt(v) + eps), g = grads * grad_scale.
 * hyper: DEVICE float[5] = {lr_t, beta1, beta2, epsilon, grad_scale}, lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t). */
int pn2_adam_step(long long n, float *params, const float *grads, float *m, float *v, const float *hyper,
                  void *stream);

/* sample_and_group's tail for layers too wide for the fused kernel (util/pointnet_util.py:39-54: group_point(xyz) -
 * tile(new_xyz), group_point(points), concat [xyz | points]) in one pass: out (b,m,nsample,3+c) with
 * out[..., :3] = xyz[idx] - new_xyz and out[..., 3:] = points[idx] (c may be 0, then points may be NULL). */
int pn2_sa_group_concat(int b, int n, int m, int nsample, int c, const float *xyz, const float *new_xyz,
                        const float *points, const int *idx, float *out, void *stream);

/* ---- the step before the path: scene sampler + voxel down-sampling ---------------- */

/* SemanticFileData._extract_z_box  dataset/semantic_dataset.py:123-163, for b centre points at once.
 * points (n,3) float64 sorted by x (as __init__ leaves them, :84-88); centers (b,3) float64; half_x/half_y =
 * box_size/2; scene_z_size = max z - min z of the scene (:132).  out_idx (b,cap) = scene indices of every column in
 * scene order, out_cnt (b) = their number (may exceed cap: then only the first cap indices were written). */
int pn2_scene_extract_z_box(int n, const double *points, int b, const double *centers, double half_x,
                            double half_y, double scene_z_size, int cap, int *out_idx, int *out_cnt,
                            void *stream);

/* _get_fix_sized_sample_mask + gathers + _center_box  semantic_dataset.py:90-121,165-186.  The reference's random
 * draw enters as `mask` (b,cap) bytes: for a column with cnt > npts its first cnt bytes are the shuffled boolean
 * sample mask (exactly npts non-zero); columns with cnt <= npts repeat their indices (i mod cnt) and ignore it.
 * labels (n) int32 / colors (n,3) float64 may be NULL (zeros out).  Outputs: out_sel (b,npts) scene indices,
 * out_centered (b,npts,3) float32 = float64 (p - shift) cast, out_raw (b,npts,3) float64, out_labels (b,npts),
 * out_colors (b,npts,3) float32 (the last three optional).  status (b): 0 ok, 1 empty column, 2 cnt > cap,
 * 3 mask does not select exactly npts points. */
int pn2_scene_sample(int b, int npts, int cap, const double *points, const int *labels, const double *colors,
                     const int *idx, const int *cnt, const unsigned char *mask, double half_x, double half_y,
                     int *out_sel, float *out_centered, double *out_raw, int *out_labels, float *out_colors,
                     int *status, void *stream);

/* down_sample  downsample.py:46-67: Open3D (IntelVCL/Open3D @33e46f7) voxel_down_sample_and_trace with
 * min_bound = min(points) - voxel_size/2 + np.bincount(labels).argmax() per voxel.  points/colors (n,3) float64,
 * labels (n) int32 in [0,64) or NULL.  Outputs sized for n voxels; *out_count (device) = number of voxels, sorted by
 * (ix,iy,iz) (Open3D's own order is unspecified).  status (device): 0 ok, 1 voxel index needs more than 21 bits,
 * 2 label outside [0,64).  workspace: 256-byte aligned, pn2_voxel_downsample_workspace_bytes(n) bytes. */
size_t pn2_voxel_downsample_workspace_bytes(int n);
int pn2_voxel_downsample(int n, const double *points, const double *colors, const int *labels, double voxel_size,
                         double *out_points, double *out_colors, int *out_labels, int *out_count, int *status,
                         void *workspace, size_t workspace_bytes, void *stream);

/* pn2_fp_mlp_fused with the first layer's product with the INTERPOLATED channels hoisted out by linearity:
 * three_interpolate(points2) @ W1a == three_interpolate(points2 @ W1a) (pointnet_util.py:300-311 followed by the first conv
 * of :312-325), and z = points2 @ W1a has the m known rows per cloud instead of the n unknown ones.  z (b*m, widths[0])
 * replaces points2; w[0] = the c1 skip-link rows of the folded first-layer weight (c1 x widths[0]; NULL when c1 == 0),
 * w[1..] / bias[0..] as in pn2_fp_mlp_fused.  nlayers = 2 or 3 (all widths multiples of 32, <= 128, LDS-resident).
 * fp32 results differ from the un-hoisted order by rounding only (~1e-7 of the activation scale). */
int pn2_fp_mlp_fused_pre(int b, int n, int m, int c1, const float *dist, const int *idx, const float *points1,
                         const float *z, int nlayers, const int *widths, const float *const *w,
                         const float *const *bias, float *y, void *stream);
/* pn2_fp_mlp_fused_pre with the skip-link rows ld_points1 floats apart (see pn2_fps_nested_ld) */
int pn2_fp_mlp_fused_pre_ld(int b, int n, int m, int c1, const float *dist, const int *idx, const float *points1,
                            int ld_points1, const float *z, int nlayers, const int *widths, const float *const *w,
                            const float *const *bias, float *y, void *stream);

/* pn2_fp_mlp_fused_pre with the kernel SCHEDULE named by the caller (stateless door for parity tests / A-B timing):
 * 0 = lockstep kernel (8 waves: gather, MFMA layers, store), 1 = software-pipelined kernel (one wave per SIMD builds the next
 * tile's first-layer accumulator between the MFMA groups of the current tile; three 128-wide layers, c1 <= 8, else
 * PN2_EUNSUP).  Same bits either way; pn2_fp_mlp_fused_pre picks 1 where it applies. */
int pn2_fp_mlp_fused_pre_schedule(int b, int n, int m, int c1, const float *dist, const int *idx, const float *points1,
                                  const float *z, int nlayers, const int *widths, const float *const *w,
                                  const float *const *bias, float *y, int schedule, void *stream);

/* pn2_sa_mlp_max_fused / pn2_sa_mlp_rows_fused with the FEATURE part of the first layer hoisted by linearity:
 * [xyz - centre | features] @ W1 = (xyz - centre) @ W1[:3] + (features @ W1[3:])[idx], and zf = features @ W1[3:] has one
 * row per source point (b*n) instead of one per grouped neighbour (b*m*nsample).  zf (b*n, widths[0]) replaces `points`,
 * w[0] = the 3 xyz rows of the folded first-layer weight.  pool != 0: max over the neighbours -> (b, m, w_last); else the
 * un-pooled rows (b, m, nsample, w_last).  nsample = 32; widths multiples of 32, <= 128 ([64,64,128] pooled,
 * [128,128] un-pooled, [128,128,128] pooled); PN2_EUNSUP otherwise. */
int pn2_sa_mlp_fused_pre(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const float *zf,
                         const int *idx, int nlayers, const int *widths, const float *const *w,
                         const float *const *bias, int pool, float *out, void *stream);

/* The wide-layer versions (pn2_fp_mlp_wide / pn2_sa_mlp_wide: widths 128 / 256 / 512, one launch per level) of the same
 * hoisting.  FP: z = points2 @ W0[:c2] (b*m, widths[0]); w[0] = the c1 skip-link rows of the folded first-layer weight
 * zero-padded to a multiple of 8 rows (c1 > 0, c1 % 4 == 0, n % 32 == 0).  SA: zf = points @ W0[feature rows]
 * (b*n, widths[0]); w[0] = the 3 xyz rows + 5 zero rows (8 x widths[0]); nsample = 32. */
int pn2_fp_mlp_wide_pre(int b, int n, int m, int c1, const float *dist, const int *idx, const float *points1,
                        const float *z, int nlayers, const int *widths, const float *const *w,
                        const float *const *bias, float *y, void *stream);
int pn2_sa_mlp_wide_pre(int b, int n, int m, int nsample, const float *xyz, const float *new_xyz, const float *zf,
                        const int *idx, int nlayers, const int *widths, const float *const *w,
                        const float *const *bias, int pool, float *y, void *stream);

/* query_ball_point with the binning hoisted out: pn2_ball_query_bin sorts every cloud of a batch into the uniform grid of
 * `radius` ONCE (one workgroup per cloud; workspace = b * pn2_ball_query_bin_bytes(n) bytes, 256-byte aligned, n <= 8192);
 * pn2_query_ball_point_binned then answers the queries from it -- its workgroups (16 per cloud at m = 1024) copy the
 * cell-sorted cloud instead of each re-binning it.  Same radius and xyz1 in both calls; results bit-identical to
 * pn2_query_ball_point (tf_grouping.cu:3-43).  PN2_EUNSUP for shapes outside the grid kernel (n > 8192, nsample > 64). */
size_t pn2_ball_query_bin_bytes(int n);
int pn2_ball_query_bin(int b, int n, float radius, const float *xyz1, void *workspace, size_t workspace_bytes, void *stream);
/* pn2_ball_query_bin with the cloud's rows ld1 floats apart (see pn2_fps_nested_ld): the bins of a dense copy, bit for bit.
 * pn2_query_ball_point_binned never reads xyz1 (the bins hold the cell-sorted cloud): it takes the strided pointer as it is. */
int pn2_ball_query_bin_ld(int b, int n, float radius, const float *xyz1, int ld1, void *workspace, size_t workspace_bytes,
                          void *stream);
int pn2_query_ball_point_binned(int b, int n, int m, float radius, int nsample, const float *xyz1, const float *xyz2,
                                const void *bins, int *idx, int *pts_cnt, int arith_mode, void *stream);

/* Pooling over the K neighbours of a group -- the `pooling=` variants of pointnet_sa_module, util/pointnet_util.py:165-191
 * (tf.reduce_max / reduce_mean / the exp(-5 |grouped_xyz|) weighted average / concat [avg, max]).  mode: 0 max, 1 avg,
 * 2 weighted_avg, 3 max_and_avg.  x (rows, k, c) float32, gxyz (rows, k, 3) (mode 2 only, else NULL) ->
 * out (rows, c), or (rows, 2c) = [avg | max] for mode 3.  The gradient is taken w.r.t. x only (grouped_xyz comes from
 * non-differentiable index ops); a maximum shared by several neighbours splits its gradient evenly (tf.reduce_max). */
int pn2_group_pool(long long rows, int k, int c, int mode, const float *x, const float *gxyz, float *out, void *stream);
int pn2_group_pool_grad(long long rows, int k, int c, int mode, const float *x, const float *gxyz, const float *dout,
                        float *dx, void *stream);

/* conv2d -> batch_norm -> relu -> conv2d of the training path (tf_util.py:186-204 twice) without writing the normalised
 * activation of the lower layer: pn2_bn_relu_forward_deferred turns the statistics of y (stats_done = 1: left in the zeroed
 * workspace by pn2_linear_bn_stats; 0: taken here, workspace zeroed by the caller) into save_mean / save_invstd, the moving
 * averages and per-channel (scale, shift) with z = relu?(fma(y, scale, shift)); pn2_linear_bn_stats_xf (the upper layer's
 * forward GEMM + statistics) and pn2_linear_wgrad_accumulate_xf (its weight gradient) apply them to y while loading it.
 * Only valid when the lower activation has no other consumer.  _xf: cin % 4 == 0, rows > 2048, 16-byte aligned operands,
 * else PN2_EUNSUP. */
int pn2_bn_relu_forward_deferred(long long rows, int c, const float *y, const float *gamma, const float *beta,
                                 const float *bias, float eps, float decay, int stats_done, float *running_mean,
                                 float *running_var, void *workspace, size_t workspace_bytes, float *save_mean,
                                 float *save_invstd, float *scale, float *shift, void *stream);
int pn2_linear_bn_stats_xf(int rows, int cin, int cout, const float *x_raw, const float *w, float *y, void *bn_workspace,
                           size_t workspace_bytes, const float *a_scale, const float *a_shift, int a_relu, void *stream);
int pn2_linear_wgrad_accumulate_xf(int rows, int cin, int cout, const float *x_raw, const float *dy, float *dw,
                                   const float *a_scale, const float *a_shift, int a_relu, void *stream);

/* y (rows, cout) = x (rows, cin) . w (cin, cout) + bias (NULL: none), no activation, for 1 <= cout <= 16: the class head
 * (model.py:145-146: conv1d(num_class), activation_fn=None) as one streaming launch instead of a zero-padded MFMA layer.
 * cin % 4 == 0, cin * cout * 4 <= 48 KB, x 16-byte aligned, else PN2_EUNSUP. */
int pn2_linear_narrow(int rows, int cin, int cout, const float *x, const float *w, const float *bias, float *y, void *stream);

/* The batch-norm gradient of the training path (tf_util.py:555-581 through tf.gradients) applied ON LOAD by the two gradient
 * GEMMs of the layer, so that dy = the gradient leaving the batch norm (+ReLU [+ max over groups of 32 rows]) is never written
 * or re-read.  pn2_bn_grad_constants turns the two per-channel sums (stats_done = 1: left in the zeroed workspace by
 * pn2_linear_dgrad_bn_grad_stats / pn2_linear_dgrad_gx of the layer above; 0: taken here with one pass over (dz, y), workspace
 * zeroed by the caller) into dgamma, dbeta and coef (6, c) = sc, sh, mean, invstd, k1, k2 with
 * dy = sc * fma(-(y - mean) * invstd, k2, g - k1), g = dz * [relu mask] -- pn2_bn_relu_backward's expressions, so both forms
 * hand the GEMMs the same bits.  pn2_linear_dgrad_gx: dx (rows, cin) = dy (rows, cout) . W^T (pn2_linear_dgrad) with dy formed
 * from y (rows, cout), dz ((rows, cout); pool = 32: the (rows / 32, cout) gradient of the pooled maxima beside zmax / ties of
 * pn2_bn_relu_forward) and coef; y_below != NULL adds pn2_linear_dgrad_bn_grad_stats' epilogue for the layer below.
 * pn2_linear_wgrad_gx: dw (cin, cout) += x^T . dy likewise; a_scale != NULL: x is the pre-normalisation output of the layer
 * below (pn2_linear_wgrad_accumulate_xf).  pool in {0, 32}; cout % 4 == 0, 16 < cout <= 512 and 16-byte aligned y / dz / coef for
 * the data gradient, else PN2_EUNSUP. */
int pn2_bn_grad_constants(long long rows, int c, const float *dz, const float *y, const float *gamma, const float *beta,
                          const float *save_mean, const float *save_invstd, int relu, int pool, const float *zmax,
                          const float *ties, const float *ysel, int stats_done, void *workspace, size_t workspace_bytes,
                          float *coef, float *dgamma, float *dbeta, void *stream);
/* pn2_bn_relu_forward with pool > 1 that also keeps ysel (rows / pool, c) = the pre-normalisation value of the first row that
 * attains each pooled maximum (stats_mode 3: sums there and folded, see pn2_linear_bn_stats_fin).  With it (pn2_bn_grad_constants(..., ysel != NULL, stats_done = 0)) the backward reduction behind
 * the max pool reads the pooled tensors only -- the gradient is non-zero on the rows attaining the maximum, all of which carry
 * zmax -- instead of making a pass over y (rows, c).  stats_mode 0 / 1 / 2 = pn2_bn_relu_forward / _ws0 / _stats. */
int pn2_bn_relu_forward_pool(long long rows, int c, const float *y, const float *gamma, const float *beta, const float *bias,
                             float eps, float decay, int relu, int pool, float *running_mean, float *running_var,
                             void *workspace, size_t workspace_bytes, int stats_mode, float *save_mean, float *save_invstd,
                             float *zmax, float *ties, float *ysel, void *stream);
int pn2_linear_dgrad_gx(int rows, int cin, int cout, const float *y, const float *dz, const float *coef, int relu, int pool,
                        const float *zmax, const float *ties, const float *w, float *dx, const float *y_below,
                        const float *gamma_below, const float *beta_below, const float *mean_below, const float *invstd_below,
                        int relu_below, void *ws_below, size_t ws_below_bytes, void *stream);
int pn2_linear_wgrad_gx(int rows, int cin, int cout, const float *x, const float *a_scale, const float *a_shift, int a_relu,
                        const float *y, const float *dz, const float *coef, int relu, int pool, const float *zmax,
                        const float *ties, float *dw, void *stream);

/* "The last workgroup finishes" (round 6): every producer of batch-norm sums used to be followed by a one-block launch that folds
 * the slot copies of the sums and derives per-channel constants (45 launches of ~5 us per training step, all on the critical
 * path).  The producers below take a two-level ticket per workgroup instead, and the workgroup that draws the last one does that
 * work inside the producing launch.
 *   pn2_linear_bn_stats_fin: pn2_linear_bn_stats (a_scale == NULL) / pn2_linear_bn_stats_xf + finish 1: fold (then
 *     pn2_bn_relu_forward_mode / pn2_bn_relu_forward_pool with stats_mode 3), or 2: fold + what pn2_bn_relu_forward_deferred
 *     publishes (save_mean, save_invstd, moving averages, scale / shift; scale == shift == NULL: not wanted).
 *   pn2_linear_dgrad_fin: pn2_linear_dgrad (dy given, y == NULL) / pn2_linear_dgrad_gx (dy == NULL) + the epilogue of
 *     pn2_linear_dgrad_bn_grad_stats for the layer below (y_below != NULL) + finish_below 0: none, 1: fold (then
 *     pn2_bn_relu_backward_mode with stats_mode 3), 3: fold + what pn2_bn_grad_constants publishes for the layer below.
 *   pn2_bn_relu_forward_mode / pn2_bn_relu_backward_mode: pn2_bn_relu_forward (pool = 0) / pn2_bn_relu_backward with the state
 *     of the workspace explicit: 0 zero it here, 1 caller zeroed it, 2 sums already there, 3 sums there and folded.
 * (pn2_bn_relu_forward_deferred with stats_done = 0, pn2_bn_relu_forward / _backward in modes 0 / 1 and pn2_bn_grad_constants
 * with stats_done = 0 finish inside their own reduction kernel; no signature changes.)  Reference: tf_util.py:186-204,555-581. */
int pn2_linear_bn_stats_fin(int rows, int cin, int cout, const float *x, const float *w, float *y, void *bn_workspace,
                            size_t workspace_bytes, const float *a_scale, const float *a_shift, int a_relu, int finish,
                            const float *gamma, const float *beta, const float *bias, float eps, float decay,
                            float *running_mean, float *running_var, float *save_mean, float *save_invstd, float *scale,
                            float *shift, void *stream);
int pn2_linear_dgrad_fin(int rows, int cin, int cout, const float *dy, const float *y, const float *dz, const float *coef,
                         int relu, int pool, const float *zmax, const float *ties, const float *w, float *dx,
                         const float *y_below, const float *gamma_below, const float *beta_below, const float *mean_below,
                         const float *invstd_below, int relu_below, void *ws_below, size_t ws_below_bytes, int finish_below,
                         float *coef_below, float *dgamma_below, float *dbeta_below, void *stream);
int pn2_bn_relu_forward_mode(long long rows, int c, const float *y, const float *gamma, const float *beta, const float *bias,
                             float eps, float decay, int relu, float *running_mean, float *running_var, void *workspace,
                             size_t workspace_bytes, int stats_mode, float *save_mean, float *save_invstd, float *z,
                             void *stream);
int pn2_bn_relu_backward_mode(long long rows, int c, const float *dz, const float *y, const float *gamma, const float *beta,
                              const float *save_mean, const float *save_invstd, int relu, int pool, const float *zmax,
                              const float *ties, void *workspace, size_t workspace_bytes, int stats_mode, float *dy,
                              float *dgamma, float *dbeta, void *stream);

/* Data gradient AND weight gradient of a narrow dense + batch-norm layer in ONE launch (csrc/pn2_bwd_fused.hip): what
 * pn2_linear_dgrad_fin and pn2_linear_wgrad_gx compute for the same arguments -- dx (rows, cin) = dy . W^T, dw (cin, cout) +=
 * x^T . dy with dy = the gradient leaving the layer's batch norm formed on load from (y, dz, coef), the epilogue and finish for
 * the layer below -- reading (y, dz) ONCE: for 32 / 64-channel layers over 10^5 .. 10^6 rows both GEMMs are HBM streams and the
 * second read is a third of the traffic.  cin, cout in {32, 64}, rows % 32 == 0, 16-byte aligned operands, else PN2_EUNSUP.
 * dw is added to.  tf_util.py:181-204,555-581 through tf.gradients. */
int pn2_linear_bwd_fused(int rows, int cin, int cout, const float *x, const float *a_scale, const float *a_shift, int a_relu,
                         const float *y, const float *dz, const float *coef, int relu, int pool, const float *zmax,
                         const float *ties, const float *w, float *dx, float *dw, const float *y_below,
                         const float *gamma_below, const float *beta_below, const float *mean_below, const float *invstd_below,
                         int relu_below, void *ws_below, size_t ws_below_bytes, int finish_below, float *coef_below,
                         float *dgamma_below, float *dbeta_below, void *stream);

/* First layer of an SA module whose points carry few channels (c <= 5: xyz + rgb of the level-0 module), training path, in ONE
 * launch: y (b,m,nsample,cout) = [group_point(xyz, idx) - new_xyz | group_point(points, idx)] . w (3 + c, cout)
 * (pointnet_util.py:39-54 + tf_util.py:181-186) written once, its batch statistics taken on the way out into the ZEROED
 * workspace and folded by the launch's last workgroup (finish 1) or folded and turned into the deferred batch norm's constants
 * (finish 2, see pn2_linear_bn_stats_fin).  xg (b,m,nsample,3+c), optional: the grouped input for the weight gradient. */
int pn2_sa_first_layer_bn(int b, int n, int m, int nsample, int c, int cout, const float *xyz, const float *new_xyz,
                          const float *points, const int *idx, const float *w, float *y, float *xg, void *workspace,
                          size_t workspace_bytes, int finish, const float *gamma, const float *beta, const float *bias, float eps,
                          float decay, float *running_mean, float *running_var, float *save_mean, float *save_invstd,
                          float *scale, float *shift, void *stream);

/* First layer of an SA / FP module of the TRAINING path with its feature half applied to the source rows (gather and
 * interpolation are linear and commute with a 1x1 conv):
 *   SA (pointnet_util.py:39-54,150-156):  y (b,m,nsample,cout) = (group_point(xyz, idx) - new_xyz) . w_xyz (3,cout) + z[b, idx]
 *      with z (b,n,cout) = points . W[3:] from pn2_linear; gxyz (b,m,nsample,3), optional: the centred coordinates.
 *   FP (pointnet_util.py:300-312):        y (b,n,cout) = three_interpolate(z, idx, w(dist)) + points1 (b,n,c1) . w1 (c1,cout)
 *      with z (b,m,cout) = points2 . W[:c2]; weights from three_nn's squared distances as in pn2_fp_interp_concat; 1 <= c1 <= 8.
 * cout % 4 == 0, <= 1024; z, y, w 16-byte aligned.  The data and weight gradients of the feature half are then GEMMs over
 * the source rows (pn2_scatter_plan_apply of dy -> dz, pn2_linear_dgrad / pn2_linear_wgrad on n resp. m rows). */
int pn2_sa_hoist_rows(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz, const int *idx,
                      const float *z, const float *w_xyz, float *y, float *gxyz, void *stream);
int pn2_fp_hoist_rows(int b, int n, int m, int c1, int cout, const float *dist, const int *idx, const float *points1,
                      const float *z, const float *w1, float *y, void *stream);
/* The same two with the batch statistics of y (tf_util.py:186-204: conv2d -> batch_norm_template) taken on the way out: every
 * workgroup adds its column sums of y and y^2 (fp64) to one of the slot copies of the ZEROED batch-norm workspace
 * (pn2_bn_workspace_bytes(cout)), so no statistics pass reads y again; finish = 0: the copies are left there
 * (pn2_bn_relu_forward_stats / _deferred with stats_done = 1 fold them), 1: the last workgroup folds them, 2: it also publishes what
 * pn2_bn_relu_forward_deferred publishes (save_mean / save_invstd, the moving averages, scale / shift) -- the arguments of
 * pn2_linear_bn_stats_fin.  r06: the hoisted first layers of SA2-SA4 / FP4 in training are one launch less each. */
int pn2_sa_hoist_rows_bn(int b, int n, int m, int nsample, int cout, const float *xyz, const float *new_xyz, const int *idx,
                         const float *z, const float *w_xyz, float *y, float *gxyz, void *bn_workspace, size_t workspace_bytes,
                         int finish, const float *gamma, const float *beta, const float *bias, float eps, float decay,
                         float *running_mean, float *running_var, float *save_mean, float *save_invstd, float *scale,
                         float *shift, void *stream);
int pn2_fp_hoist_rows_bn(int b, int n, int m, int c1, int cout, const float *dist, const int *idx, const float *points1,
                         const float *z, const float *w1, float *y, void *bn_workspace, size_t workspace_bytes, int finish,
                         const float *gamma, const float *beta, const float *bias, float eps, float decay, float *running_mean,
                         float *running_var, float *save_mean, float *save_invstd, float *scale, float *shift, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PN2_ABI_H_ */
