/*
 * mpcmax.h -- C ABI of libmpcmax.so: the contrast-maximisation (CMax) loss hot path of
 * tub-rip/MotionPriorCMax, hand-written for AMD MI355X (gfx950).
 *
 * The reference has no native code and therefore no FFI to mirror; each entry point below
 * replaces a stretch of the reference's PyTorch code (cited per function, paths relative to
 * the reference root).  The Python host side (motionpriorcmax_amd.losses) binds these with
 * ctypes -- see INTEGRATION.md for the stub a maintainer would add to the reference.
 *
 * Conventions
 *  - all tensor arguments are DEVICE pointers to densely packed fp32 / int32 arrays owned by
 *    the caller (PyTorch allocates them); the library never frees or retains them;
 *  - `ws` is a caller-provided scratch buffer of at least mpc_workspace_bytes() bytes; its
 *    content carries state from a *_fwd call to the matching *_bwd call;
 *  - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *    synchronising or allocating; the work consists of kernel launches only (no memset / memcpy nodes), so a
 *    sequence of calls can be captured into a HIP graph and replayed;
 *  - no mutable global state takes part in a result: the last-error string is thread local, and the only
 *    process-wide state is one-time, idempotent set-up (raising the dynamic-LDS limit of the kernels, tuning
 *    switches read once from the environment, listed in csrc/knn.hip);
 *  - return value: 0 ok, >0 a hipError_t, <0 an argument error (MPC_E_*).  No C++ exception
 *    crosses the ABI.
 *
 * Event record layout (one row of events[B][M][6], loader.py:156-161,360-364):
 *   0:y  1:x  2:t in [0,1]  3:polarity  4:bin index  5:valid (0 for padding rows)
 * Rows [0,Mp) are the positive block, rows [Mp,M) the negative block (loader.py:392-395); the
 * polarity split is by ROW INDEX as in focus.py:216-227.
 */
#ifndef MPCMAX_H
#define MPCMAX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_VERSION 107

/* flags (mpc_shape.flags) -- one bit per FocusLoss constructor switch (focus.py:28-45) */
#define MPC_F_SCALE_BY_DT     (1u << 0)  /* scale_iwe_by_dt        focus.py:204-206 */
#define MPC_F_MASK_BORDER     (1u << 1)  /* mask_image_border      focus.py:208-214 */
#define MPC_F_POLARITY_SPLIT  (1u << 2)  /* polarity_aware_batching focus.py:216-227 */
#define MPC_F_NORM_L2         (1u << 3)  /* focus_loss_norm == 'l2' (else 'l1') loss.py:22-25 */
#define MPC_F_OBJ_VARIANCE    (1u << 4)  /* loss_type == 'variance' loss.py:14-16 (else gradient magnitude) */
#define MPC_F_DIST_L1         (1u << 5)  /* dist_norm == 'l1' (else squared 'l2') focus.py:132-135 */
#define MPC_F_SCHEME_IWD      (1u << 6)  /* interpolation_scheme == 'iwd' (else 'mean') focus.py:155-163 */
#define MPC_F_WANT_NEXT       (1u << 7)  /* also build flow_to_next (smooth_type == 'on_flow_to_next') focus.py:170-178 */
#define MPC_F_NO_WARP         (1u << 8)  /* splat events at their own (y,x): imager.create_iwe on raw events, logging.py:76-79 */
#define MPC_F_UNIT_WEIGHT     (1u << 9)  /* weight = 1.0 for every row (create_iwe default weight) */
#define MPC_F_ATOMIC_PATH     (1u << 10) /* debugging: plain global-atomic kernels instead of the LDS-tiled ones */
#define MPC_F_NO_BWD_RECORDS  (1u << 11) /* forward only: mpc_event_splat_fwd need not keep records for _bwd */

/* argument errors */
#define MPC_E_NULL      (-1)
#define MPC_E_SHAPE     (-2)
#define MPC_E_WORKSPACE (-3)
#define MPC_E_UNSUPPORTED (-4)

typedef struct mpc_shape {
    int32_t B;      /* samples in the batch                                   */
    int32_t M;      /* padded events per sample                               */
    int32_t Mp;     /* num_pos_events: rows [0,Mp) positive, [Mp,M) negative  */
    int32_t nb;     /* num_bins                                               */
    int32_t T;      /* num_tref                                               */
    int32_t H, W;   /* image_shape                                            */
    int32_t sp;     /* lut_superpixel_size                                    */
    int32_t hq, wq; /* LUT grid = ceil(H/sp) x ceil(W/sp)                     */
    int32_t n;      /* trajectories per sample                                */
    int32_t K;      /* num_knn                                                */
    uint32_t flags; /* MPC_F_*                                                */
} mpc_shape;

/* scalars written by mpc_contrast_fwd / mpc_lut_smooth / mpc_finalize into `scal` (device,
 * MPC_SCAL_COUNT floats) */
#define MPC_SCAL_LOSS     0   /* focus + smooth                    focus.py:94      */
#define MPC_SCAL_FOCUS    1   /* 1 / val                           loss.py:12       */
#define MPC_SCAL_SMOOTH   2   /* smooth_weight * smoothness        focus.py:246     */
#define MPC_SCAL_VAL      3   /* contrast value                    loss.py:14-27    */
#define MPC_SCAL_GCOEF    4   /* d focus / d raw-IWE = GCOEF * grad_iwe_unscaled    */
#define MPC_SCAL_COUNT    8

int mpc_version(void);
const char *mpc_last_error_string(void);
/* Bounds-checked debug build (-DMPC_BOUNDS, csrc/bounds.h; the reference has no sanitizer hooks): waits for
 * the device, returns the number of out-of-range indices the checked accessors of the library's kernels recorded since the
 * last call (and clears the records; the first one's file:line, workgroup, index, extent -> mpc_last_error_string()).
 * 0 = clean; -1 = this library is the product build (the accessors compile to the bare index).                            */
int mpc_bounds_check(void);

/* Diagnostics (bench.py's instrumented pass; no reference counterpart): between mpc_profile_start() and mpc_profile_stop()
 * every kernel the library launches is bracketed by two HIP events on its launch stream.  mpc_profile_stop synchronises
 * on them and returns the number of launches recorded: their kernel names, newline separated, in `names` (names_cap
 * bytes) and their durations in milliseconds in `ms` (cap entries), in launch order.  Process-wide; not for use inside
 * a graph capture. */
int mpc_profile_start(void);
int mpc_profile_stop(char *names, int32_t names_cap, float *ms, int32_t cap);

/* Bytes of scratch needed by any call with this shape. */
int64_t mpc_workspace_bytes(const mpc_shape *s);

/* ---- A5: KNN flow look-up table (focus.py:115-180) --------------------------------------
 * traj      [B][T+nb][n][2]  (y,x); rows [0,T) are the reference times, [T,T+nb) the bin mids
 * flow_lut  [B][nb][hq][wq][T][2]                                   (out)
 * flow_next [B][nb-1][hq][wq][1][2]  or NULL unless MPC_F_WANT_NEXT (out)
 * knn_state mpc_knn_state_floats(s) floats = [3][B][nb][hq*wq] + [B][nb][ceil(hq/16)*ceil(wq/16)][5] : K-th distance
 *           (f32), K-th index (i32 bits; bit 30 set if a point at exactly the K-th distance was excluded by the index
 *           tie rule), iwd normaliser, then the largest K-th distance of every 16x16 cell tile per class of query
 *           (inner / within r+1 cells of the top, bottom, left, right border) (out; consumed by _bwd)
 * idx_out   [B][nb][hq*wq][K] int32, ascending by (distance, index), or NULL (debug/tests)  */
int mpc_knn_lut_fwd(const mpc_shape *s, const float *traj, float *flow_lut, float *flow_next,
                    float *knn_state, int32_t *idx_out, void *ws, void *stream);

/* Number of floats of the knn_state buffer of mpc_knn_lut_fwd / _bwd for this shape. */
int64_t mpc_knn_state_floats(const mpc_shape *s);

/* Diagnostics: byte offset, inside the workspace, of the list of queries that mpc_knn_lut_fwd's strip kernel handed to
 * its per-query fallback: int32 count, then count entries (query id = ((b*nb + bin)*hq + cy)*wq + cx in the low 30 bits,
 * reason in the top 2: 0 fewer than K candidates inside the first search square, 1 more candidate slots than the
 * fast path holds, 2 more keys at the K-th distance level than it ranks, 3 staging overflow).  Read it after
 * mpc_knn_lut_fwd (tests assert that the fast path serves nearly all queries of the benchmark shapes). */
int64_t mpc_knn_fail_list_offset(const mpc_shape *s);
/* Diagnostics: byte offsets of the other work lists of the KNN forward inside the workspace: out[0] strips searched again in
 * quarters, [1] strips with far queries or queries to search again, [2] / [3] the maps of such queries (bit per query; those
 * that need more rings), [4] far lists per (sample, bin), [5] tiles for the far queries' backward; -1: no such list. */
int mpc_knn_list_offsets(const mpc_shape *s, int64_t *out);
/* Diagnostics: byte offset of the counters of the KNN forward's tail launch inside the workspace (int32 each, read them after
 * mpc_knn_lut_fwd): +0 queries the main launch marked for the tail's strip workgroups (at most 1 024: the tail took them from the
 * marked list with its one-wavefront search and ran no far pass), +128 entries of the late list, +256 strip workgroups that counted
 * themselves done.  Tests assert which path a launch took. */
int64_t mpc_knn_tail_counters_offset(const mpc_shape *s);

/* Backward of the above w.r.t. traj (indices carry no gradient; 'iwd' weights are constants -- since version 107 the tile gather serves
 * 'iwd' too: a member's weight is one hardware reciprocal of (d + 1e-9), 1 ulp, on the query's gradient divided by its normaliser --,
 * focus.py:157-163).  grad_flow_next may be NULL.  grad_traj [B][T+nb][n][2] is overwritten. */
int mpc_knn_lut_bwd(const mpc_shape *s, const float *traj, const float *grad_flow_lut,
                    const float *grad_flow_next, const float *knn_state, float *grad_traj,
                    void *ws, void *stream);

/* ---- A6-A8: warp + weights + bilinear vote (focus.py:182-230, event_image_converter.py:333-391)
 * events   [B][M][6], flow_lut [B][nb][hq][wq][T][2], t_ref [T] (device)
 * iwe_raw  [B*T][P][H][W]  P = 2 with MPC_F_POLARITY_SPLIT else 1     (out, overwritten)   */
int mpc_event_splat_fwd(const mpc_shape *s, const float *events, const float *flow_lut,
                        const float *t_ref, float *iwe_raw, void *ws, void *stream);

/* ---- event-axis sharding ("optional finer split": a batch smaller than the number of ranks; the reference
 * has batch DDP only, scripts/flow_training.py:125-128).  mpc_event_splat_fwd_fixed = mpc_event_splat_fwd on THIS rank's
 * rows of `events`, but the output is the image's Q33.30 fixed-point accumulators themselves (int64 [B*T][P][H][W]).
 * Integer partial images add up exactly and in any order: all-reduce(SUM) them across the ranks, then
 * mpc_iwe_from_fixed converts to the fp32 raw IWE -- bit for bit the image one rank computes from all the events.
 * The backward per rank is mpc_event_splat_bwd on its own rows (a partial dL/dLUT; summed across ranks in fp32). */
int mpc_event_splat_fwd_fixed(const mpc_shape *s, const float *events, const float *flow_lut, const float *t_ref,
                              int64_t *iwe_fixed, void *ws, void *stream);
int mpc_iwe_from_fixed(const int64_t *iwe_fixed, float *iwe_raw, int64_t count, void *stream);

/* ---- A8 blur + A9 objective (event_image_converter.py:170-175, loss.py:4-27,58-87)
 * iwe_blur [B*T][P][H][W] (out) ; grad_iwe [B*T][P][H][W] or NULL (out: UNSCALED adjoint image
 * Blur^T Sobel^T u, or Blur^T (x - mean) for the variance objective; multiply by
 * scal[MPC_SCAL_GCOEF] to get d focus_loss / d iwe_raw).  Partial sums go to ws; call
 * mpc_finalize afterwards.                                                                  */
int mpc_contrast_fwd(const mpc_shape *s, const float *iwe_raw, float *iwe_blur, float *grad_iwe,
                     void *ws, void *stream);

/* ---- A10: smoothness of a flow field (focus.py:232-246, loss.py:29-56)
 * field [nimg][hq][wq][C] (the LUT layout: nimg = B*nb, C = 2*T; or flow_next: nimg = B*(nb-1),
 * C = 2).  grad_field (same shape, or NULL) receives d smoothness_loss / d field INCLUDING
 * smooth_weight.  Partial sums go to ws; call mpc_finalize afterwards.                       */
int mpc_lut_smooth(const mpc_shape *s, const float *field, int32_t nimg, int32_t C,
                   float smooth_weight, float *grad_field, void *ws, void *stream);

/* Reduce the partial sums of mpc_contrast_fwd and (if smooth_nimg > 0: the nimg, C, weight of
 * the preceding mpc_lut_smooth call) in fp64 and write scal[MPC_SCAL_*] (device).            */
int mpc_finalize(const mpc_shape *s, int32_t smooth_nimg, int32_t smooth_C, float smooth_weight,
                 float *scal, void *ws, void *stream);

/* ---- A11: backward of A6-A8 into the LUT.
 * grad_flow_lut[b][bin][iy][ix][tref][:] = grad_out * (scal[GCOEF] * sum over the cell's events of
 * w * bilinear-gradient of grad_iwe at the warped position  +  add_term[...]).
 * add_term: same shape as the LUT or NULL (used to fold in the smoothness gradient of
 * mpc_lut_smooth).  grad_out: device scalar or NULL (= 1).  Must follow mpc_event_splat_fwd on the
 * same workspace (the LDS-tiled path reuses the records that call left there).                 */
/* ---- UNPINNED EXTENSION (no reference code: the reference gathers a binned flow LUT, focus.py:182-195): gradient of the
 * objective with respect to the warped position of every event, i.e. autograd of create_iwe (event_image_converter.py:333-391)
 * through `warped = differences + events[..., :2]` (focus.py:191), for rows whose (y, x) columns hold positions that are warped
 * already (s->flags must carry MPC_F_NO_WARP; num_tref == 1).  grad_pos[b][i][:] = grad_out * scal[GCOEF] * w * bilinear
 * gradient of grad_iwe at the row's position, 0 for rows of weight 0.  Follows mpc_event_splat_fwd / mpc_contrast_fwd /
 * mpc_finalize of the same rows.  Used by FocusLoss.calc_per_event_basis (per-event continuous-time basis warp).          */
int mpc_event_pos_grad(const mpc_shape *s, const float *events, const float *t_ref, const float *grad_iwe,
                       const float *scal, const float *grad_out, float *grad_pos, void *stream);
/* Same extension, fused: mpc_pe_warp writes the rows with the per-event continuous-time basis warp applied --
 *   rows_out[b][i] = (y + sum_j coef[cell][0][j] phi[b][i][j], x + sum_j coef[cell][1][j] phi[b][i][j], t, p, cell, valid),
 * cell = the LUT cell of the event's own position (focus.py:186-187), coef_rows [B*hq*wq][2][k] the tile coefficients
 * (trajectories.py:15-52), phi [B][M][k] = basis_j(t_ref) - basis_j(t_event) (basis.py:18-31), or phi = NULL: the polynomial basis
 * t^(j+1) (basis.py:26-27) worked out in the kernels from t_ref[0] and the rows' time column, k <= 8 -- for mpc_event_splat_fwd with
 * MPC_F_NO_WARP; mpc_pe_grad is its backward: grad_coef_rows (zeroed here) += phi * d objective / d warped position, with float
 * atomics (the gradient of this extension is not bitwise reproducible).                                                    */
int mpc_pe_warp(const mpc_shape *s, const float *events, const float *coef_rows, const float *phi, int32_t k,
                const float *t_ref, float *rows_out, void *stream);
int mpc_pe_grad(const mpc_shape *s, const float *rows, const float *phi, int32_t k, const float *t_ref, const float *grad_iwe,
                const float *scal, const float *grad_out, float *grad_coef_rows, void *stream);
/* mpc_pe_grad for bucket-ordered events (mpc_event_bucket_order / mpc_ingest_scatter_ordered and their `offsets` table): no
 * global atomics -- every LUT strip of a sample accumulates in LDS fixed point (bitwise reproducible) and is written once.
 * `split` workgroups share the row ranges of a strip and write partial results: grad_coef_rows is [split][B*hq*wq][2][k], to be
 * summed over its first axis.  MPC_E_UNSUPPORTED if a strip's [cell][2][k] accumulators exceed the LDS (then: mpc_pe_grad).  */
int mpc_pe_grad_ordered(const mpc_shape *s, const float *rows, const int32_t *offsets, const float *phi, int32_t k,
                        const float *t_ref, const float *grad_iwe, const float *scal, const float *grad_out,
                        float *grad_coef_rows, int32_t split, void *stream);
/* 1 where mpc_pe_grad_ordered serves (shape, k), 0 where the caller takes mpc_pe_grad: the rule lives in the library only. */
int32_t mpc_pe_grad_ordered_supported(const mpc_shape *s, int32_t k);
/* Same extension: the dense <-> per-tile operators around the two calls above (csrc/tiles.hip; version 107: kernels instead of two
 * dozen torch operators -- the per-event step was bound by the host).
 *   mpc_pe_tile_rows      coef_rows [B*hq*wq][c2] = sum over the S scales of grid [B][S][c2][H][W] at the tile centres
 *                         (y, x) = (iy * tile + tile / 2, ix * tile + tile / 2): get_optical_flow_tile_mask + coeffs_grid_to_list,
 *                         src/utils/trajectories.py:3-52, scales summed as compute_basis does (basis.py:29-31); hq = ceil(H / tile)
 *   mpc_pe_tile_rows_bwd  its adjoint: EVERY element of grad_grid [B][S][c2][H][W] is written (0 off the tile centres)
 *   mpc_pe_basis_field    field [B*nb][G][2] = sum_j coef_rows[(b, cell)][d][j] * phim[t][j] -- the flow from the bin mid-times to
 *                         t_ref per tile, the field the smoothness term takes (mpc_lut_smooth: nimg = B * nb, C = 2); k <= 8, nb <= 64
 *   mpc_pe_rows_grad_finish  grad_coef_rows [B*G][2][k] = sum over the `split` partial results of mpc_pe_grad_ordered (split = 1: of
 *                         mpc_pe_grad) + grad_out[0] * (adjoint of mpc_pe_basis_field applied to grad_field, or nothing if NULL)  */
int mpc_pe_tile_rows(const float *grid, float *coef_rows, int32_t B, int32_t S, int32_t c2, int32_t H, int32_t W, int32_t tile, void *stream);
int mpc_pe_tile_rows_bwd(const float *grad_rows, float *grad_grid, int32_t B, int32_t S, int32_t c2, int32_t H, int32_t W, int32_t tile, void *stream);
int mpc_pe_basis_field(const float *coef_rows, const float *phim, float *field, int32_t B, int32_t G, int32_t k, int32_t nb, void *stream);
int mpc_pe_rows_grad_finish(const float *grad_parts, int32_t split, const float *grad_field, const float *phim, const float *grad_out,
                            float *grad_coef_rows, int32_t B, int32_t G, int32_t k, int32_t nb, void *stream);

int mpc_event_splat_bwd(const mpc_shape *s, const float *events, const float *flow_lut,
                        const float *t_ref, const float *grad_iwe, const float *scal,
                        const float *grad_out, float *grad_flow_lut, const float *add_term,
                        void *ws, void *stream);

/* ---- A4: the whole of FocusLoss.calc (focus.py:66-113) and of its backward as ONE call each: the same launch
 * sequence the stage entry points above issue (KNN LUT -> smoothness -> warp + vote -> blur + objective -> scalars;
 * event backward -> [scale] -> KNN backward), from C, so that a step costs two host calls instead of sixteen.
 * All buffers are caller-owned device memory (shapes as documented at the stage entry points):
 *   flow_next / smooth_grad : NULL unless used (MPC_F_WANT_NEXT; smooth_weight > 0 and the backward is wanted)
 *   grad_iwe                : NULL for a forward-only call (then also set MPC_F_NO_BWD_RECORDS)
 * smooth_weight > 0 applies the smoothness term to flow_next when MPC_F_WANT_NEXT is set, else to flow_lut
 * (focus.py:232-246).  mpc_focus_bwd must follow mpc_focus_fwd on the same buffers and workspace;
 * grad_lut_scratch [like flow_lut] and grad_next_scratch [like flow_next, or NULL] are scratch (version 107: the tile gather and the far
 * backward multiply the saved smoothness gradient by grad_out as they read it; grad_next_scratch is only written where the general
 * gather serves the shape -- num_tref > 1 -- but must still be given with smoothness on flow_to_next and a grad_out). */
typedef struct mpc_focus_buffers {
    const float *traj;        /* [B][T+nb][n][2]                      in  */
    const float *events;      /* [B][M][6]                            in  */
    const float *t_ref;       /* [T]                                  in  */
    float *flow_lut;          /* [B][nb][hq][wq][T][2]                out */
    float *flow_next;         /* [B][nb-1][hq][wq][1][2] or NULL      out */
    float *knn_state;         /* mpc_knn_state_floats(s)              out */
    float *smooth_grad;       /* like the smoothed field, or NULL     out */
    float *iwe_raw;           /* [B*T][P][H][W]                       out */
    float *iwe_blur;          /* [B*T][P][H][W]                       out */
    float *grad_iwe;          /* [B*T][P][H][W] or NULL               out */
    float *scal;              /* [MPC_SCAL_COUNT]                     out */
    float smooth_weight;
    const int32_t *event_offsets;  /* offsets table of mpc_event_bucket_order if `events` is ordered, else NULL   in  */
    float *scal_out;          /* [3] loss, focus, smooth once more, or NULL: the copy a caller hands out while `scal` stays saved for
                                 the backward (round 5: spares the host a copy kernel per step)                  out */
} mpc_focus_buffers;
int mpc_focus_fwd(const mpc_shape *s, const mpc_focus_buffers *io, void *ws, void *stream);
int mpc_focus_bwd(const mpc_shape *s, const mpc_focus_buffers *io, const float *grad_out,
                  float *grad_lut_scratch, float *grad_next_scratch, float *grad_traj, void *ws, void *stream);

/* ---- next row, layout half: a bucketed event layout handed from ingest to the loss.
 * mpc_event_bucket_order orders the rows of every polarity block of events [B][M][6] by (time bin, LUT strip), the key
 * of the backward buckets (it does not depend on the flow); padding rows stay at the end of their block.  Row order
 * inside a polarity block does not change the loss -- bit for bit -- so events_out is a valid `events` tensor anywhere.
 *   offsets [B][2][nb * S + 1] int32 (out), S = mpc_event_lut_strips(s): first row of bucket bin * S + strip of the
 *   block (strip = LUT row / ceil(hq / S), LUT row = clamp(int(y // sp), 0, hq - 1)), last entry = first padding row;   ws: mpc_event_order_workspace_bytes(s) bytes.
 * With the offsets, mpc_event_splat_fwd may be called with MPC_F_NO_BWD_RECORDS (it then writes no record per event
 * for the backward) and mpc_event_splat_bwd_ordered reads the event rows themselves (mpc_focus_buffers.event_offsets
 * does both). */
int32_t mpc_event_lut_strips(const mpc_shape *s);
int64_t mpc_event_order_workspace_bytes(const mpc_shape *s);
int mpc_event_bucket_order(const mpc_shape *s, const float *events_in, float *events_out, int32_t *offsets,
                           void *ws, void *stream);
int mpc_event_splat_bwd_ordered(const mpc_shape *s, const float *events, const int32_t *offsets, const float *flow_lut,
                                const float *t_ref, const float *grad_iwe, const float *scal, const float *grad_out,
                                float *grad_flow_lut, const float *add_term, void *ws, void *stream);

/* UNPINNED EXTENSION, default off (FocusLoss(pyramid_levels > 1)): helpers of an IWE pyramid, which BASELINE.json's
 * configs[2] names and the reference does not have.  mpc_pool2_fwd: out [nimg][H/2][W/2] = 2x2 average of in [nimg][H][W].
 * mpc_pool2_bwd_add: big[y][x] += (coef_small[0] / coef_big[0]) * 0.25 * small[y/2][x/2] (the adjoint of the pooling applied
 * to adjoint images that are kept in units of their own level's MPC_SCAL_GCOEF; both coefficients are device scalars). */
int mpc_pool2_fwd(const float *in, float *out, int32_t nimg, int32_t H, int32_t W, void *stream);
int mpc_pool2_bwd_add(const float *small, const float *coef_small, float *big, const float *coef_big, int32_t nimg,
                      int32_t H, int32_t W, void *stream);

/* y[i] = a[0] * x[i] (device scalar a; used to scale the smoothness gradient by grad_out). */
int mpc_scale(const float *x, const float *a, float *y, int64_t count, void *stream);

/* ---- next row: voxel-grid builder, reference src/loader/dsec/utils.py:29-77
 * (VoxelGrid.convert) as called from src/loader/dsec/loader.py:133-139.
 * xytp   [B][N][4] : x, y, t, p per raw event (p in {0,1}; t increasing within a sample; rows beyond
 *                    counts[b] are ignored)                        counts [B] int32 (device)
 * grid   [B][C][H][W] (out, overwritten).  norm: 0 none, 1 'mean_std', 2 'max'; quantile clipping before the
 * normalisation as in the reference (config/exe/flow_training/dsec.yaml uses quantile: 0). */
typedef struct mpc_vox_shape {
    int32_t B, N, C, H, W, norm;
    float quantile;   /* 0 <= quantile < 0.15: clip at the (1 - quantile) quantile of |grid| per sample (utils.py:57-61); 0 = off */
    float keep;       /* fp32(1 - quantile) with the subtraction done in DOUBLE by the caller, as `torch.quantile(x, 1 - q)` receives
                         it in the reference (utils.py:58): fp32(1 - fp64(q)) and 1 - fp32(q) differ by an ulp for ~6 % of the q.
                         0 = derive it from `quantile` */
} mpc_vox_shape;
int64_t mpc_voxel_workspace_bytes(const mpc_vox_shape *s);
int mpc_voxel_grid(const mpc_vox_shape *s, const float *xytp, const int32_t *counts, float *grid,
                   void *ws, void *stream);

/* ---- next row: event ingest, reference src/loader/dsec/loader.py:152-167 (time
 * normalisation, bin index, in-image filter, polarity split) + :360-415 (pad_events, sequence_collate_fn).
 * x, y, p [B][N] float32, t_us [B][N] int64 (increasing within a sample), counts [B] int32 (device).
 * mpc_ingest_count  -> out_max[2] (device): largest number of valid positive / negative events of a sample;
 *                      the caller reads them to size `events`  = [B][max_pos + max_neg][6]
 * mpc_ingest_scatter-> events (zero padded, valid flag in column 5, positive block first) and, if xytp is
 *                      not NULL, [B][N][4] = (x, y, t, p) as loader.py:135-138 hands them to the voxel grid. */
typedef struct mpc_ingest_shape {
    int32_t B, N, H, W, nb;
} mpc_ingest_shape;
int64_t mpc_ingest_workspace_bytes(const mpc_ingest_shape *s);
int mpc_ingest_count(const mpc_ingest_shape *s, const float *x, const float *y, const int64_t *t_us,
                     const float *p, const int32_t *counts, int32_t *out_max, void *ws, void *stream);
int mpc_ingest_scatter(const mpc_ingest_shape *s, const float *x, const float *y, const int64_t *t_us,
                       const float *p, const int32_t *counts, int32_t max_pos, int32_t max_neg,
                       float *events, float *xytp, void *ws, void *stream);

/* Ingest straight into the bucket-ordered layout: the outputs of mpc_ingest_scatter followed by mpc_event_bucket_order for
 * the loss shape `loss` (its B, nb, H, W, sp, hq, wq, flags; M = max_pos + max_neg, Mp = max_pos), without the extra pass
 * over the event tensor -- the (time bin, LUT strip) of an event is known when its row is written.  offsets
 * [B][2][nb * S + 1] as documented at mpc_event_bucket_order.  ws: the workspace mpc_ingest_count used (unchanged since);
 * ws_order: mpc_ingest_ordered_workspace_bytes(s, loss) bytes. */
int64_t mpc_ingest_ordered_workspace_bytes(const mpc_ingest_shape *s, const mpc_shape *loss);
int mpc_ingest_scatter_ordered(const mpc_ingest_shape *s, const mpc_shape *loss, const float *x, const float *y,
                               const int64_t *t_us, const float *p, const int32_t *counts, int32_t max_pos, int32_t max_neg,
                               float *events, int32_t *offsets, float *xytp, void *ws, void *ws_order, void *stream);

/* ---- next row: dense flow from tile trajectories and flow error metrics.
 * mpc_dense_flow = reference src/utils/flow.py:12-16 (dense_flow_from_traj): list_to_grid
 *   (src/utils/trajectories.py:54-76) at pixel_positions // patch, then the anti-aliased bicubic resize
 *   of src/utils/flow.py:9-10 to H x W.
 *   traj_flow [B][n][C], pixel_positions [n][2] int64 (y, x)
 *   patch_flow [B][C][H/patch][W/patch] (out), dense [B][C][H][W] (out)
 * mpc_flow_error = reference src/utils/flow.py:18-70 (calculate_flow_error).
 *   flow_gt, flow_pred [B][2][H][W]; event_mask [B][H][W] bytes (non-zero = true) or NULL;
 *   time_scale [B] or NULL; out[5] (device) = EPE, 1PE, 2PE, 3PE, AE (degrees). */
typedef struct mpc_flow_shape {
    int32_t B, C, n, patch, H, W;
} mpc_flow_shape;
int mpc_dense_flow(const mpc_flow_shape *s, const float *traj_flow, const int64_t *pixel_positions,
                   float *patch_flow, float *dense, void *stream);
typedef struct mpc_err_shape {
    int32_t B, H, W;
} mpc_err_shape;
int64_t mpc_flow_error_workspace_bytes(const mpc_err_shape *s);
int mpc_flow_error(const mpc_err_shape *s, const float *flow_gt, const float *flow_pred,
                   const uint8_t *event_mask, const float *time_scale, float *out, void *ws, void *stream);

/* ---- next row + BASELINE.json configs[3]: flow curves sampled at the tile centres as `trajectories` for the loss.
 * Reference: src/models/raft_spline/curves/base.py:88-123 (CurveBase.get_flow_from_reference: flow(t) = sum_k B_k(t) P_k, P_0 == 0),
 * bezier.py:92-113 (Bernstein basis, evaluated on the host in float64 as the reference does), polynomial.py:60-61 (dim 1 = (x, y)).
 *   params [B][2][d][n]  control points 1..d of every tile, channel 0 = x, channel 1 = y (the network's [B, 2d, h, w] output, n = h*w)
 *   basis  [T][d]        basis functions at the T reconstruction times (Bernstein: pinned by bezier.py; clamped cubic B-spline:
 *                        UNPINNED extension -- the reference has no such curve)
 *   pos    [n][2]        tile centres (y, x) (src/utils/trajectories.py:3-13)
 *   traj   [B][T][n][2]  (out)  (y, x) = pos + scale * sum_k basis[t][k] * (P_k.y, P_k.x)
 * mpc_curve_traj_bwd: grad_params [B][2][d][n] (out, overwritten) = the adjoint of the above applied to grad_traj [B][T][n][2].  */
int mpc_curve_traj_fwd(const float *params, const float *basis, const float *pos, float scale, float *traj,
                       int32_t B, int32_t d, int32_t T, int32_t n, void *stream);
int mpc_curve_traj_bwd(const float *grad_traj, const float *basis, float scale, float *grad_params,
                       int32_t B, int32_t d, int32_t T, int32_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCMAX_H */
