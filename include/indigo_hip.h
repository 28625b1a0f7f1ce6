/*
 * indigo_hip.h -- C ABI of libindigo_hip.so, the MI355X (gfx950) leaf-kernel
 * library behind the `indigo.backends.Backend` plugin surface.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes / scalars and
 * returns an int status (0 = IG_OK).  On failure a human-readable message is
 * available from ig_last_error().  All device work is enqueued asynchronously
 * on the context's HIP stream; only the entry points documented as
 * "synchronous" wait for the device.
 *
 * Each group cites the reference interface it replaces (paths relative to the
 * mbdriscoll/indigo tree).  Panels (X, Y) are column-major with an explicit
 * leading dimension in ELEMENTS, exactly like the reference's
 * `dndarray._leading_dim` (indigo/backends/backend.py:50).
 *
 * Complex numbers are interleaved (re, im) float32 pairs ("complex64").
 */
#ifndef INDIGO_HIP_H
#define INDIGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IG_ABI_VERSION 1

/* status codes */
#define IG_OK            0
#define IG_ERR_HIP       1   /* a HIP runtime call failed                    */
#define IG_ERR_ARG       2   /* invalid argument (shape, alignment, NULL)    */
#define IG_ERR_NODEVICE  3   /* no usable gfx950 device                      */
#define IG_ERR_UNSUPPORTED 4 /* valid request this build cannot serve        */
#define IG_ERR_NOMEM     5

/* copy kinds for ig_copy2d (values follow cudaMemcpyKind / hipMemcpyKind,
 * the enum the reference passes at indigo/backends/cuda.py:111-115)         */
#define IG_H2D 1
#define IG_D2H 2
#define IG_D2D 3

typedef struct ig_ctx   ig_ctx;    /* one device + one stream                */
typedef struct ig_fft   ig_fft;    /* batched C2C FFT plan                   */
typedef struct ig_event ig_event;  /* timing event on the context's stream   */
typedef struct ig_comm  ig_comm;   /* RCCL communicator of one rank (one GPU) */
typedef struct ig_graph ig_graph;  /* a recorded sequence of launches on the context's stream (hipGraph) */

/* ------------------------------------------------------------------------
 * Context.  Replaces the handle/bring-up code of CudaBackend.__init__
 * (indigo/backends/cuda.py:28-38: cudaSetDevice, cublas/cusparse handles)
 * and Backend.barrier (cuda.py:120-121).
 * ---------------------------------------------------------------------- */
int  ig_abi_version(void);
int  ig_device_count(int* count);                   /* never fails loudly: count=0 if no GPU */
int  ig_init(int device_id, ig_ctx** out);          /* creates its own non-blocking stream   */
int  ig_init_on_stream(int device_id, void* hip_stream, ig_ctx** out); /* adopt caller's stream (NULL = legacy default stream) */
void ig_destroy(ig_ctx* ctx);
const char* ig_last_error(ig_ctx* ctx);             /* ctx may be NULL: error of the last failed call on this thread */
int  ig_sync(ig_ctx* ctx);                          /* synchronous: waits for the stream     */
void* ig_stream(ig_ctx* ctx);                       /* the hipStream_t, for interop          */
int  ig_device_name(ig_ctx* ctx, char* buf, size_t len);
int  ig_mem_info(ig_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Plan options, read when a plan is made.  "fft.kernels": 0 = every kernel (default), 1 = no register-resident A x B passes
 * (160 ... 640-point axes take the multi-stage LDS kernel), 2 = only the one-stage-per-launch generic kernel -- the
 * fallback kernels stay testable on sizes the fast ones would take.  Unknown names are an error.                          */
int  ig_set_option(ig_ctx* ctx, const char* name, int64_t value);
/* Device memory the library holds on its own for this context (the SpMM kernels' repacked-panel buffer, deferred-row lists,
 * reduction scratch, solver scalars): what Backend.mem_usage() adds to the arrays the caller allocated.                    */
int  ig_library_bytes(ig_ctx* ctx, size_t* bytes);

/* ------------------------------------------------------------------------
 * Device memory.  Replaces CudaBackend.dndarray._malloc/_free/_zero/_copy*
 * (indigo/backends/cuda.py:126-181: 256-byte aligned cudaMalloc, cudaMemset,
 * cudaMemcpy2D pitch copies).  Ownership stays with the caller.
 * ---------------------------------------------------------------------- */
int  ig_malloc(ig_ctx* ctx, size_t nbytes, void** dptr);      /* >=256-B aligned; nbytes==0 gives a valid unique pointer */
int  ig_free(ig_ctx* ctx, void* dptr);                         /* synchronous w.r.t. the stream */
/* Writes zeros over the buffer as 512 rows of 256-byte segments nbytes / 512 apart and reports the fastest of three passes in ms:
 * how well THIS allocation serves passes that step megabytes per element (no reference counterpart; the backend keeps the best of a
 * few candidates for its large arrays).  Destroys the contents.  *ms = 0 for buffers too small to have such a pattern. */
int  ig_probe_placement(ig_ctx* ctx, void* dptr, size_t nbytes, double* ms);
int  ig_memset0(ig_ctx* ctx, void* dptr, size_t nbytes);
/* 2-D strided copy of `height` rows of `width_bytes` bytes.  H2D and D2H are
 * synchronous (host buffer is pageable numpy memory); D2D is asynchronous.  */
int  ig_copy2d(ig_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch,
               size_t width_bytes, size_t height, int kind);

/* ------------------------------------------------------------------------
 * Timing events on the context's stream (hipEvent).  Counterpart of the
 * barrier+wallclock `profile` context manager (indigo/util.py:33-80) without
 * its forced device syncs.
 * ---------------------------------------------------------------------- */
int  ig_event_create(ig_ctx* ctx, ig_event** out);
int  ig_event_record(ig_event* ev);
int  ig_event_elapsed_ms(ig_event* start, ig_event* stop, float* ms);   /* synchronous on `stop` */
int  ig_event_destroy(ig_event* ev);

/* Recorded launch sequences (HIP graphs).  The reference's solver loop (Backend.cg, indigo/backends/backend.py:666-686) issues
 * the same ~25 dependent launches every iteration -- one operator evaluation and the vector updates --, each paying the host's
 * launch path; recorded once and replayed as ONE graph launch the gaps between them shrink to the device's own.
 *   ig_graph_begin : every launch made through this context from now on is RECORDED on its stream, not executed.  Calls that
 *                    synchronise or allocate (ig_sync, ig_malloc, host reads, a first-use format build) are errors while recording:
 *                    run the sequence once unrecorded first.  Profile mode must be off.
 *   ig_graph_end   : stops recording and instantiates the graph; ig_graph_abort : stops recording and drops what was recorded
 *   ig_graph_launch: enqueues the whole sequence on the context's stream (same buffers, same scalars as when it was recorded)  */
int  ig_graph_begin(ig_ctx* ctx);
int  ig_graph_end(ig_ctx* ctx, ig_graph** out);
int  ig_graph_abort(ig_ctx* ctx);
int  ig_graph_launch(ig_graph* graph);
int  ig_graph_destroy(ig_graph* graph);

/* Profile mode: while enabled, every kernel launch made through this context
 * is bracketed by two events on the stream (no host sync).  ig_prof_report
 * synchronises, writes one line per kernel name
 *     "<name> <launches> <total_ms> <algorithmic_bytes>\n"
 * into buf (truncated to len), and clears the records.  This is how bench.py
 * measures a kernel's average launch duration live (the reference's `profile`
 * hooks, indigo/operators.py:259,334,353, carry the same bytes model).       */
int  ig_prof_enable(ig_ctx* ctx, int on);
int  ig_prof_report(ig_ctx* ctx, char* buf, size_t len);

/* ------------------------------------------------------------------------
 * BLAS-1 glue.  Replaces Backend.axpby/scale/dot/norm2/max
 * (indigo/backends/backend.py:453-467,734; numpy oracle np.py:53-74,141-145;
 *  CUDA: cublasCscal+cublasCaxpy cuda.py:239-248, cublasCdotc :260-277,
 *  cublasScnrm2 :279-296, cu_max _customgpu.cu:7-13).
 * n counts complex elements.  beta == 0 means y is not read (BLAS rule).
 * ---------------------------------------------------------------------- */
int  ig_caxpby(ig_ctx* ctx, int64_t n, float beta_re, float beta_im, void* y,
               float alpha_re, float alpha_im, const void* x);              /* y = beta*y + alpha*x */
int  ig_cscal(ig_ctx* ctx, int64_t n, float alpha_re, float alpha_im, void* x); /* x *= alpha */
int  ig_cdotc(ig_ctx* ctx, int64_t n, const void* x, const void* y, double out[2]); /* sum conj(x)*y ; synchronous */
int  ig_scnrm2sq(ig_ctx* ctx, int64_t n, const void* x, double* out);       /* ||x||_2^2 ; synchronous */
int  ig_cmax(ig_ctx* ctx, int64_t nfloats, float val, void* arr);           /* arr[i] = max(arr[i], val) over floats */
/* Device-resident solver scalars: the same reductions and updates with their scalar results / factors left in
 * device memory, so that a Krylov iteration (Backend.cg, indigo/backends/backend.py:666-686: alpha = rr / <p, Ap>,
 * beta = r2 / rr) enqueues without a single host synchronisation.  ig_scalars hands out the context's block of
 * doubles (zeroed); any device double pointers may be used.
 *   ig_cdotc_dev / ig_scnrm2sq_dev : results to d_out[0..1] / d_out[0]
 *   ig_scalar_ratio : *d_out = scale * *d_num / *d_den (0 when *d_den == 0)   ig_scalar_copy : d_dst[0..count) = d_src[0..count)
 *   ig_scalar_ratio_gated : the same, forced to 0 once *d_gate_num < gate_tol * *d_gate_den (a solver's step length after
 *                     its residual passed the tolerance: iterations enqueued beyond convergence become no-ops; the
 *                     reference breaks out of its loop on the host instead, backend.py:683-685)
 *   ig_caxpby_dev   : y = (beta_scale * *d_beta) * y + (alpha_scale * *d_alpha) * x   (a NULL pointer stands for 1)
 *   ig_scalar_read  : device -> host, synchronous (the residual history, every so many iterations)           */
int  ig_scalars(ig_ctx* ctx, double** d_slots, int* nslots);
int  ig_cdotc_dev(ig_ctx* ctx, int64_t n, const void* x, const void* y, double* d_out);
int  ig_scnrm2sq_dev(ig_ctx* ctx, int64_t n, const void* x, double* d_out);
int  ig_scalar_ratio(ig_ctx* ctx, double* d_out, const double* d_num, const double* d_den, double scale);
int  ig_scalar_ratio_gated(ig_ctx* ctx, double* d_out, const double* d_num, const double* d_den, double scale,
                           const double* d_gate_num, const double* d_gate_den, double gate_tol);
int  ig_scalar_copy(ig_ctx* ctx, double* d_dst, const double* d_src, int64_t count);
int  ig_scalar_read(ig_ctx* ctx, const double* d_src, int64_t count, double* host);
/* One CG iteration's vector work in three passes (reference loop: indigo/backends/backend.py:666-686):
 *   ig_cg_dot      Ap += lamda p (if lamda != 0) and the block partials of Re<p, Ap>
 *   ig_cg_step_r   alpha = rr / <p, Ap> -- 0 once rr < tol2 * r0, or when <p, Ap> == 0 --, r -= alpha Ap, block partials of ||r||^2;
 *                  *d_alpha = alpha
 *   ig_cg_step_xp  beta = r2 / rr, x += alpha p, p = r + beta p; *d_rr_next = r2 = ||r||^2 (a slot other than d_rr),
 *                  *d_hist = r2 / r0 (optional)
 * issued in this order with the same n on one context (they hand the block partials to each other through the context's
 * reduction scratch; every block of the consumer sums them itself: no separate reduction kernels, no host synchronisation). */
int  ig_cg_dot(ig_ctx* ctx, int64_t n, const void* p, void* Ap, float lamda);
int  ig_cg_step_r(ig_ctx* ctx, int64_t n, void* r, const void* Ap, const double* d_rr, const double* d_r0, double tol2, double* d_alpha);
int  ig_cg_step_xp(ig_ctx* ctx, int64_t n, void* x, void* p, const void* r, const double* d_alpha, const double* d_rr, double* d_rr_next,
                   const double* d_r0, double* d_hist);
int  ig_caxpby_dev(ig_ctx* ctx, int64_t n, const double* d_beta, float beta_scale, void* y,
                   const double* d_alpha, float alpha_scale, const void* x);
/* y(rows) = beta*y + alpha * sum_j X[:, j]  for a column-major rows x ncols panel: the coil
 * combination that VStack._eval_adjoint performs with one scale + ncols axpby-like passes
 * (indigo/operators.py:440-447), in one pass.                                              */
int  ig_csum_cols(ig_ctx* ctx, int64_t rows, int64_t ncols, const void* X, int64_t ldx,
                  float alpha_re, float alpha_im, float beta_re, float beta_im, void* y);
/* the same sum over a row-major (coil-interleaved) panel: y[k] = beta*y[k] + alpha * sum_j X_il[k*ncols + j] */
int  ig_csum_il(ig_ctx* ctx, int64_t rows, int64_t ncols, const void* X_il,
                float alpha_re, float alpha_im, float beta_re, float beta_im, void* y);

/* ------------------------------------------------------------------------
 * CSR x dense-panel SpMM.  Replaces Backend.ccsrmm
 * (indigo/backends/backend.py:514-519; oracle np.py:120-127;
 *  cusparseCcsrmm cuda.py:582-596; custom_ccc_csrmm _customcpu.c:14-114;
 *  cu_exw_csrmm_H _customgpu.cu:49-81,182-216).
 *
 *   adjoint == 0 :  Y(MxN) = alpha *  A   * X(KxN) + beta * Y
 *   adjoint != 0 :  Y(KxN) = alpha * A^H * X(MxN) + beta * Y
 *
 * A is M x K, CSR, 0-based int32 rowptr (M+1) / colind (nnz), complex64
 * values, sorted or unsorted columns.  X, Y column-major, leading dims in
 * elements.  The adjoint is a scatter over A's rows: with exwrite != 0 the
 * caller asserts every column of A holds at most one nonzero (plain stores,
 * the reference's `_exwrite` property, backend.py:555-567); otherwise float
 * atomics are used (correct, order-nondeterministic, slow).  For a fast
 * deterministic adjoint use ig_ccsrmm_t with a transposed copy.
 * ---------------------------------------------------------------------- */
int  ig_ccsrmm(ig_ctx* ctx, int adjoint, int exwrite,
               int64_t M, int64_t K, int64_t N, int64_t nnz,
               float alpha_re, float alpha_im,
               const void* vals, const int32_t* colind, const int32_t* rowptr,
               const void* X, int64_t ldx,
               float beta_re, float beta_im,
               void* Y, int64_t ldy);

/* Y(KxN) = alpha * A^H * X(MxN) + beta * Y  in GATHER form, given the CSR of
 * A^T (K rows, M columns; values NOT conjugated -- the kernel conjugates).
 * Same role as the reference's "store the transpose, wrap in Adjoint" recipe
 * (examples/pics.py:104-109).                                              */
int  ig_ccsrmm_t(ig_ctx* ctx,
                 int64_t M, int64_t K, int64_t N, int64_t nnz,
                 float alpha_re, float alpha_im,
                 const void* vals_t, const int32_t* colind_t, const int32_t* rowptr_t,
                 const void* X, int64_t ldx,
                 float beta_re, float beta_im,
                 void* Y, int64_t ldy);

/* ig_ccsrmm_t restricted to the support of a 3-D grid of output rows, row = kx + n0*(km + nm*ks), nm a
 * multiple of 16 and <= 512: only the 16-row segments (16 consecutive kx) flagged in the THIRD part of
 * the support table are computed and written; all other rows are left untouched (they hold no nonzero by
 * construction of the table; support may be NULL = all rows).  Same table as ig_fft_exec_cropped, which
 * reads exactly the flagged segments and nothing else.                                            */
int  ig_ccsrmm_t_grid(ig_ctx* ctx,
                      int64_t M, int64_t K, int64_t N, int64_t nnz,
                      float alpha_re, float alpha_im,
                      const void* vals_t, const int32_t* colind_t, const int32_t* rowptr_t,
                      const void* X, int64_t ldx,
                      float beta_re, float beta_im,
                      void* Y, int64_t ldy,
                      const int16_t* support, int64_t n0, int64_t nm, const int32_t* xrow_perm);

/* Coil-interleaved panels (the grid side of the fused transform's grid_layout 2): the K x N panel is stored
 * row-major, element (k, j) at [k*N + j], so that one grid point's N coil values are one contiguous N*8-byte
 * row.  The reference's KronI(C, G) batches the same products over column-major panels
 * (indigo/operators.py:236-265 KronI._eval); only the memory order of the grid-side panel differs.
 *   ig_ccsrmm_il        Y = alpha * A * X_il + beta * Y          A: M x K CSR, X_il interleaved, Y column-major
 *   ig_ccsrmm_t_grid_il Y_il = alpha * A^H * X                   through the CSR of A^T (K rows), Y_il interleaved,
 *                       restricted to the flagged segments of `support` exactly like ig_ccsrmm_t_grid
 * N must be a power of two (ig_ccsrmm_il: <= 64; ig_ccsrmm_t_grid_il: 2, 4 or 8).                      */
int  ig_ccsrmm_il(ig_ctx* ctx,
                  int64_t M, int64_t K, int64_t N, int64_t nnz,
                  float alpha_re, float alpha_im,
                  const void* vals, const int32_t* colind, const int32_t* rowptr,
                  const void* X_il,
                  float beta_re, float beta_im,
                  void* Y, int64_t ldy);
/* ... for a matrix whose weights are all real (a gridding matrix times the +-1 modulation of a centred transform on an even
 * grid): vals_re = the real parts as floats (nnz x 4 bytes), read instead of the complex values by the several-rows-per-wave
 * gather over panels of 2, 4 or 8 columns -- a third less matrix traffic, half the multiply-adds. */
int  ig_ccsrmm_il_rw(ig_ctx* ctx,
                     int64_t M, int64_t K, int64_t N, int64_t nnz,
                     float alpha_re, float alpha_im,
                     const void* vals, const float* vals_re, const int32_t* colind, const int32_t* rowptr,
                     const void* X_il,
                     float beta_re, float beta_im,
                     void* Y, int64_t ldy);
int  ig_ccsrmm_t_grid_il(ig_ctx* ctx,
                         int64_t M, int64_t K, int64_t N, int64_t nnz,
                         float alpha_re, float alpha_im,
                         const void* vals_t, const int32_t* colind_t, const int32_t* rowptr_t,
                         const void* X, int64_t ldx,
                         void* Y_il,
                         const int16_t* support, int64_t n0, int64_t nm);

/* Brick-binned adjoint gridding: Y_il = alpha * A^H * X as a race-free SCATTER, for a matrix whose columns index a 3-D
 * grid, col = kx + n0*(km + nm*ks) -- what the reference's cu_exw_csrmm_H does under its exwrite promise
 * (indigo/backends/_customgpu.cu:49-81), here made safe for any matrix by binning.  The grid is cut into bricks of
 * 16 x bm x bs points (powers of two, bm*bs <= 64); on the host the nonzeros are sorted by the brick of their column
 * (stable) into 12-byte entries {uint32 cell inside the brick, float re, float im}, the entries one row has in one brick
 * padded with {0xffffffff, 0, 0} to a multiple of `unit` = 64/N, and `round_rows`: the row of every group of `unit`
 * entries (ig_grid_bricks_count: entries per brick; ig_grid_bricks_fill: entries and round_rows at the caller's prefix
 * sums; all pointers HOST memory).  `brick_table` (device, int32
 * x 2 per NON-EMPTY brick in brick order: brick id, end of its entries) and `tasks` (device, int32 x 4 per task: lo, hi,
 * first table row, number of table rows | shared << 16; lo and hi multiples of `unit`) give each wave either a run of
 * consecutive whole bricks (at most 64 bricks and 512 segments; [lo, hi) is exactly their entries) or a piece of
 * ONE heavy brick, marked shared: its tasks add with float atomics and the brick is listed in `shared_bricks` (zeroed
 * first).  Only the 16-row segments flagged in `support` (third part, as ig_ccsrmm_t_grid; NULL = all segments of bricks
 * that hold a nonzero) are written; everything else is left untouched.  `support_tile` = kx points per entry of the support
 * table (16 as built for ig_ccsrmm_t_grid; 8 or 4: a finer table, see ig_fft_set_support_tile -- segments are then 8 / 4
 * grid rows).  N in {4, 8}; (16/support_tile)*bm*bs <= 32; Y_il row-major; X
 * column-major.  A wave keeps one brick image in LDS, walks its run with entries and panel rows requested two trips
 * ahead (the rows come from round_rows, so no load depends on another), and accumulates with plain read-add-write in
 * entry order: deterministic except for shared bricks (float atomics).                                              */
int  ig_grid_bricks_count(int64_t M, const int32_t* rowptr, const int32_t* colind, int64_t n0, int64_t nm, int64_t ns,
                          int bm, int bs, int unit, int32_t* brick_entries);
int  ig_grid_bricks_fill(int64_t M, const int32_t* rowptr, const int32_t* colind, const void* vals, int64_t n0, int64_t nm,
                         int64_t ns, int bm, int bs, int unit, const int64_t* brick_ptr, void* entries,
                         uint32_t* round_rows);
int  ig_ccsrmm_t_bricks(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float alpha_re, float alpha_im,
                        const void* entries, const uint32_t* round_rows, const void* X, int64_t ldx, void* Y_il,
                        const int16_t* support, int64_t n0, int64_t nm, int bm, int bs, const int32_t* tasks, int64_t ntasks,
                        const int32_t* brick_table, const int32_t* shared_bricks, int64_t nshared, int support_tile, int support_zwords,
                        int entry_words);
/* (support_zwords: words per entry of the support table's bitmaps = zw_in of ig_grid_support; 16 for 256- / 512-point km axes.
 *  entry_words: 3 = the 12-byte entries above; 2 = 8-byte entries {cell, re} for a matrix whose weights are all real -- a gridding
 *  matrix times the +-1 modulation of a centred transform on an even grid: a third less of the format to read per evaluation.) */

/* The brick scatter for interleaved panels of 1, 2 or 4 columns (the ranks of a coil-sharded run that hold few coils): a
 * lane is an ENTRY and loops over the columns; race-freedom comes from the ORDER of the entries.  ig_grid_slots_build (host)
 * reorders the unpadded brick format (ig_grid_bricks_count / _fill with unit = 1) inside every brick by (occurrence of the
 * cell, cell) and cuts it into SLOTS of at most 64 entries with distinct cells: 16-byte entries {cell, re, im, row of X},
 * slots per brick, nslots + 1 offsets (slot_ptr: room for nentries + 1).  ig_ccsrmm_t_slots runs it: tasks / brick_table /
 * shared_bricks as for ig_ccsrmm_t_bricks with slots in place of entries.  No padding, no transposed matrix.              */
int  ig_grid_slots_build(int64_t nbricks, const int64_t* brick_ptr, const void* entries12, const uint32_t* entry_rows, int ncell,
                         void* entries16, int32_t* brick_slots, int32_t* slot_ptr, int64_t* nslots);
int  ig_ccsrmm_t_slots(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float alpha_re, float alpha_im,
                       const void* entries16, const int32_t* slot_ptr, const void* X, int64_t ldx, void* Y_il,
                       const int16_t* support, int64_t n0, int64_t nm, int bm, int bs, const int32_t* tasks, int64_t ntasks,
                       const int32_t* brick_table, const int32_t* shared_bricks, int64_t nshared, int support_tile, int support_zwords,
                       int entry_words /* 4 = the 16-byte entries; 3 = {cell, re, row}, 12 bytes: a matrix whose weights are all real */);

/* Gridding from the SEPARABLE form of the matrix (records of ig_interp3_sep, on the device): the taps are computed, not
 * streamed -- 64 bytes per sample instead of 27 stored taps (216 .. 324 bytes), 128 instead of 125 taps at the reference's
 * default kernel width 3 (Backend.NUFFT, indigo/backends/backend.py:403).  Replaces, for the coil-interleaved grid panel of the
 * fused SENSE leaf, the products Backend.csr_matrix.forward / .adjoint (indigo/backends/backend.py:569-585) run through ccsrmm.
 *   ig_grid_gather_sep   Y (M x NC, column-major, ldy) = alpha * G * X_il + beta * Y;  X_il: n0 * nm * ns grid points x NC
 *                        interleaved coils (axes in memory order, as the records').  NC in {2, 4, 8}.                       */
int  ig_grid_gather_sep(ig_ctx* ctx, int64_t M, int64_t NC, int tw, const void* records, int64_t rec_stride, const void* X_il,
                        int64_t n0, int64_t nm, int64_t ns, float alpha_re, float alpha_im, float beta_re, float beta_im,
                        void* Y, int64_t ldy, const uint32_t* group_order);
/* (rec_stride: 32-bit words from one sample's record to the next, >= ig_interp3_sep_words(tw), a multiple of 4.
 *  group_order, optional (NULL: trajectory order), device memory: a workgroup works on a group of ig_grid_gather_sep_group(NC, tw)
 *  consecutive samples; group_order is a permutation of the ceil(M / group) groups that says which workgroup takes which -- sorted by
 *  where the groups' samples lie on the grid, neighbouring workgroups read neighbouring grid rows, and a densely sampled trajectory stops
 *  re-fetching them from HBM.  Results do not depend on it, bit for bit.) */
int  ig_grid_gather_sep_group(int64_t NC, int tw);                     /* host; 0: no such kernel */

/*   ig_grid_scatter_sep  Y_il = alpha * G^H * X  (X: M x NC column-major, ldx; NC = 4 or 8) as a race-free scatter of SHARES: a share =
 *                        (sample, brick of 16 x bm x bs grid cells its footprint meets; bm, bs <= 4), 8 bytes {sample | slow-axis cells of
 *                        the brick that hold a tap << 28, ox + 8 | (om + 8) << 5 | (os + 8) << 10 | blo << 15 | bhi << 18 | clo << 22 |
 *                        chi << 25}: tap (a, b, c) sits at brick cell (ox + a, om + b, os + c), taps b in [blo, bhi), c in [clo, chi) are
 *                        inside the brick.  n0 is a multiple of 16; bm and bs need NOT divide nm and ns: the bricks of an axis number
 *                        ceil(n / b), the part of a last brick outside the grid holds no tap and must not be flagged (277 = 69 * 4 + 1).
 *                        ig_grid_shares_count / _fill (host) bin the shares by brick (sample order inside a brick).  tasks as
 *                        for ig_ccsrmm_t_bricks with shares in place of entries; brick_table: 16 bytes per non-empty brick {brick, end of
 *                        its shares, uint64 flagged segments: bit xs + (16 / support_tile) * (im + bm * is)} -- only flagged segments are
 *                        written; shared_table: the table rows of the bricks several tasks add into (float atomics; zeroed first).
 *                        A wave keeps the brick image in REGISTERS and accumulates on the matrix cores: one v_mfma_f32_16x16x1_4b_f32
 *                        (fp32 in, fp32 accumulate) adds a share's taps on all 16 x 4 cells of one slow-axis plane of the brick for all
 *                        coils, whatever the number of taps -- the scatter is bound by instruction issue, not by HBM.
 *                        `records` must leave room behind every record for the sample's panel row (rec_stride >= words + 2 NC): the call
 *                        writes X[t, :] there, so that a share's record and panel row are one line.  What the cu_exw_csrmm_H scatter of
 *                        the reference (indigo/backends/_customgpu.cu:49-81) does under its exwrite promise, made safe by binning.     */
int  ig_grid_shares_count(int64_t M, const uint32_t* records, int tw, int64_t n0, int64_t nm, int64_t ns, int bm, int bs, int32_t* brick_shares);
int  ig_grid_shares_fill(int64_t M, const uint32_t* records, int tw, int64_t n0, int64_t nm, int64_t ns, int bm, int bs,
                         const int64_t* brick_ptr, uint32_t* shares);
int  ig_grid_scatter_sep(ig_ctx* ctx, int64_t M, int64_t NC, int tw, void* records, int64_t rec_stride, const void* shares, const void* X, int64_t ldx,
                         void* Y_il, int64_t n0, int64_t nm, int64_t ns, int bm, int bs, const int32_t* tasks, int64_t ntasks,
                         const int32_t* brick_table, const int32_t* shared_table, int64_t nshared, int support_tile,
                         float alpha_re, float alpha_im);

/* The same scatter for the reference's own panel layout, 64 columns: Y(K x 64, column-major, ldy) = alpha * A^H * X(M x 64,
 * column-major, ldx) for ANY CSR matrix with K a multiple of 16 (beta == 0: Y is zeroed first).  Bricks are 16 consecutive
 * rows of Y: ig_grid_bricks_count / _fill with (n0, nm, ns) = (K, 1, 1), bm = bs = 1, unit = 1 give `entries` (12 bytes:
 * {row of Y inside the brick, re, im}) and `entry_rows` (= round_rows: the row of X of every entry), both on the device here;
 * brick_table / tasks as for ig_ccsrmm_t_bricks (at most 64 bricks per run).  M * 512 < 2^31.  BASELINE config 3's adjoint.
 * owned_tiles (optional, device): one bit per 16-row tile of Y, set iff exactly one NON-shared task holds the tile's brick --
 *   that task stores the whole tile, only the other tiles are zeroed first (NULL: all of Y is zeroed first).               */
int  ig_ccsrmm_t_bricks_wide(ig_ctx* ctx, int64_t M, int64_t K, float alpha_re, float alpha_im,
                             const void* entries, const uint32_t* entry_rows, const void* X, int64_t ldx, void* Y, int64_t ldy,
                             const int32_t* tasks, int64_t ntasks, const int32_t* brick_table, const uint32_t* owned_tiles);
/* ... with bricks of 16 x bm x bs points of a grid n0 x nm x (K / n0 / nm) whose first axis runs fastest along the rows of Y
 * (bm * bs = 2 or 4; bm = bs = 1 is the call above).  Entries from ig_grid_bricks_count / _fill with that geometry and
 * unit = 4 (cell = x + 16 (m + bm s)): a row's share of a brick is a whole number of QUADS, entry_rows holds one row per quad,
 * padding entries must name a valid cell (< 16 bm bs; the caller replaces the fill's 0xffffffff) with weight zero; task and
 * brick boundaries are multiples of four entries.  One row of X is loaded per quad, and a 27-tap gridding row falls into 4.5
 * such bricks instead of 10 bricks of 16 rows.  The brick image lives in registers (k_bricks_wide64r).  owned_tiles as above:
 * a brick a single non-shared task holds stores all of its bm * bs tiles.                                                  */
int  ig_ccsrmm_t_bricks_wide_grid(ig_ctx* ctx, int64_t M, int64_t K, float alpha_re, float alpha_im,
                                  const void* entries, const uint32_t* entry_rows, const void* X, int64_t ldx, void* Y, int64_t ldy,
                                  const int32_t* tasks, int64_t ntasks, const int32_t* brick_table, const uint32_t* owned_tiles,
                                  int64_t n0, int64_t nm, int bm, int bs,
                                  int entry_words /* 3 = {cell, re, im}; 2 = {cell, re}: every weight real (grid bricks only) */);

/* Locality-ordered variants.  The caller may store A with its ROWS reordered (row r of the stored
 * matrix is row perm[r] of A; e.g. k-space samples sorted by the grid cell they touch, so that
 * neighbouring rows gather neighbouring panel rows and share cache lines):
 *   ig_ccsrmm_rowperm : forward product of the stored matrix, result row r written to Y[yrow_perm[r], :]
 *   ig_ccsrmm_t_grid  : xrow_perm (may be NULL) -- panel row k of the stored transpose's product is
 *                       X[xrow_perm[k], :] (folded into the panel repacking pass; needs N <= 8).      */
int  ig_ccsrmm_rowperm(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                       float alpha_re, float alpha_im,
                       const void* vals, const int32_t* colind, const int32_t* rowptr,
                       const void* X, int64_t ldx,
                       float beta_re, float beta_im,
                       void* Y, int64_t ldy,
                       const int32_t* yrow_perm);

/* Forward product over a SUBSET of the panel's rows: colind_c indexes the list `xrows` (device, nxrows ascending row
 * numbers of X) instead of X itself, i.e. Y = beta*Y + alpha * A' * X[xrows, :] with A' = A restricted to its non-empty
 * columns.  The repacking pass of a wide panel then reads and writes only the rows some nonzero touches (a gridding
 * matrix touches 30 % of its 256^3 grid: BASELINE config 3, col_frac of operators.py:246-256).  2 <= N <= 64.        */
int  ig_ccsrmm_xrows(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, int64_t nnz,
                     float alpha_re, float alpha_im,
                     const void* vals, const int32_t* colind_c, const int32_t* rowptr,
                     const void* X, int64_t ldx,
                     float beta_re, float beta_im,
                     void* Y, int64_t ldy,
                     const int32_t* xrows, int64_t nxrows);

/* The same product for 64 panel columns through the RUN format of A' (rows of A' in runs of 16, the nonzeros of a run grouped by
 * column: a run of 16 consecutive gridding samples touches ~100 distinct panel rows with its 432 nonzeros, and each is then
 * loaded once; the run's 16 result rows stay in registers).  ig_csr_runs_build (host): first call with dcols == NULL fills
 * run_dptr[ceil(M/16) + 1] (prefix sums of the runs' distinct columns), second call fills dcols (run_dptr[last] words: column |
 * (entries - 1) << 27) and entries (nnz x {row in run, re, im}, 12 bytes, a run's entries where its nonzeros sit in the CSR).
 * IG_ERR_UNSUPPORTED when a row holds a column twice or K > 2^27 (callers then keep ig_ccsrmm_xrows).  nxrows * 512 < 2^32.
 * Reference contract: indigo/backends/backend.py:514-519.                                                                  */
int  ig_csr_runs_build(int64_t M, int64_t K, const int32_t* rowptr, const int32_t* colind, const void* vals,
                       int32_t* run_dptr, uint32_t* dcols, void* entries, int* all_real);
int  ig_ccsrmm_xrows_runs(ig_ctx* ctx, int64_t M, int64_t K, int64_t nnz, float alpha_re, float alpha_im, const int32_t* rowptr,
                          const int32_t* run_dptr, const uint32_t* dcols, const void* entries, int all_real,
                          const int32_t* run_order /* optional: a permutation of the runs, e.g. by the grid brick they start in */,
                          const void* X, int64_t ldx, float beta_re, float beta_im, void* Y, int64_t ldy,
                          const int32_t* xrows, int64_t nxrows);

/* Host-side structure analysis.  Replaces `inspect`
 * (indigo/backends/_customcpu.c:179-215): number of non-empty rows / columns
 * and exwrite = "every column has <= 1 nonzero".  Pointers are HOST memory. */
int  ig_csr_inspect(const int32_t* rowptr, const int32_t* colind, int64_t M, int64_t K,
                    int64_t* nzrow, int64_t* nzcol, int* exwrite);

/* Host-side CSR transpose (counting sort, stable => sorted columns).
 * All pointers are HOST memory; outputs sized K+1 / nnz / nnz.              */
int  ig_csr_transpose(int64_t M, int64_t K, int64_t nnz,
                      const int32_t* rowptr, const int32_t* colind, const void* vals,
                      int32_t* rowptr_t, int32_t* colind_t, void* vals_t);

/* ------------------------------------------------------------------------
 * The remaining leaves of the Backend contract (outside the SENSE tree): matrix of ones, DIA sparse
 * matrices, dense matrices.  Panels column-major with leading dimensions in elements, as everywhere.
 *   ig_conemm : Y(MxN) = beta*Y + alpha * ones(M,K) * X(KxN)      Backend.onemm  (indigo/backends/backend.py:528-533;
 *               oracle np.py:94-97; CUDA cu_onemm _customgpu.cu:15-47)
 *   ig_cdiamm : adjoint == 0: Y(MxN) = beta*Y + alpha * A   * X(KxN);  adjoint != 0: Y(KxN) = beta*Y + alpha * A^H * X(MxN)
 *               A is M x K with `ndiag` stored diagonals: data is ld_data x ndiag column-major with
 *               data[j + d*ld_data] = A[j - offsets[d], j] (scipy's dia_matrix.data transposed, the form the reference
 *               uploads, backend.py:607-610).  Backend.cdiamm (backend.py:521-526; np.py:129-136; cu_diamm / cu_diammH
 *               _customgpu.cu:83-143)
 *   ig_cgemm  : M is rows_m x cols_m dense, column-major, leading dimension ldm; op(M) = M or M^H (adjoint != 0).
 *               right == 0: Y(r x p) = beta*Y + alpha * op(M)(r x c) * X(c x p)
 *               right != 0: Y(p x c) = beta*Y + alpha * X(p x r) * op(M)(r x c)
 *               Backend.cgemm / csymm (backend.py:481-491; np.py:76-90; cuBLAS cgemm/csymm cuda.py:314-392)
 * ---------------------------------------------------------------------- */
int  ig_conemm(ig_ctx* ctx, int64_t M, int64_t K, int64_t N, float alpha_re, float alpha_im,
               const void* X, int64_t ldx, float beta_re, float beta_im, void* Y, int64_t ldy);
int  ig_cdiamm(ig_ctx* ctx, int adjoint, int64_t M, int64_t K, int64_t N, int64_t ndiag, const int32_t* offsets,
               const void* data, int64_t ld_data, float alpha_re, float alpha_im, const void* X, int64_t ldx,
               float beta_re, float beta_im, void* Y, int64_t ldy);
int  ig_cgemm(ig_ctx* ctx, int adjoint, int right, int64_t rows_m, int64_t cols_m, int64_t p,
              float alpha_re, float alpha_im, const void* M, int64_t ldm, const void* X, int64_t ldx,
              float beta_re, float beta_im, void* Y, int64_t ldy);

/* Host-side construction of the 3-D gridding (interpolation) matrix, CSR.  Replaces the numba loop nest
 * _interp3_mat / lin_interp (indigo/interp.py:8-80) behind Backend.Interp (indigo/backends/backend.py:392-401):
 * same arithmetic in the same order in double precision, float32 weights out (the NUFFT factory stores float32,
 * backend.py:438).  m samples, coord = 3 x m doubles (all x, then all y, then all z) in units of the field of view,
 * N = grid dims, width = kernel HALF-width, table = ntable samples of the kernel on [0, 1).
 *   ig_interp3_count : rowptr[0..m]   (number of taps of every sample, prefix-summed)
 *   ig_interp3_fill  : colind / weights, columns sorted within a row; grid_order 0 numbers the grid (x, y, z) like the
 *                      reference, 1 numbers it (x, z, y) (the order of the fused transform's grid layouts 1 and 2).
 * All pointers are HOST memory.                                                                               */
int  ig_interp3_count(int64_t m, const int64_t* N, double width, const double* coord, int32_t* rowptr);
int  ig_interp3_fill(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable,
                     const double* coord, const int32_t* rowptr, int32_t* colind, float* weights, int grid_order);
/* The same matrix with COMPLEX values (float)w * exp(2 pi i (phase_x[kx] + phase_y[ky] + phase_z[kz])) * scale: interpolation
 * times the centred transform's modulation (Backend.FFTc, indigo/backends/backend.py:355-369) and normalisation -- the G'
 * factor of the -O3 SENSE tree (examples/pics.py:104-177) in one pass.  phase_x/y/z: N[0] / N[1] / N[2] doubles (turns).   */
int  ig_interp3_fill_modulated(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable,
                               const double* coord, const int32_t* rowptr, int32_t* colind, void* values, int grid_order,
                               const double* phase_x, const double* phase_y, const double* phase_z, double scale);
/* The SEPARABLE form of the same matrix (round 6): the reference's weights are products of per-axis factors by construction
 * (w = wz * wy * wx, indigo/interp.py:42-52) and what `pics.py -O3` folds into the stored values on an even grid -- the centred
 * transform's modulation exp(i pi k) and 1 / sqrt(P), examples/pics.py:104-177 -- is a sign per axis and cell and a constant.  One
 * record of ig_interp3_sep_words(tw) 32-bit words per sample, axes in MEMORY order of the grid (0 = x, 1 = middle, 2 = slow:
 * (x, y, z) for grid_order 0, (x, z, y) for grid_order 1):
 *   words [0, tw) / [tw, 2 tw) / [2 tw, 3 tw)   float32 weights of the taps on axis 0 / 1 / 2 (times sign_*[cell]; axis 2 also
 *                                               times `scale`); unused weights are 0
 *   word 3 tw       first tap (wrapped) on axis 0 | first tap on axis 1 << 16
 *   word 3 tw + 1   first tap on axis 2 | taps on axis 0 << 16 | taps on axis 1 << 20 | taps on axis 2 << 24
 * tw = 4 (kernel half-width <= 2), 6 (<= 3) or 8 (<= 4): 16 / 32 / 32 words.  sign_x / _y / _z: +-1 per grid cell of the
 * REFERENCE axes x, y, z (N[0] / N[1] / N[2] doubles) or NULL.  IG_ERR_UNSUPPORTED when a sample has more than tw taps on an axis.
 * Host memory.  The gridding kernels that compute their taps from these records: ig_grid_gather_sep, ig_grid_scatter_sep.   */
int  ig_interp3_sep_words(int tw);
int  ig_interp3_sep(int64_t m, const int64_t* N, double width, const double* table, int64_t ntable, const double* coord,
                    int grid_order, const double* sign_x, const double* sign_y, const double* sign_z, double scale,
                    int tw, uint32_t* records);
/* k-space support table (host) of a gridding matrix whose columns number the grid as kx + n0*(kz + n2*ky): for every
 * (ky, kx tile of `tile` points) the kz hull and one bit per kz that holds a nonzero, and the ky hull of every kx tile --
 * the table ig_fft_exec_padded / _cropped and the gridding kernels take.  The bitmaps come with zw_in words per (ky, kx tile)
 * -- bit kz / zw_in of word kz % zw_in -- and, where zw_out differs, once more with zw_out words: the forms the z pass of the
 * transform reads on its input and on its output side (ig_fft_support_words(n2): 16 / 16 for 256- and 512-point axes, B / A
 * for an axis the A x B kernel transforms).  table: 2*(n1*nt + nt) int16 + n1*nt*(zw_in + (zw_out != zw_in ? zw_out : 0))
 * uint32, nt = n0 / tile.  n2 <= 32 * zw_in, 32 * zw_out.                                                                */
int  ig_grid_support(int64_t nnz, const int32_t* colind, int64_t n0, int64_t n1, int64_t n2, int tile, int zw_in, int zw_out, int16_t* table);

/* ------------------------------------------------------------------------
 * Batched complex-to-complex FFT.  Replaces Backend.fftn/ifftn
 * (indigo/backends/backend.py:497-512; oracle np.py:102-115; cuFFT plan
 *  cache + workspace cuda.py:470-498).
 * Fortran-ordered array dims[0] (contiguous) .. dims[rank-1], then `batch`
 * contiguous volumes.  Both directions are UNNORMALISED (forward e^{-i..},
 * inverse e^{+i..}, no 1/N).  x == y (in place) is allowed.
 * ---------------------------------------------------------------------- */
int  ig_fft_plan(ig_ctx* ctx, int rank, const int64_t* dims, int64_t batch,
                 ig_fft** plan, size_t* workspace_bytes);
int  ig_fft_exec(ig_fft* plan, const void* x, void* y, int direction /* -1 fwd, +1 inv */,
                 void* workspace /* >= workspace_bytes, may be NULL if 0 */);
int  ig_fft_describe(ig_fft* plan, char* buf, size_t len);   /* kernel / radix schedule, for logs and tests */
/* Bytes of workspace that make an IN-PLACE ig_fft_exec (x == y) as fast as an out-of-place one (>= the plan's
 * workspace_bytes).  The two-launch 256^3 transform cannot run in place and stages through it; called in place without a
 * workspace it runs the three in-place axis passes instead.  The library never allocates device memory inside an exec. */
int  ig_fft_inplace_workspace(ig_fft* plan, size_t* bytes);

/* Zero-padded forward / cropped inverse 3-D transforms: the fusion of the reference's
 * Zpad . diag . FFT chain (indigo/backends/backend.py:371-387 Zpad, :355-369 FFTc, :403-442 NUFFT;
 * the `-O3` tree's S' matrix, examples/pics.py:104-193) into the transform's first / last pass.
 * The image occupies the box box_lo[a] .. box_lo[a]+box_dims[a] of a dims[0] x dims[1] x dims[2]
 * grid (each grid axis 256 or 512, else IG_ERR_UNSUPPORTED).  Compact arrays are F-ordered
 * box_dims (x batch); `w` (optional, may be NULL) holds one complex weight per compact element and
 * batch member; `x_bstride` is the element distance between batch members of x (0: one image
 * shared by all members, as in SENSE where the coils share the image).
 *   padded :  Y[..,c] = FFT3( zeropad( w[..,c] .* x[..,c] ) )            unnormalised, forward
 *   cropped:  x[..,c] = conj(w[..,c]) .* crop( IFFT3( Y[..,c] ) )        unnormalised, inverse
 * Y is a full grid x batch array; the cropped transform leaves Y intact.  Both need
 * workspace_bytes of scratch (a grid x batch array plus a compact intermediate; the padded
 * transform only uses it for grid_layout 1).
 * grid_layout: memory order of Y.  0 = (x, y, z), the reference's Fortran order; 1 = (x, z, y), i.e.
 * element (kx, ky, kz) at kx + n0*kz + n0*n2*ky -- an internal order that keeps the largest axis pass
 * at a 4 KB stride; the consumer of Y (the gridding matrix) must be indexed the same way.
 * 2 = (c, x, z, y): layout 1 with the batch (coils) interleaved below it, element (c, kx, ky, kz) at
 * c + batch*(kx + n0*kz + n0*n2*ky), batch in {1, 2, 4, 8, 16}; the gridding products then move one contiguous
 * row per grid point (ig_ccsrmm_il / ig_ccsrmm_t_grid_il).  With layout 2 the weights and, for the cropped
 * transform, the compact result x are interleaved too: w[i*batch + c], x[i*batch + c] (x_bstride is ignored
 * by the cropped transform; sum the coils with ig_csum_il).                                         */
int  ig_fft_plan_padded(ig_ctx* ctx, const int64_t* dims, const int64_t* box_lo, const int64_t* box_dims,
                        int64_t batch, int grid_layout, ig_fft** plan, size_t* workspace_bytes);
/* support (optional, grid_layouts 1 and 2; may be NULL): the k-space support of the gridding matrix that
 * consumes / produced Y, one contiguous int16 buffer in three parts (nt = n0/16 kx tiles; layout 2 plans may switch to
 * tiles of 8, 4 or 2 kx points with ig_fft_set_support_tile -- read 16 below as that tile):
 *   1. n1*nt pairs  [z_lo, z_hi)  at 2*(ky*nt + kx/16): the kz hull of the tile's column (empty: skip it);
 *   2. nt pairs     [y_lo, y_hi)  per kx tile: the ky range outside which every z hull is empty (y pass);
 *   3. n1*nt*16 uint32 words (4-byte aligned, every pair before it being 4 bytes):
 *      bit m of word 16*(ky*nt + kx/16) + t is set iff the 16-row segment (kx tile, ky, kz = t + 16*m)
 *      holds a nonzero of the gridding matrix.  (A z axis that is not 256 or 512 points long -- an A x B axis -- has zw_in = B
 *      words per entry, kz = t + B*m, followed by the same bits with zw_out = A words per entry: ig_fft_support_words tells
 *      the two numbers for an axis length, ig_grid_support writes the table.)
 * The padded transform only guarantees the flagged segments of Y (the rest is undefined and must not be
 * read); the cropped transform reads only flagged segments (everything else counts as zero and may hold
 * anything).  A radial trajectory covers a ball (52 % of the cube) with gaps between its outer spokes:
 * 30 % of the segments of the 512^3 grid of the 256^3 SENSE problem are flagged.                  */
/* Granularity of the k-space support table of a coil-interleaved (layout 2) padded plan: `tile` kx points per table entry
 * (16 by default; 8, 4 or 2 as long as coils * tile >= 16).  The table then has n0/tile entries per (ky) row in each of its
 * three parts; a finer table flags fewer grid bytes (BASELINE config 4: 30.5 % of the grid at 16, 22.2 % at 8, 16.4 % at 4).   */
int  ig_fft_set_support_tile(ig_fft* plan, int tile);
/* (round 6) A circular shift by `shift` points on the IMAGE side of a y or z axis of a coil-interleaved (layout 2) padded plan --
 * equivalently the modulation exp(2 pi i k shift / n) on its k-space side:
 *     ig_fft_exec_padded            Y[k] = exp(+2 pi i k shift / n) * FFT(zeropad(...))[k]        along that axis
 *     ig_fft_exec_cropped[_sum...]  x    = crop(IFFT(exp(-2 pi i k shift / n) * Y))               (its adjoint)
 * This is what the reference's centred transform (Backend.fftc_mod, indigo/backends/backend.py:352-366: exp(2 pi i (k - c/2) c / n),
 * c = n // 2) puts on an ODD axis; on an even axis it is the sign (-1)^k, which a gridding matrix absorbs in its weights.  Folded
 * into the transform pass, the gridding matrix of a grid with odd axes keeps real weights (8-byte entries, separable records).
 * Only chirp-z axes (ig_fft_padded_axis_kind == 5: the odd lengths int(N * osf) produces have a large prime factor more often than
 * not -- 277) take it: the shift moves the origin of their input (forward) or output (inverse) weights and of the convolution kernel,
 * at no cost per pass.  IG_ERR_UNSUPPORTED on any other axis (shift 0 is always accepted).                                          */
int  ig_fft_set_axis_shift(ig_fft* plan, int axis, int64_t shift);
int  ig_fft_support_words(int64_t n2, int* zw_in, int* zw_out);        /* host; IG_ERR_UNSUPPORTED: no zero-pad-aware z pass */
/* what ig_fft_plan_padded would run an axis of n points with: 3 = the power-of-two kernel (256, 512; any layout), 4 = the A x B
 * kernel (smooth lengths 128 ... 640; coil-interleaved layout), 5 = chirp-z (Bluestein) over an A x B length m >= 2 n - 1 for
 * lengths with a prime factor above 7 (y and z axes of the coil-interleaved layout; no support table on such a grid), 0 = none.
 * The lengths int(N * osf) of the reference's driver (indigo/backends/backend.py:427-430) are of kinds 4 and 5.               */
int  ig_fft_padded_axis_kind(int64_t n, int* kind);
int  ig_fft_exec_padded(ig_fft* plan, const void* x, int64_t x_bstride, const void* w, void* y, void* workspace,
                        const int16_t* support);
int  ig_fft_exec_cropped(ig_fft* plan, const void* y, const void* w, void* x, int64_t x_bstride, void* workspace,
                         const int16_t* support);
/* Cropped transform fused with the coil combination (grid_layout 2 only, w required):
 *   x = sum_c conj(w[.., c]) .* crop( IFFT3( Y[.., c] ) )          x: ONE compact box_dims array
 * i.e. the adjoint of the SENSE map stack S'^H (examples/pics.py:104-193 builds it as a VStack of Diag(maps);
 * indigo/operators.py:440-447 VStack._eval_adjoint sums the per-coil results) inside the transform's last pass. */
int  ig_fft_exec_cropped_sum(ig_fft* plan, const void* y, const void* w, void* x, void* workspace,
                             const int16_t* support);
/* The same transform in two phases, so that the image can leave the GPU slab by slab while later slabs are still
 * being transformed (multi-GPU: ig_allreduce_sum_f32_side per slab):
 *   phase 0          : the z pass over the whole grid (z0, z1 ignored)
 *   phase 1          : the y pass and the coil-summing x pass for the image planes z0 <= z' < z1 only; writes
 *                      x[.., .., z0:z1].  Phases 1 over a partition of [0, box_dims[2]) after one phase 0 give exactly
 *                      ig_fft_exec_cropped_sum.                                                              */
int  ig_fft_exec_cropped_sum_slab(ig_fft* plan, const void* y, const void* w, void* x, void* workspace,
                                  const int16_t* support, int phase, int64_t z0, int64_t z1);
/* The same slab-by-slab schedule for the per-coil grid layout 1 (the one-coil ranks of a coil-sharded run): phase 0 = the z
 * pass, phase 1 = the y and x passes of the image planes z0 <= z' < z1 into x (as ig_fft_exec_cropped).                   */
int  ig_fft_exec_cropped_slab(ig_fft* plan, const void* y, const void* w, void* x, int64_t x_bstride, void* workspace,
                              const int16_t* support, int phase, int64_t z0, int64_t z1);
int  ig_fft_destroy(ig_fft* plan);

/* ------------------------------------------------------------------------
 * Multi-GPU: the one collective of the coil-sharded path.  KronI(C, B) never mixes coils
 * (indigo/operators.py:374-375); VStack._eval_adjoint sums them (operators.py:440-447).  With the coils
 * sharded over ranks that sum is finished by ONE all-reduce of the image per A^H A evaluation; CG's vectors
 * stay replicated, so the reference's pdot/pnorm2 hook (indigo/backends/backend.py:469-479) needs no
 * collective.  One process per GPU: rank 0 creates an id (ig_comm_unique_id, 128 opaque bytes), hands it to the
 * other ranks by any out-of-band channel, and every rank calls ig_comm_init_rank (collective).  RCCL is loaded at
 * run time on the first of these calls (IG_ERR_UNSUPPORTED if it cannot be found); nothing else in the library
 * needs it.
 * ---------------------------------------------------------------------- */
#define IG_COMM_ID_BYTES 128
/* loads RCCL and resolves its entry points: everything of the bring-up that can fail on one rank ALONE.  The ranks agree on
 * its outcome (out of band) before any of them enters ig_comm_init_rank, which returns only when all have entered it  */
int  ig_comm_preflight(void);
int  ig_comm_unique_id(void* id_out /* IG_COMM_ID_BYTES, host */);
int  ig_comm_init_rank(ig_ctx* ctx, int nranks, int rank, const void* id, ig_comm** out);   /* collective */
/* The DIRECT communicator (round 6; no RCCL): the ranks of ONE node (one process per GPU) each expose a window of device memory to
 * the others through hipIpcGetMemHandle / hipIpcOpenMemHandle, and ig_allreduce_sum_f32 becomes a reduce-scatter + all-gather over
 * peer-mapped memory: every rank sums ITS 1 / nranks slab of all windows (rank order: the same bits everywhere) and writes it back
 * into all of them -- its traffic runs over all of the GPU's xGMI links at once, where a ring all-reduce is bound by one.
 * `name`: a POSIX shared-memory name ("/...") the ranks agree on, unique to the communicator -- it carries the IPC handles, a barrier
 * and the host scalars (ig_allreduce_max_f64_host / _sum_ / ig_comm_barrier work on it); window_bytes: a multiple of 4096, messages
 * larger than the window go in pieces.  Collective; every rank fails within timeout_s if one never arrives.  The all-reduce of such
 * a communicator is HOST-synchronous (stream sync + shared-memory barrier between its phases: no kernel ever spins on a flag another
 * process must set); ig_allreduce_sum_f32_side equals ig_allreduce_sum_f32.  At most 16 ranks.                                   */
int  ig_comm_init_direct(ig_ctx* ctx, int nranks, int rank, const char* name, size_t window_bytes, double timeout_s, ig_comm** out);
int  ig_comm_info(ig_comm* comm, int* rank, int* nranks, char* rccl_lib, size_t len);
/* in-place sum over the ranks of nfloats float32 (an image of N complex64 is 2N floats), in order with the context's
 * stream like every other call -- no host synchronisation.  (All collectives of a communicator run on the communicator's
 * own stream, bracketed by event dependencies: one communicator, one stream.)                                    */
int  ig_allreduce_sum_f32(ig_comm* comm, void* buf, int64_t nfloats);
/* the same on the communicator's own stream, ordered after everything enqueued so far on the context's stream;
 * ig_comm_join makes the context's stream wait for all such all-reduces (call it before anything reads the buffers) */
int  ig_allreduce_sum_f32_side(ig_comm* comm, void* buf, int64_t nfloats);
int  ig_comm_join(ig_comm* comm);
/* one host double reduced over the ranks (step timing, convergence scalars); synchronous */
int  ig_allreduce_max_f64_host(ig_comm* comm, double* inout);
int  ig_allreduce_sum_f64_host(ig_comm* comm, double* inout);
int  ig_comm_barrier(ig_comm* comm);                                   /* synchronous */
int  ig_comm_destroy(ig_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* INDIGO_HIP_H */
