/*
 * onmf_hip.h - C-ABI of libonmf_hip.so: the online-NMF minibatch step on MI355X (gfx950).
 *
 * The reference (HanbaekLyu/ONMF_ONTF_NDL) is pure Python and has no FFI of its own; its boundary
 * for this path is the `Online_NMF` / `Online_NTF` class surface.  Every entry point below names
 * the reference function it stands in for.  A Python host binds them with ctypes
 * (onmf_ontf_ndl_amd/_lib.py); nothing in these signatures is a torch type.
 *
 * Conventions
 *   - All matrix pointers are DEVICE pointers to float32, row-major (NumPy C order), owned by the
 *     caller.  The context owns scratch memory only.
 *   - Shapes follow the reference: X (d x b, leading dimension ldx >= b), W (d x r), H (r x b),
 *     A (r x r), B (r x d), C (d x d).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue work on it;
 *     entry points that return host scalars (sweeps_done, dist_trace) synchronise that stream.
 *   - Return value: ONMF_OK (0) or a negative ONMF_ERR_* code; the text is onmf_last_error(ctx).
 *     Nothing throws across the ABI.  A context is not thread-safe; use one per GPU/stream.
 */
#ifndef ONMF_HIP_H
#define ONMF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ONMF_OK 0
#define ONMF_ERR_ARG (-1)       /* bad shape / null pointer / unsupported size          */
#define ONMF_ERR_HIP (-2)       /* a HIP runtime call failed                             */
#define ONMF_ERR_NOMEM (-3)     /* scratch allocation failed                             */
#define ONMF_ERR_TIMEOUT (-4)   /* a bounded device-side wait gave up and no retry was possible */

#define ONMF_ABI_VERSION 3

/* coder_mode of onmf_sparse_code */
#define ONMF_CODER_DAMPED 0 /* reference step 1/(sqrt(i+10)(G_kk+1)), src/onmf.py:257       */
#define ONMF_CODER_EXACT_CD 1 /* step 1/G_kk, run to a fixed point: the convex problem that   \
                                 src/ontf.py:79-86 hands to sklearn LassoLars                 */

typedef struct onmf_ctx onmf_ctx;

int onmf_abi_version(void);

/* Scratch owner for one GPU.  No counterpart in the reference (NumPy allocates temporaries). */
int onmf_create(int device, onmf_ctx** out);
int onmf_destroy(onmf_ctx* ctx);
const char* onmf_last_error(onmf_ctx* ctx);

/*
 * update_code_within_radius(X, W, H0, r=None, alpha, sub_iter, stopping_diff)  src/onmf.py:233-271
 * and Online_NMF.sparse_code (src/onmf.py:51-90, which calls it with sub_iter=10, stopping_diff=0.01).
 *   H0 == NULL: H0 ~ U[0,1) from a counter-based device generator keyed by `seed`
 *               (the reference draws it from NumPy's global generator, onmf.py:245-246; a host
 *               that needs that exact stream draws it itself and passes H0).
 *   stopping_diff: the loop ends when sigma_max(H - H_prev)/sigma_max(H_prev) <= stopping_diff
 *               (spectral norms, onmf.py:265) or after sub_iter sweeps.
 *   sweeps_done (host int*, nullable), dist_trace (host float[sub_iter], nullable): dist per
 *               sweep where the test was evaluated in full; where rigorous bounds decided it, the
 *               deciding bound (a lower bound of dist if the loop went on, an upper bound if it
 *               stopped); NaN for sweeps that did not run.
 *   The radius argument r of the reference: onmf_sparse_code_within_radius below (every caller of the
 *               reference passes r=None, which is this entry point).
 *   coder_mode ONMF_CODER_EXACT_CD: the loop ends when ||H - H_prev||_F / ||H_prev||_F <= stopping_diff
 *               (Frobenius norms over the whole batch), tested on the last sweep of each chunk of
 *               8, 16, 32, 64, 64, ... sweeps (one launch and one host round trip per chunk), or after
 *               sub_iter sweeps; dist_trace holds the tested values and NaN for the sweeps in between.
 *               onmf_sparse_code_each is the form with one test per column and no host round trip.
 */
int onmf_sparse_code(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W,
                     const float* H0, uint64_t seed, float alpha, int sub_iter,
                     float stopping_diff, int coder_mode, float* H, int* sweeps_done,
                     float* dist_trace, void* stream);

/*
 * update_code_within_radius with r given (src/onmf.py:260-263): after every row's update the reference
 * pulls the iterate back to within spectral distance r of H0, H1 = H0 + r / max(r, ||H1 - H0||_2) (H1 - H0),
 * and then rebinds H0 to the iterate (line 263) - so from the second row on the distance is zero and
 * nothing is pulled.  What the branch does is therefore: the first row's move in the first sweep is scaled
 * by radius / max(radius, its Euclidean norm over the batch); everything else is onmf_sparse_code with the
 * damped step.  radius must be positive and finite (0 divides by zero in the reference), r <= 640.  The
 * stopping test is evaluated sweep by sweep (one host round trip each): no caller of the reference takes
 * this branch, it is here for completeness, not for speed.
 */
int onmf_sparse_code_within_radius(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W,
                                   const float* H0, uint64_t seed, float radius, float alpha, int sub_iter,
                                   float stopping_diff, float* H, int* sweeps_done, float* dist_trace,
                                   void* stream);

/*
 * Aggregate recursion of Online_NMF.step, src/onmf.py:150-158 (ontf form: src/ontf.py:147-148):
 *   A_out = (1-w) A_in + w H H^T,  B_out = (1-w) B_in + w H X^T,  [C_out = (1-w) C_in + w X X^T]
 * with w = t^(-beta) computed by the caller.  C_in/C_out may be NULL (two-aggregate form).
 * In/out pointers may alias.  A_in/B_in/C_in NULL means zeros.
 */
int onmf_aggregate(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* H,
                   double w, const float* A_in, const float* B_in, const float* C_in, float* A_out,
                   float* B_out, float* C_out, void* stream);

/*
 * Online_NMF.update_dict(W, A, B), src/onmf.py:92-116 (same body: src/ontf.py:91-115).
 * For d > 1024 the pass runs on several workgroups that wait for each other's part of every atom's
 * norm (bounded waits).  If one gives up - the workgroups were not resident together - the pass is
 * redone atom by atom without such waits (slower, same result to rounding) and the event is counted
 * (ONMF_CNT_DICT_TIMEOUTS).  That happens on the device, by a kernel enqueued behind the pass: the call
 * does not wait for the pass (ONMF_OPT_DICT_KERNEL = 1, the cross-check kernel: the host checks and redoes,
 * and the call synchronises `stream`).
 */
int onmf_update_dict(onmf_ctx* ctx, int d, int r, const float* W_in, const float* A, const float* B,
                     float* W_out, void* stream);

/*
 * onmf_update_dict for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments,
 * same kernels and bit-identical results as onmf_update_dict, but the pass's workspace (the running
 * product Q, the padded A, the atom deltas, W^T in and out) lives in a static device array of the
 * library - 4.25 MB per GPU the library runs on, held until the process ends - instead of buffers the
 * context allocates on demand.  Limits: d <= 1024 and r <= 256 (larger shapes return ONMF_ERR_ARG:
 * use onmf_update_dict).  Calls neither hipMalloc nor hipFree nor any other allocating routine; asynchronous on `stream`,
 * returns no host scalars.  Every call on a GPU uses that GPU's one array.  Calls from this process are
 * ordered behind each other on the GPU whatever stream, context or thread they come from (an event per
 * device, created by onmf_create, recorded behind each call and waited for by the next) - except calls
 * on a stream that is being captured, which are left unordered: calls that overlap on one GPU under
 * capture, or from two processes, race.  Nothing carries over between calls - each call rewrites every
 * part of the array that it reads.
 */
int onmf_update_dict_static(onmf_ctx* ctx, int d, int r, const float* W_in, const float* A, const float* B,
                            float* W_out, void* stream);

/*
 * Online_NMF.step(X, aggregates, W, t), src/onmf.py:119-167: code, aggregates, then the
 * dictionary pass with the aggregates from BEFORE this step (onmf.py:161).  w = t^(-beta).
 * H_out may be NULL when the caller does not need the code.  Outputs may alias inputs.
 */
int onmf_step(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W_in,
              const float* A_in, const float* B_in, const float* C_in, const float* H0,
              uint64_t seed, double w, float alpha, int sub_iter, float stopping_diff,
              float* H_out, float* A_out, float* B_out, float* C_out, float* W_out,
              int* sweeps_done, float* dist_trace, void* stream);

/*
 * The step for a minibatch whose columns are split over several GPUs (one process each).  The reference
 * is single-process; what couples the columns of a minibatch is only the sums H H^T and H X^T of
 * src/onmf.py:155-156.  Every rank holds the same (W, A, B) and its own columns X:
 *   onmf_step_partial  codes the local columns (src/onmf.py:144), leaves the BARE local sums
 *                      P_out = [ H H^T (r x r) | pad to a multiple of 4 floats | H X^T (r x d) ]
 *                      (onmf_partial_floats(d, r) floats) and runs the dictionary pass on the aggregates
 *                      from before the step (A_in, B_in; src/onmf.py:161) - in the same three-stream
 *                      schedule as onmf_step.
 *   (the host adds P over the ranks: one all-reduce)
 *   onmf_step_apply    A_out = (1-w) A_in + w P[H H^T],  B_out = (1-w) B_in + w P[H X^T]
 *                      (src/onmf.py:155-156 with the summed products); outputs may alias inputs.
 * state_ready_event (hipEvent_t as void*, nullable): the dictionary pass and the aggregate products -
 * the only parts that touch A_in, B_in and P_out - wait for it; the code solve does not.  A host that
 * all-reduces the previous step's P and applies it on a stream of its own records the event there and so
 * hides that exchange under this step's W^T X and sweeps (legal because the dictionary pass of step t
 * needs the aggregates of step t-1 only).  With stopping_diff > 0 every rank tests its own columns.
 */
/*
 * With ONMF_OPT_DEFER_VERDICT set, an onmf_step call that asks for no host scalars (sweeps_done == NULL and
 * dist_trace == NULL) does not wait for the GPU at all: the sweeps decide the stopping test on the device and
 * everything behind them is enqueued on the assumption that their verdicts are conclusive.  The host looks at
 * the verdict records at the NEXT entry point called on the context (or onmf_settle); where a verdict was not
 * conclusive - the bounds straddled stopping_diff in the middle of the solve - the solve and the aggregate
 * products are run again there, on `stream`, before anything else.  The caller's side of the bargain: the
 * call's input buffers stay untouched, and nothing but this library reads its outputs H_out, A_out, B_out,
 * C_out, until that next entry point has returned.  W_out is final at once (it does not depend on the code).
 * Same results as without the option.
 */
int onmf_settle(onmf_ctx* ctx);

size_t onmf_partial_floats(int d, int r);
int onmf_step_partial(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W_in,
                      const float* A_in, const float* B_in, const float* H0, uint64_t seed, float alpha,
                      int sub_iter, float stopping_diff, float* H_out, float* P_out, float* W_out,
                      int* sweeps_done, float* dist_trace, void* state_ready_event, void* stream);
int onmf_step_apply(onmf_ctx* ctx, int d, int r, const float* P, double w, const float* A_in,
                    const float* B_in, float* A_out, float* B_out, void* stream);

/* X_batch = X[:, idx]  (src/onmf.py:214, src/ontf.py:231); idx: device int64[b]. */
int onmf_gather_columns(onmf_ctx* ctx, int d, int n, const float* X, int ldx, const int64_t* idx,
                        int b, float* Xb, void* stream);

/*
 * code[:, idx] += H with NumPy fancy-index semantics (src/onmf.py:221): for a repeated index only
 * its last occurrence contributes.  idx == NULL means idx = arange(n) (dense add, b == n).
 */
int onmf_code_accumulate(onmf_ctx* ctx, int r, int n, float* code, const int64_t* idx, int b,
                         const float* H, void* stream);

/*
 * tensorly.unfold(T, mode) for a 3-way tensor (src/ontf.py:204), optionally transposed
 * (learn_joint_dict, src/ontf.py:207).  T: (n0 x n1 x n2) C order.  out: (n_mode x rest), or its
 * transpose.
 */
int onmf_unfold3(onmf_ctx* ctx, int n0, int n1, int n2, const float* T, int mode, int transpose,
                 float* out, void* stream);

/*
 * Many independent one-column solves in one call: what reconstruct_image_color does patch by patch,
 * image_reconstruction.py:375-384 - update_code_within_radius(patch, W, H0=None, r=None, alpha,
 * sub_iter, stopping_diff) (src/onmf.py:233-271) once per d x 1 patch.  Column c of X is solved with
 * its OWN stopping test (for one column the spectral norms of src/onmf.py:265 are vector 2-norms):
 * it takes part in sweep i only while dist_c of sweep i-1 exceeded stopping_diff.
 *   H0 == NULL: U[0,1) from the device generator (the reference draws rand(r,1) per patch, :245-246;
 *               a host that needs that stream passes H0 with one draw per column).
 *   sweeps_each (device int32[b], nullable): sweeps each column ran.
 * Asynchronous on `stream` (returns no host scalars).  H may be H0 (in place).
 */
int onmf_sparse_code_each(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W,
                          const float* H0, uint64_t seed, float alpha, int sub_iter,
                          float stopping_diff, int coder_mode, float* H, int32_t* sweeps_each,
                          void* stream);

/*
 * Online_NTF.joint_sparse_code_tensor(X, W), src/ontf.py:59-89: scikit-learn's
 *   SparseCoder(dictionary=W.T, transform_alpha=alpha, transform_algorithm='lasso_lars', positive_code=True)
 *     .transform(X.T)
 * followed event by event, one column per wavefront: the positive LARS-lasso path of
 * sklearn.linear_model._least_angle._lars_path_solver (scikit-learn 1.7.2: method="lasso", positive=True,
 * Gram = W^T W and Xy = W^T x precomputed, alpha_min = alpha / d, return_path=False) including the way it
 * ends - the library's path stops short of the optimum of the convex problem once no atom with a positive
 * correlation is left, and the reference's aggregates are built from those codes.  W^T W and W^T X are
 * accumulated in double from the float operands, the path itself runs in double.
 *   alpha     the reference's `alpha` (2 when it is None, src/ontf.py:79-81); must be >= 0
 *   max_iter  the library's event limit (SparseCoder's transform_max_iter, 1000 by default)
 *   H         r x b out (the reference returns the transpose, b x r)
 *   info      device int32[4*b], nullable: per column {events, end (1: alpha reached, 2: every atom active
 *             or max_iter, 3: the library's "alpha is increasing" bail-out, 4: safety limit), active atoms at
 *             the end, atoms that left the active set on the way}
 * r <= 256.  Asynchronous on `stream`, returns no host scalars.
 */
int onmf_sparse_code_lars(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W,
                          double alpha, int max_iter, float* H, int32_t* info, void* stream);

/*
 * Online_NTF.step(X, A, B, W, t), src/ontf.py:117-154: the coder above, the aggregate recursion
 * (A_out = (1-w) A_in + w H H^T, B_out = (1-w) B_in + w H X^T with H as r x b; src/ontf.py:147-148 in the
 * reference's b x r convention) and the dictionary pass on the aggregates from BEFORE the step
 * (src/ontf.py:151), which runs on a second stream beside the coder when no output is one of the inputs.
 * w = t^(-beta).  H_out (r x b) may be NULL.  Asynchronous on `stream`.
 */
int onmf_ntf_step(onmf_ctx* ctx, int d, int r, int b, const float* X, int ldx, const float* W_in,
                  const float* A_in, const float* B_in, double w, double alpha, int max_iter, float* H_out,
                  float* A_out, float* B_out, float* W_out, int32_t* info, void* stream);

/* R = W H (d x b, leading dimension ldr): np.dot(self.W, code), image_reconstruction.py:352, :385. */
int onmf_patch_product(onmf_ctx* ctx, int d, int r, int b, const float* W, const float* H, float* R,
                       int ldr, void* stream);

/*
 * Patches of an image as the columns of X (d x b, d = k*k*ch).  img: h x w x ch, C order; feature
 * order = window.reshape(-1), i.e. f = (x0*k + x1)*ch + cc.
 *   corners != NULL: device int32[2*b], (row, column) of each patch's top-left corner -
 *               extract_random_patches, image_reconstruction.py:173-206 (the host draws the corners,
 *               :185-186 / :197-198, so the RNG stream stays the caller's).
 *   corners == NULL: the regular grid of grid_rows x grid_cols patches `stride` pixels apart, row by
 *               row (b = grid_rows*grid_cols) - sklearn extract_patches_2d as used for reconstruction,
 *               image_reconstruction.py:163-166 (stride 1, all patches), and the window loop of
 *               reconstruct_image_color, :375-378.
 */
int onmf_extract_patches(onmf_ctx* ctx, const float* img, int h, int w, int ch, int k,
                         const int32_t* corners, int b, int grid_rows, int grid_cols, int stride,
                         float* X, int ldx, void* stream);

/*
 * Overlap-average of patch values back into an image: the painting loop of reconstruct_image_color,
 * image_reconstruction.py:388-392 (running mean over the patches that cover a pixel), and sklearn
 * reconstruct_from_patches_2d at :354.  R: d x (grid_rows*grid_cols) patch values (leading dimension
 * ldr) on the same grid as onmf_extract_patches.  out: h x w x ch; pixels that no patch covers are 0
 * (the reference's canvas starts as zeros).  count (nullable): h x w, patches over each pixel
 * (A_overlap_count).
 */
int onmf_fold_patches(onmf_ctx* ctx, const float* R, int ldr, int h, int w, int ch, int k,
                      int grid_rows, int grid_cols, int stride, float* out, float* count, void* stream);

/*
 * Surrogate loss of the current dictionary against the aggregates,
 *   tr(W A W^T) - 2 tr(W B) + tr(C)      (ising_reconstruction.py:133, :164),
 * accumulated in double.  C may be NULL (its trace is then left out).  *loss is a HOST double: the
 * call synchronises `stream`.
 */
int onmf_surrogate_loss(onmf_ctx* ctx, int d, int r, const float* W, const float* A, const float* B,
                        const float* C, double* loss, void* stream);

/*
 * Patch source of network dictionary learning: Markov chains of motif embeddings into a graph and
 * the k x k adjacency matrix each embedding induces (network_reconstruction_nx.py:280-329,
 * get_patches_glauber / chd_gen_mx with iterations=1).
 *   graph     CSR, device int32: row_ptr[n_nodes+1], nbr[row_ptr[n_nodes]] with every row sorted
 *             ascending (an undirected graph stores both directions).
 *   motif     device int32[k*k], row-major 0/1 adjacency of the rooted tree motif in depth-first
 *             order (path_adj, :86-95), 1 <= k <= 64; for every i >= 1 the first 1 of column i must
 *             lie in a row < i (find_parent, :100-106).
 *   emb       device int32[n_chains*k], in/out: node that motif node q of chain c sits on.
 *   chain_kind  ONMF_NDL_GLAUBER (glauber_gen_update, :136-173) or ONMF_NDL_PIVOT (Pivot_update,
 *             :265-278 = RW_update :175-199 + tree_sample :108-134).
 * Every chain advances `steps` times; the patch after step s of chain c is column c*steps + s of
 * X (k*k x ldx, feature q*k + r = has_edge(emb[q], emb[r]), values 0/1).  n_chains = 1 is the
 * reference's single sequential chain; more chains are independent replicas of it.
 * Randomness is Philox4x32-10 keyed by `seed`, a pure function of (chain, step_offset + s, slot):
 * a later call continues the same chains when it passes step_offset = steps already taken.
 * (The reference draws from NumPy's global generator; its stream cannot be followed on the device.)
 * emb_trace (device int32[n_chains*steps*k], nullable): the embedding behind every patch, row
 * c*steps + s - what reconstruct_network needs to put a patch back on the graph (:477-479).
 * onmf_ndl_init_embeddings: emb[c] = tree_sample(motif, pivots[c]); pivots == NULL draws each pivot
 * uniformly (train_dict, :351-352).
 * onmf_ndl_accumulate_edges: the edge-wise averaging of reconstruct_network (:477-491) as dense sums:
 * for every patch t and entry (q, r), sum[a*n + b] += R[(q*k + r)*ldr + t] and count[a*n + b] += 1 with
 * (a, b) = (emb_trace[t][q], emb_trace[t][r]); sum and count are n x n device floats that the caller
 * zeroes first (the reference's running mean is sum / count).  Float atomics: the order of the
 * additions, hence the last bits of sum, can differ between runs.
 */
#define ONMF_NDL_GLAUBER 0
#define ONMF_NDL_PIVOT 1
int onmf_ndl_init_embeddings(onmf_ctx* ctx, int n_nodes, const int32_t* row_ptr, const int32_t* nbr,
                             int k, const int32_t* motif, int n_chains, const int32_t* pivots,
                             uint64_t seed, int32_t* emb, void* stream);
int onmf_ndl_sample_patches(onmf_ctx* ctx, int n_nodes, const int32_t* row_ptr, const int32_t* nbr,
                            int k, const int32_t* motif, int chain_kind, int n_chains, int steps,
                            int32_t* emb, uint64_t seed, uint64_t step_offset, float* X, int ldx,
                            int32_t* emb_trace, void* stream);
int onmf_ndl_accumulate_edges(onmf_ctx* ctx, int n_nodes, int k, int n_patches, const int32_t* emb_trace,
                              const float* R, int ldr, float* sum, float* count, void* stream);

/*
 * np.random.rand(n) out of NumPy's legacy global generator (MT19937), continued on the device: the draw the
 * reference makes for the start code of every solve, H0 = np.random.rand(r, b) (src/onmf.py:245-246; also
 * the initial dictionary, :190), written to device memory as float32 - the float64 value NumPy would
 * return, rounded once.
 *   key, pos      HOST: the generator's 624 state words and position, as np.random.get_state() gives them
 *   key_out, pos_out  HOST: the state after the draw, for np.random.set_state() - the host's stream goes on
 *                 exactly where NumPy would be after np.random.rand(n)
 * Synchronises `stream` (it returns host data).  With a jump table (onmf_numpy_set_jump_table: rows j = 1..n
 * hold x^(j * chunk_words - 1) modulo the generator's minimal polynomial, 624 words each, computed by
 * onmf_ontf_ndl_amd/mt_jump.py) chunks of chunk_words = 131 040 stream words are produced by as many
 * workgroups at once; without one, chunk after chunk by a single workgroup.  Same output either way.
 */
int onmf_numpy_set_jump_table(onmf_ctx* ctx, const uint32_t* table_host, int rows, long long chunk_words);
int onmf_numpy_rand(onmf_ctx* ctx, const uint32_t* key, int pos, size_t n, float* out, uint32_t* key_out,
                    int* pos_out, void* stream);

/* U[0,1) fill, the device stand-in for np.random.rand (src/onmf.py:190, :246). */
int onmf_fill_uniform(onmf_ctx* ctx, float* out, size_t n, uint64_t seed, void* stream);

/*
 * Host-side helper of the stopping test: largest eigenvalue (double) of a symmetric positive
 * semi-definite n x n float matrix in HOST memory.  sigma_max(M)^2 of src/onmf.py:265 is this
 * value for M M^T.  Touches no GPU.
 */
double onmf_host_lambda_max(const float* S_host, int n);

/*
 * Options (no counterpart in the reference).
 *   ONMF_OPT_PROFILE       != 0: bracket the sweep kernels of later sparse-code calls with HIP
 *                          events on the caller's stream; read back with onmf_get_profile.
 *   ONMF_OPT_EXACT_STOP    != 0: always evaluate the spectral-norm stopping test in full (Gram
 *                          matrices + eigenvalues) instead of deciding by norm bounds when those
 *                          are conclusive; dist_trace then holds every dist value.
 *   ONMF_OPT_SWEEP_KERNEL  0 = pick by shape, 1 = column-per-lane kernel, 2 = MFMA tile kernel
 *                          (3 / 4: tile kernel pinned to 1 / 2 column tiles of 16 per wave).
 *   ONMF_OPT_OVERLAP_DICT  != 0 (default): onmf_step runs the dictionary pass on a second stream
 *                          beside the two big products when no output aliases an input, and the
 *                          code solve's small set-up work (and the last columns of W^T X) on a
 *                          third; 0: everything in order on the caller's stream.
 *   ONMF_OPT_SCREEN_STOP   != 0 (default): decide the stopping test on the device from rigorous
 *                          bounds that the sweep kernel produces as by-products, all sweeps
 *                          enqueued without host round trips; the full test runs only where the
 *                          bounds straddle stopping_diff.  0: one host decision per sweep.
 *   ONMF_OPT_DICT_KERNEL   0 = pick by shape, 1 = cooperating one-wave workgroups (any d <= 16384),
 *                          2 = panel kernel: one workgroup per 1024 rows (d <= 16384).
 *   ONMF_OPT_PERSISTENT_SWEEPS != 0 (default): with the screened stopping test, run all sweeps of
 *                          a solve in one launch when its grid (one workgroup per 256 columns)
 *                          fits the chip's CUs; 0: one launch per sweep.
 *   ONMF_OPT_DEVICE_RESOLVE  where the bounds cannot decide a stopping test, it is evaluated in full: by the host
 *                          (two launches, a copy, two eigenvalue problems, a new sweep launch: about 60 us per
 *                          test at r = 25), or by a kernel enqueued ahead of time behind a one-launch solve - both
 *                          Gram matrices and their top eigenvalues on the GPU (Householder tridiagonalisation
 *                          for r <= 32, Lanczos above, multisection), the sweeps continued from there - as
 *                          many rounds (four at most) as the previous solve needed.  For r <= 32 a round settles
 *                          the inconclusive sweep AND the one after it (whose move is still in the work
 *                          buffers).  < 0 (default): device rounds for r <= 32 behind a solve that stopped
 *                          early and needed a full evaluation (measured: 0.23 against 0.30 ms per step at d = 441,
 *                          r = 25, b = 1000); 0: the host evaluates every such test; > 0: device rounds for every
 *                          r <= 64 behind any solve that needed a full evaluation (at r = 64 a round costs
 *                          about what the host's does).  Same sweep counts and codes in every setting.
 *   ONMF_OPT_LARS_SLOTS    onmf_sparse_code_lars keeps, per column, the inverse Gram matrix of the active atoms in
 *                          LDS.  Room for all r atoms lets only a few wavefronts be resident per CU, so a call
 *                          sizes its launch by how many columns of the PREVIOUS call needed more than 16, 32,
 *                          48 and 64 slots (the smallest of these that all but 1 column in 64 fitted in; a first
 *                          call tries 64) and codes the columns that need more in a second launch with room
 *                          for r (ONMF_CNT_LARS_RECODED counts them).  0 (default):
 *                          that rule; n > 0: n slots (at most 64, or r) in the first launch whatever the
 *                          history.  Same codes either way.
 *   ONMF_OPT_DEFER_VERDICT != 0: onmf_step calls that return no host scalars do not wait for the stopping test's
 *                          verdicts; see onmf_settle.  0 (default): every call's outputs are final when it returns
 *                          (in stream order).
 *   ONMF_OPT_VERDICT_LAG3  != 0 (default): one-launch solves on the 64-column kernel (batches of up to 64 columns per
 *                          CU) act on a sweep's stopping test three sweeps later instead of two, so that no wave
 *                          waits for a verdict: a stop leaves two sweeps run in vain instead of one, and a test
 *                          the bounds cannot decide can no longer be evaluated in full (its input is
 *                          overwritten) - the solve is then run again in the plain form (counted by
 *                          ONMF_CNT_LAG3_REDONE / ONMF_CNT_DEFERRED_REDONE).  Used only when the start code can be
 *                          had again and the previous solve needed no host decision.  Same results either way.
 *   ONMF_OPT_GEMM_TAIL     experiment, off (0) by default: 100 * p + n computes the last p per cent of the tiles of
 *                          the big W^T X launch in n pieces along the contraction, summed by a small kernel
 *                          (measured: no gain at any setting; results differ from the default in the last bits
 *                          of those columns of W^T X).
 *   ONMF_OPT_SPIN_LIMIT_MS bound, in milliseconds, of the waits of workgroups for each other - a
 *                          one-launch solve's, and the dictionary pass's for d > 1024 (default 2000;
 *                          -1: give up at the first look if what is waited for is not there yet; -2: give
 *                          up at the first look in any case, which exercises the recovery paths - every
 *                          kernel still ends normally).  After a timeout the solve is run again
 *                          with one launch per sweep, and so are the next 64 solves (128 after the next
 *                          timeout, and so on); then the one-launch mode is tried again.  The one-launch
 *                          mode assumes that one workgroup per CU is resident at once, which holds when
 *                          the process has the GPU to itself; another process's kernels on the same GPU
 *                          can break it, and these counters show when that happens.
 */
#define ONMF_OPT_PROFILE 1
#define ONMF_OPT_EXACT_STOP 2
#define ONMF_OPT_SWEEP_KERNEL 3
#define ONMF_OPT_OVERLAP_DICT 4
#define ONMF_OPT_SCREEN_STOP 5
#define ONMF_OPT_DICT_KERNEL 6
#define ONMF_OPT_PERSISTENT_SWEEPS 7
#define ONMF_OPT_SPIN_LIMIT_MS 8
#define ONMF_OPT_DEVICE_RESOLVE 9
#define ONMF_OPT_LARS_SLOTS 10
#define ONMF_OPT_DEFER_VERDICT 11
#define ONMF_OPT_GEMM_TAIL 12
#define ONMF_OPT_VERDICT_LAG3 13
int onmf_set_option(onmf_ctx* ctx, int key, int value);
/*
 * Event counters since onmf_create (no counterpart in the reference):
 *   ONMF_CNT_SWEEP_TIMEOUTS   one-launch solves whose workgroups gave up waiting for each other
 *   ONMF_CNT_DICT_TIMEOUTS    dictionary passes redone atom by atom for the same reason (counted by the
 *                             kernel that redid them: a pass still in flight is not in the count yet)
 *   ONMF_CNT_FALLBACK_SOLVES  solves that ran with one launch per sweep because of an earlier timeout
 */
#define ONMF_CNT_SWEEP_TIMEOUTS 1
#define ONMF_CNT_DICT_TIMEOUTS 2
#define ONMF_CNT_FALLBACK_SOLVES 3
#define ONMF_CNT_DEVICE_RESOLVES 4   /* inconclusive stopping tests settled on the device (no host round trip) */
#define ONMF_CNT_LARS_RECODED 5      /* columns of onmf_sparse_code_lars coded by its second launch (calls still in
                                        flight are not in the count yet) */
#define ONMF_CNT_SWEEPS_TOTAL 7       /* sweeps run by all settled solves (what sweeps_done would have summed to) */
#define ONMF_CNT_DEFERRED_REDONE 8   /* deferred steps (ONMF_OPT_DEFER_VERDICT) whose solve had to be run again */
#define ONMF_CNT_LAG3_REDONE 9       /* waited-for solves run again because a test acted on three sweeps later came out
                                        inconclusive (ONMF_OPT_VERDICT_LAG3; deferred steps count under 8) */
#define ONMF_CNT_LARS_SLOTS_USED 6   /* not a count: the most atoms a column of the last completed
                                        onmf_sparse_code_lars call held at a time */
int onmf_get_counter(onmf_ctx* ctx, int which, long long* value);
/* Summed device time (ms) and number of sweep-kernel launches since the previous call. */
int onmf_get_profile(onmf_ctx* ctx, double* sweep_ms, int* sweep_launches);

#ifdef __cplusplus
}
#endif
#endif /* ONMF_HIP_H */
