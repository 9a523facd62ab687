/* libscilmm_hip C ABI -- the drop-in boundary of the MI355X-native sparse-Cholesky REML engine.
 *
 * The reference has no FFI: its boundary is the duck-typed Python protocol
 *     cholesky_func(V) -> factor ;  factor(b), factor.L(), factor.P(), factor.logdet()
 * implemented by sksparse.cholmod (reference scilmm/SparseCholesky.py:16-26, used at :30,:32,:40,:50,
 * :52,:93,:100; twin scilmm/Estimation/LMM.py:14-24).  Each entry point below names the reference call it
 * replaces.  The Python shim (scilmm_amd/_lib.py, scilmm_amd/factor.py) binds exactly these symbols through
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only; every function returns an int status (0 = ok, <0 = error);
 * opaque handles; caller-allocated outputs; int64 offsets, int32 indices, float64 values; dense
 * right-hand sides are ROW-major n x r (NumPy C order, what the reference passes); one handle may be used
 * from one thread at a time; no global state.  Functions suffixed _dev take DEVICE pointers (HBM) and
 * enqueue on the handle's stream without synchronising.
 */
#ifndef SCILMM_HIP_H
#define SCILMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
  SCILMM_OK = 0,
  SCILMM_ERR_ARG = -1,        /* bad argument / handle */
  SCILMM_ERR_NOT_PD = -2,     /* V is not positive definite (sksparse: CholmodNotPositiveDefiniteError) */
  SCILMM_ERR_DEVICE = -3,     /* HIP runtime error (no device, launch failure, out of memory) */
  SCILMM_ERR_STATE = -4       /* call order violated (e.g. factorize before values_upload) */
};

typedef struct scilmm_symbolic scilmm_symbolic; /* pattern analysis + device-resident A_k values */
typedef struct scilmm_factor scilmm_factor;     /* numeric factor L of V[P][:,P]                  */

/* Ordering / supernode options; the counterpart of the reference's constructor kwargs
 * SparseCholesky(use_long=False, mode='supernodal', ordering_method='nesdis') (SparseCholesky.py:17-20).
 * Negative / zero fields mean "library default". */
typedef struct scilmm_options {
  int32_t ordering;     /* 0 = approximate minimum degree, 1 = natural, 2 = perm_in, 3 = nested dissection ('nesdis',
                           SparseCholesky.py:17), 4 = whichever of 0 / 3 gives fewer factor flops */
  int32_t relax_small;  /* relaxed amalgamation: always merge when merged width <= this */
  int32_t relax_w1, relax_w2;
  double relax_z1, relax_z2, relax_z3;
  double amd_dense;
  int32_t max_width;    /* split supernodes wider than this (0 = library default) */
  double nd_oksep;      /* nested dissection: accept a separator only below this share of its subgraph (0 = default 0.1;
                           1.0 = always dissect, CHOLMOD's nd_oksep default) */
  double dense_relax;   /* the top of the elimination tree is padded to a dense block-column matrix while padded / true
                           flops stay below this (0 = default: 1.10, and up to 1.25 once the tail is >= 32768 columns
                           wide, where the dense kernel takes it over; negative = never pad) */
} scilmm_options;

typedef struct scilmm_info {
  int32_t n, K, nsuper, nlevels;
  int64_t nnzL;         /* sum_j colcount_j : algorithmic nonzeros of L (BASELINE metric nnz(L)/s) */
  int64_t nnzL_stored;  /* doubles of supernodal panel storage */
  int64_t nnz_pattern;  /* entries of tril(union pattern of the A_k) */
  double flops;         /* sum_j colcount_j^2 (CHOLMOD "fl" convention) */
  int64_t n_rows_total; /* sum_s m_s */
  int64_t n_updates;    /* number of (target, descendant) update pairs */
  double update_flops;  /* algorithmic flops of the supernodal update kernels (lower-triangular count, true structure) */
  double solve_flops_per_rhs; /* 4 nnz(L_stored) : forward + backward sweep per right-hand side */
  double update_flops_executed; /* update_flops plus the explicit zeros of the padded dense tail */
  int32_t dense_first;  /* fronts [dense_first, nsuper) form the dense tail (nsuper: none) */
  double dense_flops;   /* algorithmic update flops among the fronts of the dense tail (true structure): k_dense's work */
} scilmm_info;

/* --- symbolic phase: replaces cholmod_analyze inside sk_cholesky (SparseCholesky.py:23-26), but once per
 * pattern instead of once per evaluation.  indptr[k]/indices[k] is the CSR pattern of mats[k]; VALUES are only ever
 * read from the entries with column <= row (the matrices are symmetric).  Inputs that store both halves with
 * strictly ascending rows (what SciPy hands over) are analysed without any scatter pass -- verified by a hash of the
 * mirrored entries; anything else (one half only, unsorted rows, an unsymmetric pattern, whose upper entries are
 * ignored) takes a slower path with the same result.  perm_in[new] = old or NULL; a given permutation is used exactly
 * as it is, otherwise the library's ordering may be followed by moving the dense tail's fronts to the end of the
 * order (same fill). */
int scilmm_symbolic_create(int32_t n, int32_t K, const int64_t* const* indptr, const int32_t* const* indices,
                           const int32_t* perm_in, const scilmm_options* opts, int32_t ngpus,
                           scilmm_symbolic** out);  /* ngpus: must be >= 1; the analysis itself does not depend on it (the
                                                       distribution is chosen by scilmm_dist_init) */
int scilmm_symbolic_info(const scilmm_symbolic* sym, scilmm_info* info);
/* Copy a named int32/int64 array of the analysis ("perm" replaces factor.P(), SparseCholesky.py:93).
 * Call with out == NULL to get the element count.  Names: perm, iperm, parent, colcount, sn_start, sn_parent,
 * sn_rowptr, sn_rows, sn_loff, sn_level, level_ptr, level_fronts, dense_first, tail_blk_ptr / tail_blk (block pattern
 * of the dense tail's true structure), asm_dst, diag_dst, pat_colptr, pat_row, val_slot:k / val_src:k (value-assembly
 * maps of input matrix k), upd_*, tile_*, combo_* (built on demand), level_tile_*, level_pair_*, inv_off, child_*. */
int scilmm_symbolic_get(const scilmm_symbolic* sym, const char* what, void* out, int64_t* count);
const char* scilmm_symbolic_error(const scilmm_symbolic* sym);
void scilmm_symbolic_free(scilmm_symbolic* sym);
/* Image of an analysis on disk or in /dev/shm (the reference keeps its stage artefacts as files,
 * scilmm/IBDCompute.py:82-84): the 20 s analysis of the 1M config is done once per pattern and NODE instead of once per
 * process (8 ranks of a multi-GPU run, repeated fits).  `key` = the caller's 64-bit hash of everything the analysis
 * depends on (patterns, permutation, options); scilmm_symbolic_load returns SCILMM_ERR_STATE unless the file exists, was
 * written by this build and carries that key -- the caller then analyses afresh.  Host only. */
int scilmm_symbolic_save(const scilmm_symbolic* sym, const char* path, uint64_t key);
int scilmm_symbolic_load(const char* path, uint64_t key, scilmm_symbolic** out);

/* The ordering step of cholmod_analyze alone (SparseCholesky.py:17 ordering_method): fill-reducing permutation of a
 * symmetric CSR pattern (only entries with column < row are read).  method 0 = approximate minimum degree,
 * 1 = nested dissection (graph bisection, minimum degree on the leaves and wherever no separator below 10 % of the
 * subgraph exists), 2 = nested dissection that accepts every separator.  perm_out[new] = old. */
int scilmm_order(int32_t n, const int64_t* indptr, const int32_t* indices, int32_t method, int32_t* perm_out);
/* nnz(L), sum colcount^2 and the largest column count of the factor of pattern[perm][:,perm] (perm NULL = natural):
 * elimination tree + column counts only -- what the ordering study in profiles/ and bench.py's fill figures use. */
int scilmm_fill_count(int32_t n, const int64_t* indptr, const int32_t* indices, const int32_t* perm, int64_t* nnzL,
                      double* flops, int32_t* max_colcount);

/* Frees the host copies of the value-assembly maps (asm_dst, pat_row, val_slot / val_src: 13 GB at the 1M config) once the
 * values are in HBM and the device plan exists; afterwards scilmm_values_upload and the corresponding scilmm_symbolic_get
 * names are no longer available on this handle.  For processes that only evaluate (8 ranks of one node each hold a copy). */
int scilmm_symbolic_release_host_maps(scilmm_symbolic* sym);

/* --- values: one upload per A_k replaces the per-evaluation CSR arithmetic of matrices_weighted_sum
 * (SparseCholesky.py:55-59).  data_k is the CSR data array matching indptr[k]/indices[k]. */
int scilmm_values_upload(scilmm_symbolic* sym, int32_t k, const double* data_k);

/* --- numeric phase: V = sum_k sigma2[k] A_k assembled on the device, then L L^T = V[P][:,P].
 * Replaces matrices_weighted_sum + cholesky_func(V) (SparseCholesky.py:88,92).  A factor handle can be
 * re-used: scilmm_refactorize overwrites its values for new sigma2. On SCILMM_ERR_NOT_PD,
 * *bad_col receives the failing (permuted) column. */
int scilmm_factorize(scilmm_symbolic* sym, const double* sigma2, scilmm_factor** out, int32_t* bad_col);
int scilmm_refactorize(scilmm_factor* fac, const double* sigma2, int32_t* bad_col);
/* Queue the same work and return at once: the host can draw the next normal matrix (np.random.randn(n,100),
 * SparseCholesky.py:50) while the device factorizes.  scilmm_factor_wait -- or any call that uses the factor --
 * completes it and reports SCILMM_ERR_NOT_PD / *bad_col exactly like scilmm_refactorize. */
int scilmm_refactorize_async(scilmm_factor* fac, const double* sigma2);
int scilmm_factor_wait(scilmm_factor* fac, int32_t* bad_col);
void scilmm_factor_free(scilmm_factor* fac);

/* --- multi-GPU: one process per GPU, ONE cohort factored by all of them.  The fronts of the dense
 * tail at the top of the block elimination tree (the separator supernodes: > 99 % of the flops and of the storage of a
 * pedigree factor) are owned 1-D block-cyclically (tail front dense_first + j belongs to rank j % world); everything
 * below (the "prelude") is replicated.  FAN-OUT with rank-local storage: a rank keeps the prelude, its OWN tail panels
 * and a ring of a few panel slots through which the other ranks' panels pass -- a finished panel is broadcast by its
 * owner, applied by every rank to those of its own targets that need it (the sources of a whole group of panels at a
 * time, so the accumulators stay in registers over K = group x 128) and dropped.  Per rank nnz(L) / world + ring instead
 * of nnz(L): this is what lets BASELINE configs[4] (3M individuals, ~1.3 TB of factor) fit 8 x 288 GB.  The triangular
 * sweeps run with one small collective per tail block (forward: all-reduce of the block's accumulated contributions;
 * backward: broadcast of the block's solution); L*R is summed by one all-reduce.  DESIGN.md section 7 has the numbers.
 * The library links no communication library: it calls `fn` at ENQUEUE time, in the same order on every rank, and
 * the callback issues the collective on `comm_stream` (torch.distributed over RCCL in production, gloo in rehearsals):
 *   op 0 = broadcast from `root`, 1 = all-reduce (sum), 2 = all-reduce (min);
 *   buffer 0 = L, 1 = invD, 2 = logd (the arrays given to scilmm_factor_create_external), 3 = the sweeps' work
 *   buffer (scilmm_dist_set_work);  offset / count in doubles, offsets are RANK-LOCAL (the same panel lives at
 *   different offsets on its owner and in a receiver's ring).
 * The engine orders its own streams against `comm_stream` with events (before the call: comm_stream waits for the
 * producer's kernels; after it: an event recorded on comm_stream releases the consumers).  Must be called before the
 * first numeric call on the handle.  A non-positive pivot is reported by EVERY rank (the status word is all-reduced).
 * The reference has no counterpart (single-process CHOLMOD). */
typedef int (*scilmm_comm_fn)(void* ctx, int32_t op, int32_t buffer, int64_t offset, int64_t count, int32_t root);
int scilmm_dist_init(scilmm_symbolic* sym, int32_t rank, int32_t world, void* comm_stream, scilmm_comm_fn fn, void* ctx);
/* Factor storage owned by the caller (so that the communication layer can address it, e.g. as torch tensors):
 * sizes in doubles incl. the slack the kernels over-read -- of THIS rank's share once scilmm_dist_init has been called;
 * then create the handle and use scilmm_refactorize. */
int scilmm_factor_sizes(const scilmm_symbolic* sym, int64_t* L_doubles, int64_t* invD_doubles, int64_t* logd_doubles);
int scilmm_factor_create_external(scilmm_symbolic* sym, double* L, double* invD, double* logd, scilmm_factor** out);
/* Work buffer of the distributed sweeps (right-hand sides, solutions, the forward sweep's accumulator), caller-owned
 * for the same reason: scilmm_dist_work_size doubles; required before the first solve / L*R when world > 1. */
int scilmm_dist_work_size(const scilmm_symbolic* sym, int64_t* doubles);
int scilmm_dist_set_work(scilmm_symbolic* sym, double* work);
/* The distribution rule as data (tests cross-check the CPU rehearsal engine against it): owner[nsuper] (-1 = replicated),
 * loff[nsuper + 1] rank-local panel offsets for `rank` of `world` (last entry: doubles of local panel storage),
 * params = {first distributed front, group size, ring slots}.  Host only. */
int scilmm_dist_layout(const scilmm_symbolic* sym, int32_t rank, int32_t world, int32_t* owner, int64_t* loff, int32_t* params);

/* factor.logdet()  (SparseCholesky.py:40) */
int scilmm_logdet(scilmm_factor* fac, double* out);
/* factor(b): X = V^{-1} B, B and X row-major n x r  (SparseCholesky.py:30,32,52,100,149,153) */
int scilmm_solve(scilmm_factor* fac, const double* B, int32_t r, double* X);
/* (factor.L() @ R)[argsort(P)] : Z = P^T L R  (simulate_vector, SparseCholesky.py:50-51) */
int scilmm_lmul(scilmm_factor* fac, const double* R, int32_t r, double* Z);
/* factor.L() as CSC of the permuted factor (SparseCholesky.py:50); call with vals == NULL to get nnz.
 * On a DISTRIBUTED factor (scilmm_dist_init with world > 1) it returns SCILMM_ERR_STATE with an explanatory message and leaves
 * the factor untouched: a rank holds the prelude and its own tail panels only, and nothing on the hot path needs the gathered
 * factor (simulate_vector goes through scilmm_lmul, which IS collective). */
int scilmm_export_L(scilmm_factor* fac, int64_t* colptr, int32_t* rowidx, double* vals, int64_t* nnz);
/* out[j] = sum_i (A_k U)_ij U_ij  (compute_gradients, SparseCholesky.py:65-66,70); U row-major n x r */
int scilmm_quadforms(scilmm_symbolic* sym, int32_t k, const double* U, int32_t r, double* out);
/* Y = A_k X, row-major n x r (SparseCholesky.py:66,70,160,163) */
int scilmm_spmm(scilmm_symbolic* sym, int32_t k, const double* X, int32_t r, double* Y);
/* the same with DEVICE pointers, asynchronous on the engine's stream (the residual of a refinement sweep without a host round trip) */
int scilmm_spmm_dev(scilmm_symbolic* sym, int32_t k, const double* dX, int32_t r, double* dY);

/* SURVEY section 8f rank 4 -- the exact tr(V^-1 A_k) of the gradient instead of the reference's Monte-Carlo estimate
 * (scilmm/SparseCholesky.py:49-52, :65).  scilmm_selected_inverse replaces, IN PLACE, every stored entry of the factor by
 * the entry of Z = (V[P][:,P])^-1 at the same position (Takahashi recursion over the supernodes from the last level
 * down: Z_RC = -Z_RR (L21 L11^-1), Z_CC = L11^-T L11^-1 - (L21 L11^-1)^T Z_RC; twice the factorization's flops, no
 * second copy of the factor).  The handle is consumed: refactorize before the next solve.  scilmm_inverse_traces then
 * returns out[k] = tr(V^-1 A_k) = sum over A_k's pattern of Z_ij A_k,ij for every matrix k (one streaming pass each);
 * scilmm_export_L on an inverted handle returns the entries of Z on L's pattern.  No counterpart in the reference. */
/* Single-GPU handles only: on a distributed factor both return SCILMM_ERR_STATE with an explanatory message before touching
 * anything -- the factor stays valid and usable (the Takahashi recursion reads, for every front, the inverse entries of ALL
 * later fronts: with rank-local storage that is a second fan-out in the opposite direction, not built).  The default
 * Monte-Carlo trace of the reference needs neither. */
int scilmm_selected_inverse(scilmm_factor* fac);
int scilmm_inverse_traces(scilmm_factor* fac, double* out);

/* BASELINE configs[4] ("fp64 factor with fp32 MFMA fronts"): bits = 32 runs the products of the dense-tail update on
 * the fp32 matrix pipe: the finished tail panels get an fp32 shadow (one rounding per entry; + 50 % tail storage -- of a rank's own
 * panels and ring slots when the tail is distributed --, dropped when the device has no room for it: the operands are then
 * rounded while they are staged), products are
 * summed in fp32 over 256 (staged form: 16) of them and those sums in fp64;
 * everything else -- the subtraction from the panel, potrf, trsm, the solves -- stays fp64.  The factor then has a
 * relative backward error of ~1e-7: callers refine their solves against the exact V (scilmm_spmm), as
 * scilmm_amd.factor.Factor does.  bits = 64 (default) restores the all-fp64 path.  No counterpart in the reference. */
int scilmm_set_front_precision(scilmm_symbolic* sym, int32_t bits);

/* Haseman-Elston moments on the device (reference HE, SparseCholesky.py:192-246, REML's starting
 * point at :121): *frob = sum_ij (A_k1 o A_k2)_ij over the full symmetric matrices, *diag_dot = diag(A_k1) . diag(A_k2),
 * from the value arrays already resident in HBM (one streaming pass).  y'A_k y comes from scilmm_quadforms. */
int scilmm_he_moments(scilmm_symbolic* sym, int32_t k1, int32_t k2, double* frob, double* diag_dot);

/* Dominance relationship matrix on the pattern of the IBD matrix, built on the device (replaces
 * reference scilmm/Matrices/Dominance.py:12-43 `dominance(rel, ibd)`):
 *   out[t] = 1/4 (A[f_i,f_j] A[m_i,m_j] + A[f_i,m_j] A[m_i,f_j])  for the stored entry t = (i, j), i != j;  1 on the diagonal.
 * A: CSR with both halves stored, rows strictly ascending (canonical), explicit zeros removed (the reference calls
 * eliminate_zeros first); parents: n x 2 int32, -1 = unknown (contributes 0).  out has A's layout (same indptr /
 * indices).  Values equal the reference's NumPy arithmetic bit for bit (no fma contraction).
 * scilmm_dominance takes host buffers (upload, one kernel, download); scilmm_dominance_dev device buffers and a HIP
 * stream (0 = default) and only enqueues.  scilmm_dominance_error(): text of the calling thread's last failure. */
int scilmm_dominance(int32_t n, const int64_t* indptr, const int32_t* indices, const double* data, const int32_t* parents,
                     double* out);
int scilmm_dominance_dev(int32_t n, const int64_t* d_indptr, const int32_t* d_indices, const double* d_data,
                         const int32_t* d_parents, double* d_out, void* stream);
const char* scilmm_dominance_error(void);

/* Y = A X for a CSR matrix (both halves stored; any square sparse matrix) and a row-major n x r block, all DEVICE pointers,
 * enqueued on `stream` (0 = default): needs NO symbolic analysis.  The n x 100 products of the Haseman-Elston standard error
 * (reference SparseCholesky.py:259-278 `mat_j.dot(sim_y)`, `H.dot(t)`: four per matrix pair on the host) -- HE is what the
 * reference's authors run above 250k individuals (README.md:63) and it never factorizes.  X and Y must not alias.
 * scilmm_csr_spmm_error(): text of the calling thread's last failure. */
int scilmm_csr_spmm_dev(int32_t n, const int64_t* d_indptr, const int32_t* d_indices, const double* d_data, const double* d_X,
                        int32_t r, double* d_Y, void* stream);
const char* scilmm_csr_spmm_error(void);

/* --- device-pointer variants used by bench.py and by callers that keep data resident in HBM */
int scilmm_solve_dev(scilmm_factor* fac, const double* dB, int32_t r, double* dX);
int scilmm_lmul_dev(scilmm_factor* fac, const double* dR, int32_t r, double* dZ);
int scilmm_quadforms_dev(scilmm_symbolic* sym, int32_t k, const double* dU, int32_t r, double* d_out);
int scilmm_sync(scilmm_symbolic* sym);

/* --- measurement hooks (bench.py): HIP-event time of the last factorize / solve on the handle's
 * stream, in milliseconds, split by phase. */
typedef struct scilmm_timing {
  double assemble_ms, factor_ms, solve_fwd_ms, solve_bwd_ms, lmul_ms, quad_ms;
  int64_t n_launches;
  /* filled when profiling is on: HIP-event time summed per kernel class over the last factorize */
  double update_ms, potrf_ms, trsm_ms;
  int64_t n_update_launches;
  double reduce_cells_ms;
  /* wall time during which at least one update launch was running (early launches of consecutive levels
   * overlap on the two side streams, so update_ms -- the SUM of launch durations -- counts that time twice) */
  double update_union_ms;
  /* the dense-tail kernel alone (k_dense_b / k_dense_h / k_dense32): summed launch durations and launch count of the last factorize */
  double dense_ms;
  int64_t n_dense_launches;
  /* multi-GPU: own tail targets of the last factorize whose late update was issued in two parts -- the sources that had
   * already arrived first, the newest source panel's items after the wait for its broadcast (look-ahead on the chain's
   * critical path); 0 on one GPU */
  int64_t n_late_split;
} scilmm_timing;
int scilmm_last_timing(const scilmm_symbolic* sym, scilmm_timing* out);
/* Bracket every kernel class of the factorization with HIP events on the handle's stream (bench.py's
 * live roofline figure).  Off by default.  on = 2 additionally queues every look-ahead ("early") update launch on ONE
 * side stream instead of alternating between two, so that the durations of consecutive launches of the dominant kernel
 * do not overlap each other (the clean per-launch figure; slower as a whole -- a measurement mode, same results). */
int scilmm_set_profiling(scilmm_symbolic* sym, int32_t on);

/* --- SURVEY section 8(f) rank 1 ("next"): pedigree -> IBD matrix, the producer of the hot path's input.
 * Replaces Numerator.LD + create_numerator (scilmm/Matrices/Numerator.py:5-38) and, with count_only != 0,
 * Relationship.count_IBD_nonzero (scilmm/Matrices/Relationship.py:38-61).  parents: n x 2 int32, -1 = unknown,
 * individuals in topological order (parents first).  Host C++/OpenMP; no device needed. */
typedef struct scilmm_ibd scilmm_ibd;
/* count_only: 0 = pattern + values, 1 = the number of structural nonzeros only, 2 = the PATTERN only (no values: what the
 * symbolic analysis needs when the values are then computed on the device, scilmm_ibd_values_device) */
int scilmm_ibd_build(int32_t n, const int32_t* parents, int32_t count_only, scilmm_ibd** out, int64_t* nnz);
int scilmm_ibd_sizes(const scilmm_ibd* h, int64_t* nnz_A, int64_t* nnz_L);
/* A: symmetric CSR (both triangles, sorted); L: ancestor-weight rows; D, F: length n.  NULL pointers are skipped. */
int scilmm_ibd_export(const scilmm_ibd* h, int64_t* a_indptr, int32_t* a_indices, double* a_data, int64_t* l_indptr,
                      int32_t* l_indices, double* l_data, double* D, double* F);
void scilmm_ibd_free(scilmm_ibd* h);
/* The VALUES of the IBD matrix computed on the device, straight into the HBM-resident value slots of matrix k of `sym` --
 * they never exist on the host and never cross PCIe (replaces create_numerator's L D L^T, Numerator.py:37-38, and the
 * 8 bytes per entry of scilmm_values_upload: 7.3 GB at the 1M config).  Tabular recursion on the analysed pattern,
 * A[i,j] = 1/2 (A[f_i,j] + A[m_i,j]) for i > j, A[i,i] = 1 + 1/2 A[f_i,m_i], one launch per generation sum (every entry
 * depends on entries with a smaller gen(a) + gen(b) only).  parents: n x 2, -1 = unknown, individuals in pedigree order
 * (the row order of the matrices given to scilmm_symbolic_create); the analysed pattern must contain every pair with a
 * common ancestor (scilmm_ibd_build's pattern does).  Exact: the values are dyadic rationals. */
int scilmm_ibd_values_device(scilmm_symbolic* sym, int32_t k, int32_t n, const int32_t* parents);
/* The VALUES of the dominance relationship matrix computed on the device from the IBD values of matrix k_src already
 * resident in HBM, straight into the value slots of matrix k_dst (replaces reference scilmm/Matrices/Dominance.py:12-43
 * `dominance(rel, ibd)` and the upload of its result; scilmm_dominance / scilmm_dominance_dev are the CSR-layout forms):
 *   D[a,b] = 1/4 (A[f_a,f_b] A[m_a,m_b] + A[f_a,m_b] A[m_a,f_b]) for a != b, 1 on the diagonal, on every slot of the
 * analysed pattern (matrix k_dst is given to scilmm_symbolic_create with matrix k_src's pattern, as the reference builds
 * it).  Bit-identical to the reference's NumPy arithmetic.  With scilmm_ibd_values_device, BASELINE configs[4]'s two
 * variance components are built where they are used: only the pattern and the parent table reach a rank. */
int scilmm_dominance_values_device(scilmm_symbolic* sym, int32_t k_dst, int32_t k_src, int32_t n, const int32_t* parents);
/* Matrix k's device-resident values in PATTERN-SLOT order (scilmm_symbolic_get "pat_colptr" / "pat_row": permuted CSC of
 * the lower triangle, diagonal first; n values for a diagonal-only matrix) -- for tests and diagnostics. */
int scilmm_values_download(scilmm_symbolic* sym, int32_t k, double* slots_out);

/* --- the on-disk format in front of the path.  Replaces scipy.io.mmread(path).tocsr()
 * (scilmm/SparseCholesky.py:399) by a memory-mapped, all-cores parse of the MatrixMarket coordinate file (real /
 * integer / pattern; general / symmetric / skew-symmetric; duplicates summed, symmetric storage expanded): CSR with
 * sorted indices.  Host C++/OpenMP; no device needed. */
typedef struct scilmm_mm scilmm_mm;
int scilmm_mm_read(const char* path, scilmm_mm** out, int32_t* nrows, int32_t* ncols, int64_t* nnz);
int scilmm_mm_export(const scilmm_mm* m, int64_t* indptr, int32_t* indices, double* data);
const char* scilmm_mm_error(const scilmm_mm* m);
void scilmm_mm_free(scilmm_mm* m);

const char* scilmm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SCILMM_HIP_H */
