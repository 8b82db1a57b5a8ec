/*
 * msweep_core.h -- C ABI of the MI355X-native abundance-estimation core (libmsweep_core.so).
 *
 * This is the drop-in boundary for mSWEEP's estimation hot path.  Every entry point cites
 * the reference interface (file:line under /root/reference) it replaces.  Plain pointers
 * and sizes only; all pointers are HOST pointers unless a name ends in `_dev`.  Return
 * value 0 = success, non-zero = error (text via msw_last_error); the C++ shim
 * (msweep_amd/cpp/rcgpar_hip.hpp) turns non-zero into std::runtime_error so that
 * mSWEEP's try/catch blocks (src/mSWEEP.cpp:400-406, 506-511) behave as with rcgpar.
 *
 * Threading: one caller thread per handle (the reference calls from its single main
 * thread, src/mSWEEP.cpp:402,507).  A handle is bound to one HIP device.
 */
#ifndef MSWEEP_CORE_H
#define MSWEEP_CORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msw_core *msw_handle;

/* --algorithm (src/mSWEEP.cpp:127,192-204): rcggpu -> MSW_ALGO_RCG, emgpu -> MSW_ALGO_EM. */
enum { MSW_ALGO_RCG = 0, MSW_ALGO_EM = 1 };
/* --emprecision (src/mSWEEP.cpp:129,202; MSW_ALGO_EM only -- RCG ignores it, as rcgpar's rcg_optl_* have no such
 * argument).  MSW_PREC_FLOAT (round 5) is REAL fp32 arithmetic (msweep_amd/csrc/em_f32_kernels.hpp): likelihood values,
 * weights, row sums and quotients in fp32, the log-likelihood rounded to float once per iteration for the stop rule
 * (which is why a float run stops after a few hundred iterations where double reaches --max-iters, as the reference's
 * does: docs/gpubenchmarks.md:20-22), column sums in exact 64-bit fixed point.  Served by fp32 kernels for 4-byte
 * offset records with table and group vectors in LDS, up to 6144 groups, one GPU; other layouts (8-byte / index /
 * value records, dense matrices without background structure, EC-sharded solves) run the fp64 kernels under
 * MSW_PREC_FLOAT as until round 4 -- msw_timing::em_float_kernels says which it was. */
enum { MSW_PREC_DOUBLE = 0, MSW_PREC_FLOAT = 1 };

/* ---- lifetime ---------------------------------------------------------------------- */
int msw_core_create(int device, msw_handle *out);
void msw_core_destroy(msw_handle h);
/* last error text of the handle (or of the failed msw_core_create when h == NULL) */
const char *msw_last_error(msw_handle h);
/* library / device identification string ("msweep_core <ver> gfx950 ...") */
const char *msw_core_version(void);

/* ---- likelihood upload -------------------------------------------------------------
 * Replaces the materialised `seamat::DenseMatrix<double> log_likelihoods`
 * (include/Likelihood.hpp:85,176-185) that rcg_optl receives as `ll_mat`
 * (src/mSWEEP.cpp:176,402,507).  The likelihood stays resident on the device across the
 * 1 + --iters solves of one grouping (same `logl` object at :402 and :507).            */

/* Dense G x E, rows = groups, element (g, j) at L[g*ld + j] -- the seamat layout read via
 * operator()(row, col) (include/Likelihood.hpp:182,265); needed for --read-likelihood
 * (include/Likelihood.hpp:224-253).
 * A matrix of the shape fill_ll_mat writes (include/Likelihood.hpp:176-185: one background value,
 * log(zi), in at least 3/4 of the cells and at most 65536 distinct values elsewhere) is re-expressed
 * on the device as the CSR-of-ECs form below -- the same numbers, bit for bit
 * (msw_core_get_dense_logl returns the input), any group count, solved by the sparse sweeps.
 * Any other matrix is kept dense up to n_groups = 8192; beyond, it is re-expressed whatever its shape (every
 * cell that differs from the most frequent value listed, up to 2^28 of them).  msw_core_shape's nnz tells which. */
int msw_core_set_dense_logl(msw_handle h, const double *L, size_t n_groups, size_t n_ecs,
                            size_t ld);

/* CSR-of-ECs form of the same matrix: for EC j the cells k in [rowptr[j], rowptr[j+1])
 * name group grp[k] hit cnt[k] >= 1 times; L(g, j) = lut[g*lut_ld + cnt] and every cell
 * not listed is `logzi` (= lut[g*lut_ld + 0] for every g, include/Likelihood.hpp:98).
 * `lut` is the precalc_lls table (include/Likelihood.hpp:92-107). */
int msw_core_set_csr(msw_handle h, const uint64_t *rowptr, const uint32_t *grp,
                     const uint32_t *cnt, const double *lut, size_t lut_ld, double logzi,
                     size_t n_groups, size_t n_ecs);

/* Builds the CSR-of-ECs likelihood on the device straight from the pseudoalignment:
 * replaces LL_WOR21::fill_ll_mat + fill_ec_counts (include/Likelihood.hpp:109-195) and
 * ConstructAdaptiveLikelihood (:333-380).
 *   ec_tptr/ec_targets : for EC i the targets it aligned to (the set bits of
 *                        Alignment::operator()(i, j), include/mSWEEP_alignment.hpp:223)
 *   target_group[T]    : Alignment::get_groups() (include/mSWEEP_alignment.hpp:241)
 *   group_sizes[G]     : Grouping::get_sizes()   (include/Grouping.hpp:35-50)
 *   ec_counts[E]       : Alignment::reads_in_ec  (include/mSWEEP_alignment.hpp:220)
 *   q, e               : -q / -e flags (bb_constants, include/Likelihood.hpp:212-214)
 *   min_hits           : --min-hits mask (include/Likelihood.hpp:141-171)
 * Outputs: *n_groups_out = groups kept, mask_out[G] = groups_considered() (:331),
 * logc_out[E] = log_counts() (:328); either may be NULL.
 * EC-sharded (msw_core_set_comm called first; every rank passes its own block of ECs and the same targets /
 * groups): the --min-hits counts are all-reduced (G unsigned 64-bit integers, once), so every rank keeps
 * the same groups in the same order. */
int msw_core_build_likelihood(msw_handle h, const uint64_t *ec_tptr,
                              const uint32_t *ec_targets, size_t n_ecs,
                              const uint32_t *target_group, size_t n_targets,
                              const uint64_t *group_sizes, size_t n_groups,
                              const uint64_t *ec_counts, double q, double e,
                              double zero_inflation, size_t min_hits,
                              size_t *n_groups_out, uint8_t *mask_out, double *logc_out);

/* Downloads the dense G' x E log-likelihood (rows = groups) of the resident likelihood --
 * what Likelihood::log_mat() (include/Likelihood.hpp:325) would hold; used by
 * --write-likelihood (include/Likelihood.hpp:255-273) and the parity tests. */
int msw_core_get_dense_logl(msw_handle h, double *L_out, size_t ld);
/* FNV-1a hash of the resident CSR-of-ECs layout (EC order, slice geometry, records): test hook that
 * holds the device packer against its host reference implementation (MSWEEP_HOST_PACK=1). */
int msw_core_layout_hash(msw_handle h, uint64_t *hash_out);
/* shape of the resident likelihood */
int msw_core_shape(msw_handle h, size_t *n_groups, size_t *n_ecs, size_t *nnz);
/* How the resident CSR-of-ECs likelihood is laid out for the sweeps: reporting only (bench.py, the
 * timing tools); no reference counterpart. */
typedef struct msw_layout_info {
  int32_t record_bytes;        /* 4 or 8 bytes per listed cell; 12: value records (the cell's log-likelihood inline) */
  int32_t index_records;       /* 1: (group, entry) index records + hybrid slot area; 0: byte-offset records */
  int32_t groups_in_lds;       /* the per-group vectors of both sweeps live in LDS */
  int32_t table_in_lds;        /* the whole slot area lives in LDS */
  int32_t passB_mode;          /* k_passB GMODE (sweep_kernels.hpp) */
  uint32_t slot_entries;       /* 16-byte entries of the slot area */
  uint32_t slot_entries_in_lds;
  uint32_t n_slices, n_long_ecs;
  uint64_t rows;               /* slice rows (64 records each), padding included */
  uint64_t rows_from_memory;   /* ... whose table entries are gathered from memory (cold segments, whole slices) */
  uint32_t slices_by_lanes[7]; /* slices whose ECs take 64, 32, 16, 8, 4, 2, 1 lanes each (ECs of 513..1024, 257..512,
                                * .., 17..32, <= 16 cells; MSWEEP_MULTILANE=0: all in the last) */
  uint32_t max_rows;           /* rows of the longest slice (<= 16: every slice on the register path of the sweeps) */
  int32_t bank_scheduled;      /* the cells of every slice were ordered for the LDS banks (msw_core_set_pack_schedule) */
  int32_t passB_reg_cells;     /* records per slice lane pass B holds in registers: 16, or 8 = its short-slice instantiation
                                * with 16 wavefronts per workgroup (slices of > 8 rows hold <= 1/20 of the rows) */
  uint64_t rows_over_8;        /* rows of the slices of more than 8 rows (0 when not counted: index / wide / value records) */
} msw_layout_info;
int msw_core_layout_info(msw_handle h, msw_layout_info *out);
/* Whether the NEXT likelihood made resident on the handle (msw_core_set_csr, msw_core_build_likelihood, a dense
 * matrix re-expressed as CSR-of-ECs) gets its cells ordered for the LDS banks of the sweeps (the order of
 * the cells inside an EC is free).  enabled = 1 (default): an iteration is 5-6 % faster (cfg3: 0.177 against 0.187 ms)
 * and the upload 10 ms slower (cfg3; 28 ms at cfg5) -- pays from about the 1 000th iteration on the same likelihood,
 * i.e. for bootstrap runs (src/mSWEEP.cpp:496-518); enabled = 0: the cells keep their CSR order -- the faster way to
 * ONE solve.  Same results either way (the order of the additions inside an EC changes: rounding).  The drivers set
 * it from --iters.  No reference counterpart. */
int msw_core_set_pack_schedule(msw_handle h, int enabled);

/* ---- solver options ------------------------------------------------------------------
 * The knobs of the optimiser loops that are restated from memory of rcgpar v1.2.1 (the library is an un-vendored
 * FetchContent dependency of the reference, CMakeLists.txt:274-311).  The defaults are the
 * restatement; every alternative reading is selectable, here and in the oracle (oracle/msweep_oracle.h
 * orc_rcg_opts / orc_em_opts), so that a result from a real mSWEEP run can decide between them
 * (tests/golden/external/README.md).  Options persist on the handle and apply to every later solve and bootstrap.
 *   MSW_OPT_CHECK_EVERY n  the stop rule `bound - oldbound < tol && !didreset` (and EM's) is tested after iterations
 *                          n, 2n, ... only.  Default 1.  (Every iteration count the reference publishes is a multiple
 *                          of 5, docs/gpubenchmarks.md:15-25: rcgpar logs every 5th iteration -- the default's
 *                          reading -- or tests on that grid, n = 5.)
 *   MSW_OPT_INIT_BOUND  b  the value `bound` holds before the first iteration; default -100000 (a first bound below
 *                          it sends iteration 0 through the steepest-descent retry).
 *   MSW_OPT_EM_PRIOR 0|1   em_torch's M-step: 0 = MAP, the Dirichlet prior as pseudo-counts alpha0 - 1 (default);
 *                          1 = ML, theta_g = sum_j c_j q_gj / sum_j c_j (alpha0 ignored).
 *   MSW_OPT_EM_STOP  0|1   em_torch's stop: 0 = gain of the count-weighted log-likelihood < tol (default);
 *                          1 = largest move of a mixture weight in the M-step < tol. */
enum { MSW_OPT_CHECK_EVERY = 0, MSW_OPT_INIT_BOUND = 1, MSW_OPT_EM_PRIOR = 2, MSW_OPT_EM_STOP = 3 };
int msw_core_set_option(msw_handle h, int option, double value);
int msw_core_get_option(msw_handle h, int option, double *value);

/* ---- solve --------------------------------------------------------------------------
 * Replaces rcgpar::rcg_optl_torch / rcg_optl_omp / em_torch as called from rcg_optl()
 * (src/mSWEEP.cpp:176-205) followed by rcgpar::mixture_components[_torch]
 * (src/mSWEEP.cpp:419-423, 512-516):
 *   logc[E]   = log_times_observed (natural log of EC counts; -inf allowed,
 *               src/BootstrapSample.cpp:70)
 *               NULL = the log counts msw_core_build_likelihood left on the device (no 8 * E
 *               byte upload per solve; an error for likelihoods that were uploaded instead)
 *   alpha0[G] = prior_counts (src/mSWEEP.cpp:391-398)
 *   tol, max_iters = --tol / --max-iters (src/mSWEEP.cpp:123-125)
 * theta_out[G] receives mixture_components(gamma, logc).  iters_out / bound_out optional. */
int msw_core_solve(msw_handle h, const double *logc, const double *alpha0, double tol,
                   size_t max_iters, int algo, int prec, double *theta_out,
                   size_t *iters_out, double *bound_out);

/* The same call split in two for callers that keep the inputs resident (bench.py's timed
 * region, the bootstrap driver): msw_core_prepare uploads logc / alpha0 and forms the EC
 * multiplicities on the device; msw_core_run executes the optimiser on the prepared inputs.
 * msw_core_solve == msw_core_prepare + msw_core_run. */
int msw_core_prepare(msw_handle h, const double *logc, const double *alpha0);
int msw_core_run(msw_handle h, double tol, size_t max_iters, int algo, int prec,
                 double *theta_out, size_t *iters_out, double *bound_out);

/* The G x E log-responsibility matrix gamma of the last solve (the DenseMatrix rcg_optl
 * returns, src/mSWEEP.cpp:195,199,203; consumed by Sample::store_probs / write_probs /
 * mGEMS binning, src/mSWEEP.cpp:402,451,478-487).  Row-major, rows = groups, leading
 * dimension ld >= E.  Materialised on demand from the structured state. */
int msw_core_gamma(msw_handle h, double *gamma_out, size_t ld);

/* The columns [ec_begin, ec_end) of the same matrix: gamma_out[g * ld + (j - ec_begin)], ld >= ec_end -
 * ec_begin.  What Sample::write_probs (src/Sample.cpp:63-85: one output line per EC) and the mGEMS binning
 * input (src/mSWEEP.cpp:437-469) consume EC by EC: at the size of BASELINE's configurations the whole matrix
 * is hundreds of GB (G x E x 8 B = 375 GB at 10 M reads x 5 k groups), a block of it is not. */
int msw_core_gamma_block(msw_handle h, size_t ec_begin, size_t ec_end, double *gamma_out, size_t ld);

/* Per-iteration diagnostics of the last solve (what rcgpar logs every 5th iteration to
 * the verbose stream, src/mSWEEP.cpp:198): arrays of length n (<= max recorded, 4096);
 * theta_trace is n x G or NULL.  Returns the number of iterations recorded via *n_out. */
int msw_core_trace(msw_handle h, size_t n, double *bound, double *newnorm, double *beta,
                   int32_t *didreset, double *theta_trace, size_t *n_out);
/* how many leading iterations keep a theta snapshot (default 0) */
int msw_core_set_trace_theta(msw_handle h, size_t n_iters);

/* ---- bootstrap ----------------------------------------------------------------------
 * Replaces BootstrapSample::init_bootstrap / construct / resample_counts
 * (src/BootstrapSample.cpp:33-73) and the replicate loop (src/mSWEEP.cpp:496-518).
 *   ec_counts[E]      Alignment::reads_in_ec as uint32 (src/BootstrapSample.cpp:38-42)
 *   seed              --seed narrowed to int32 (include/Sample.hpp:169,172)
 *   bootstrap_count   draws per replicate (src/BootstrapSample.cpp:56)
 *   replicates [rep_begin, rep_end) of the ONE sequential mt19937_64 stream are solved
 *   (a rank of an N-GPU job passes its own slice; the stream position of replicate b is
 *   b * bootstrap_count draws).
 * theta_out is (rep_end - rep_begin) x G, row b = abundances of replicate rep_begin+b,
 * normalised by the resampled total as src/mSWEEP.cpp:513 does.  iters_out optional.
 * A replicate whose solve fails numerically (likelihood underflow, non-finite bound: the cases in which
 * the reference writes NaN abundances) yields a row of NaN and the call still returns 0 -- the other
 * replicates are unaffected; device errors and bad arguments fail the whole call. */
int msw_core_bootstrap(msw_handle h, const uint32_t *ec_counts, int32_t seed,
                       size_t bootstrap_count, size_t rep_begin, size_t rep_end,
                       const double *alpha0, double tol, size_t max_iters, int algo,
                       int prec, double *theta_out, size_t *iters_out);

/* The whole replicate loop of src/mSWEEP.cpp:496-518 over the GPUs of one node: rank r of `comm`
 * (msw_comm_create_rccl: one process per GPU; msw_comm_create_local: thread-ranks) solves the contiguous
 * block [n_replicates * r / P, n_replicates * (r + 1) / P) of the one sequential stream and ONE all-gather
 * (RCCL over xGMI) leaves the complete n_replicates x G block in replicate order -- the rows
 * 1..n_replicates of bootstrap_results (include/Sample.hpp:157,180) -- on EVERY rank; results do not
 * depend on the number of ranks (more ranks than replicates: the surplus ranks solve nothing and still take
 * part in the exchange).  Every rank holds the same resident likelihood and passes the same
 * arguments.  iters_out[n_replicates] optional.  A rank whose block fails still joins the all-gather and
 * reports through a status word: the call then returns non-zero on EVERY rank and no rank is left waiting. */
struct msw_comm;
int msw_core_bootstrap_dist(msw_handle h, struct msw_comm *comm, const uint32_t *ec_counts, int32_t seed,
                            size_t bootstrap_count, size_t n_replicates, const double *alpha0, double tol,
                            size_t max_iters, int algo, int prec, double *theta_out, size_t *iters_out);

/* msw_core_bootstrap and msw_core_bootstrap_dist solve the replicates on solver states of their own:
 * msw_core_gamma / msw_core_trace afterwards still describe the last msw_core_solve on the handle (the
 * reference writes the probabilities of the un-resampled estimate, src/mSWEEP.cpp:437-493). */

/* Only the resampling step: counts_out is (rep_end-rep_begin) x E uint32, bit-exact with
 * std::discrete_distribution<uint32_t> driven by std::mt19937_64(seed). */
int msw_core_resample_counts(msw_handle h, const uint32_t *ec_counts, size_t n_ecs,
                             int32_t seed, size_t bootstrap_count, size_t rep_begin,
                             size_t rep_end, uint32_t *counts_out);

/* ---- EC-sharded single solve over several GPUs --------------------------------------
 * The reference scales one solve only by OpenMP threads; rcgpar's MPI variant partitions the ECs
 * (columns) over ranks and all-reduces N, |g|^2 and the bound every iteration.
 * Here every rank holds a handle with a contiguous block of ECs (set_csr / build_likelihood /
 * set_dense_logl on the LOCAL ECs; all ranks pass the same groups and alpha0).  After
 * msw_core_set_comm, msw_core_solve / _prepare / _run take the local logc, exchange one scalar
 * and one (G + 4)-vector per iteration and return the SAME theta / iteration count on every rank.
 *   msw_comm_create_rccl : one process per GPU, RCCL over xGMI; `id` from msw_comm_unique_id on
 *                          rank 0, distributed by the caller (e.g. torch.distributed broadcast).
 *   msw_comm_create_local: the ranks are host threads of ONE process; out[] receives nranks
 *                          communicators (host-staged exchange, summed in rank order). */
typedef struct msw_comm *msw_comm_t;
int msw_comm_unique_id(unsigned char id_out[128]);
int msw_comm_create_rccl(const unsigned char id[128], int rank, int nranks, int device, msw_comm_t *out);
int msw_comm_create_local(int nranks, msw_comm_t *out);
/* msw_comm_create_shm: the ranks are PROCESSES of one host that meet in a POSIX shared-memory segment `name` ("/..."),
 * host-staged like the thread-ranks.  Test infrastructure for the process-per-rank paths on a box with one GPU (RCCL
 * refuses two ranks on one device): with MSWEEP_ALLREDUCE=peer the inboxes travel as hipIpc handles, as under RCCL. */
int msw_comm_create_shm(const char *name, int rank, int nranks, int device, msw_comm_t *out);
void msw_comm_destroy(msw_comm_t c);
/* ranks of the communicator and this rank's index (either pointer may be NULL) */
int msw_comm_size(msw_comm_t c, int *nranks, int *rank);
/* what RCCL itself reports for the communicator (ncclCommCount); 0 for an in-process communicator */
int msw_comm_rccl_count(msw_comm_t c, int *count);
/* recv[r * n .. (r + 1) * n) = rank r's send[0 .. n); host buffers, blocking (ncclAllGather through
 * device staging buffers).  The exchange step of the bootstrap (src/mSWEEP.cpp:513-517 stores every
 * replicate's abundances in one table, include/Sample.hpp:157). */
int msw_comm_allgather(msw_comm_t c, const double *send, size_t n, double *recv);
/* The per-iteration exchange of the sharded solve on host buffers (staged through device memory, blocking): ints[0..ni)
 * and reals[0..nr) are replaced by their sums over the ranks -- integers exactly, doubles in rank order, the same bits
 * on every rank.  Either part may be empty.  The transport is the communicator's: ncclAllReduce, the in-process
 * staging of thread-ranks, or -- MSWEEP_ALLREDUCE=peer in the environment when the communicator is created -- one
 * kernel that writes the message into every peer's inbox (msweep_amd/csrc/peer_comm.hpp).  Tests and timing. */
int msw_comm_allreduce(msw_comm_t c, uint64_t *ints, size_t ni, double *reals, size_t nr, int repeats,
                       double *ms_per_call);
/* comm == NULL detaches.  The communicator must outlive the solves that use it. */
int msw_core_set_comm(msw_handle h, msw_comm_t comm);
/* last error text of the msw_comm_* calls of this thread */
const char *msw_comm_last_error(void);

/* ---- pseudoalignment input (host code, no GPU) -------------------------
 * Replaces mSWEEP::Alignment::read + collapse (include/mSWEEP_alignment.hpp:97-215) for Themisto
 * plaintext files ("read_id target target ..." per line; one file per strand): reads -> sets of
 * targets, strands merged by intersection / union (--themisto-mode, :123-133), unaligned reads
 * dropped, reads keyed by the reference's 64-bit hash of their target set (:152-156) and collapsed
 * into equivalence classes in ascending hash order.  The export is what msw_core_build_likelihood
 * takes: ec_tptr[n_ecs + 1], ec_targets[n_hits] (ascending target ids of each EC's first read),
 * ec_counts[n_ecs] (Alignment::reads_in_ec, :220), plus ec_rptr[n_ecs + 1] / ec_reads[n_aligned]
 * (Alignment::reads_assigned_to_ec, :229: the read ids of every EC, ascending).  n_reads is the line
 * count of the last strand (Alignment::n_reads, :219).  Any output pointer may be NULL.
 * gzip-compressed files (Themisto --gzip-output; the reference opens its inputs through bxzstr, which detects
 * the compression by its magic bytes) are inflated with zlib; bzip2 / xz files are refused by name.
 * The compact alignment-writer format is not supported (BitMagic): convert to plaintext. */
typedef struct msw_alignment *msw_alignment_t;
#define MSW_MERGE_INTERSECTION 0
#define MSW_MERGE_UNION 1
int msw_alignment_read(const char *const *paths, size_t n_paths, size_t n_targets, int merge_mode,
                       msw_alignment_t *out);
int msw_alignment_shape(msw_alignment_t a, size_t *n_ecs, size_t *n_reads, size_t *n_hits, size_t *n_aligned);
int msw_alignment_export(msw_alignment_t a, uint64_t *ec_tptr, uint32_t *ec_targets, uint64_t *ec_counts,
                         uint64_t *ec_rptr, uint32_t *ec_reads);
/* The same five arrays WITHOUT a copy: pointers into the handle's own storage, valid until msw_alignment_destroy (round 5:
 * the export of cfg3's alignment -- 0.9 GB into freshly mapped memory, on one thread -- was a fifth of the whole
 * text-to-abundances time).  Any output pointer may be NULL.  msw_alignment_export copies on the reader's threads. */
int msw_alignment_view(msw_alignment_t a, const uint64_t **ec_tptr, const uint32_t **ec_targets, const uint64_t **ec_counts,
                       const uint64_t **ec_rptr, const uint32_t **ec_reads);
void msw_alignment_destroy(msw_alignment_t a);
/* The same reader ON THE DEVICE of handle h (round 5; msweep_amd/csrc/host_reader.inc): the text goes to device memory
 * as it is read and the parse, the rows by read id, the paired-end merge, the reference's hash, the sort and the
 * classes are kernels; the five arrays stay in device memory (msw_core_build_likelihood_aln consumes them there) and
 * msw_alignment_view / _export copy them out on first use.  Same outcome as msw_alignment_read, array for array.
 * Text the host parser would not take silently (anything but digits, blanks and line ends, ids beyond 32 bits,
 * target ids out of range) is handed to msw_alignment_read, whose result or error message stands
 * (msw_alignment_last_error / msw_last_error). */
int msw_alignment_read_device(msw_handle h, const char *const *paths, size_t n_paths, size_t n_targets, int merge_mode,
                              msw_alignment_t *out);
/* 1: the handle's arrays are device-resident (msw_alignment_read_device served the text with its kernels); 0: host
 * arrays (msw_alignment_read, or the host parser behind the device entry). */
int msw_alignment_on_device(msw_alignment_t a);
/* msw_alignment_read_device keeps its device temporaries (~5 bytes per byte of text) on the handle between calls: a
 * repeated read allocates nothing, and nothing is freed in front of the likelihood build (memory given back is scrubbed
 * before it is handed out again: 0.2 s at 10 M reads).  msw_core_trim gives them back; msw_core_destroy does too. */
int msw_core_trim(msw_handle h);
/* msw_core_build_likelihood on an alignment handle: arrays resident on h's device are read where they lie (no copy
 * through the host); any other handle goes through its host arrays.  ec_counts = the classes' read counts. */
int msw_core_build_likelihood_aln(msw_handle h, msw_alignment_t a, const uint32_t *target_group, size_t n_targets,
                                  const uint64_t *group_sizes, size_t n_groups, double q, double e,
                                  double zero_inflation, size_t min_hits, size_t *n_groups_out, uint8_t *mask_out,
                                  double *logc_out);
/* error text of the last failed msw_alignment_read of this thread (the reference's messages) */
const char *msw_alignment_last_error(void);

/* ---- measurement hooks (used by bench.py; no effect on results) ---------------------- */
/* Device time (ms, HIP events on the solve stream) and launch counts of the dominant
 * kernels during the last solve: pass A (gradient norm sweep) and pass B (softmax /
 * column-sum / ELBO sweep). */
typedef struct msw_timing {
  double solve_ms;       /* whole solve loop, events on the solve stream */
  double passA_ms;       /* summed over launches (only when profiling is enabled) */
  double passB_ms;
  uint64_t passA_launches;
  uint64_t passB_launches;
  uint64_t iters;
  uint64_t bytes_passA;  /* algorithmic bytes per launch */
  uint64_t bytes_passB;
  double collective_ms;  /* EC-sharded solve, profiling enabled: the all-reduces of the solve (one scalar after pass A,
                          * one (3 G + 4)-word vector after pass B, per iteration) with the small kernels that pack
                          * them, summed over the launches; events on the solve stream.  0 without a communicator. */
  uint64_t collectives;  /* how many */
  uint64_t em_float_kernels; /* 1: the last solve was an EM run under MSW_PREC_FLOAT served by the fp32 kernels */
} msw_timing;
int msw_core_set_profiling(msw_handle h, int enabled);
int msw_core_last_timing(msw_handle h, msw_timing *out);
/* Host wall-clock split of the last msw_core_bootstrap / msw_core_bootstrap_dist on the handle (bench.py's
 * cfg4 leg): where a rank's time went.  No reference counterpart. */
typedef struct msw_bootstrap_timing {
  double table_ms;      /* init_bootstrap (src/BootstrapSample.cpp:33-44): cumulative table on the host + upload */
  double solve_ms;      /* this rank's block of replicates: seek / jump-ahead, resampling (under the solves), solves */
  double gather_ms;     /* msw_core_bootstrap_dist: the all-gather, waiting for the slowest rank included */
  uint64_t replicates;  /* solved on this rank */
  uint64_t iterations;  /* summed over them */
  uint64_t table_reused; /* 1: the call brought the EC counts of the resident table again (compared, not rebuilt) */
} msw_bootstrap_timing;
int msw_core_last_bootstrap_timing(msw_handle h, msw_bootstrap_timing *out);
/* Reporting (tests, diagnostics): how many equivalence classes pass B has evaluated through the cancellation guard
 * (Z formed over every group instead of background + listed cells) since the likelihood became
 * resident on this handle, summed over the iterations.  No reference counterpart. */
int msw_core_guarded_visits(msw_handle h, uint64_t *out);
/* Measurement only (bench.py's roofline object): the streaming rates this device reaches on n_bytes of HBM --
 * a read-only 16-byte-load sweep (the shape of the sweeps' record stream) and the triad a = b + 3 c -- best of
 * `reps` launches after two warm-up launches, in GB/s (1e9).  SURVEY.md 8(d): the practical ceiling beside the
 * 8 TB/s specification.  No reference counterpart. */
int msw_core_hbm_stream_rates(msw_handle h, size_t n_bytes, int reps, double *read_gbs, double *triad_gbs);
/* fixed-iteration mode for benchmarking: run exactly max_iters iterations (tol ignored) */
int msw_core_set_fixed_iters(msw_handle h, int enabled);
/* n_iters MORE iterations of the fixed-iteration RCG solve that last ran on the handle (msw_core_run in
 * fixed-iteration mode): the optimiser state carries on where it stood.  bench.py's W warm-up steps are
 * msw_core_run(W), its K timed steps msw_core_continue(K).  *iters_out = total iterations so far. */
int msw_core_continue(msw_handle h, size_t n_iters, double *theta_out, size_t *iters_out, double *bound_out);

#ifdef __cplusplus
}
#endif
#endif
