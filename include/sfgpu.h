/*
 * sfgpu.h -- C ABI of the MI355X-native Sailfish quantification core (libsfgpu.so).
 *
 * This is the drop-in boundary behind Sailfish's C++ host code: every entry point replaces one
 * seam of the reference (file:line under the reference tree given next to each declaration;
 * INTEGRATION.md shows the adaptor a maintainer would compile into `sailfish quant`).
 *
 * Conventions
 *   - plain C: opaque handles, plain pointers and sizes, int error codes, no exceptions.
 *   - pointers named d_* are DEVICE pointers (HBM of the current HIP device), h_* are HOST
 *     pointers.  All buffers are caller-owned; handles own only their internal scratch.
 *   - `stream` is a hipStream_t passed as void* (NULL = the HIP default stream).  Work is
 *     enqueued on that stream; functions documented as "synchronous" wait for it.
 *   - errors: 0 = ok, otherwise an SFGPU_ERR_* code; sfgpu_last_error() gives the text
 *     (thread-local).  Nothing here ever falls back to a CPU implementation: without a usable
 *     gfx950 device every compute entry point returns SFGPU_ERR_HIP.
 */
#ifndef SFGPU_H
#define SFGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFGPU_VERSION 100 /* 0.1.0 */
#define SFGPU_API __attribute__((visibility("default")))

enum {
    SFGPU_OK = 0,
    SFGPU_ERR_INVALID = 1,   /* bad argument */
    SFGPU_ERR_HIP = 2,       /* HIP runtime / device failure */
    SFGPU_ERR_NO_ACTIVE = 3, /* "no transcripts expressed" -- optimize() returns false,
                                src/CollapsedEMOptimizer.cpp:794-798 */
    SFGPU_ERR_ALPHA_SUM = 4, /* "total alpha weight was too small" -- :877-881 */
    SFGPU_ERR_RANGE = 5,     /* a size exceeds what the device layout holds (see each call) */
    SFGPU_ERR_STATE = 6,     /* call order violated (e.g. export before finish) */
    SFGPU_ERR_UNSUPPORTED = 7 /* reserved: an option of the reference this build does not implement (none at present) */
};

typedef void* sfgpu_stream;          /* hipStream_t */
typedef struct sfgpu_eq sfgpu_eq;    /* EquivalenceClassBuilder on the device */
typedef struct sfgpu_em sfgpu_em;    /* CollapsedEMOptimizer state on the device */

SFGPU_API int sfgpu_version(void);
/* 1 in builds made with -DSFGPU_VARIANTS (tools/em_variants.sh, tools/eq_variants.sh): they also hold the kernel forms that lost their
 * measurements (the ring / quad / pipelined class build, the graph-replayed EM loops) and read the tuning switches; the product answers 0.
 * Environment switches the PRODUCT library reads (everything else is a variants-only tuning knob):
 *   SFGPU_EM_FUSED=0|1      0: sweep + update kernels per EM iteration; 1: one kernel per iteration wherever it can run
 *   SFGPU_EM_PERSIST=0      never run the EM loop as one persistent launch (also read when a problem's plan is made)
 *   SFGPU_EM_GATHER=0       phase C of the sweep scatters with LDS atomics (rounds 1 - 2) instead of the gather form
 *   SFGPU_EM_EXACT_NORM=1   VBEM: psi(sum alpha) from the summed vector instead of the constant psi(M prior + numMapped)
 *   SFGPU_EM_NO_RENUMBER=1  the EM plan keeps the caller's transcript order whatever the labels look like
 *   SFGPU_EM_COVER_SORT=1 / SFGPU_EM_COVER_CHECK=1   tests: cover lists by sorting / both forms compared
 *   SFGPU_EQ_SUBBATCH=n     reads per sub-batch of the class build; SFGPU_EQ_HOST_CHUNK=n reads per staged chunk of a host batch
 *   SFGPU_BS_LANES=n        concurrent bootstrap replicates (1 .. 8; default: 1 where a replicate's EM loop runs as one persistent launch, else 3)
 *   SFGPU_BS_PERSIST=1      several lanes keep the persistent loop (dev: they disturb each other's launches, profiles/r6_em_notes.md 4)
 *   SFGPU_EM_XBUF=pool      the persistent loop's exchange buffer in ordinary pool memory instead of uncached device memory
 *   SFGPU_EM_COOP=1         the persistent loop is launched with hipLaunchCooperativeKernel (a launch-time check of the grid's residency)
 *   SFGPU_MN_TREE=levels    the bootstrap's multinomial tree with one launch per level (tests: the two-launch form gives the same counts)
 *   SFGPU_POOL_LARGE_LIMIT_GB=g   cached device blocks >= 1 GiB kept per device
 *   SFGPU_TIMING=1          plans described on stderr */
SFGPU_API int sfgpu_has_variants(void);
SFGPU_API const char* sfgpu_last_error(void);
/* Forwarded to sopt.jointLog by the adaptor (level: 0 info, 1 warn, 2 error).  NULL = silent. */
SFGPU_API void sfgpu_set_logger(void (*log)(int level, const char* msg));
/* Scratch device memory is cached inside the library (hipMalloc/hipFree are slow and hipFree
 * synchronises the device); this returns every cached block to the driver. */
SFGPU_API int sfgpu_pool_trim(void);
/* Cached blocks of >= 1 GiB (the Gibbs sampler's chain state: 4 x nnz x chains bytes, 38 GB for 1.6 M classes and 1024
 * chains) are kept only up to this many bytes per device -- memory parked in this cache is invisible to other allocators in
 * the process (e.g. torch's).  Default: a quarter of the device's memory, at most 64 GiB (SFGPU_POOL_LARGE_LIMIT_GB overrides);
 * 0 = never cache them (every call pays the mapping: ~15 ms per GB); negative = back to the default. */
SFGPU_API int sfgpu_pool_set_large_limit(long long bytes);
/* Device name / CU count / HBM bytes of the current device (any pointer may be NULL). */
SFGPU_API int sfgpu_device_info(char* name, int name_len, int* n_cu, uint64_t* hbm_bytes);

/* ---------------------------------------------------------------------------------------------
 * a1. TranscriptGroup hash: XXH64(label bytes, 4*n, seed 0)
 *     replaces: TranscriptGroup::TranscriptGroup(std::vector<uint32_t>)  src/TranscriptGroup.cpp:9-12
 *               XXH64                                                    src/xxhash.c:346-455, 458-484
 * Packed batch: label r = d_ids[d_offsets[r] .. d_offsets[r+1]).  Asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------- */
SFGPU_API int sfgpu_xxh64_labels(const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads,
                       uint64_t* d_hashes, sfgpu_stream stream);

/* ---------------------------------------------------------------------------------------------
 * a2-a5. EquivalenceClassBuilder   include/EquivalenceClassBuilder.hpp:53-119
 * ------------------------------------------------------------------------------------------- */
/* ctor: reserves room for `expected_classes` (the reference reserves 1e6, :57). 0 = default. */
SFGPU_API int sfgpu_eq_create(sfgpu_eq** out, uint64_t expected_classes, sfgpu_stream stream);
SFGPU_API int sfgpu_eq_destroy(sfgpu_eq* eq);
/* start() :62 -- also clears any previous contents so a builder can be reused. */
SFGPU_API int sfgpu_eq_start(sfgpu_eq* eq);
/* addGroup() :90-108, batched: one call carries the hit lists of many reads (the reference's
 * unit is the <=1000-read parser job, src/SailfishQuantify.cpp:73,399-416,608-625).
 * label r = ids[offsets[r] .. offsets[r+1]); the ORDERED list is the key (equality ==
 * vector equality, src/TranscriptGroup.cpp:53-55); empty lists are skipped like the call
 * site's `if (txpIDs.size() > 0)` guard.  Thread-safe (calls are serialised per builder).
 * offsets are uint32: one batch holds < 2^32 ids and < 2^31 reads (else SFGPU_ERR_RANGE); a larger
 * experiment is handed over in several batches (counts are uint64 and accumulate across batches).
 * The EXPORT has the same ceiling: rowptr is uint32, so the finished table must hold < 2^32 label
 * ids in total (sum of class sizes; sfgpu_eq_export_* return SFGPU_ERR_RANGE beyond) -- ~460x the
 * 9.3 M of the 400 M-read / 200 k-transcript configuration.
 * _device reads a device-resident batch and returns after it has been folded in -- or, for a batch of fewer than 8 M
 * reads, after it has been copied (device to device) behind earlier small batches, which are built 16 M reads at a time: a
 * build costs ~0.25 ms of launches and round trips whatever its size (100 M reads in 1 M-read batches: 34 -> 17 ms).
 * _host takes the caller's (pageable or pinned) host arrays: small batches are copied into a pinned
 * accumulation buffer (thread-safe: only the reservation of the range is serialised) and built
 * 2 M reads at a time, large ones are staged directly; offsets may start at a non-zero base (the ids are
 * read from h_ids + h_offsets[0]).  Either way the caller may reuse its buffers on return, and
 * sfgpu_eq_finish() folds in whatever is still accumulated. */
SFGPU_API int sfgpu_eq_add_batch_host(sfgpu_eq* eq, const uint32_t* h_ids, const uint32_t* h_offsets, uint32_t n_reads);
SFGPU_API int sfgpu_eq_add_batch_device(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets, uint32_t n_reads);
/* insertGroup(TranscriptGroup, count) :82-88, batched and with upsert semantics: group g is added
 * with multiplicity d_counts[g] (equal labels accumulate).  Used to merge class tables built on
 * different GPUs; same limits and synchronisation as sfgpu_eq_add_batch_device. */
SFGPU_API int sfgpu_eq_add_weighted_device(sfgpu_eq* eq, const uint32_t* d_ids, const uint32_t* d_offsets,
                                 const uint64_t* d_counts, uint32_t n_groups);
/* ---- the class-table exchange of a multi-GPU run (the reference has one table in one process) ----------
 * One process / thread per GPU builds the table of ITS reads; afterwards every rank must hold the table a single
 * builder would have produced from all reads.  The library does the device work on class tables in CSR form (the
 * sfgpu_eq_export_device arrays); the HOST moves the byte blocks (RCCL, MPI, ...):
 *   1. owner(class) = a function of its XXH64 mod N.  sfgpu_eqvec_owner_sizes -> classes / ids per owner;
 *      sfgpu_eqvec_pack_by_owner -> N blocks, block d = [counts u64[c_d] | lens u32[c_d] | ids u32[l_d] | pad to 8 B]
 *      at d_blocks + h_block_off[d] (SFGPU_BLOCK_BYTES(c_d, l_d) bytes; classes keep their canonical order).
 *   2. all-to-all: block d goes to rank d.  The owner folds what it received with sfgpu_eq_add_block_device (upsert:
 *      equal labels add their counts), finish()es and exports ITS partition of the merged table.
 *   3. sfgpu_eqvec_export_block -> the partition as one block [counts u64[C] | hashes u64[C] | lens u32[C] | ids u32[L]]
 *      (SFGPU_GATHER_BYTES(C, L) bytes); all-gather of the blocks: the partitions are DISJOINT.
 *   4. sfgpu_eqvec_merge_disjoint -> the union in the canonical order (first id, XXH64, length, label), as CSR: a sort
 *      of (first id, hash) keys and a gather, nothing is hashed again.  *same_key_twice = 1 (and no output) if two
 *      different labels share first id and XXH64: fold the blocks through a builder instead (never seen).
 * Integer work throughout: the result equals the single-builder table class for class, in order.  n_owners <= 256,
 * n_parts <= 64; the h_* arrays are host arrays; calls are synchronous on `stream`. */
#define SFGPU_BLOCK_BYTES(c, l) ((12ull * (uint64_t)(c) + 4ull * (uint64_t)(l) + 7ull) & ~7ull)
#define SFGPU_GATHER_BYTES(c, l) (20ull * (uint64_t)(c) + 4ull * (uint64_t)(l))
SFGPU_API int sfgpu_eqvec_owner_sizes(const uint32_t* d_rowptr, const uint64_t* d_hashes, uint64_t n_classes, uint32_t n_owners,
                                      uint64_t* h_classes, uint64_t* h_ids, sfgpu_stream stream);
SFGPU_API int sfgpu_eqvec_pack_by_owner(const uint32_t* d_rowptr, const uint32_t* d_ids, const uint64_t* d_counts, const uint64_t* d_hashes,
                                        uint64_t n_classes, uint32_t n_owners, const uint64_t* h_classes, const uint64_t* h_ids,
                                        void* d_blocks, uint64_t* h_block_off /* [n_owners + 1] */, sfgpu_stream stream);
SFGPU_API int sfgpu_eq_add_block_device(sfgpu_eq* eq, const void* d_block, uint64_t n_classes, uint64_t n_ids, sfgpu_stream stream);
SFGPU_API int sfgpu_eqvec_export_block(const uint32_t* d_rowptr, const uint32_t* d_ids, const uint64_t* d_counts, const uint64_t* d_hashes,
                                       uint64_t n_classes, uint64_t n_ids, void* d_block, sfgpu_stream stream);
SFGPU_API int sfgpu_eqvec_merge_disjoint(const void* const* d_blocks, const uint64_t* n_classes, const uint64_t* n_ids, uint32_t n_parts,
                                         uint32_t* d_rowptr, uint32_t* d_ids, uint64_t* d_counts, uint64_t* d_hashes,
                                         int* same_key_twice, sfgpu_stream stream);

/* Builder counters since the last start(): device time of the insert kernel (HIP events on the
 * builder's stream), launches, table growths, deferred-and-replayed reads, current table slots. */
typedef struct {
    double insert_ms;
    uint64_t insert_launches;
    uint64_t table_grows;
    uint64_t deferred_reads;
    uint64_t table_slots;
    uint64_t hot_reads;       /* reads whose class was already hot and that the route pass counted itself */
    uint64_t spilled_reads;   /* reads of the partitioned passes that went to the generic kernel (bin overflow, long labels) */
    uint64_t pipeline_drains; /* times the pipelined partition passes had to run dry (table growth, deferred reads, reallocation) */
} sfgpu_eq_stats;
SFGPU_API int sfgpu_eq_get_stats(sfgpu_eq* eq, sfgpu_eq_stats* out);
/* finish() :64-80: snapshot into the canonical class order (first id, XXH64, length, label --
 * the reference's order is hash-table order and run dependent).  Reports what the reference
 * logs: #classes and sum(count); nnz = sum of label lengths. Synchronous. */
SFGPU_API int sfgpu_eq_finish(sfgpu_eq* eq, uint64_t* n_classes, uint64_t* nnz, uint64_t* total_reads);
/* eqVec() :110-112 as CSR: rowptr[C+1], ids[nnz], counts[C] (uint64 like TGValue::count),
 * hashes[C] (TranscriptGroup::hash; may be NULL).  nnz must be < 2^32 (SFGPU_ERR_RANGE).
 * _device is asynchronous on the builder's stream; _host is synchronous. */
SFGPU_API int sfgpu_eq_export_device(sfgpu_eq* eq, uint32_t* d_rowptr, uint32_t* d_ids, uint64_t* d_counts, uint64_t* d_hashes);
SFGPU_API int sfgpu_eq_export_host(sfgpu_eq* eq, uint32_t* h_rowptr, uint32_t* h_ids, uint64_t* h_counts, uint64_t* h_hashes);

/* ---------------------------------------------------------------------------------------------
 * a14. fragment-length distribution -> effective lengths   src/SailfishQuantify.cpp
 * The 1000-entry correction tables are serial prefix sums and are built on the host in the
 * reference's evaluation order; the O(M) transform runs on the device.
 * ------------------------------------------------------------------------------------------- */
/* getNormalFragLengthDist :648-673 (mean/sd are integers in SailfishOpts.hpp:34-35) */
SFGPU_API int sfgpu_cf_gaussian(uint32_t max_frag_len, uint64_t mean, uint64_t sd, double* h_cf);
/* correctionFactorsFromCounts :769-807 */
SFGPU_API int sfgpu_cf_counts(const uint32_t* h_fl_counts, uint32_t max_frag_len, double* h_cf);
/* computeSmoothedEffectiveLengths :809-838 ; setEffectiveLengthsDirect :706-715 when h_cf == NULL.
 * Asynchronous on `stream`: the table h_cf is copied to the device by a stream-ordered copy, so a PINNED h_cf must stay valid and
 * unchanged until the stream has passed this call (pageable memory is staged by the runtime before the call returns). */
SFGPU_API int sfgpu_efflen_smoothed(const uint32_t* d_ref_len, uint64_t M, const double* h_cf, uint32_t max_frag_len,
                          double* d_eff_len, sfgpu_stream stream);
/* --unsmoothedFLD: computeEmpiricalEffectiveLengths :717-767 over EmpiricalDistribution
 * (src/EmpiricalDistribution.cpp:29-118; float pdf table, cut where the cumulative mass passes 1 - 1e-6,
 * two-ended median walk).  h_fl_counts[i] = observed fragments of length i, i in [0, max_frag_len).
 * eff = RefLength when RefLength <= median, else sum_l pdf(l) * (RefLength - l + 1).  Synchronous. */
SFGPU_API int sfgpu_efflen_empirical(const uint32_t* h_fl_counts, uint32_t max_frag_len, const uint32_t* d_ref_len, uint64_t M,
                           double* d_eff_len, sfgpu_stream stream);

/* ---------------------------------------------------------------------------------------------
 * a6-a12. CollapsedEMOptimizer   src/CollapsedEMOptimizer.cpp:711-893
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t M;               /* transcripts() .size(); <= 2^31 */
    const double* d_len;      /* per transcript: RefLength if noEffectiveLengthCorrection else
                                 EffectiveLength (:736-737); clamped to >= 1 internally (:738) */
    uint64_t C;               /* eqVec().size() */
    const uint32_t* d_rowptr; /* C+1 */
    const uint32_t* d_ids;    /* rowptr[C] */
    const uint64_t* d_counts; /* C; every count must be < 2^31 (SFGPU_ERR_RANGE) */
    uint64_t num_mapped;      /* ReadExperiment::numMappedFragments() (:792) */
} sfgpu_problem;

typedef struct {
    int use_vbem;             /* sopt.useVBOpt (:784) */
    double tol;               /* relDiffTolerance: 0.01 at the call site, src/SailfishQuantify.cpp:1343 */
    uint32_t min_iter;        /* 50 in optimize() (:716); 0 in doBootstrap (:486) */
    uint32_t max_iter;        /* 10000 at the call site */
    int check_mode;           /* 0: gate on alphasPrime > 1e-2 (:852); 1: on alphas > 1e-2 (:499) */
    uint32_t iters_per_launch;/* iterations enqueued between host polls of the device-side stop
                                 latch (the stop iteration is exact regardless). 0 = default (32) */
} sfgpu_em_opts;

typedef struct {
    uint32_t iters;           /* itNum when the loop stopped */
    uint32_t converged;
    double max_rel_diff;      /* as logged at :871-872 */
    double alpha_sum;         /* after truncateCountVector (:875) */
    uint64_t n_active;        /* activeTranscriptIDs.size() (:774-782) */
    double loop_ms;           /* device time of the iteration loop (HIP events on `stream`) */
    uint32_t fused;           /* 1: the loop ran as one kernel per iteration (update folded into the sweep), 0: sweep + update */
    uint32_t persistent;      /* 1: the whole loop ran as ONE launch (csrc/em_persist.h); implies fused */
} sfgpu_em_stats;

SFGPU_API int sfgpu_em_create(sfgpu_em** out, const sfgpu_problem* prob, sfgpu_stream stream);
SFGPU_API int sfgpu_em_destroy(sfgpu_em* em);
/* optimize() :711-893 without the bias branch: writes estCount (alpha after truncation) to
 * d_alpha_out[M] and, if non-NULL, mass = alpha/alphaSum to d_mass_out[M].  Synchronous.
 * Returns SFGPU_ERR_NO_ACTIVE / SFGPU_ERR_ALPHA_SUM where the reference returns false. */
SFGPU_API int sfgpu_em_optimize(sfgpu_em* em, const sfgpu_em_opts* opts, double* d_alpha_out, double* d_mass_out,
                      sfgpu_em_stats* stats);

/* The same loop in pieces, for callers that own the iteration (multi-GPU: classes sharded over
 * ranks, alphaOut all-reduced between sweep and update).  All asynchronous on the stream.
 *   begin : zero alphaOut, then alphaOut[t] = 1 for every transcript in a local class (:774-782)
 *           -> caller may SUM-all-reduce alphaOut across ranks
 *   init  : n_active = #(alphaOut > 0); alpha = active ? numMapped/n_active : 0 (:800-803); it = 0
 *   sweep : alphaOut += E-step contributions of the local classes (EMUpdate_ :224-281 /
 *           VBEMUpdate_ :322-367); no-op once the stop latch is set
 *           -> caller may SUM-all-reduce alphaOut across ranks
 *   update: [VBEM: alphaOut += prior] convergence test + alpha <- alphaOut, alphaOut <- 0
 *           (:849-861), ++it, evaluates the loop condition of :820 into the stop latch
 *   poll  : synchronous; reads the latch and counters
 *   finish: truncate (:875), alphaSum, write outputs; synchronous */
SFGPU_API int sfgpu_em_begin(sfgpu_em* em, const sfgpu_em_opts* opts);
SFGPU_API int sfgpu_em_init(sfgpu_em* em);
SFGPU_API int sfgpu_em_sweep(sfgpu_em* em);
SFGPU_API int sfgpu_em_update(sfgpu_em* em);
SFGPU_API int sfgpu_em_poll(sfgpu_em* em, int* done, sfgpu_em_stats* stats);
SFGPU_API int sfgpu_em_finish(sfgpu_em* em, double* d_alpha_out, double* d_mass_out, sfgpu_em_stats* stats);
/* device pointer of alphaOut (M doubles) for the caller's collective */
SFGPU_API double* sfgpu_em_alpha_out(sfgpu_em* em);
/* The piecewise loop with doBiasCorrect (src/CollapsedEMOptimizer.cpp:814-840): run to a recompute iteration by
 * lowering the stop bounds (set_bounds, between polls), hand the current abundances (sfgpu_em_alpha) and lengths
 * (sfgpu_em_lengths: effLens as the loop holds them, clamped at 1) to sfgpu_bias_update, then give the new lengths
 * back with rebase -- updateEqClassWeights (:527-555): the lengths are replaced (clamped at 1) and x is rebuilt from
 * the current alpha -- raise the bounds again and continue.  The next begin() restores the problem's own lengths. */
SFGPU_API double* sfgpu_em_alpha(sfgpu_em* em);
SFGPU_API double* sfgpu_em_lengths(sfgpu_em* em);
SFGPU_API int sfgpu_em_set_bounds(sfgpu_em* em, uint32_t min_iter, uint32_t max_iter);
/* Process-wide: may optimize() / the bootstrap run the EM loop as ONE persistent launch (csrc/em_persist.h; the loop of
 * src/CollapsedEMOptimizer.cpp:818-861)?  Default 1.  The launch needs every one of its blocks resident, i.e. the device to itself while
 * it starts: processes or ranks that share a device call this with 0 (one kernel per iteration then; the same results).  Takes effect
 * for runs that begin afterwards; handles planned while it was 0 have no tables for the loop and keep one kernel per iteration. */
SFGPU_API int sfgpu_em_allow_persistent(int on);
SFGPU_API int sfgpu_em_rebase(sfgpu_em* em, const double* d_len);
/* The sharded loop as ONE call (classes partitioned over the GPUs, alpha replicated, one SUM all-reduce of
 * alphaOut per iteration): `em` holds THIS rank's slice of the classes; `allreduce` must leave the element-wise sum over
 * all ranks in d_buf on every rank (in place; it may enqueue on `stream`, the stream the loop runs on, or synchronise --
 * e.g. ncclAllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, comm, stream)).  Runs begin -> all-reduce (union of the
 * active sets) -> init -> { sweep, all-reduce, update } with the stop latch polled every `poll_every` iterations ->
 * finish: every rank stops at the same iteration with the same alpha (the reference's stop iteration for the union of
 * the classes).  Same outputs and return codes as sfgpu_em_optimize.  Replaces the reference's
 * CollapsedEMOptimizer::optimize call (src/SailfishQuantify.cpp:1343) in a multi-GPU host. */
typedef int (*sfgpu_allreduce_fn)(double* d_buf, uint64_t n, void* user, sfgpu_stream stream);
SFGPU_API int sfgpu_em_optimize_sharded(sfgpu_em* em, const sfgpu_em_opts* opts, sfgpu_allreduce_fn allreduce, void* user,
                                        uint32_t poll_every, double* d_alpha_out, double* d_mass_out, sfgpu_em_stats* stats);
/* the stream the handle's kernels run on (what to pass to a collective that must be ordered with them) */
/* The sharded loop with ONE sweep kernel per iteration (the update folded into the head of the next sweep, as in optimize()): sweep +
 * fold + all-reduce per iteration instead of sweep + fold + all-reduce + update.  Every rank must run the same form: ask each rank
 * (sfgpu_em_sharded_fused_ok: 1 if its plan allows it), agree on the minimum, tell each rank (sfgpu_em_set_sharded_fused). */
SFGPU_API int sfgpu_em_sharded_fused_ok(sfgpu_em* em);
SFGPU_API int sfgpu_em_set_sharded_fused(sfgpu_em* em, int on);
SFGPU_API sfgpu_stream sfgpu_em_stream(sfgpu_em* em);

/* ---------------------------------------------------------------------------------------------
 * (e) multi-GPU transport: RCCL over xGMI, bound at run time (csrc/comm.hip).  No counterpart in the reference (a
 * shared-memory program: the atomic adds of src/CollapsedEMOptimizer.cpp:224-281 are what the all-reduce replaces).
 * libsfgpu.so does not link librccl: it is dlopen'ed on first use (the copy the process already holds, else /opt/rocm/lib).
 * A communicator is made the NCCL way: one rank calls sfgpu_comm_unique_id and hands the SFGPU_COMM_ID_BYTES bytes to the
 * others by any channel (MPI_Bcast, a torch.distributed broadcast, a file); every rank then calls sfgpu_comm_create with
 * its own device current.  sfgpu_comm_allreduce_fn() is the callback for sfgpu_em_optimize_sharded (user = the
 * communicator): ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm, stream), enqueued on the loop's stream, so that
 * no host code runs between two EM iterations. */
#define SFGPU_COMM_ID_BYTES 128
typedef struct sfgpu_comm sfgpu_comm;
SFGPU_API int sfgpu_comm_available(void);                                   /* 1 if librccl.so could be loaded */
SFGPU_API int sfgpu_comm_unique_id(void* id_out /* SFGPU_COMM_ID_BYTES */);
SFGPU_API int sfgpu_comm_create(sfgpu_comm** out, const void* id /* SFGPU_COMM_ID_BYTES */, int world, int rank);
SFGPU_API int sfgpu_comm_count(sfgpu_comm* c, int* ranks);                     /* ncclCommCount: the ranks the communicator really spans */
SFGPU_API int sfgpu_comm_destroy(sfgpu_comm* c);
SFGPU_API int sfgpu_comm_allreduce_sum_f64(sfgpu_comm* c, double* d_buf, uint64_t n, sfgpu_stream stream);   /* in place, on `stream` */
SFGPU_API sfgpu_allreduce_fn sfgpu_comm_allreduce_fn(void);
/* average duration (us) of one such all-reduce, `reps` back to back on `stream` (HIP events): what a host's choice between
 * the replicated and the sharded EM rests on */
SFGPU_API int sfgpu_comm_time_allreduce(sfgpu_comm* c, double* d_buf, uint64_t n, uint32_t reps, sfgpu_stream stream, double* avg_us);
/* Launch the E-step sweep kernel `n` times back to back (state untouched afterwards) and
 * return its average duration from HIP events on the stream: the live roofline measurement. */
SFGPU_API int sfgpu_em_time_sweep(sfgpu_em* em, const sfgpu_em_opts* opts, uint32_t n, double* avg_ms);

/* ---------------------------------------------------------------------------------------------
 * a15/a17. Bootstrap   src/CollapsedEMOptimizer.cpp:438-525 (doBootstrap), :557-709 (gatherBootstraps),
 *                      include/MultinomialSampler.hpp:13-64
 * Each draw b: class counts ~ Multinomial(N = sum(count) [uint32, as in the reference], p = count/N)
 * (exact; tree of conditional binomials, Philox4x32-10 streams keyed by (seed, b)), alpha
 * re-initialised uniformly over the active transcripts, EM/VBEM to convergence with doBootstrap's
 * loop (no 50-iteration floor, gate on alphas > 1e-2), truncation.  The reference seeds from
 * std::random_device, so only the DISTRIBUTION of the outputs is comparable.
 *   d_out   : n_bootstraps x M doubles on the device, or NULL
 *   cb      : writeBootstrap (std::function<bool(const std::vector<double>&)>): called once per
 *             draw with a host copy of alpha; return 0 to abort.  May be NULL.
 *   h_iters : per-draw iteration counts (host), may be NULL
 * opts->use_vbem / tol / max_iter are honoured; min_iter and check_mode are forced to doBootstrap's.
 * Where a replicate's EM loop runs as one persistent launch the draws run one after the other; elsewhere up to three
 * run concurrently (the handle plus internal clones of it, each on its own stream and host thread);
 * SFGPU_BS_LANES=1..8 overrides; draw b is the same whichever lane computes it and
 * `cb` is still called one draw at a time, in draw order.
 * Synchronous.  The handle's counts are restored afterwards.
 * ------------------------------------------------------------------------------------------- */
typedef int (*sfgpu_sample_cb)(const double* h_alpha, uint64_t M, void* user);
SFGPU_API int sfgpu_bootstrap(sfgpu_em* em, const sfgpu_em_opts* opts, uint32_t n_bootstraps, uint64_t seed,
                    double* d_out, sfgpu_sample_cb cb, void* user, uint32_t* h_iters);
/* One multinomial resample of the class counts (the sampCounts of doBootstrap :468) into
 * d_counts_out[C] (uint32), for draw index `draw` of `seed`.  Synchronous. */
SFGPU_API int sfgpu_bootstrap_counts(sfgpu_em* em, uint64_t seed, uint64_t draw, uint32_t* d_counts_out);

/* ---------------------------------------------------------------------------------------------
 * a16. CollapsedGibbsSampler::sample<ReadExperiment>   src/CollapsedGibbsSampler.cpp:198-291
 * (initCountMap_ :35-94, sampleRound_ :96-186).  Requires Transcript::mass from a prior optimize().
 * Runs `n_chains` independent chains (0 = default: min(n_samples, 1024) rounded up to 64); every
 * chain is initialised like initCountMap_ and then yields one sample per sampleRound_, so sample s
 * comes from chain s % n_chains after s / n_chains + 1 rounds (the reference runs one chain per TBB
 * chunk of the sample range and one round per sample).  The reference seeds from std::random_device:
 * parity is distributional.  Unlike the reference this call does NOT overwrite Transcript::mass
 * (:219-221 mutate it in place); the same transformed values are used internally.
 *   d_mass : M doubles, mass = alpha / alphaSum as written by optimize()
 *   d_out  : n_samples x M int32 on the device, or NULL
 *   cb     : writeSample (std::function<bool(const std::vector<int>&)>), host copy per sample; may be NULL
 * Device memory: 4 * nnz * n_chains + 4 * M * n_chains bytes of chain state.  Synchronous.
 * ------------------------------------------------------------------------------------------- */
typedef int (*sfgpu_gibbs_cb)(const int32_t* h_counts, uint64_t M, void* user);
SFGPU_API int sfgpu_gibbs_sample(const sfgpu_problem* prob, const double* d_mass, uint32_t n_samples, uint32_t n_chains,
                       uint64_t seed, int32_t* d_out, sfgpu_gibbs_cb cb, void* user, sfgpu_stream stream);

/* ---------------------------------------------------------------------------------------------
 * (next) A quasi-mapping front end: reads in, sfgpu_hit records out -- what the reference gets from RapMap's
 * SACollector inside processReadsQuasi (src/SailfishQuantify.cpp:141-142, 192-213 paired end, :487-488, 526-528 single
 * end), so that the hit lists the path consumes have a producer on the device.  RapMap (COMBINE-lab/RapMap @ sf-v0.10.1,
 * scripts/fetchRapMap.sh:20) is not in the reference tree; this is NOT its algorithm and parity with it is unpinned.
 * The contract (csrc/mapper.hip; oracle/mapper_oracle.py restates it): exact k-mer seeds at read offsets 0 and len - k,
 * forward strand then reverse complement; the first occurrence seen for a (transcript, strand) fixes the position;
 * hits sorted by (transcript, strand); mates on one transcript with opposite strands pair up (PAIRED_END_PAIRED,
 * fragment length = max end - min start), otherwise both mates' hits are kept as orphans (left run, right run).
 *   sfgpu_index_build : d_seq / d_seq_off / d_ref_len as for sfgpu_bias_create (transcript t = d_seq[d_seq_off[t] ..
 *                       + d_ref_len[t])); 8 <= k <= 31; max_occ = occurrences kept per lookup (0: 1000).
 *   sfgpu_map_reads   : reads of one batch, read r = d_seq1[d_off1[r] .. d_off1[r + 1]) (bytes; any case; other letters
 *                       than ACGT never match); d_seq2 / d_off2 = the mates or NULL.  Writes d_hit_offsets[n_reads + 1]
 *                       and, if they fit hit_capacity, the records (else SFGPU_ERR_RANGE with *n_hits set: size and
 *                       call again).  The output feeds sfgpu_filter_hits directly.  Synchronous. */
typedef struct sfgpu_index sfgpu_index;
struct sfgpu_hit;
SFGPU_API int sfgpu_index_build(sfgpu_index** out, const char* d_seq, const uint64_t* d_seq_off, const uint32_t* d_ref_len, uint64_t M,
                                uint32_t k, uint32_t max_occ, sfgpu_stream stream);
SFGPU_API int sfgpu_index_destroy(sfgpu_index* idx);
/* Seeds per strand (default 2: offsets 0 and len - k, every (transcript, strand) either seed hits is kept).  With S > 2 the
 * seeds sit at offsets floor(j (len - k) / (S - 1)), j = 0 .. S-1, and a mate keeps only the (transcript, strand) pairs that the
 * most seeds hit: more sensitive on reads with errors (one clean k-mer is enough) without keeping what a single repeat k-mer
 * drags in.  2 <= S <= 8. */
SFGPU_API int sfgpu_index_set_seeds(sfgpu_index* x, uint32_t seeds_per_strand);
/* The mapping mode.  seed_len != 0 (the DEFAULT after sfgpu_index_build: min(19, k)): SCAN mode, modelled on RapMap's maximal
 * mappable prefixes -- the sorted k-mer table is used as a suffix array of depth k, a seed of seed_len <= k bases is a prefix
 * range of it.  A mate is walked once per strand (forward first; the reverse complement is skipped when a forward match
 * covered the whole read): window at i -> prefix range; no occurrence, more than max_occ, or a non-ACGT base -> i += 1;
 * otherwise every occurrence is extended base by base on the transcripts' text (the index keeps a copy), L = the longest
 * extension, the occurrences that reach L form a group, i += L - seed_len + 1 (at most 8 groups per mate).  A (transcript,
 * strand) is positioned by the first group that holds it and gets a vote per group; with several candidates only those with
 * the most votes are kept.  A read with substitutions maps as long as seed_len clean bases remain somewhere that an indexed
 * k-mer starts in: seeds are prefix ranges of WHOLE k-mers, so a seed that begins within the last k - seed_len bases of a
 * transcript, or within k - 1 bases upstream of a non-ACGT base, is not in the index.  Mates of >= 2^24 bases are left unmapped.
 * seed_len == 0: the END-SEED contract above (exact k-mers at offsets 0 and len - k, or sfgpu_index_set_seeds' S seeds) -- the
 * baseline of rounds 1-2.  8 <= seed_len <= k.  Parity with RapMap is unpinned in both modes. */
SFGPU_API int sfgpu_index_set_scan(sfgpu_index* x, uint32_t seed_len);
SFGPU_API int sfgpu_index_info(const sfgpu_index* idx, uint32_t* k, uint64_t* n_positions, uint64_t* n_kmers);
SFGPU_API int sfgpu_map_reads(const sfgpu_index* idx, const char* d_seq1, const uint64_t* d_off1, const char* d_seq2, const uint64_t* d_off2,
                              uint32_t n_reads, struct sfgpu_hit* d_hits, uint64_t hit_capacity, uint32_t* d_hit_offsets, uint64_t* n_hits,
                              sfgpu_stream stream);

/* ---------------------------------------------------------------------------------------------
 * (next) Per-read hit filtering: the loop bodies of processReadsQuasi
 * (src/SailfishQuantify.cpp:215-417 paired end, :530-626 single end) between "the mapper returned
 * jointHits for a read" and eqBuilder.addGroup -- maxReadOccs cut (:217, :532), orphan policy (:226),
 * orphan merge by transcript id (:231-246), library-type compatibility (sailfish::utils::compatibleHit /
 * hitType, src/SailfishUtils.cpp:157-289; pinned by the reference's tests/LibraryTypeTests.cpp), the
 * "compatible hits if any, else all hits unless enforceLibCompat" rule (:324-341, :355-368, :395-416) and
 * the fragment-length sampling of unique proper pairs (:419-434).  Bias / GC sampling: sfgpu_sample_bias below.
 * One record per hit, reads in CSR form; the output is the packed hit lists sfgpu_eq_add_batch_device
 * takes (reads that end up unmapped get an empty list), so labels never visit the host.
 * Enum values: mate_status 0 SINGLE_END, 1 PAIRED_END_LEFT, 2 PAIRED_END_RIGHT, 3 PAIRED_END_PAIRED (the
 * adaptor maps rapmap::utils::MateStatus); sfgpu_libfmt fields as include/LibraryFormat.hpp:7-9
 * (type 0 SE / 1 PE; orientation 0 SAME, 1 AWAY, 2 TOWARD, 3 NONE; strandedness 0 SA, 1 AS, 2 S, 3 A, 4 U).
 * ------------------------------------------------------------------------------------------- */
typedef struct sfgpu_hit {
    uint32_t tid;          /* QuasiAlignment::transcriptID() */
    int32_t  pos;          /* h.pos */
    int32_t  mate_pos;     /* h.matePos */
    uint32_t frag_len;     /* h.fragLen */
    uint16_t read_len;     /* h.readLen */
    uint16_t mate_len;     /* h.mateLen */
    uint8_t  fwd;          /* h.fwd */
    uint8_t  mate_fwd;     /* h.mateIsFwd */
    uint8_t  mate_status;  /* h.mateStatus */
    uint8_t  pad_;
} sfgpu_hit;               /* 24 bytes */
typedef struct sfgpu_libfmt { uint8_t type, orientation, strandedness, pad_; } sfgpu_libfmt;
typedef struct sfgpu_filter_opts {
    uint32_t max_read_occs;     /* sfOpts.maxReadOccs */
    uint32_t max_frag_len;      /* sfOpts.maxFragLen: size of the fragment-length histogram */
    int32_t  paired_library;    /* 1: the paired-end loop (:215-417), 0: the single-end loop (:530-626) */
    int32_t  discard_orphans;   /* !sfOpts.allowOrphans (:139) */
    int32_t  ignore_compat;     /* sfOpts.ignoreLibCompat (:156) */
    int32_t  enforce_compat;    /* sfOpts.enforceLibCompat (:160) */
    int32_t  can_dovetail;      /* sfOpts.allowDovetail (:165) */
    sfgpu_libfmt expected;      /* rl.format() (:163) */
} sfgpu_filter_opts;
typedef struct sfgpu_filter_stats {   /* all ACCUMULATED by the call */
    uint64_t n_observed;        /* numObservedFragments */
    uint64_t n_mapped;          /* validHits: reads handed to addGroup */
    uint64_t total_hits;        /* totalHits (after the maxReadOccs / orphan cuts) */
    uint64_t upper_bound_hits;  /* upperBoundHits: reads with at least one hit before the cuts */
    uint64_t n_fwd, n_rc;       /* readExp.addNumFwd / addNumRC */
    uint64_t fl_sampled;        /* fragment lengths added to the histogram by this call */
} sfgpu_filter_stats;
/* d_hits[d_hit_offsets[r] .. d_hit_offsets[r+1]) are read r's hits in the mapper's order (for orphans: left
 * mate's hits first, each run ascending in tid, as mergeLeftRightHits leaves them).
 * d_ids_out needs room for d_hit_offsets[n_reads] ids; d_offsets_out for n_reads + 1.
 * d_fl_counts (max_frag_len uint32, may be NULL) and *remaining_fl_ops (may be NULL) carry the
 * fragment-length histogram and its sample budget across calls: the first *remaining_fl_ops qualifying
 * reads in read order are counted, exactly what one mapping thread does.  Synchronous. */
SFGPU_API int sfgpu_filter_hits(const sfgpu_hit* d_hits, const uint32_t* d_hit_offsets, uint32_t n_reads,
                      const sfgpu_filter_opts* opts, uint32_t* d_ids_out, uint32_t* d_offsets_out,
                      uint32_t* d_fl_counts, int64_t* remaining_fl_ops, sfgpu_filter_stats* stats, sfgpu_stream stream);

/* The samples the same loop collects when bias correction is on (one more pass over the hit records, for the
 * callers that need it): for every read, the 6-mer context of the FIRST hit that yields one (needBiasSample,
 * src/SailfishQuantify.cpp:270-287 / :559-581; ReadKmerDist<6>::update, include/ReadKmerDist.hpp:35-73), while
 * the budget sfOpts.numBiasSamples lasts -- "the first N successful reads in read order", what one mapping thread
 * does; and, in a paired library, the fragment-GC percentage of every properly paired hit (:375-389; no budget).
 * The reads and hits considered are those that survive the maxReadOccs / orphan cuts of sfgpu_filter_hits (same
 * opts).  Counters are ACCUMULATED into d_read_bias (4096 uint32) / d_observed_gc (101 uint32); either may be NULL.
 * d_gc_prefix: the per-transcript inclusive G/C counts laid out like d_seq (sfgpu_gc_prefix; 4 bytes per base),
 * required with d_observed_gc -- Transcript::GCCount_ for gcSampFactor 1 (include/Transcript.hpp:183-196).
 * Synchronous. */
typedef struct sfgpu_bias_sampler {
    const char* d_seq;                /* RapMapSAIndex::seq */
    const uint64_t* d_seq_off;        /* [M] txpOffsets */
    const uint32_t* d_ref_len;        /* [M] */
    uint32_t* d_read_bias;            /* [4096] ReadKmerDist<6>::counts, or NULL (biasCorrect off) */
    int64_t* remaining_bias_samples;  /* sfOpts.numBiasSamples (host), decremented */
    uint32_t* d_observed_gc;          /* [101] ReadExperiment::observedGC, or NULL (gcBiasCorrect off) */
    const uint32_t* d_gc_prefix;      /* see above */
    uint64_t n_bias_sampled, n_gc_sampled;   /* ACCUMULATED by the call */
    uint32_t gc_size_samp;            /* SailfishOpts::gcSampFactor: 0 or 1 = exact counts, > 1 = the reference's interpolation (bin clamped to [0,100]) */
    uint32_t pad_;
} sfgpu_bias_sampler;
SFGPU_API int sfgpu_gc_prefix(const char* d_seq, const uint64_t* d_seq_off, const uint32_t* d_ref_len, uint64_t M,
                      uint32_t* d_gc_prefix, sfgpu_stream stream);
SFGPU_API int sfgpu_sample_bias(const sfgpu_hit* d_hits, const uint32_t* d_hit_offsets, uint32_t n_reads,
                      const sfgpu_filter_opts* opts, sfgpu_bias_sampler* sampler, sfgpu_stream stream);

/* ---------------------------------------------------------------------------------------------
 * (next) Bias-aware effective lengths: sailfish::utils::updateEffectiveLengths
 * (src/SailfishUtils.cpp:611-926) -- the sequence-specific (--biasCorrect; 6-mer context model,
 * include/ReadKmerDist.hpp, include/UtilityFunctions.hpp:40-148) and fragment-GC (--gcBiasCorrect;
 * Transcript::gcFrac, include/Transcript.hpp:85-95) corrections -- and the optimize() variant that calls it
 * at iterations 50, 500 and 1000 (src/CollapsedEMOptimizer.cpp:814-840) and hands the corrected lengths back
 * (:888).  A handle holds what the function reads from ReadExperiment / SailfishOpts and is constant over the
 * run; sfgpu_bias_update is one call of updateEffectiveLengths.
 *   d_seq / d_seq_off : RapMapSAIndex::seq on the device and txpOffsets (Transcript::Sequence() =
 *                       seq + txpOffsets[i], include/ReadExperiment.hpp:115); bytes A C G T U in either case,
 *                       any other byte counts as 'A' in a k-mer (what nextKmerIndex does; a first k-mer
 *                       with such a byte is undefined behaviour in the reference)
 *   d_txp_eff_len     : Transcript::EffectiveLength as the FLD correction left it (:703, :821)
 *   h_fl_counts       : the vector ReadExperiment::setFragLengthDist received (counts of lengths
 *                       0..max_frag_len-1); the EmpiricalDistribution is rebuilt from it
 *   h_read_bias       : 4096 ReadKmerDist<6>::counts, pseudo-count included (may be NULL when !seq_bias)
 *   h_observed_gc     : 101 ReadExperiment::observedGC counts, pseudo-count included (may be NULL when !gc_bias)
 *   gc_speed_samp     : SailfishOpts::pdfSampFactor (--gcSpeedSamp)
 *   gc_size_samp      : SailfishOpts::gcSampFactor (--gcSizeSamp).  Above 1 the reference keeps the G/C count at every
 *                       gc_size_samp-th base only and interpolates (Transcript::gcCountInterp_, include/Transcript.hpp:
 *                       133-162, lambda on the LEFT sample as written); the same values are used here (slow path:
 *                       per-base table built for the duration of sfgpu_bias_create).  The interpolated difference
 *                       can leave [0, fragment length], where the reference indexes outside its 101 bins: the bin is
 *                       clamped to [0,100]
 * As in the reference, seq_bias and gc_bias together, or num_fwd + num_rc == 0, make every update a copy
 * of its input (status 2 / 1).  GC correction needs fld_low >= 1 (the reference divides by the fragment
 * length) and a 0.995 quantile below 16000 (SFGPU_ERR_RANGE).  Device memory: 808 bytes per transcript in
 * GC mode (the per-transcript GC-bin profile, built once at create), else O(1).
 * A handle carries the state of its last update (the expectation vectors): use one per optimize() that runs at a time.
 * Floating point: sums run in a different order than the reference's serial loops (and the 4096-bin
 * expectation is accumulated with atomics), so lengths agree to ~1e-12 relative, not bit for bit.
 * ------------------------------------------------------------------------------------------- */
typedef struct sfgpu_bias sfgpu_bias;
typedef struct sfgpu_bias_inputs {
    uint64_t M;
    const char* d_seq;
    const uint64_t* d_seq_off;      /* [M] */
    const uint32_t* d_ref_len;      /* [M] */
    const double* d_txp_eff_len;    /* [M] */
    const uint32_t* h_fl_counts;    /* [max_frag_len] */
    uint32_t max_frag_len;
    uint32_t gc_speed_samp;
    const uint32_t* h_read_bias;    /* [4096] */
    const uint32_t* h_observed_gc;  /* [101] */
    int64_t num_fwd, num_rc;        /* ReadExperiment::numFwd() / numRC() */
    int32_t seq_bias, gc_bias;      /* SailfishOpts::biasCorrect / gcBiasCorrect */
    uint32_t gc_size_samp;
    uint32_t pad_;
} sfgpu_bias_inputs;
typedef struct sfgpu_bias_stats {
    int32_t status;                 /* 0 recomputed, 1 no mappings (:625-630), 2 both models on (:633-638) */
    int32_t fld_low, fld_high;      /* the 0.005 / 0.995 quantiles of the FLD (:669-681), GC mode */
    int32_t pad_;
    uint64_t n_corrected, n_uncorrected;   /* numCorrected / numUncorrected (:806-807) of the last update */
} sfgpu_bias_stats;
SFGPU_API int sfgpu_bias_create(sfgpu_bias** out, const sfgpu_bias_inputs* in, sfgpu_stream stream);
SFGPU_API int sfgpu_bias_destroy(sfgpu_bias* b);
/* effLensOut = updateEffectiveLengths(sopt, readExp, effLensIn, alphas); d_eff_out may alias d_eff_in.
 * Asynchronous on `stream` when stats is NULL, else synchronises and fills *stats. */
SFGPU_API int sfgpu_bias_update(sfgpu_bias* b, const double* d_eff_in, const double* d_alpha, double* d_eff_out,
                      sfgpu_bias_stats* stats, sfgpu_stream stream);
/* ReadExperiment::expectedSeqBias() (4096) / expectedGCBias() (101) after the last update (the aux/
 * expected_bias, expected_gc files, src/GZipWriter.cpp:140-165); either pointer may be NULL.  Synchronous. */
SFGPU_API int sfgpu_bias_expected(sfgpu_bias* b, double* h_expected_seq, double* h_expected_gc);
/* CollapsedEMOptimizer::optimize with doBiasCorrect (:717): as sfgpu_em_optimize, plus the recompute hook
 * at iterations 50, 500, 1000 and d_eff_len_out [M] = the lengths to store back into
 * Transcript::EffectiveLength (:888; may be NULL).  n_recomputes (may be NULL) = hooks taken. */
SFGPU_API int sfgpu_em_optimize_bias(sfgpu_em* em, const sfgpu_em_opts* opts, sfgpu_bias* bias, double* d_alpha_out,
                      double* d_mass_out, double* d_eff_len_out, uint32_t* n_recomputes, sfgpu_em_stats* stats);

/* ---------------------------------------------------------------------------------------------
 * a13. quant.sf columns   src/GZipWriter.cpp:216-245
 *   TPM_t = ((estCount_t/numMapped)/len_t) / sum_u((estCount_u/numMapped)/len_u) * 1e6
 * d_len as in sfgpu_problem.  Asynchronous on `stream`.
 * ------------------------------------------------------------------------------------------- */
SFGPU_API int sfgpu_tpm(const double* d_est_count, const double* d_len, uint64_t M, double num_mapped,
              double* d_tpm, sfgpu_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* SFGPU_H */
