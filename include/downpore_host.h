/*
 * downpore_host.h — C ABI of libdownpore_host.so: the whole `downpore overlap` / `downpore map` pipeline above the kernels of
 * libdownpore_hip.so (include/downpore_hip.h).
 *
 * downpore_hip.h is the boundary of the kernels: one call per batched body of overlap.Overlapper / seeds.SeedIndex /
 * mapping.Mapper.  A host that drives those calls one synchronous round at a time gets the kernels, not the throughput: the
 * reference's own command loop (commands/overlap.go:119-195: PrepareQueries -> AddSequences -> FindOverlaps -> finalCheckWorker,
 * round after round) leaves the GPU idle between dependent launches.  What bench.py measures (BENCH_r*.json) is the pipeline in
 * this library: a planner that runs the PrepareQueries chain ahead on speculative lanes, a window cache that selects every edge
 * window's seeds on the device once, executor slots that run rounds concurrently and commit
 * them in order with a speculation check, consensus + PAF numbers on the device, text on formatter threads.  This header is
 * that pipeline's boundary: a Go `commands/overlap.go` that calls dph_overlap_open / _init / _step / _round_paf (cgo,
 * INTEGRATION.md, integration/commands/gpu_overlap.go) prints the reference's PAF at the measured rate.
 *
 * Conventions: plain pointers and sizes; handles are opaque; int results are 0 / a count on success and negative on failure
 * (dph_last_error gives the text); returned text / byte pointers belong to the handle and stay valid until the next call of the
 * same function on it.  One caller thread per handle (the library runs its own threads behind it).  No CPU fallback: the calls
 * fail without a GPU.
 */
#ifndef DOWNPORE_HOST_H
#define DOWNPORE_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPH_API __attribute__((visibility("default")))

/* h: an overlap handle, or NULL for the calling thread's last failed constructor (reads / open / create / map_run) */
DPH_API const char* dph_last_error(void* h);

/* ---- read sets: sequence.NewFastaSequenceSet (sequence/seqio.go:45-300) --------------------------------------------------
 * One-line FASTA / FASTQ records; a line is a sequence iff its first byte is in ['A','T'], kept iff len(line incl. '\n') >=
 * min_len; ids = file order; himem != 0 = the cache the commands use (later passes serve SubSequence views, seqio.go:111-118).
 * from_arrays: read r = bases[off[r] .. off[r+1]) (ASCII), n reads; _q: raw FASTQ quality characters at the same offsets. */
DPH_API void* dph_reads_from_arrays(const char* bases, const int64_t* off, int64_t n, int64_t min_len, int himem);
DPH_API void* dph_reads_from_arrays_q(const char* bases, const char* quals, const int64_t* off, int64_t n, int64_t min_len, int himem);
DPH_API void* dph_reads_from_fasta(const char* path, int64_t min_len, int himem);
DPH_API void dph_reads_free(void* reads);
DPH_API int64_t dph_reads_count(void* reads);
DPH_API int64_t dph_reads_total_bases(void* reads);
/* SetIgnore flags (seqio.go:375) as the last job left them: out[n reads] */
DPH_API void dph_reads_get_ignore(void* reads, uint8_t* out);
DPH_API void dph_reads_reset_ignore(void* reads);

/* ---- `downpore overlap` (commands/overlap.go:96-233) -----------------------------------------------------------------------
 * dph_overlap_open   device context on HIP device `device`; the reads are uploaded and packed (and stay resident for every job
 *                    on this handle); FASTQ qualities travel with them.
 * dph_overlap_init   everything the command does between "Counting all k-mers" and its first round: k-mer position index,
 *                    value table (on the device unless `values` - 4^k doubles, the -seed_values table - is given), executor
 *                    slots, planner, window cache.  params[8] = overlap_size, k, num_seeds, seed_batch_size, chunk_size,
 *                    query_batch_size, himem | query_type << 8 (overlap.QueryEdges 1 = the overlap command, QueryCentre 2,
 *                    QueryAll 4, + WeightEdges 8; overlap/overlap.go:18-21), executor slots (rounds in flight; 8 = what bench.py
 *                    uses).  min_hits = -min_hits.
 * dph_overlap_step   commits the next finished round(s) in round order; returns how many (0 = the command is finished).  The
 *                    PAF lines of exactly those rounds: dph_overlap_round_paf (dph_overlap_step_lines of them); every line so
 *                    far: dph_overlap_all_paf; the reference's stderr lines: dph_overlap_errtext.
 * dph_overlap_reset  ends the job, keeps the handle, the resident reads and the executor contexts: dph_overlap_init starts the
 *                    next job (other k, other parameters).  dph_overlap_create = open + init. */
DPH_API void* dph_overlap_open(void* reads, int device);
DPH_API int dph_overlap_init(void* h, const int64_t* params, double min_hits, const double* values);
DPH_API int dph_overlap_reset(void* h);
DPH_API void* dph_overlap_create(void* reads, int device, const int64_t* params, double min_hits, const double* values);
DPH_API void dph_overlap_destroy(void* h);
DPH_API int dph_overlap_step(void* h);
DPH_API int64_t dph_overlap_step_lines(void* h);
DPH_API const char* dph_overlap_round_paf(void* h, int64_t* n);
DPH_API const char* dph_overlap_all_paf(void* h, int64_t* n);
DPH_API const char* dph_overlap_errtext(void* h, int64_t* n);
/* multi-rank runs: keep == 0 on a rank that does not print the PAF - gathered rounds are committed with their counts, flags and read
 * lists, their text is dropped as it arrives (default: kept on every rank) */
DPH_API void dph_overlap_keep_text(void* h, int keep);
DPH_API int dph_overlap_done(void* h);
DPH_API int64_t dph_overlap_round(void* h);                     /* rounds committed so far */
DPH_API void dph_overlap_set_round_limit(void* h, int64_t n);   /* dph_overlap_step commits no round >= n (-1: no limit) */
DPH_API void dph_overlap_drain(void* h);                        /* drops the rounds in flight (they are executed again) */
DPH_API int dph_overlap_slots(void* h);
/* the job's k-mer value table (4^k doubles, commands/overlap.go:55-93) */
DPH_API const double* dph_overlap_values(void* h, int64_t* n);
/* the device context of the handle (a dp_ctx* of downpore_hip.h) */
DPH_API void* dph_overlap_ctx(void* h);
/* out[3]: seconds spent creating the context, uploading + packing the reads, in the last dph_overlap_init */
DPH_API void dph_overlap_setup_times(void* h, double* out);
/* statistics of the last committed round / summed over the job: out[31] doubles in the order of downpore_amd/overlap.py
 * STAT_FIELDS (host seconds per phase, kernel milliseconds, algorithmic bytes, counts) */
DPH_API void dph_overlap_stats(void* h, double* out);
DPH_API void dph_overlap_stats_total(void* h, double* out);

/* ---- multi-GPU: one handle per GPU (process, or thread of one process) ---------------------------------------------------------
 * scan-shard (reads partitioned, survivors' seed index all-gathered): every rank runs every round on its read
 * range [lo, hi); the survivors are exchanged inside the library (dp_comm: RCCL over xGMI, or peer copies between the handles
 * of one process).  dph_comm_unique_id on rank 0, the 128 bytes to every rank, dph_overlap_comm_init (one communicator) or
 * _comm_init_slots (one per executor slot: ids = n_slots x 128 bytes) BEFORE dph_overlap_init; then dph_overlap_set_shard and
 * dph_overlap_round_sharded (one round) / dph_overlap_rounds_sharded (one round per slot, concurrently) - collective calls.
 * dph_overlap_round_scan / _local / _round_finish: the same round in two halves with the exchange left to the caller.
 * round-parallel (reads + index on every GPU, rounds dealt to the ranks): dph_overlap_set_ranks, then per superstep
 * dph_overlap_wait_owned_many (this rank's finished rounds, serialised) -> the caller all-gathers the blobs ->
 * dph_overlap_commit_gathered on every rank (commits the valid prefix in round order; rejected rounds are executed again).
 * dph_overlap_exec_round / dph_overlap_commit_blobs: the batch-synchronous form of the same. */
DPH_API int dph_comm_unique_id(uint8_t* id128);
DPH_API int dph_overlap_comm_init(void* h, int n_ranks, int rank, const uint8_t* id128);
DPH_API int dph_overlap_comm_init_slots(void* h, int n_ranks, int rank, const uint8_t* ids, int n_slots);
DPH_API int dph_overlap_comm_init_local(void** handles, int n);
DPH_API int dph_overlap_comm_init_local_slots(void** handles, int n, int n_slots);
DPH_API void dph_overlap_set_shard(void* h, int64_t lo, int64_t hi);
DPH_API int dph_overlap_round_sharded(void* h);
DPH_API int dph_overlap_rounds_sharded(void* h);
DPH_API int dph_overlap_round_scan(void* h);
DPH_API void dph_overlap_local(void* h, const uint32_t** read, const uint32_t** n_seeds, const uint64_t** seg_off, const int32_t** segs,
                               uint64_t* n, uint64_t* n_segs);
DPH_API int dph_overlap_round_finish(void* h, const uint32_t* read, const uint32_t* n_seeds, const int32_t* segs, uint64_t n);
DPH_API void dph_overlap_set_ranks(void* h, int rank, int world);
DPH_API const uint8_t* dph_overlap_wait_owned(void* h, uint64_t* n);
DPH_API const uint8_t* dph_overlap_wait_owned_many(void* h, int max_rounds, uint64_t* n);
DPH_API int dph_overlap_commit_gathered(void* h, const uint8_t* blobs, const uint64_t* sizes, int count);
/* the same superstep with the exchange inside the library (dp_allgather_blobs on the communicator of dph_overlap_comm_init):
 * rounds committed; 0 = the superstep's first round was rejected and runs again (or the command is finished: dph_overlap_done) */
DPH_API int dph_overlap_superstep(void* h, int max_rounds);
/* root >= 0: a superstep gathers the rounds' PAF text to rank `root` alone (the rank that prints: commands/overlap.go:225-228 prints
 * in one process) and all-gathers only the rounds' control records (flags, read lists, counts: a few KB per round); every rank of
 * the job must pass the same root, before the first superstep.  -1 (the default): text and control travel together to every rank. */
DPH_API void dph_overlap_text_root(void* h, int root);
DPH_API const uint8_t* dph_overlap_exec_round(void* h, int64_t first_round, uint64_t* n);
DPH_API int dph_overlap_commit_blobs(void* h, const uint8_t* blobs, const uint64_t* sizes, int count);

/* ---- `downpore map` (commands/map.go:33-116, mapping/mapping.go) -------------------------------------------------------------
 * The whole command: reference = first sequence of `ref` (a read set opened with himem = 0), reads top-level.  params[6] =
 * circular, k, query_size, min_length, chunk_size, seed_rate.  DP_MAP_SHARDS / DP_MAP_DEVICES in the environment spread the
 * reference index over several contexts / GPUs (BASELINE config 5).  Returns a handle holding the PAF (read order) and the
 * reference's stderr lines, or NULL.  dph_map_stats: out[13] = chunks, seeds, windows, chains, batches, scan kernel ms, map
 * kernel ms, seconds of set-up / window scans / dp_map_windows / host, algorithmic bytes of the map kernels (index query +
 * prefilter + chaining) and of the window scans (packed bases). */
DPH_API void* dph_map_run(void* ref, void* reads, const int64_t* params, int device);
DPH_API void dph_map_free(void* m);
DPH_API const char* dph_map_paf(void* m, int64_t* n);
DPH_API const char* dph_map_errtext(void* m, int64_t* n);
DPH_API void dph_map_stats(void* m, double* out);

/* ---- test hooks (host logic without a GPU, counters) ------------------------------------------------------------------------ */
DPH_API const char* dph_reads_dump(void* reads, int64_t* n);
DPH_API void dph_values_from_counts(uint64_t* counts, int k, double* out);

/* ---- test hooks (not part of the boundary): decision rules of the host side on bare numbers, held by tests/test_hand_known_answers.py
 * to answers worked by hand from the reference's Go text (the files under tests/golden/hand).
 * dph_hand_is_consistent   mapping.isConsistent (mapping/mapping.go:131-160): left5 = {RC, Query.Len(), QueryInset, Start, End},
 *                          right4 = {RC, QueryOffset, Start, End}
 * dph_hand_remove_dominated  removeDominated (mapping.go:387-428): maps3 = n x {QueryOffset, QueryInset, ids}; kept[] = the survivors'
 *                          indices in the order the function returns them; returns their number
 * dph_hand_trim_indices    step 1 of trimToBestSeed (overlap/combine.go:24-58): match i's MatchA = match_a[off[i] .. off[i + 1]);
 *                          out2 = {bestIndex, backIndex}
 * dph_test_coroutines     the mapper's read tasks are stackful coroutines (host/host_coro.hpp): n_tasks of them on recycled stacks,
 *                          each resumed `yields` times; returns the sum of id x yields over the tasks, -1 if a frame came back damaged.
 */
DPH_API long dph_test_coroutines(int n_tasks, int yields);
/* packBytes of a whole read as `map` hands its reads to the device (sequence/sequence.go:59-93); out: ceil(n / 4) bytes; scalar_only:
   without the AVX2 path */
DPH_API void dph_pack_bases(const char* bases, int64_t n, uint8_t* out, int scalar_only);
/* SeedIndex.AddSeeds (seeds/seeds.go:62-156) of one top-level sequence into an empty index, as the planner's host selection does it: the index's
   seedMap (k-mers in seed-id order); returns their number, -1 when cap is too small */
DPH_API int dph_hand_add_seeds(const char* bases, int64_t len, int k, int num_seeds, const double* values, uint32_t* seed_map, int cap);
/* SeedMatch.GetBasesCovered (seeds/sequence.go:830-858: the PAF line's tenth column) on raw segment arrays and matched seed indices;
   out2 = {countA, countB}; returns 1 where the reference would panic */
DPH_API int dph_hand_bases_covered(const int32_t* a_seg, int a_n, const int32_t* b_seg, int b_n, const int32_t* match_a, const int32_t* match_b, int n,
                                   int k, int64_t* out2);
/* multiAligner.Consensus of the host's consensus path (seeds/alignment.go:23-268) on raw segment arrays: sequence i = segs[off[i] .. off[i + 1]);
   the consensus' segments, the indices of the sequences whose match was kept (>= 3 pairs) in the order returned, their pairs */
DPH_API int dph_hand_consensus(const int32_t* segs, const int64_t* off, int n_seqs, int k, int32_t* cons_out, int64_t cons_cap, int64_t* cons_n,
                               int* kept, int64_t* out_counts, int32_t* out_a, int32_t* out_b, int64_t cap, int64_t* n_matches);
DPH_API int dph_hand_is_consistent(const int64_t* left5, const int64_t* right4, int circular, int64_t ref_len);
DPH_API int dph_hand_remove_dominated(const int64_t* maps3, int n, int64_t query_len, int* kept);
DPH_API void dph_hand_trim_indices(int upto, const int32_t* match_a, const int64_t* off, int n_matches, int min_match, int length, int* out2);
DPH_API void dph_profile_print(void);
/* process-wide pipeline counters since the process started: 0 plans computed, 1 thrown away, 2 erased by a commit's flags, 3 rounds
   executed, 4 rejected at the commit, 5 committed, 6 us in plan computes (wall, all lanes), 7 us the slots waited for plans,
   8 us the committing thread (the caller of dph_overlap_step) waited for the next round in order, 9 us it spent on a round's text
   (13: of which waiting for a formatter thread), 10 us on flags + planner bookkeeping, 11 us keeping the step's text,
   12 us the formatter threads spent formatting (all threads together), 14 the planner's lanes at the moment (a state, not a sum);
   -1 for any other index */
DPH_API int64_t dph_planner_counter(int which);
/* The library keeps some host buffers between jobs (process-wide, shared by all handles, surviving dph_overlap_destroy): PAF text
 * strings and record arrays of finished rounds (at most 512 MB), window-cache chunks (at most 16 x 11.5 MB), the staging block of
 * the last `map` command (reference + reads, ~400 MB at BASELINE config 3).  A long-lived embedder calls this after its last job
 * (or whenever it wants the memory back; a running job simply allocates again).  Returns the bytes released. */
DPH_API int64_t dph_release_caches(void);
DPH_API int dph_selftest_planner_flags(void* reads, int k, int64_t seed_batch_size, const double* values);
DPH_API int dph_selftest_planner_lanes(void* reads, int k, int64_t seed_batch_size, const double* values, int lanes, int flag_every,
                                       int64_t* n_rounds);
/* the plan chain of a round-parallel run of `world` ranks whose planners compute only their own rounds' plans and guess the rest
 * (Planner::setOwnership), played against a commit that accepts a plan iff it starts at the committed firstSequence */
DPH_API int dph_selftest_planner_sparse(void* reads, int k, int64_t seed_batch_size, const double* values, int world, int flag_every,
                                        int64_t* n_rounds, int64_t* n_redone);
DPH_API int dph_selftest_touch(int k, const uint32_t* seeds, int64_t n_seeds, const uint32_t* kmers, int64_t n_windows, int64_t stride,
                               uint8_t* res, int* isa_mask);
/* finalCheckWorker (commands/overlap.go:197-233) over externally supplied queries, indexed sequences and matches (flat arrays
 * in the reference's segment layout): returns the round's PAF text and applies SetIgnore to `reads` */
DPH_API const char* dph_finalcheck(void* reads, int k, int64_t overlap_size, const uint32_t* seed_kmers, int64_t n_seeds,
                                   const int32_t* q_segs, const int64_t* q_off, const int64_t* q_id, const int64_t* q_seq_id,
                                   const int64_t* q_len, const int64_t* q_offset, const int64_t* q_inset, int64_t n_q,
                                   const int32_t* i_segs, const int64_t* i_off, const int64_t* i_id, const int64_t* i_len,
                                   const int64_t* i_offset, const int64_t* i_inset, int64_t n_i, const int64_t* m_query,
                                   const int64_t* m_target, const int64_t* m_off, const int32_t* m_a, const int32_t* m_b,
                                   int64_t n_m, int64_t num_query_seqs, int64_t* out_len, int64_t* out_stats);

#ifdef __cplusplus
}
#endif
#endif /* DOWNPORE_HOST_H */
