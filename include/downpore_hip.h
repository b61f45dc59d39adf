/*
 * downpore_hip.h — C ABI of libdownpore_hip.so: the MI355X (gfx950) implementation of downpore's
 * seed-index + seed-chaining overlap/map hot path.
 *
 * This is the drop-in boundary.  The reference's native boundary is a set of
 * per-slice Plan-9 assembly routines called millions of times (util/bitset.go:197,248-254;
 * sequence/sequence.go:65,326,327,438) — far too fine for a GPU — so the boundary moves one level up,
 * to the batched bodies of the Go interfaces `overlap.Overlapper` (overlap/overlap.go:24-29),
 * `mapping.Mapper` (mapping/mapping.go:22-26) and `seeds.SeedIndex` (seeds/seeds.go:11-21).  A Go type
 * implementing those interfaces marshals to the calls below through cgo.
 *
 * Conventions
 *   - every function returns 0 on success, a negative dp_status on failure; dp_last_error(ctx) gives text.
 *   - plain pointers and sizes only.  Inputs are borrowed for the duration of the call (the library copies;
 *     no caller pointer is retained — cgo-safe).  Output pointers inside the *_batch structs point into
 *     library-owned pinned host buffers that stay valid until the next call of the SAME function on the same
 *     context (or dp_ctx_destroy).
 *   - one host thread per context; a context owns one HIP stream on one device.
 *   - all results are bit-exact restatements of the reference's arithmetic (integer / bitset work only).
 */
#ifndef DOWNPORE_HIP_H
#define DOWNPORE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Only the entry points declared here are exported from libdownpore_hip.so (the library is built with
 * -fvisibility=hidden; tests/test_abi_symbols.py checks the dynamic symbol table against this header). */
#define DP_API __attribute__((visibility("default")))

typedef struct dp_ctx dp_ctx;

enum dp_status {
    DP_OK = 0,
    DP_ERR_HIP = -1,       /* a HIP runtime call failed */
    DP_ERR_ARG = -2,       /* invalid argument */
    DP_ERR_STATE = -3,     /* call order violated (e.g. dp_scan before dp_round_begin) */
    DP_ERR_CAPACITY = -4,  /* a reference capacity limit was hit (the reference would panic) */
    DP_ERR_NODEVICE = -5   /* no usable GPU */
};

/* Library/ABI version and the code-object architecture this build targets ("gfx950"). */
DP_API const char* dp_version(void);

/* Create a context on HIP device `device`.  Fails (DP_ERR_NODEVICE) when no GPU is present: there is no CPU
 * fallback. */
DP_API int dp_ctx_create(int device, dp_ctx** out);
/* A second context on the same device that BORROWS the resident reads of `src` (no copy): lets several host threads
 * drive independent rounds concurrently, each with its own stream and per-round state.  Destroy it before `src`. */
DP_API int dp_ctx_create_shared(dp_ctx* src, dp_ctx** out);
/* Give the context's stream the device's highest (high != 0) or default scheduling priority.  For the context of a
 * latency-critical caller next to throughput work - the goroutine that runs PrepareQueries (overlap/overlap.go:157) and
 * waits for dp_select_seeds while other contexts keep the GPU busy with whole rounds.  Call it while the context is idle. */
DP_API int dp_ctx_set_priority(dp_ctx* ctx, int high);
/* How calls of this library wait for their stream: spin != 0 = the runtime's busy wait (immediate wake-up, the waiting thread
 * occupies a core), 0 = poll an event every 20 us (the default: a waiting thread costs nothing).  Process-wide; a caller with
 * one executor thread per context and cores to spare wants the first (eight rounds in flight: 0.44 against 0.54 ms per round). */
DP_API void dp_set_stream_wait(int spin);
/* Kernel times in the per-call statistics (dp_scan_batch.kernel_ms, dp_paf_batch.*_kernel_ms ...) come from events recorded
 * around the launches; every event is a packet of its own for the command processor (five rounds in flight: 0.307 ms per
 * round with every round timed, 0.294 with every eighth, 0.286 with none).  every = N: each context times every N-th of its
 * rounds and reports 0 ms for the others; 0 = never; the default is 8 (or DP_KERNEL_TIMING).  Process-wide. */
DP_API void dp_set_kernel_timing(int every);
DP_API void dp_ctx_destroy(dp_ctx* ctx);
/* The device blocks (from 4 KiB on) and pinned host buffers of a destroyed context stay parked in a process-wide cache, so that the
 * next context - the next `map` run, the next job's index build - does not pay the driver for them again (a context holds some eighty
 * of them; releasing four contexts one hipFree at a time was a quarter of a config-3 `map` run).  What stays parked once the context
 * that owned the reads has gone is capped: DP_DEV_CACHE_MB (default 16384) and DP_PIN_CACHE_MB (default 2048).  This call gives
 * everything parked back to the driver and returns the bytes released; safe at any time.  (Infrastructure of the device path: the
 * reference, a CPU program, has no counterpart.) */
DP_API int64_t dp_release_device_caches(void);
DP_API const char* dp_last_error(const dp_ctx* ctx); /* ctx may be NULL: error of the last failed dp_ctx_create */

/* ---- A1: reads resident in HBM ------------------------------------------------------------------------
 * Replaces NewPackedSequence/packBytes (sequence/sequence.go:67-93, sequence/asm_amd64.s:33-78) for the whole
 * read set: `bases` is the concatenated ASCII of all reads, read r = bases[off[r] .. off[r+1]).  Packing
 * (2 bit/base, ((b>>1)^((b&4)>>2))&3, first base in the top bits of each byte) runs on the device.  Reads stay
 * resident for every later round. */
DP_API int dp_reads_upload(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_reads);
/* The same, but every read r >= first_paired is stored TWICE: as device read first_paired + 2*(r - first_paired) and,
 * reverse-complemented (sequence.go:179-198: reversed, 3 - code per base), as the next id.  `downpore map` scans every
 * query window on both strands (mapping.go:497-499); the reverse strands are produced by the pack kernel instead of
 * being built on the host and sent over PCIe. */
DP_API int dp_reads_upload_rc(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_reads, uint32_t first_paired);
/* dp_reads_upload_rc for a host that holds its reads the way the reference does - packed (sequence.packedSequence,
 * sequence/sequence.go:22-31: 2 bits per base, ceil(len / 4) bytes, first base in a byte's top bits, made by packBytes when the file is
 * read) - so that a quarter of the bytes cross PCIe: `packed` holds the reads one after another, read r at the sum over the reads before
 * it of ceil(lens[r'] / 4) rounded up to a multiple of 16 bytes (the device's own layout of a forward-only read set); the padding bytes
 * are not looked at.  The device puts the reads in place and makes the reverse strands of the reads >= first_paired from the packed
 * forward ones (sequence.go:179-198).  From memory of dp_host_alloc (pinned) the bytes travel in one copy, from anywhere else through
 * the staging ring of dp_reads_upload. */
DP_API int dp_reads_upload_packed_rc(dp_ctx* ctx, const uint8_t* packed, const uint32_t* lens, uint32_t n_reads, uint32_t first_paired);
/* Pinned host memory from the library's block cache (kept across contexts up to DP_PIN_CACHE_MB; NULL: none to be had). */
DP_API void* dp_host_alloc(size_t bytes);
DP_API void dp_host_free(void* p);
/* dp_reads_upload_rc that returns while the reads still travel (round 5: `map` maps its first reads while its last ones are on the link -
 * 400 MB of ASCII are 11 ms of a 59 ms config-3 run).  The call itself makes the new read set's tables resident and returns once host
 * reads [0, ready_first) are packed on the device, both strands; a thread of the library sends the rest on piece by piece, on a stream
 * of its own, and packs every read as soon as its bases have arrived.  `bases` must stay valid until dp_reads_upload_wait(ctx,
 * 0xffffffff) has returned.  Kernels that read packed reads (dp_scan and everything behind it) may only be given reads a wait has
 * covered.  Contexts made with dp_ctx_create_shared may be created while the upload runs.  Reference: the same as dp_reads_upload_rc
 * (sequence/seqio.go:106 `readFasta` + sequence/sequence.go:59-80 `packBytes`; the reverse strands: sequence.go:185-189). */
DP_API int dp_reads_upload_rc_begin(dp_ctx* ctx, const uint8_t* bases, const int64_t* off, uint32_t n_reads, uint32_t first_paired,
                                    uint32_t ready_first);
/* Blocks until host reads [0, host_read_hi) of a dp_reads_upload_rc_begin are packed on the device (at once when no upload is pending).
 * Callable from any thread, on the owner or on a context that borrows its reads.  host_read_hi = 0xffffffff (owner only): until the
 * whole set is resident and the library's thread has ended; returns the upload's error, if it had one. */
DP_API int dp_reads_upload_wait(dp_ctx* ctx, uint32_t host_read_hi);
/* Copy back the packed bytes of one read (ceil(len/4) bytes) — test hook. */
DP_API int dp_reads_packed(dp_ctx* ctx, uint32_t read, uint8_t* out, uint64_t cap, uint64_t* n_bytes);
DP_API uint32_t dp_reads_count(const dp_ctx* ctx);
DP_API uint64_t dp_reads_total_bases(const dp_ctx* ctx);

/* ---- A22: k-mer histogram -----------------------------------------------------------------------------
 * Replaces sequtil.KmerOccurrences/countWorker (util/sequtil/kmers.go:34-69): counts_out[kmer] (4^k entries)
 * = occurrences of every k-mer over all uploaded reads. */
DP_API int dp_kmer_histogram(dp_ctx* ctx, int k, uint64_t* counts_out);

/* ---- A22 + A23: the k-mer value table -----------------------------------------------------------------------
 * Replaces the block between "Counting all k-mers" and "Counting complete" of the commands (commands/overlap.go:55-93,
 * commands/map.go:45-71): KmerOccurrences, value = 1 - |frequency - 0.000005| for k-mers seen at least 3 times,
 * then util/sequtil/kmers.go:87-112: counts merged with their reverse complements (in place, so every pair ends at
 * twice its sum) and the n/100 k-mers with the highest merged counts set to 0 (ties at the threshold: highest k-mer id
 * first), values[0] = 0.  Everything runs on the device; the table stays resident as the context's value table (what
 * dp_values_upload would have installed) and is copied to values_out (4^k doubles) unless that is NULL.  float64,
 * bit-identical to the host computation. */
DP_API int dp_kmer_values(dp_ctx* ctx, int k, double* values_out);
/* The resident table (n = 4^k doubles) copied to the host on the CALLING context's stream - a context that borrows the
 * reads may do it while the owner goes on with dp_scan_prepare. */
DP_API int dp_values_download(dp_ctx* ctx, double* values_out, uint64_t n);
/* The same table at 2 bytes per k-mer.  A value is a function of the k-mer's own count and the total count alone
 * (commands/overlap.go:73-88), so codes_out[i] = count of k-mer i where the table holds a value and 0 where it holds 0
 * (count < 3, the 1 % cut of kmers.go:98-110, k-mer 0); *total_out = tot of overlap.go:60-63.  value(i) is then
 * f(codes_out[i], total) - 65536 evaluations of overlap.go's expression rebuild every entry bit for bit.
 * *overflow_out = 1 when a k-mer that holds a value was seen more than 65535 times (its code saturates: fetch the table
 * with dp_values_download instead).  DP_ERR_STATE for a table installed by dp_values_upload (no histogram behind it). */
/* The same with one byte per k-mer (0 = value 0, 1..254 = the count): *overflow_out = 1 when a valued k-mer counts 255 or more -
 * codes_out is not usable then and the caller takes the two-byte form.  With the 1 %% cut in force nearly every table fits. */
DP_API int dp_values_download_codes8(dp_ctx* ctx, uint8_t* codes_out, uint64_t n, uint64_t* total_out, int* overflow_out);
DP_API int dp_values_download_codes(dp_ctx* ctx, uint16_t* codes_out, uint64_t n, uint64_t* total_out, int* overflow_out);

/* ---- round state: the seed set --------------------------------------------------------------------------
 * Mirrors the per-round SeedIndex tables kmers/kmerMap/seedMap (seeds/seeds.go:13-18): seed id = position in
 * seed_kmers.  Builds the 4^k-bit membership table and the k-mer -> seed-id map on the device (sparse
 * set/clear: only the entries of the previous round's seeds are touched). */
DP_API int dp_round_begin(dp_ctx* ctx, int k, const uint32_t* seed_kmers, uint32_t n_seeds);

/* ---- A2 + A10: batched packed k-mer scan ---------------------------------------------------------------
 * Replaces SeedIndex.NewSeedSequence = packedCountKmers + packedWriteSegments + kmerMap translate
 * (seeds/seeds.go:33-50; sequence/asm_amd64.s:81-203,206-394) for MANY sequence views in one call.
 * An item is the run of k-mers that the reference's scan examines for one view:
 *     k-mer start positions [start, start + n_kmers) of read `read` (0 = the read's first base).
 * For a cached whole read view (seqio.go:115) n_kmers = len-k+1; for a top-level read with len%4==0 the
 * reference examines 4 fewer k-mers and the caller passes n_kmers = len-k+1-4.
 * Items whose hit count is >= min_seeds get their seed sequence written out as the reference's interleaved
 * segments array [gap0, seed0, gap1, ..., gapN] (int32); others only report their count. */
typedef struct {
    uint32_t read;
    uint32_t start;
    uint32_t n_kmers;
    uint32_t min_seeds;
} dp_scan_item;

typedef struct {
    uint32_t n_items;
    const uint32_t* n_seeds; /* [n_items] hits per item */
    const uint64_t* seg_off; /* [n_items+1] offsets into segs (int32 units); empty range if below min_seeds */
    const int32_t* segs;     /* host copy of all written segments */
    uint64_t n_segs;
    double kernel_ms;        /* device time of all scan kernels of this call (HIP events on the context's stream) */
    double count_kernel_ms;  /* ... of the count pass alone (the kernel that streams every item once) */
    double write_kernel_ms;  /* ... of the write pass alone (survivors only) */
    uint64_t bases_scanned;  /* sum of n_kmers + k - 1 over items */
} dp_seedseq_batch;

DP_API int dp_scan(dp_ctx* ctx, const dp_scan_item* items, uint32_t n_items, dp_seedseq_batch* out);

/* AddSequences-level form of the scan (overlap.go:217-250): every read r in [lo, hi) whose ignore[r] is 0 is scanned
 * as the view a later pass receives (top_level != 0: re-read top-level sequences incl. the len%4==0 quirk; 0: cached
 * views) and only the SURVIVORS (>= min_seeds hits; chunkWorker drops the rest, overlap.go:259-261) come back, compacted
 * on the device; `extra` items (the query windows) are appended and always returned.  The ignore array is uploaded only
 * when ignore_epoch differs from the previous call on this context.  The device-resident segments keep the layout of
 * `segs` (survivors in read order, then the extra items), so dp_index_build can reference them directly. */
typedef struct {
    uint32_t n_survivors;
    const uint32_t* read;          /* [n_survivors] ascending read ids */
    const uint32_t* n_seeds;       /* [n_survivors] */
    const uint64_t* seg_off;       /* [n_survivors] offset of each survivor's segments in segs */
    uint32_t n_extra;
    const uint32_t* extra_n_seeds; /* [n_extra] */
    const uint64_t* extra_seg_off; /* [n_extra] */
    const int32_t* segs;
    uint64_t n_segs;
    double kernel_ms, count_kernel_ms, write_kernel_ms;
    uint64_t bases_scanned;
    uint32_t reads_scanned;
    uint32_t index_mode;           /* 1: counts and segments came from the resident k-mer position index, not a scan */
    uint64_t index_hits;           /* index mode: occurrences of this round's seed k-mers in the whole read set */
} dp_survivor_batch;

/* Optional, once per read set and k before the rounds: does the one-off work dp_scan_reads would otherwise do inside its
 * first call - building the resident k-mer position index when the read set is large enough for it (>= 1 Gbase, or
 * DP_SCAN_INDEX=1).  Right after dp_kmer_values the build reuses that call's k-mer histogram. */
DP_API int dp_scan_prepare(dp_ctx* ctx, int k);
/* Releases what dp_scan_prepare / dp_scan_reads / dp_kmer_values left resident for the rounds of ONE job on the context
 * that owns the reads: the k-mer position index (8 B per base) and the k-mer histogram.  The reads stay.  For a caller
 * that is done with `overlap` and goes on to something else, or runs another job (other k) on the same reads.  No
 * borrowing context may be inside a dp_scan_reads call. */
DP_API int dp_scan_release(dp_ctx* ctx);
DP_API int dp_scan_reads(dp_ctx* ctx, const uint8_t* ignore, uint64_t ignore_epoch, uint32_t lo, uint32_t hi, int top_level,
                  uint32_t min_seeds, const dp_scan_item* extra, uint32_t n_extra, dp_survivor_batch* out);

/* ---- A9 (selection part): AddSeeds' block-winner / top-N selection on the device -------------------------
 * seeds.AddSeeds (seeds/seeds.go:62-129) walks a query window in blocks of k rolling k-mers followed by 2k skipped
 * bases, looks every evaluated k-mer up in the 4^k-entry value table (`kmerRanks`), keeps the block maximum
 * (strict >, first wins, initial 0.0 / k-mer 0) and feeds it to an ascending top-`num_seeds` insertion list.  The
 * table probes are random 8-byte reads over 4^k*8 bytes (512 MiB at k=13): latency-bound on a CPU, cheap in HBM.
 * dp_values_upload keeps the table resident (borrowing contexts created AFTERWARDS share it); dp_select_seeds runs
 * the selection for `n` windows assuming no evaluated k-mer is a seed yet ("speculative" form: the caller re-runs a
 * window on the host when that assumption fails, i.e. when AddSeeds would have abandoned a block, seeds.go:94-97)
 * and writes num_seeds k-mers per window to `top_out` in list order (untouched slots hold k-mer 0, as in the
 * reference).  win[i].read/start/n_kmers describe the window (n_kmers = window length in BASES here); min_seeds is
 * ignored.  num_seeds <= 64. */
DP_API int dp_values_upload(dp_ctx* ctx, const double* values, uint64_t n);
/* FASTQ input: AddSeeds weights a k-mer's value by the quality byte of its middle base, value *= q[nextIndex - k/2]
 * (seeds.go:99-101; the bytes are phred - 33 as sequence/seqio.go:169-173,231-236 stores them).  qual[off[r] + p] belongs
 * to base p of read r (the offsets of dp_reads_upload), has_qual[r] = 0 marks reads whose record had no usable quality
 * line.  After this call dp_select_seeds / dp_select_windows apply the weight; contexts created with dp_ctx_create_shared
 * AFTERWARDS share the bytes. */
DP_API int dp_quality_upload(dp_ctx* ctx, const uint8_t* qual, const int64_t* off, const uint8_t* has_qual, uint32_t n_reads);
DP_API int dp_select_seeds(dp_ctx* ctx, const dp_scan_item* win, uint32_t n, int k, int num_seeds, uint32_t* top_out);
/* dp_select_seeds that also hands back every k-mer the selection loop evaluated, `stride` slots per window (slot = block *
 * k + position inside the block; unused slots are 0xffffffff): exactly the k-mers AddSeeds tests against the seed set
 * (seeds.go:94-97), so the caller's "did the speculation hold" test is a run of probes of resident k-mers instead of a
 * re-walk of the window's bases.  *evaluated_out points into a library-owned pinned buffer (valid until the next
 * selection call on this context).  A window of L bases evaluates at most ceil((L - 2k) / 3k) * k k-mers. */
DP_API int dp_select_windows(dp_ctx* ctx, const dp_scan_item* win, uint32_t n, int k, int num_seeds, uint32_t* top_out,
                             const uint32_t** evaluated_out, uint32_t stride);


/* ---- A13: seed index build ------------------------------------------------------------------------------
 * Replaces SeedIndex.AddSequence + IndexSequences/index (seeds/seeds.go:272-305,372-384).  Indexed sequence i
 * is a view segs[seg_off .. seg_off + 2*n_seeds + 1) into the device-resident output of the last dp_scan
 * (the chunks produced by overlap.chunkWorker are exactly such views, overlap/overlap.go:253-318).  Builds the
 * posting bit-matrix (seed -> set of sequence indices) and the per-sequence seed bitsets. */
typedef struct {
    uint64_t seg_off;
    uint32_t n_seeds;
    uint32_t reserved;
} dp_seq_ref;

DP_API int dp_index_build(dp_ctx* ctx, const dp_seq_ref* seqs, uint32_t n_seqs);

/* ---- A14 + A5 + A6 + A7 + A8: index query and overlap chaining ---------------------------------------------
 * Replaces overlapper.matchWorker (overlap/overlap.go:346-387) for all queries of a round:
 *   SeedIndex.Matches -> util.GetSharedIDs (+ getSoftUnion{4,8,16}Asm semantics incl. threshold saturation and
 *   the 16-ladder's step-8 behaviour) -> IntSet.CountIntersectionTo prefilter -> seedAligner.PairwiseAlignments
 *   -> best-chain pick and the minMatches ratchet, candidates in ascending index order.
 * Query q has segments q_segs[q_off[q] .. q_off[q+1]) (reference layout).  max_query_len is the aligner's buffer
 * size NewSeedAligner(overlap/2) (overlap/overlap.go:349). */
typedef struct {
    uint32_t n_matches;
    const uint32_t* query;    /* [n_matches] query index, ascending */
    const uint32_t* target;   /* [n_matches] indexed-sequence index, ascending within a query */
    const uint64_t* off;      /* [n_matches+1] offsets into match_a / match_b */
    const int32_t* match_a;   /* query seed indices of the chain */
    const int32_t* match_b;   /* target seed indices of the chain */
    const int32_t* target_anchor; /* [2*n_matches]: base offset of the chain's first target seed from the target's start
                                   * (GetSeedOffset(match_b[first])) and of its last one from the end
                                   * (GetSeedOffsetFromEnd(match_b[last])), seeds/sequence.go - what Trimmed()
                                   * (overlap/combine.go:171-181) would otherwise sum over the whole chunk; INT32_MIN = not
                                   * computed.  A chunk's first / last gap may be negative (overlapping seeds), so an anchor may
                                   * be too: the host mirror sums for itself whenever it sees a negative one */
    /* test hooks */
    uint32_t n_queries;
    const uint64_t* cand_off; /* [n_queries+1] */
    const uint32_t* cand;     /* SeedIndex.Matches() output per query (ascending ids) */
    double query_kernel_ms;   /* device time of the soft-union (index query) kernel */
    double chain_kernel_ms;   /* device time of the prefilter+chaining kernel */
    uint64_t query_bytes;     /* algorithmic bytes of the index query (posting words inside windows * 8) */
    uint64_t chain_bytes;     /* ... of the prefilter + chaining kernel: 2 bitset rows per candidate, both segment arrays
                               * per chained pair, the chains written (counted by the kernel itself) */
} dp_match_batch;

/* want_candidates: bit 0 = also return Matches()' candidate lists (test hook); bit 1 = leave the matches on the device
 * (only the kernel times / byte counts of `out` are filled in): for dp_consensus_paf, or a later dp_fetch_overlaps; bit 2
 * (with bit 1) = do not even wait: the stage stays pending, the next call on the context must be dp_consensus_paf, which
 * evaluates it in the wait it needs anyway (overflows of either stage are repeated there) and reports its times and bytes. */
DP_API int dp_find_overlaps(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t n_queries,
                     double hit_fraction, int k, uint32_t max_query_len, int want_candidates, dp_match_batch* out);

/* Optional: announces the queries of the round's coming dp_find_overlaps before the index is built (matchWorker's queries are
 * known as soon as the scan is: overlap/overlap.go:200-214 builds them from the scan's output).  The library stages them now
 * and the index build's first launch (dp_index_build_chunked) carries them to the device, so that the query stage starts with
 * its kernel.  dp_find_overlaps must then be given arrays of the same content; with any other content, or when no launch came
 * in between, it simply uploads as it always did.  The arrays are borrowed for the duration of the call only. */
DP_API int dp_query_prestage(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t n_queries, double hit_fraction);

/* ---- A19 + A20: map-flavour query (mapping.performMapping core) -----------------------------------------------
 * For each window: Matches(0.25) candidates, CountIntersectionTo prefilter and SeedSequence.Match
 * (Reduced x2 -> dynamicMatch -> extendChain; seeds/sequence.go:361-576) with the minMatches ratchet of
 * mapping/mapping.go:494-549 (window pairs fwd/rc are linked: the fwd ratchet also raises the rc threshold).
 * Windows come in (fwd, rc) pairs: window 2i is the forward query, 2i+1 its reverse complement; w_len[w] is the
 * window's length in bases (SeedSequence.Len(), used by the 2/3 flank test mapping.go:536,576).  Only chains that pass
 * that test are returned, in the order performMapping appends them. */
typedef struct {
    uint32_t n_chains;
    const uint32_t* window;  /* [n_chains] window index */
    const uint32_t* target;  /* [n_chains] indexed-sequence (reference chunk) index */
    const uint64_t* off;     /* [n_chains+1] */
    const int32_t* match_a;  /* window seed indices */
    const int32_t* match_b;  /* target seed indices */
    double kernel_ms;
    double alg_bytes;        /* algorithmic bytes of the call's index query + prefilter + chaining: what its roofline is priced on */
} dp_chain_batch;

DP_API int dp_map_windows(dp_ctx* ctx, const int32_t* w_segs, const uint64_t* w_off, const uint32_t* w_len, uint32_t n_windows,
                   int k, dp_chain_batch* out);

/* ---- the reference index spread over several contexts / GPUs (BASELINE config 5: mapping/mapping.go:67-109 for a 3 Gb
 * reference) -----------------------------------------------------------------------------------------------------------
 * Every shard holds the same seeds (dp_round_begin) and a contiguous range of the reference chunks whose first chunk id is
 * a multiple of 64 (dp_index_build on that range: posting / seed-set words [word_base, word_base + W) of the whole index).
 * util.GetSharedIDs works on the sets' windows - a posting set takes part from the scan's first word to its own last word
 * (util/bitset.go:323-353: the early return, the drops, the gather order of the 16-ladder) - so a shard must know the WHOLE
 * sets: dp_index_meta returns a shard's rows {count, first word, last word, last + 1} (local word numbers), the caller
 * combines them over the shards (sum, min, max in global word numbers; an empty set keeps start 1 / end 0) and hands the
 * result to every shard with dp_index_set_global together with the shard's word_base and the total number of chunks.
 * dp_map_windows_shard is dp_map_windows for one strand (phase 0 = the forward windows, 1 = the reverse complements) of
 * every window pair against one shard.  performMapping's ratchets (mapping.go:543-549, 583-586) run over a window's
 * candidates in ascending chunk id, forward strand first, and the forward ratchet also raises the reverse threshold:
 * thr_io[2 * pair + strand] carries minMatches / minRCMatches from shard to shard (initialise to -1 = "take the window's
 * own"); call phase 0 on the shards in ascending chunk order, then phase 1 in the same order.  `target` in `out` is the chunk's
 * index inside the shard.  Chains of a window: phase 0 results of shard 0, 1, ..., then phase 1 results of shard 0, 1, ... */
DP_API int dp_index_meta(dp_ctx* ctx, uint32_t* meta_out, uint32_t n_seeds);
DP_API int dp_index_set_global(dp_ctx* ctx, const uint32_t* meta_global, uint32_t n_seeds, uint32_t word_base, uint32_t n_seqs_global);
DP_API int dp_map_windows_shard(dp_ctx* ctx, const int32_t* w_segs, const uint64_t* w_off, const uint32_t* w_len, uint32_t n_windows,
                         int k, int phase, int32_t* thr_io, dp_chain_batch* out);
/* The parallel part of SeedIndex.AddSingleSeeds (seeds/seeds.go:160-200; NewMapper calls it on the mapping reference,
 * mapping/mapping.go:67-109) for resident read `read` (a top-level sequence) and the resident value table of this k: for every
 * window of seed_rate bases - for (i = 0; i < len - seed_rate; i += seed_rate) - its best-valued k-mer (what the reference adds as a
 * seed when no k-mer of the window's count region is a seed yet) and the k-mers of its count region that are the best of ANY window
 * (only those can ever be seeds).  The caller walks the windows in order: a window none of whose candidates is a seed so far adds its
 * best k-mer - the sequential rule, with five or six probes per window instead of seed_rate.  The arrays are the library's (ordinary
 * host memory, valid until the context's next dp_single_seed_candidates or its destruction). */
typedef struct {
    uint32_t n_windows;
    const uint32_t* best;      /* [n_windows] */
    const uint32_t* cand_off;  /* [n_windows + 1] offsets into cand */
    const uint32_t* cand;      /* candidate k-mers of every window's count region */
} dp_single_seed_batch;
DP_API int dp_single_seed_candidates(dp_ctx* ctx, uint32_t read, int k, int64_t seed_rate, dp_single_seed_batch* out);


/* ---- A16 (part): seed-space multiple alignment of multiAligner.Consensus (seeds/alignment.go:52-247) ------------
 * For each of `n_groups` groups (one per query window) the caller passes the Reduced() seed sequences of the trimmed
 * matched targets (seeds shared by >= 2 of them, alignment.go:45-50), flattened: sequence s = segs[seq_off[s] ..
 * seq_off[s+1]) in the usual [gap, seed, ..., gap] layout (an empty range = the sequence has no shared seed), group g =
 * sequences group_off[g] .. group_off[g+1].  The device grows each group's consensus ([dist, seed, ..., 0]) and, per
 * sequence, the list of (consensus index, index into the REDUCED sequence) pairs; the caller maps the second component
 * back through Reduced()'s index map, drops sequences with < 3 pairs (alignment.go:258-266) and goes on with
 * trimToBestSeed.  Sequence s owns match_a/match_b[seq_off[s] .. +match_len[s]).  flags[g] != 0: the group was not
 * computed (more than 64 sequences, more than 6144 ints, or a value outside the 32-bit safe range) and must be done
 * by the caller. */
typedef struct dp_consensus_batch {
    uint32_t n_groups;
    const int32_t* cons;        /* consensus of group g: cons[cons_off[g] .. + cons_len[g]) */
    const uint64_t* cons_off;
    const uint32_t* cons_len;
    const int32_t* match_a;
    const int32_t* match_b;
    const uint32_t* match_len;  /* per sequence */
    const uint32_t* flags;      /* per group */
    double kernel_ms;
} dp_consensus_batch;
DP_API int dp_consensus_align(dp_ctx* ctx, const int32_t* segs, const uint64_t* seq_off, const uint32_t* group_off, uint32_t n_groups,
                       int k, dp_consensus_batch* out);

/* ---- A16 + A17 on the device: BuildConsensus (overlap/combine.go:163-193) with multiAligner.Consensus
 * (seeds/alignment.go:23-268), NewSeedContig / trimToBestSeed (combine.go:21-133) and the numbers finalCheckWorker prints
 * (commands/overlap.go:197-233), for every query window of the round whose chaining stage ran last on this context
 * (dp_find_overlaps with bit 1 of want_candidates set keeps the matches on the device and skips their download).  Queries
 * must be (forward, reverse complement) pairs 2g, 2g+1 as PrepareQueries emits them (overlap.go:189-201); a "group" is
 * one such window.  metas[i] = id (read), Len(), GetOffset(), GetInset() of indexed sequence i as AddSequences built it
 * (overlap.go:253-318); rc_of[s] = kmerMap[ReverseComplement(seedMap[s], k)] (seeds/sequence.go:125-159).
 * Per group the caller gets the lines of the contig's parts in order - everything `Fprintf` prints except the names -
 * and the reads finalCheckWorker passes to SetIgnore (:203-205, 217-223), in call order.  flag != 0: the group does not
 * fit the device layout (> 64 trimmed sequences, > 4096 ints, a value beyond 2^28, or a state in which the reference
 * itself would panic); the caller then runs the host path for that group on the matches of dp_fetch_overlaps. */
typedef struct {
    uint32_t read;                   /* SeedSequence.id */
    int32_t length, offset, inset;   /* Len(), GetOffset(), GetInset() */
} dp_seq_meta;
typedef struct {
    uint32_t q_read, t_read;         /* Parts[0], Parts[id]: names are the caller's */
    int32_t q_len, q_start, q_end;   /* SeqLengths[0], Offsets[0], Offsets[0] + Lengths[0] */
    int32_t t_len, t_start, t_end;   /* SeqLengths[id], Offsets[id], Offsets[id] + Lengths[id] */
    int32_t ident;                   /* Matches[id-1].GetBasesCovered(k), consensus side; 0 where the reference would panic */
    uint32_t minus;                  /* ReverseComplement[0] != ReverseComplement[id] */
} dp_paf_rec;
typedef struct {
    uint32_t slot;                   /* lines of the group: paf[slot .. slot + n_lines), ignores: ignore_ids[slot .. slot + n_ignore) */
    uint32_t n_lines, n_ignore;
    uint32_t bad_back;               /* "Bad back:" diagnostics suppressed (combine.go:93-102) */
    uint32_t empty_match;            /* lines whose GetBasesCovered would have panicked (ident printed as 0) */
    uint32_t flag;
    uint32_t n_matches;              /* hits of the two queries of the window (commands/overlap.go:158-173) */
    uint32_t reserved;               /* flag == 0: algorithmic bytes the window's consensus read and wrote (records, chains, anchors,
                                      * trimmed segments, query segments in; PAF records, ignore ids, this record out);
                                      * flag == 1: why the device left the window to the caller (diagnosis) */
} dp_group_meta;
typedef struct {
    uint32_t n_groups;
    const dp_group_meta* groups;
    const dp_paf_rec* paf;
    const uint32_t* ignore_ids;
    double kernel_ms;
    uint32_t n_indexed;              /* sequences in the round's index (exact also after dp_index_build_chunked) */
    double query_kernel_ms, chain_kernel_ms;  /* the chaining stage this call finished (dp_find_overlaps, want_candidates bit 2) */
    uint64_t query_bytes, chain_bytes;
    double index_kernel_ms;          /* device time of the round's index build (dp_index_build_chunked: chunk, seed-set rows, posting matrix, row meta) */
} dp_paf_batch;
DP_API int dp_consensus_paf(dp_ctx* ctx, const dp_seq_meta* metas, uint32_t n_seqs, const int32_t* rc_of, uint32_t n_seeds, int k,
                            int overlap_size, dp_paf_batch* out);

/* A12 + A4/A13 without the host: overlap.chunkWorker (overlap/overlap.go:253-318: a read with fewer than 3 * min_seeds hits or of
 * less than one chunk_size goes in whole; otherwise chunks of >= min_seeds seeds and ~chunk_size bases that back up overlap / 2
 * bases, the tail from 150 seeds before the end in one piece) for the first `n_survivors` survivors of the context's last
 * dp_scan_reads, in file order, then AddSequence + IndexSequences as dp_index_build does.  The chunks never leave the device:
 * *n_seqs_cap is an upper bound of their number (it sizes the bit matrices; dp_consensus_paf reports the exact count and takes
 * the chunks' {read, length, offset, inset} from the device when its `metas` is NULL; `inset` = the served view's inset);
 * dp_index_chunks copies them to the host for a caller that needs them there. */
DP_API int dp_index_build_chunked(dp_ctx* ctx, int64_t chunk_size, int64_t overlap, uint32_t min_seeds, int32_t inset, uint32_t n_survivors,
                           uint32_t* n_seqs_cap);
DP_API int dp_index_chunks(dp_ctx* ctx, dp_seq_ref* refs_out, dp_seq_meta* metas_out, uint32_t cap, uint32_t* n_out);
/* The dp_index_build_chunked call the caller will make after its NEXT dp_scan_reads, announced: a scan answered from the resident
 * k-mer index then launches the chunk stage itself, directly behind its own kernels and before anybody waits - for the survivors
 * the device finds, in buffers sized from the context's previous round - and dp_scan_reads returns as soon as its own output has
 * arrived, so the chunk stage runs while the host prepares the queries (the reference's chunkWorker goroutines likewise start on
 * a sequence as soon as AddSequences has produced it, overlap/overlap.go:217-250).  dp_index_build_chunked with the same
 * parameters then only checks the guesses against the exact bound, and launches the old way for the rare round that outgrew them
 * (or whenever nothing was launched: a scanned round, the first round of a context).  dp_index_prechained: 1 if the last
 * dp_scan_reads did launch the chunk stage (dp_query_prestage has nothing to ride on then). */
DP_API int dp_index_prechain(dp_ctx* ctx, int64_t chunk_size, int64_t overlap, uint32_t min_seeds, int32_t inset);
DP_API int dp_index_prechained(const dp_ctx* ctx);
/* The match lists of the last dp_find_overlaps on this context (what that call returns itself unless bit 1 of
 * want_candidates asked it not to). */
DP_API int dp_fetch_overlaps(dp_ctx* ctx, dp_match_batch* out);

/* ---- introspection for tests ---------------------------------------------------------------------------------- */
/* posting row of `seed` (n_words = ceil(n_seqs/64)) and its popcount/start/end as the reference's IntSet holds. */
DP_API int dp_index_posting_row(dp_ctx* ctx, uint32_t seed, uint64_t* words, uint32_t cap_words, uint32_t* n_words,
                         uint32_t* count, uint32_t* start, uint32_t* end);
DP_API int dp_index_seedset_row(dp_ctx* ctx, uint32_t seq, uint64_t* words, uint32_t cap_words, uint32_t* n_words);

/* ---- multi-GPU: scan sharded by read, survivors all-gathered over xGMI -------------------------------
 * Ranks own ascending contiguous read ranges [lo, hi) (dp_scan_reads) of the same resident read set; after a round's scan
 * dp_allgather_survivors concatenates every rank's survivors (read id, hit count, segments) in rank order = file order,
 * device to device on the context's stream (counts first, then the payloads), and installs the result as the context's
 * scan output, so that dp_index_build / dp_find_overlaps see what a single GPU scanning every read would hold.  The
 * `extra` items of dp_scan_reads (the query windows) are scanned by every rank and stay local.
 *   dp_comm_unique_id + dp_comm_init : one process per GPU; an RCCL communicator (librccl is loaded at run time).  Rank 0
 *       creates the 128-byte id, the caller hands it to the other ranks by whatever means it has.
 *   dp_comm_init_local : one process driving n contexts (on n devices, or several on one): peer copies, no RCCL.  Every
 *       rank calls dp_allgather_survivors from its own host thread; the call is collective. */
typedef struct dp_comm dp_comm;
DP_API int dp_comm_unique_id(uint8_t* id_out /* [128] */);
DP_API int dp_comm_init(dp_ctx* ctx, int n_ranks, int rank, const uint8_t* unique_id /* [128] */, dp_comm** out);
DP_API int dp_comm_init_local(dp_ctx* const* ctxs, int n, dp_comm** out /* [n] */);
DP_API void dp_comm_destroy(dp_comm* comm);
/* The caller gives up on the job (its own work of a round failed before it reached the exchange): peers inside or entering
 * dp_allgather_survivors return an error instead of waiting for this rank for ever.  The communicator is dead afterwards
 * (every exchange on it fails); dp_allgather_survivors does the same by itself when it fails on a rank. */
DP_API void dp_comm_abort(dp_comm* comm);
DP_API int dp_comm_rank(const dp_comm* comm);
DP_API int dp_comm_size(const dp_comm* comm);
/* `local` = what dp_scan_reads just returned on `ctx`; `all` = the same description of the gathered set (host arrays in
 * library-owned pinned memory, valid until the next call on this communicator). */
DP_API int dp_allgather_survivors(dp_comm* comm, dp_ctx* ctx, const dp_survivor_batch* local, dp_survivor_batch* all);

/* All-gather of one variable-size byte string per rank on the same communicator: the result exchange of the round-parallel
 * layout (every GPU holds the reads and their index, the rounds - commands/overlap.go:119's loop iterations - are dealt to the
 * ranks, and a round's PAF text, SetIgnore ids and read lists travel to every rank, which commits them in round order).  *all_out
 * = the ranks' strings back to back in rank order, *sizes_out[n_ranks] their lengths (library-owned, valid until the next call on
 * this communicator).  Collective. */
DP_API int dp_allgather_blobs(dp_comm* comm, dp_ctx* ctx, const uint8_t* blob, uint64_t n, const uint8_t** all_out, const uint64_t** sizes_out);
/* The resident k-mer position index of a multi-GPU job (what dp_scan_prepare builds: 4 - 5 bytes per base of the read set, identical on
 * every rank) is built in shares once a communicator is announced: every rank radix-sorts the k-mers of 1 / n_ranks of the first-digit
 * buckets (all ranks hold all reads) and the shares - index entries, bucket offsets, k-mer counts - are all-gathered in place, device to
 * device (RCCL over xGMI, or copies between the contexts of one process).  BASELINE.json's "RCCL all-gather of the seed index".
 * Call on the context that owns the reads, on every rank, before dp_scan_prepare / dp_kmer_values; comm = NULL: every rank builds
 * the whole index on its own again.  dp_scan_prepare is then collective. */
DP_API int dp_kindex_set_comm(dp_ctx* ctx, dp_comm* comm);
/* Test hook: an order-independent digest of the resident index of `ctx` (built for k): out[0] = entries, out[1] = sum over the k-mers of
 * mix(k-mer, bucket start), out[2] = sum over the entries of mix(k-mer of the entry's bucket, read, position).  Two builds of the same
 * reads agree on all three whatever the order of the entries inside a bucket. */
DP_API int dp_kindex_digest(dp_ctx* ctx, int k, uint64_t* out /* [3] */);
/* Gather of one variable-size byte string per rank to rank `root` only: the PAF text of the round-parallel layout's rounds, which only
 * the rank that prints needs (finalCheckWorker's fmt.Print, commands/overlap.go:225-228, happens in one process; at 8 ranks every rank
 * otherwise receives and copies all 235 MB of a config-2 job's text).  sizes[n_ranks] = every rank's length, the same array on every
 * rank (the lengths travel inside the control blobs of dp_allgather_blobs: no size exchange here).  *all_out = the concatenation in
 * rank order on the root (library-owned, valid until the next blob call on this communicator), NULL on the other ranks.  Collective. */
DP_API int dp_gather_blobs(dp_comm* comm, dp_ctx* ctx, const uint8_t* blob, uint64_t n, const uint64_t* sizes, int root, const uint8_t** all_out);

/* Device pointers of the last dp_scan output, for a multi-GPU exchange driven by the caller (RCCL all-gather of
 * the survivors).  segs_dev: int32[n_segs]. */
DP_API int dp_scan_device_buffers(dp_ctx* ctx, void** segs_dev, uint64_t* n_segs);
/* extras_only != 0: dp_scan_reads leaves the surviving reads' segments on the device (for dp_index_build_chunked) and copies only
 * the extra items' (the query windows') to the host; `segs` of its result is then valid at the extra items' offsets only.
 * dp_scan_fetch_segments fetches the whole array of the last scan after all (host consensus path of flagged windows). */
DP_API int dp_scan_fetch_mode(dp_ctx* ctx, int extras_only);
DP_API int dp_scan_fetch_segments(dp_ctx* ctx, const int32_t** segs_out, uint64_t* n_segs);
/* Replace the device-resident scan output with externally gathered segments (host pointer). */
DP_API int dp_scan_import_segments(dp_ctx* ctx, const int32_t* segs, uint64_t n_segs);

#ifdef __cplusplus
}
#endif
#endif /* DOWNPORE_HIP_H */
