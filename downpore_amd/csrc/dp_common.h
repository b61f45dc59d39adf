// Internal declarations shared by the translation units of libdownpore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "downpore_hip.h"

// make COPYLOG=1: every hipMemcpyAsync of the library is counted per call site (file:line, direction, calls, bytes) and the table is
// printed when the process ends - the tool that names the call sites behind the runtime's copy kernels in a profile
#ifdef DP_COPY_LOG
hipError_t dp_copy_logged(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s, const char* file, int line);
#define hipMemcpyAsync(d_, s_, n_, k_, st_) dp_copy_logged((void*)(d_), (const void*)(s_), (n_), (k_), (st_), __FILE__, __LINE__)
#endif

#define DP_WAVE 64

// Growable device / pinned-host buffers owned by a context.
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct dp_kindex;

struct FindState;
struct dp_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    bool destroy_pending = false;    // dp_ctx_destroy of an owner whose reads are still borrowed: carried out by the last borrower
    dp_ctx* owner = nullptr;     // context whose reads (and k-mer position index) this one borrows
    std::atomic<int> n_borrowers{0};  // live contexts created from this one with dp_ctx_create_shared
    dp_kindex* kidx = nullptr;   // resident k-mer position index (dp_kindex.hip), owned by the reads' owner
    struct dp_comm* kx_comm = nullptr;  // dp_kindex_set_comm: the ranks of this communicator build the index in shares and all-gather them
    DevBuf d_kx_sz, d_kx_lo, d_kx_tmp, d_kx_keys, d_kx_vals;  // per-round scratch of the index path
    uint64_t kx_hits = 0;
    uint64_t kx_prev_hits = 0, kx_prev_segs = 0;  // the previous index-mode round of this context: what the one-go step is sized from
    uint32_t kx_prev_max = 0, kx_prev_surv = 0;
    uint64_t kx_oneshot_rounds = 0, kx_oneshot_redone = 0;
    // dp_index_prechain: the chunk stage launched behind an un-waited scan
    bool pc_armed = false, pc_launched = false;
    int64_t pc_chunk_size = 0, pc_overlap = 0;
    uint32_t pc_min_seeds = 0, pc_cap = 0, pc_tiles = 0, pc_prev_cap = 0, pc_prev_ns = 0;
    int32_t pc_inset = 0;
    uint64_t pc_hits = 0, pc_misses = 0;
    std::vector<uint32_t> ss_host;  // dp_single_seed_candidates' result (ordinary host memory: read by a sequential host walk)
    uint32_t last_n_extra = 0;    // extra items of the last dp_scan_reads
    uint32_t kx_seq = 0;          // sequence number the sort pass of a one-go index step stores into h_total[15] when its output is complete
    uint32_t kx_maxlen = 0;       // longest read (hit records hold 24 bits of position)
    size_t kx_maxlen_reads = 0;   // ... of a read set of this many reads
    uint32_t kx_head_reads = 0;  // reads the zeroed extra-item list heads (d_kx_lo) are sized for
    hipStream_t stream = nullptr;
    hipEvent_t ev[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [8], [9]: around the index build's kernels
    bool index_marked = false;  // marks 8 / 9 were recorded for the round's index build (dp_consensus_paf reads them)
    hipEvent_t ev_sync = nullptr;  // blocking-sync event: waiting host threads sleep instead of polling
    bool stream_priority_set = false;  // (dp_ctx_set_priority: such a stream is destroyed with its context instead of being parked)
    std::string err;
    bool borrowed_reads = false;  // d_packed/d_boff/d_len belong to another context
    struct ReadsUpload;           // a read set still on its way to the device (dp_reads_upload_rc_begin, dp_scan.hip)
    std::shared_ptr<ReadsUpload> upload;  // (under upload_mu: a borrower's wait holds its own reference while the owner's final wait lets go)
    std::mutex upload_mu;

    // ---- reads (A1)
    uint32_t n_reads = 0;
    uint64_t total_bases = 0;
    uint64_t packed_bytes = 0;  // incl. alignment padding, excl. tail pad
    DevBuf d_packed;            // 2-bit packed reads, each read starting on a 16-byte boundary
    DevBuf d_boff;              // uint64 byte offset of each read in d_packed  [n_reads+1]
    DevBuf d_len;               // uint32 length in bases                      [n_reads]
    DevBuf d_values;            // 4^k doubles (kmerRanks) for dp_select_seeds; shared like the reads
    uint64_t n_values = 0;
    bool values_computed = false;  // the resident table came from dp_kmer_values (its histogram and total are known)
    uint64_t values_total = 0;  // sum of the k-mer counts the resident table was computed from (0: uploaded table)
    DevBuf d_qual, d_qualoff, d_hasq;  // FASTQ quality bytes of the reads (dp_quality_upload); shared like the reads
    DevBuf d_selwin, d_seltop;
    PinBuf h_seltop;
    DevBuf d_cin, d_cout;       // dp_consensus_align
    DevBuf d_cretry;            // dp_consensus_paf: windows the small LDS layout could not hold ([0] count, [1..] window numbers)
    PinBuf h_cin, h_cout;
    std::vector<uint64_t> h_boff;
    std::vector<uint32_t> h_len;

    // ---- round state
    int k = 0;
    int table_k = 0;            // k the dense tables are currently sized/zeroed for
    DevBuf d_bits;              // 4^k-bit membership table (uint32 words)
    DevBuf d_kmap;              // int32 k-mer -> seed id
    DevBuf d_seeds;             // uint32 seed k-mers of this round
    uint32_t n_seeds = 0;
    bool round_open = false;

    // ---- scan (A2+A10)
    DevBuf d_items, d_counts, d_segoff, d_segs, d_total;
    PinBuf h_counts, h_segoff, h_segs, h_total;
    PinBuf h_seeds;                   // the round's seed list as dp_round_begin received it (pinned: kernels may read it in place)
    bool seeds_uploaded = false;      // d_seeds holds it too (only the scan kernels' tables need that, see dp_seeds_ptr)
    uint64_t n_segs = 0;
    uint32_t scan_items = 0;
    DevBuf d_ignore, d_surv;          // dp_scan_reads: ignore mask; compacted survivor lists
    PinBuf h_surv;
    bool extras_staged = false;       // ... and have not been brought to d_items yet (the index step's first kernel or a fetch launch does)
    PinBuf h_extra;                   // the extra scan items (query windows) of a round, staged for the device to fetch
    PinBuf h_spack;                   // the survivor list as the compaction kernel writes it (pinned, written by the device)
    uint64_t ignore_epoch = ~0ull;
    PinBuf h_ignore;              // the flags as the device holds them (pinned: the device's copy is refreshed from here by a kernel of the stream)
    bool ignore_shadow_valid = false;
    uint64_t cached_bases = 0;
    uint32_t cached_reads = 0, cached_lo = 0, cached_hi = 0;
    uint32_t chunk_lo = 0;            // what turns d_surv's survivor entries into read ids: the scanned range's first read, or 0 once dp_allgather_survivors installed the gathered list
    int cached_top = -1, cached_k = 0;
    // the read items on the device (make_read_items_kernel) are regenerated only when what they are made from changes
    const void* items_ptr = nullptr;
    uint64_t items_epoch = ~0ull;
    uint32_t items_lo = 0, items_hi = 0, items_min = 0;
    int items_top = -1, items_k = 0;
    // pageable staging of borrowed host inputs whose copies are still queued when a call returns (dp_stage); emptied by every
    // dp_stream_sync
    std::vector<uint8_t> stage_buf;
    size_t stage_used = 0;
    size_t mcover_off = 0;                   // ints from d_manchor to the matches' covered bases (dp_match_anchors_launch)
    uint32_t cons_large_rounds = 0;          // rounds for which the large consensus layout is still launched at once (a recent round listed a window for it)
    bool cons_huge = false;                  // a window of an earlier round did not fit the large consensus layout: the huge one follows it from now on
    uint32_t cons_prev_pairs = 0;            // pairs of the previous round's chaining stage (output bound of a pending one)
    struct FindState* find_state = nullptr;  // dp_overlap.hip: the chaining stage between launch and evaluation
    bool timing_on = true;
    uint64_t round_serial = 0;

    // ---- index (A13)
    uint32_t n_seqs = 0, W = 0, SW = 0;
    // dp_index_build_chunked: the chunks were made on the device - n_seqs is then an upper bound (buffer and row sizes), the
    // exact number lives at d_nseqs[0], the chunks' {read, length, offset, inset} in d_chunk_meta
    DevBuf d_chunk_meta, d_nseqs;
    bool chunks_on_device = false;
    int scan_fetch_extras_only = 0;  // dp_scan_fetch_mode
    uint32_t last_surv_all = 0;      // survivors + extra items of the last dp_scan_reads (layout of h_surv)
    uint32_t map_stage_windows = 0;
    bool map_stage_valid = false;  // dp_map_windows_shard: the query stage of the forward pass is reused by the reverse pass
    uint32_t word_base = 0, global_n_seqs = 0;  // dp_index_set_global: this index is the words [word_base, word_base + W) of a larger one
    DevBuf d_seqrefs, d_posting, d_seedsets, d_pmeta;  // pmeta: uint32 {count,start,end,lens} per seed

    // ---- overlaps (A14..A8)
    DevBuf d_seeds_applied;     // the seed list currently written into d_bits / d_kmap (they are brought up to date lazily)
    uint32_t n_applied = 0;
    bool tables_dirty = false;
    void* d_kcounts = nullptr;  // k-mer histogram (uint32 [4^kcounts_k]) left behind by dp_kmer_values for the k-mer index build
    int kcounts_k = 0;
    std::vector<void*> retired_dev, retired_pin;  // outgrown buffers, released with the context (dev_reserve / pin_reserve)
    const int32_t* qsegs_dev = nullptr;             // the query segments and offsets of the last dp_query_stage (inside d_qsegs)
    const uint64_t* qoff_dev = nullptr;
    DevBuf d_qsegs, d_qoff, d_qsets, d_qmeta, d_cand, d_pool, d_mrec, d_ma, d_mb, d_cursor, d_sched, d_manchor;
    DevBuf d_pbase, d_pspec, d_clist, d_sa, d_sb;  // chaining stage: pair offsets, proposals, candidate lists, scratch columns
    uint32_t n_pairs = 0;                          // (query, candidate) pair slots of the last dp_find_overlaps
    uint32_t last_nq = 0, last_ni = 0;             // its queries / packed chain ints
    size_t q_pre_bytes = 0;                        // dp_query_prestage: bytes of the announced query block in h_qup (0: none)
    uint32_t q_pre_nq = 0;
    double q_pre_hf = 0;
    bool q_pre_fetched = false;                    // ... and a launch has brought it to d_qsegs
    uint32_t max_seq_seeds = 0;                    // longest indexed sequence of the last dp_index_build (the map kernel's BIG variant sizes its slices by it)
    DevBuf d_qbig;                                 // set lists of the queries that hold more posting sets than query_kernel's LDS (QBig)
    uint64_t last_sints = 0;                       // scratch ints of the last chaining stage (all pairs' columns)
    bool chains_packed = true;                     // false: the final chains sit in the scratch columns (d_sa/d_sb), see ChainArgs.pack
    uint32_t chain_open_ahead[2] = {~0u, ~0u};     // pairs still open ahead of proposal pass 1 / pass 2 in this context's previous chaining stage (~0: unknown)
    int last_k = 0;
    bool find_valid = false;                       // the device still holds that call's records
    PinBuf h_mrec, h_ma, h_mb, h_ta, h_tb, h_qm, h_qup, h_cursor, h_cand, h_cand_off, h_cand_list, h_mq, h_mt, h_moff, h_manchor, h_manout;
};

int dp_fail(dp_ctx* ctx, int code, const char* what, hipError_t e = hipSuccess);
// Waits for everything queued on the context's stream.  The host thread blocks on an interrupt (several executor
// contexts share few host cores); DP_SPIN_SYNC=1 restores the runtime's polling wait.
hipError_t dp_stream_sync(dp_ctx* ctx);
int dev_reserve(dp_ctx* ctx, DevBuf& b, size_t bytes, bool keep = false);
int pin_reserve(dp_ctx* ctx, PinBuf& b, size_t bytes);
// The round's seed list for a kernel that reads every seed once or twice (the k-mer index walk): the device copy if one was
// made, else the pinned host block read in place over the link - 80 KB per round that then never travel as a copy (as a
// pageable hipMemcpyAsync they cost the calling thread 80 us per round inside the runtime, HISTORY.md 5.3).
static inline const uint32_t* dp_seeds_ptr(const dp_ctx* ctx) {
    return (const uint32_t*)(ctx->seeds_uploaded ? ctx->d_seeds.p : ctx->h_seeds.p);
}
// A caller's borrowed buffer copied into memory of the context that stays untouched until the context's next dp_stream_sync,
// so that the H2D copy out of it can still be queued when the call returns.  PAGEABLE on purpose: the runtime stages small
// pageable sources itself and copies them with a blit kernel on the stream's own queue, which measured faster under eight
// concurrent slots than a pinned source (SDMA engines); should it pin the pages and copy later instead, they are still ours.
const void* dp_stage(dp_ctx* ctx, const void* src, size_t bytes);
// hipMalloc / hipFree for the library's large device blocks (k-mer index and its build buffers, value tables: hundreds of MB
// to tens of GB).  Blocks of 32 MiB and more (round 5: 4 KiB and more) that are freed stay in a process-wide cache and satisfy later requests of about
// their size: releasing and re-acquiring gigabytes from the driver costs 0.3-0.6 s every now and then (measured: a 400 MB
// hipMalloc of 622 ms at the start of a job).  The cache is dropped when an allocation fails and by dp_dev_trim().
hipError_t dp_dev_malloc(void** p, size_t bytes);
hipError_t dp_dev_free(void* p);
void dp_dev_trim();
// Round 5: the cache takes blocks from 4 KiB on, and a second one of the same shape holds the contexts' pinned host buffers
// (dp_pin_malloc / dp_pin_free) - a context that goes parks its blocks instead of paying a hipFree / hipHostFree for each; what
// stays parked is capped (DP_DEV_CACHE_MB / DP_PIN_CACHE_MB) and dp_release_device_caches() (downpore_hip.h) empties both
hipError_t dp_pin_malloc(void** p, size_t bytes);
void dp_pin_free(void* p);
size_t dp_dev_cached_bytes();  // device memory parked in that cache (it counts as free for the library's own capacity checks)
// kernel timing events of the per-round calls: recorded while ctx->timing_on (DP_KERNEL_TIMING, see dp_round_begin)
inline hipError_t dp_mark(dp_ctx* ctx, int i) { return ctx->timing_on ? hipEventRecord(ctx->ev[i], ctx->stream) : hipSuccess; }
inline float dp_elapsed(dp_ctx* ctx, int a, int b) {
    float ms = 0;
    if (ctx->timing_on && hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) ms = 0;
    return ms;
}
// up to four device regions (8-byte aligned, sizes rounded up to 8 bytes) set to zero by ONE launch on the context's stream
struct dp_zero_region {
    void* p;
    size_t bytes;
};
int dp_zero_regions(dp_ctx* ctx, const dp_zero_region* r, int n);
// The same launch also fetches up to two blocks from PINNED host memory into device buffers (8-byte words; both sides need
// slack up to the next multiple of 8): an upload done by a kernel of the stream itself instead of a copy handed to the
// runtime - copies, not kernels, are what the runtime makes expensive with several rounds in flight (HISTORY.md 5.3).
struct dp_fetch_region {
    void* dst;        // device
    const void* src;  // pinned host (hipHostMalloc)
    size_t bytes;
};
int dp_zero_fetch_regions(dp_ctx* ctx, const dp_zero_region* z, int nz, const dp_fetch_region* f, int nf);  // pinned copy of a caller buffer, valid until the next dp_stream_sync

#define DP_HIP(call)                                                          \
    do {                                                                      \
        hipError_t _e = (call);                                               \
        if (_e != hipSuccess) return dp_fail(ctx, DP_ERR_HIP, #call, _e);     \
    } while (0)

// resident k-mer position index (dp_kindex.hip)
struct dp_kindex_fast {   // the count walk's short cut for views served whole (dp_kindex.hip, KxRec)
    const uint8_t* ign;   // device: one byte per read, != 0 = ignored (null: the walk reads the items)
    uint32_t qlo, qspan;  // reads that may carry extra items
};
struct dp_kindex_oneshot {  // the index step of a round launched in one go, sized from guesses (dp_kindex_count)
    uint64_t hits_guess;    // seed occurrences expected (sizes the record shards)
    uint32_t surv_guess;    // survivors expected (grid of the sort pass)
    uint32_t sort_cap;      // 256 / 1024 / 4096: hits of the largest survivor the sort pass is launched for
    uint32_t min_seeds;     // chunkWorker's filter for read items (extra items carry their own)
    int32_t* d_segs;        // segment buffer, seg_cap ints
    uint64_t seg_cap;
    int32_t* host_segs;     // pinned mirror for the extra items' segments (or null)
    uint32_t* done_flag;    // pinned: the chunk stage launched behind the step (dp_index_prechain) stores done_seq here as it starts
    uint32_t done_seq;
};
int dp_histogram_device(dp_ctx* ctx, int k, uint32_t* d_counts);  // dp_scan.hip
int dp_kindex_ensure(dp_ctx* ctx, int k);
struct dp_comm;
int dp_comm_allgather_ranges(dp_comm* c, dp_ctx* ctx, void* dst, size_t elem, const uint64_t* first, const void* src);  // dp_comm.hip
bool dp_comm_is_rccl(const dp_comm* c);  // dp_comm.hip
// DP_DEBUG=a,b,c: diagnosis output of the named parts (no change of behaviour).  DP_TUNE=key=value,...: the numbers experiments vary
// (grids, pools, polls; the defaults are what the measurements chose).  Both are read once per process.  (dp_scan.hip)
bool dp_debug(const char* what);
long dp_tune(const char* key, long dflt);
// A rank's share of a k-mer position index built by several ranks (round 5): filled by dp_kindex_build_sorted from the first-digit
// counts every rank computes alike - rank q sorts the k-mers [digit_first[q] << kmer_shift, digit_first[q + 1] << kmer_shift), which are
// the entries [entry_first[q], entry_first[q + 1]) of the whole index; the offsets it writes are relative to its own first entry.
struct dp_kindex_shard {
    int rank = 0, n_ranks = 1;
    std::vector<uint32_t> digit_first;  // [n_ranks + 1]
    std::vector<uint64_t> entry_first;  // [n_ranks + 1]
    int kmer_shift = 0;
    uint64_t total = 0;
};
int dp_kindex_build_sorted(dp_ctx* ctx, dp_ctx* ow, int k, uint32_t* d_counts, uint64_t* d_off, void** d_pos_out, void** d_pos_hi_out,
                           int* fmt_out, int* pbits_out, uint64_t* n_pos_out, float* ms_out, dp_kindex_shard* shard = nullptr);  // dp_kbuild.hip
void dp_kindex_free(dp_ctx* ctx);
int dp_kindex_count(dp_ctx* ctx, int k, const dp_scan_item* d_items, uint32_t lo, uint32_t hi, uint32_t n_read_items, uint32_t n_extra,
                    uint32_t* d_counts, uint64_t* d_segoff, uint32_t* s_item, uint32_t* s_count, uint64_t* s_off, uint4* s_pack, uint64_t* d_totals,
                    unsigned long long* host_totals /* pinned, 64 bytes: the totals are stored there by the last kernel (or null) */,
                    const struct dp_kindex_oneshot* one = nullptr, const struct dp_kindex_fast* fast = nullptr);
int dp_kindex_refill(dp_ctx* ctx, const dp_scan_item* d_items, uint32_t n_read_items, uint32_t n_extra);
int dp_index_prechain_launch(dp_ctx* ctx, const uint64_t* scan_totals, uint32_t n_extra, uint64_t seg_cap, uint32_t* done_flag,
                             uint32_t done_seq);  // dp_overlap.hip
void dp_index_prechain_cancel(dp_ctx* ctx);
int dp_kindex_write(dp_ctx* ctx, int k, const dp_scan_item* d_items, uint32_t lo, uint32_t hi, uint32_t n_read_items, uint32_t n_extra,
                    const uint32_t* d_sel, uint32_t n_sel, uint32_t max_count, const uint32_t* d_counts, const uint64_t* d_segoff,
                    const uint64_t* d_totals, int32_t* d_segs, int32_t* host_segs = nullptr);

// kernels implemented in other translation units
#define DP_NO_ANCHOR ((int32_t)0x80000000)  // match_anchor_kernel: the chain's indices are not inside the target (anchors may be negative)
int dp_match_anchors_launch(dp_ctx* ctx, const dp_fetch_region* fetch = nullptr, uint32_t* zero_word = nullptr);  // dp_overlap.hip: GetSeedOffset / GetSeedOffsetFromEnd anchors of the last chaining stage's records
int dp_index_build_impl(dp_ctx* ctx, const dp_seq_ref* seqs, uint32_t n_seqs);
int dp_find_overlaps_impl(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t nq, double hf, int k,
                          uint32_t max_query_len, int want_candidates, dp_match_batch* out);
// phase 2 = both strands (whole index); 0 / 1 = forward / reverse windows of a shard, thresholds through thr_io[nw]
int dp_map_windows_impl(dp_ctx* ctx, const int32_t* w_segs, const uint64_t* w_off, const uint32_t* w_len, uint32_t nw, int k,
                        dp_chain_batch* out, int phase = 2, int32_t* thr_io = nullptr);
void dp_find_state_free(dp_ctx* ctx);
bool dp_find_pending(const dp_ctx* ctx);
// where the chaining stage's final chains are: MRec.off indexes these two arrays (a side, b side)
static inline const void* dp_chain_a(const dp_ctx* ctx) { return ctx->chains_packed ? ctx->d_ma.p : ctx->d_sa.p; }
static inline const void* dp_chain_b(const dp_ctx* ctx) { return ctx->chains_packed ? ctx->d_mb.p : ctx->d_sb.p; }
uint32_t dp_find_pair_cap(const dp_ctx* ctx);
int dp_find_complete(dp_ctx* ctx, bool* reran);
void dp_find_stats(const dp_ctx* ctx, double* query_ms, double* chain_ms, uint64_t* query_bytes, uint64_t* chain_bytes);
int dp_query_stage(dp_ctx* ctx, const int32_t* q_segs, const uint64_t* q_off, uint32_t nq, double hf, uint32_t** d_qmeta_out,
                   uint64_t** d_words_out, int32_t** d_mc_out, uint32_t* mc_n_out, uint32_t** d_qcnt_out = nullptr);

// ---- device helpers ---------------------------------------------------------------------------------------
// An entry of the resident k-mer position index (dp_kindex.hip; written by dp_kbuild.hip): fmt 8 = read << 32 | position in a
// 64-bit word; 4 = read << pbits | position in 32 bits; 5 = the same in 40 bits, the top byte in a stream of its own.
struct KxPos {
    const void* lo;
    const uint8_t* hi;
    uint32_t fmt, pbits;
};
#ifdef __HIPCC__
__device__ __forceinline__ uint64_t kx_entry(const KxPos P, uint64_t i) {  // -> read << 32 | position
    if (P.fmt == 8) return ((const uint64_t*)P.lo)[i];
    uint64_t v = ((const uint32_t*)P.lo)[i];
    if (P.fmt == 5) v |= (uint64_t)P.hi[i] << 32;
    return ((v >> P.pbits) << 32) | (v & (((uint64_t)1 << P.pbits) - 1));
}
// Four CONSECUTIVE entries i .. i + 3 (the caller masks what lies beyond its stretch: the arrays end in 64 bytes of slack): one 16-byte
// load of the low words - and one 4-byte window of the byte stream, cut out of two aligned words - instead of four (eight) loads.
struct __attribute__((packed, aligned(4))) KxU4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ void kx_entry4(const KxPos P, uint64_t i, uint64_t e[4]) {
    if (P.fmt == 8) {
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = ((const uint64_t*)P.lo)[i + u];
        return;
    }
    const KxU4 q = *(const KxU4*)((const uint32_t*)P.lo + i);
    uint32_t hi4 = 0;
    if (P.fmt == 5) {
        const uintptr_t a = (uintptr_t)(P.hi + i);
        const uint32_t* w = (const uint32_t*)(a & ~(uintptr_t)3);
        const uint32_t w0 = w[0], w1 = w[1];
        hi4 = __builtin_amdgcn_alignbyte(w1, w0, (uint32_t)(a & 3));  // bytes a .. a + 3
    }
    const uint32_t lo4[4] = {q.x, q.y, q.z, q.w};
    const uint64_t pm = ((uint64_t)1 << P.pbits) - 1;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint64_t v = (uint64_t)lo4[u] | ((uint64_t)((hi4 >> (8 * u)) & 0xffu) << 32);
        e[u] = ((v >> P.pbits) << 32) | (v & pm);
    }
}
__device__ __forceinline__ int dp_lane() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_incl_sum(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (dp_lane() >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_incl_max(int v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (dp_lane() >= d) v = max(v, t);
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
// The same sum over the data-parallel-primitive path of the vector unit (six VALU instructions and a read-out instead of six
// trips through the LDS crossbar, which is what __shfl_xor compiles to: ~60 cycles against ~400).  Needs every lane of the wave
// active (rows take their neighbours' values whether those lanes are live or not).
// Inclusive prefix sum over the 64 lanes on the same path (the classic row_shr / row_bcast sequence).  Every lane active.
__device__ __forceinline__ int wave_incl_sum_dpp(int v) {
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, false);  // row_shr:3: sums of up to four neighbours inside a row
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xe, false);  // row_shr:4 into lanes 4-15 of each row
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xc, false);  // row_shr:8 into lanes 8-15: inclusive sums inside rows of 16
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);   // quad_perm:[1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false);   // quad_perm:[2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8: every lane of a row of 16 holds the row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3: lane 63 holds the total
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_dpp(int v) {  // (as wave_sum_dpp; every lane active)
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0xb1, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x4e, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
// Exclusive prefix of tile `tile` from the status words of its predecessors (flag << 62 | value; flag 1 = a tile's own value, 2 = its
// inclusive prefix), looked back at by a whole wave at once: lane l reads tile - 1 - l.  With a few dozen tiles that all start
// together - the per-round scans here - a single thread walking back hop by hop (a dependent trip to memory each) WAS the kernel:
// tile 24 made 24 hops.  The caller's lane 0 has published the tile's own value before; every lane of the wave takes part.
__device__ __forceinline__ unsigned long long dp_wave_lookback(const unsigned long long* status, uint32_t tile, int lane) {
    unsigned long long excl = 0;
    for (long long base = (long long)tile - 1; base >= 0; base -= 64) {
        const long long t = base - lane;
        unsigned long long v = 0;
        if (t >= 0) {
            do {
                v = __hip_atomic_load(&status[t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            } while ((v >> 62) == 0);
        }
        const unsigned long long done = __ballot(t >= 0 && (v >> 62) == 2);
        const int stop = done ? __builtin_ctzll(done) : 63;
        const unsigned long long x = (t >= 0 && lane <= stop) ? (v & ((1ull << 62) - 1)) : 0ull;
        // (a 62-bit sum over 64 lanes as three sums of 25 + 25 + 12 bits)
        const unsigned long long lo = (unsigned)wave_sum_dpp((int)(x & 0x1ffffffull)), mid = (unsigned)wave_sum_dpp((int)((x >> 25) & 0x1ffffffull)),
                                 hi = (unsigned)wave_sum_dpp((int)(x >> 50));
        excl += lo + (mid << 25) + (hi << 50);
        if (done) break;
    }
    return excl;
}
#endif
