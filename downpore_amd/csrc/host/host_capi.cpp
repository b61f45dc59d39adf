// C entry points of libdownpore_host.so: lets Python (bench.py, tests) drive the product's host pipeline round by
// round, interposing the multi-GPU survivor exchange between the local scan and the index build.
#include <unistd.h>

#include <algorithm>
#include <cstring>

#include "downpore_host.h"
#include "host_util.hpp"

using namespace dph;

namespace {
struct ReadsH {
    ReadSet set;
};
struct OverlapH {
    OverlapRun run;
    dp_ctx* ctx = nullptr;
    dp_comm* comm = nullptr;
    dp_ctx* xctx = nullptr;           // round-parallel mode: the context (own stream) the result exchange runs on
    std::vector<dp_comm*> slotComms;  // scan-shard with executor slots: one communicator per slot
    ReadSet* reads = nullptr;
    int textRoot = -1;                // dph_overlap_text_root: >= 0 = a superstep gathers the rounds' PAF text to that rank only
    bool keepText = true;             // dph_overlap_keep_text(0): gathered rounds of other ranks are committed without their PAF text
    double tCtx = 0, tUpload = 0, tInit = 0;
    std::string err;
    // PAF of every committed round so far.  Kept as one chunk per commit and joined only when somebody asks for the
    // whole text: appending to one growing std::string re-copied up to 128 MB at every doubling (20-30 ms stalls of the
    // committing thread at rounds ~80, ~165, ~330 of a config-2 job).
    // (round 4: a step's text is MOVED in - it is the buffer a formatter thread filled, never copied, never freed before the job
    // ends: copying 400 KB per round into fresh memory and freeing the original was an mmap + munmap pair per round on the
    // committing thread)
    std::vector<std::string> pafChunks;
    std::string allPafJoined;
    std::string lastStepCopy;  // text of the last step once allPaf() has folded the chunks away
    bool lastStepHasText = false, lastStepInChunks = false;
    void addPaf(std::string& s) {  // takes s's buffer; s is empty afterwards
        lastStepHasText = !s.empty();
        lastStepInChunks = lastStepHasText;
        if (lastStepHasText) {
            pafChunks.push_back(std::move(s));
            s.clear();
        }
    }
    const std::string& lastStepText() const {
        static const std::string none;
        if (!lastStepHasText) return none;
        return lastStepInChunks ? pafChunks.back() : lastStepCopy;
    }
    const std::string& allPaf() {
        if (!pafChunks.empty()) {
            size_t n = allPafJoined.size();
            for (const std::string& c : pafChunks) n += c.size();
            allPafJoined.reserve(n);
            for (const std::string& c : pafChunks) allPafJoined += c;
            if (lastStepHasText && lastStepInChunks) {
                lastStepCopy = pafChunks.back();
                lastStepInChunks = false;
            }
            pafChunks.clear();
        }
        return allPafJoined;
    }
};
thread_local std::string g_err;
}  // namespace

extern "C" {

const char* dph_last_error(void* h) { return h ? ((OverlapH*)h)->err.c_str() : g_err.c_str(); }

void* dph_reads_from_arrays(const char* bases, const int64_t* off, int64_t n, int64_t minLen, int himem) {
    return new ReadsH{ReadSet::fromArrays(bases, off, (size_t)n, minLen, himem != 0)};
}
// bases + raw FASTQ quality characters at the same offsets: the set behaves as if it had been read from a FASTQ file
void* dph_reads_from_arrays_q(const char* bases, const char* quals, const int64_t* off, int64_t n, int64_t minLen, int himem) {
    return new ReadsH{ReadSet::fromArrays(bases, off, (size_t)n, minLen, himem != 0, quals)};
}
void* dph_reads_from_fasta(const char* path, int64_t minLen, int himem) {
    ReadsH* h = new ReadsH();
    if (!ReadSet::fromFile(path, minLen, himem != 0, h->set, g_err)) {
        delete h;
        return nullptr;
    }
    return h;
}
void dph_reads_free(void* h) { delete (ReadsH*)h; }
// "name\tACGT...\n" per read, bases spelled from their 2-bit codes: lets tests compare this reader with the oracle's
const char* dph_reads_dump(void* h, int64_t* n) {
    static thread_local std::string out;
    out.clear();
    ReadSet& s = ((ReadsH*)h)->set;
    for (size_t i = 0; i < s.size(); i++) {
        out += s.names[i];
        out += '\t';
        const char* b = s.seq(i);
        for (i64 j = 0; j < s.length(i); j++) out += "ACGT"[baseCode((unsigned char)b[j])];
        if (const uint8_t* q = s.quality(i)) {  // FASTQ: the stored quality bytes (phred - 33), two hex digits each
            out += '\t';
            for (i64 j = 0; j < s.length(i); j++) {
                out += "0123456789abcdef"[q[j] >> 4];
                out += "0123456789abcdef"[q[j] & 15];
            }
        }
        out += '\n';
    }
    *n = (int64_t)out.size();
    return out.data();
}
int64_t dph_reads_count(void* h) { return (int64_t)((ReadsH*)h)->set.size(); }
void dph_reads_get_ignore(void* h, uint8_t* out) {
    auto& ig = ((ReadsH*)h)->set.ignore;
    memcpy(out, ig.data(), ig.size());
}
void dph_reads_reset_ignore(void* h) {
    auto& ig = ((ReadsH*)h)->set.ignore;
    std::fill(ig.begin(), ig.end(), 0);
}
int64_t dph_reads_total_bases(void* h) { return (int64_t)((ReadsH*)h)->set.bases.size(); }

static OverlapParams paramsFrom(const int64_t* params, double minHits) {
    OverlapParams p;
    p.overlapSize = params[0];
    p.k = (int)params[1];
    p.numSeeds = (int)params[2];
    p.seedBatchSize = params[3];
    p.chunkSize = params[4];
    p.queryBatchSize = params[5];
    p.himem = (params[6] & 1) != 0;
    p.queryType = (int)(params[6] >> 8) ? (int)(params[6] >> 8) : 1;  // bits 8.. of the himem word: overlap.Query* flags
    p.minHits = minHits;
    return p;
}

// The command in two halves, so that a caller can keep the reads resident and run (or time) whole jobs on them:
// dph_overlap_open = device context + the reads uploaded and packed; dph_overlap_init = everything `downpore overlap`
// does between "Counting all k-mers" and its first round (value table, k-mer position index, executor slots, planner);
// dph_overlap_reset = back to the state after dph_overlap_open (the job's resident tables are released).
void* dph_overlap_open(void* reads, int device) {
    OverlapH* h = new OverlapH();
    ReadSet& rs = ((ReadsH*)reads)->set;
    h->reads = &rs;
    const double tc0 = now();
    int rc = dp_ctx_create(device, &h->ctx);
    if (rc != 0) {
        g_err = dp_last_error(nullptr);
        delete h;
        return nullptr;
    }
    const double tc1 = now();
    rc = dp_reads_upload(h->ctx, (const uint8_t*)rs.bases.data(), rs.off.data(), (uint32_t)rs.size());
    h->tCtx = tc1 - tc0;
    h->tUpload = now() - tc1;
    if (g_prof.on) fprintf(stderr, "[setup] context %.1f ms, upload + pack %.1f ms\n", 1e3 * h->tCtx, 1e3 * h->tUpload);
    if (rc == 0 && rs.isFastq && !rs.qual.empty()) {
        // FASTQ: the selection kernels weight values by the quality bytes (seeds.go:99-101).  Once per read set, here - the bytes
        // belong to the reads, not to a job, and dp_quality_upload refuses a context that already lends its reads to others
        // (the contexts a reset() keeps for the handle's next job do)
        rc = dp_quality_upload(h->ctx, (const uint8_t*)rs.qual.data(), rs.off.data(), rs.hasQual.data(), (uint32_t)rs.size());
        if (rc == 0) h->run.qualityCtx = h->ctx;
    }
    if (rc != 0) {
        g_err = dp_last_error(h->ctx);
        dp_ctx_destroy(h->ctx);
        delete h;
        return nullptr;
    }
    return h;
}
// params: overlapSize,k,numSeeds,seedBatchSize,chunkSize,queryBatchSize,himem,nSlots
int dph_overlap_init(void* hh, const int64_t* params, double minHits, const double* valuesOrNull) {
    OverlapH* h = (OverlapH*)hh;
    const double t0 = now();
    int rc = h->run.init(h->ctx, h->reads, paramsFrom(params, minHits), valuesOrNull, (int)params[7]);
    h->tInit = now() - t0;
    if (rc != 0) h->err = h->run.error.empty() ? dp_last_error(h->ctx) : h->run.error;
    return rc;
}
int dph_overlap_reset(void* hh) {
    OverlapH* h = (OverlapH*)hh;
    h->run.shutdown(true);  // (the slots' and the planner's contexts serve the handle's next job)
    TextJobBuffers::giveTexts(h->pafChunks);  // (the rounds' text buffers serve the next job's rounds)
    h->pafChunks.clear();
    h->allPafJoined.clear();
    h->lastStepCopy.clear();
    h->lastStepHasText = h->lastStepInChunks = false;
    std::fill(h->reads->ignore.begin(), h->reads->ignore.end(), 0);
    int rc = dp_scan_release(h->ctx);
    if (rc != 0) h->err = dp_last_error(h->ctx);
    return rc;
}
// out: seconds spent creating the context, uploading + packing the reads, and in the last dph_overlap_init
void dph_overlap_setup_times(void* hh, double* out) {
    OverlapH* h = (OverlapH*)hh;
    out[0] = h->tCtx;
    out[1] = h->tUpload;
    out[2] = h->tInit;
}
void* dph_overlap_create(void* reads, int device, const int64_t* params, double minHits, const double* valuesOrNull) {
    void* hh = dph_overlap_open(reads, device);
    if (!hh) return nullptr;
    if (dph_overlap_init(hh, params, minHits, valuesOrNull) != 0) {
        OverlapH* h = (OverlapH*)hh;
        g_err = h->err;
        h->run.shutdown();
        dp_ctx_destroy(h->ctx);
        delete h;
        return nullptr;
    }
    return hh;
}
void dph_overlap_destroy(void* hh) {
    OverlapH* h = (OverlapH*)hh;
    if (!h) return;
    h->run.shutdown();
    if (h->ctx) dp_kindex_set_comm(h->ctx, nullptr);
    if (h->comm) dp_comm_destroy(h->comm);
    for (dp_comm* c : h->slotComms) dp_comm_destroy(c);
    if (h->xctx) dp_ctx_destroy(h->xctx);
    dp_ctx_destroy(h->ctx);
    delete h;
}
void dph_overlap_set_shard(void* hh, int64_t lo, int64_t hi) {
    OverlapH* h = (OverlapH*)hh;
    h->run.shardLo = (size_t)lo;
    h->run.shardHi = (size_t)hi;
}
const double* dph_overlap_values(void* hh, int64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    const double* v = h->run.fullValues();
    *n = (int64_t)h->run.values.size();
    return v;
}
// 1 = a round was prepared and the local shard scanned; 0 = no more rounds; <0 error
int dph_overlap_round_scan(void* hh) {
    OverlapH* h = (OverlapH*)hh;
    int rc = h->run.roundPrepareAndScan();
    if (rc < 0) h->err = h->run.error;
    return rc;
}
void dph_overlap_local(void* hh, const uint32_t** read, const uint32_t** nseeds, const uint64_t** segoff,
                       const int32_t** segs, uint64_t* n, uint64_t* nsegs) {
    Survivors& s = ((OverlapH*)hh)->run.local();
    *read = s.read.data();
    *nseeds = s.n_seeds.data();
    *segoff = s.seg_off.data();
    *segs = s.segData();
    *n = s.read.size();
    *nsegs = s.segCount();
}
// survivors == all ranks' lists concatenated in rank order (read ids ascending); read == NULL: use the local list
int dph_overlap_round_finish(void* hh, const uint32_t* read, const uint32_t* nseeds, const int32_t* segs, uint64_t n) {
    OverlapH* h = (OverlapH*)hh;
    int rc;
    if (!read) {
        rc = h->run.roundFinish(h->run.local());
    } else {
        Survivors all;
        all.read.assign(read, read + n);
        all.n_seeds.assign(nseeds, nseeds + n);
        all.seg_off.assign(1, 0);
        uint64_t pos = 0;
        for (uint64_t i = 0; i < n; i++) {
            pos += 2ull * nseeds[i] + 1;
            all.seg_off.push_back(pos);
        }
        all.segsView = segs;  // the caller's gathered array outlives the call
        all.segsViewLen = pos;
        rc = h->run.roundFinish(all);
    }
    if (rc < 0) h->err = h->run.error;
    else h->addPaf(h->run.paf);
    return rc;
}
// ---- multi-GPU with the exchange inside the library (dp_comm): RCCL across processes, or in-process peers
int dph_comm_unique_id(uint8_t* id128) { return dp_comm_unique_id(id128); }
int dph_overlap_comm_init(void* hh, int nRanks, int rank, const uint8_t* id128) {
    OverlapH* h = (OverlapH*)hh;
    dp_kindex_set_comm(h->ctx, nullptr);
    if (h->comm) dp_comm_destroy(h->comm);
    h->comm = nullptr;
    int rc = dp_comm_init(h->ctx, nRanks, rank, id128, &h->comm);
    if (rc != 0) h->err = dp_last_error(h->ctx);
    h->run.comm = h->comm;
    // (round 5: with a communicator in place the job's k-mer position index is built in shares, one per rank, and all-gathered over it)
    if (rc == 0) dp_kindex_set_comm(h->ctx, h->comm);
    return rc;
}
int dph_overlap_comm_init_local(void** handles, int n) {
    std::vector<dp_ctx*> ctxs;
    for (int i = 0; i < n; i++) ctxs.push_back(((OverlapH*)handles[i])->ctx);
    std::vector<dp_comm*> comms((size_t)n, nullptr);
    int rc = dp_comm_init_local(ctxs.data(), n, comms.data());
    if (rc != 0) return rc;
    for (int i = 0; i < n; i++) {
        OverlapH* h = (OverlapH*)handles[i];
        dp_kindex_set_comm(h->ctx, nullptr);
        if (h->comm) dp_comm_destroy(h->comm);
        h->comm = comms[(size_t)i];
        h->run.comm = h->comm;
        dp_kindex_set_comm(h->ctx, h->comm);
    }
    return 0;
}
// One communicator per executor slot (ids = nSlots x 128 bytes, made by rank 0 with dph_comm_unique_id and handed to every
// rank in the same order).  Call before dph_overlap_init: the slots pick their communicators up when they are created.
int dph_overlap_comm_init_slots(void* hh, int nRanks, int rank, const uint8_t* ids, int nSlots) {
    OverlapH* h = (OverlapH*)hh;
    if (!h->comm) dp_kindex_set_comm(h->ctx, nullptr);
    for (dp_comm* c : h->slotComms) dp_comm_destroy(c);
    h->slotComms.clear();
    for (int i = 0; i < nSlots; i++) {
        dp_comm* c = nullptr;
        int rc = dp_comm_init(h->ctx, nRanks, rank, ids + (size_t)i * 128, &c);
        if (rc != 0) {
            h->err = dp_last_error(h->ctx);
            return rc;
        }
        h->slotComms.push_back(c);
    }
    h->run.slotComms = h->slotComms;
    for (size_t i = 0; i < h->run.slots.size() && i < h->slotComms.size(); i++) h->run.slots[i]->comm = h->slotComms[i];
    if (!h->comm && !h->slotComms.empty()) dp_kindex_set_comm(h->ctx, h->slotComms[0]);  // (the index is built in shares over slot 0's)
    return 0;
}
int dph_overlap_comm_init_local_slots(void** handles, int n, int nSlots) {
    std::vector<dp_ctx*> ctxs;
    for (int i = 0; i < n; i++) ctxs.push_back(((OverlapH*)handles[i])->ctx);
    for (int i = 0; i < n; i++) {
        OverlapH* h = (OverlapH*)handles[i];
        if (!h->comm) dp_kindex_set_comm(h->ctx, nullptr);
        for (dp_comm* c : h->slotComms) dp_comm_destroy(c);
        h->slotComms.clear();
    }
    for (int s = 0; s < nSlots; s++) {
        std::vector<dp_comm*> comms((size_t)n, nullptr);
        int rc = dp_comm_init_local(ctxs.data(), n, comms.data());
        if (rc != 0) return rc;
        for (int i = 0; i < n; i++) ((OverlapH*)handles[i])->slotComms.push_back(comms[(size_t)i]);
    }
    for (int i = 0; i < n; i++) {
        OverlapH* h = (OverlapH*)handles[i];
        h->run.slotComms = h->slotComms;
        for (size_t j = 0; j < h->run.slots.size() && j < h->slotComms.size(); j++) h->run.slots[j]->comm = h->slotComms[j];
        if (!h->comm && !h->slotComms.empty()) dp_kindex_set_comm(h->ctx, h->slotComms[0]);
    }
    return 0;
}
// the next slots.size() rounds of this rank, executed concurrently and committed in order (collective: every rank calls it,
// from its own thread / process): rounds committed, 0 = finished, <0 error
int dph_overlap_rounds_sharded(void* hh) {
    OverlapH* h = (OverlapH*)hh;
    int rc = h->run.roundsShardedBatch();
    if (rc < 0) h->err = h->run.error;
    else h->addPaf(h->run.paf);
    return rc;
}
// one whole round of this rank (collective: every rank calls it, from its own thread / process): 1 ran, 0 finished
int dph_overlap_round_sharded(void* hh) {
    OverlapH* h = (OverlapH*)hh;
    int rc = h->run.roundSharded();
    if (rc < 0) h->err = h->run.error;
    else h->addPaf(h->run.paf);
    return rc;
}

const char* dph_overlap_round_paf(void* hh, int64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    const std::string& t = h->lastStepText();
    *n = (int64_t)t.size();
    return t.data();
}
const char* dph_overlap_all_paf(void* hh, int64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    const std::string& all = h->allPaf();
    *n = (int64_t)all.size();
    return all.data();
}
const char* dph_overlap_errtext(void* hh, int64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    *n = (int64_t)h->run.errText.size();
    return h->run.errText.data();
}
// out[0..] t_prepare,t_scan,t_index,t_query,t_consensus,k_scan_ms,k_query_ms,k_chain_ms,scan_bases,scan_items,
// scan_bytes,query_bytes,n_queries,n_indexed,n_hits,n_matches,n_paf,n_seeds,round,badBack,emptyMatch,k_count_ms,
// k_write_ms,count_bytes,k_cons_ms,idx_rounds,idx_hits,chain_bytes
static void statsOut(OverlapH* h, const RoundStats& s, double* out);
void dph_overlap_stats(void* hh, double* out) { statsOut((OverlapH*)hh, ((OverlapH*)hh)->run.last, out); }
// the same fields summed over every committed round of the job
void dph_overlap_stats_total(void* hh, double* out) { statsOut((OverlapH*)hh, ((OverlapH*)hh)->run.total, out); }
static void statsOut(OverlapH* h, const RoundStats& s, double* out) {
    double v[] = {s.t_prepare, s.t_scan, s.t_index, s.t_query, s.t_consensus, s.k_scan_ms, s.k_query_ms, s.k_chain_ms,
                  (double)s.scan_bases, (double)s.scan_items, (double)s.scan_bytes, (double)s.query_bytes, (double)s.n_queries,
                  (double)s.n_indexed, (double)s.n_hits, (double)s.n_matches, (double)s.n_paf, (double)s.n_seeds,
                  (double)h->run.round, (double)h->run.badBack, (double)h->run.emptyMatch, s.k_count_ms, s.k_write_ms,
                  (double)s.count_bytes, s.k_cons_ms, (double)s.idx_rounds, (double)s.idx_hits, (double)s.chain_bytes, (double)s.timed_rounds, (double)s.gang_members, (double)s.cons_bytes, s.k_index_ms};
    memcpy(out, v, sizeof v);
}
void* dph_overlap_ctx(void* hh) { return ((OverlapH*)hh)->ctx; }

// ---- whole round on this process (planner thread prefetches the next query batches): 1 ran, 0 finished, <0 error
int dph_overlap_step(void* hh) {
    OverlapH* h = (OverlapH*)hh;
    int rc = h->run.step();
    if (rc < 0) h->err = h->run.error;
    else {
        const double t0 = now();
        h->addPaf(h->run.paf);
        g_prof.commitKeepUs += (long long)((now() - t0) * 1e6);
    }
    return rc;
}
// step() stops committing at round n (tests that compare the first n rounds with a fixture); -1 lifts the limit
void dph_overlap_set_round_limit(void* hh, int64_t n) { ((OverlapH*)hh)->run.roundLimit = n; }
int64_t dph_overlap_round(void* hh) { return ((OverlapH*)hh)->run.round; }
void dph_profile_print() { profilePrint(); }
// Buffers the library keeps between jobs so that a handle's next job need not map and zero-fill them again (PAF text strings and
// record arrays: up to 512 MB; window-cache chunks: up to 16 x 11.5 MB; the map command's staging block: ~400 MB at config 3)
// go back to the allocator.  Safe at any time (a running job simply allocates again); returns the bytes released.
int64_t dph_release_caches() {
    return (int64_t)(TextJobBuffers::releaseAll() + WindowCache::releaseSpares() + releaseMapStaging()) + dp_release_device_caches();  // (round 5: + the device library's parked blocks)
}
// process-wide planner counters (tests): 0 plans computed, 1 computed plans thrown away (stale flags, or started from a wrong
// guess of where the plan before them ends), 2 finished plans erased by a commit's flags
int64_t dph_planner_counter(int which) {
    // 0 plans computed, 1 thrown away, 2 erased by flags, 3 rounds executed, 4 rejected at the commit, 5 committed, 6 microseconds in
    // plan computes (wall), 7 slots' microseconds waiting for a plan (planComputes etc. count since the process started)
    switch (which) {
        case 0: return g_prof.planComputes.load();
        case 1: return g_prof.planDiscarded.load();
        case 2: return g_prof.planErased.load();
        case 3: return g_prof.executed.load();
        case 4: return g_prof.rejected.load();
        case 5: return g_prof.committed.load();
        case 6: return g_prof.planUs.load();
        case 7: return g_prof.getWaitUs.load();
        case 8: return g_prof.commitWaitUs.load();   // the committing thread (whoever calls dph_overlap_step): waiting for the next round in order,
        case 9: return g_prof.commitTextUs.load();   // ... joining its text,
        case 10: return g_prof.commitStateUs.load();  // ... flags + planner bookkeeping,
        case 11: return g_prof.commitKeepUs.load();   // ... keeping the step's text for dph_overlap_all_paf
        case 12: return g_prof.formatUs.load();       // formatter threads: time in TextJob::format, all threads together
        case 13: return g_prof.textWaitUs.load();
        case 14: return g_prof.planLanes.load();     // lanes of the planner that was given lanes last (not a sum: bench.py reports it as is)     // the committing thread's share of `text` spent waiting for a formatter thread
        default: return -1;
    }
}
int64_t dph_overlap_step_lines(void* hh) { return ((OverlapH*)hh)->run.pafLines; }  // PAF lines of the last step
// host-logic test hook: the value table (commands/overlap.go:55-92) from a k-mer histogram; counts is overwritten with the
// merged forward + reverse-complement counts like the reference's in-place loop
// ---- test hooks: three decision rules of the host side on bare numbers, held to answers worked by hand from the reference's Go
// text (tests/golden/hand/, tests/test_hand_known_answers.py) - not part of the pipeline's boundary
int dph_hand_is_consistent(const int64_t* left5, const int64_t* right4, int circular, int64_t ref_len) {
    return dph::handIsConsistent((const dph::i64*)left5, (const dph::i64*)right4, circular != 0, (dph::i64)ref_len) ? 1 : 0;
}
int dph_hand_remove_dominated(const int64_t* maps3, int n, int64_t query_len, int* kept) {
    return dph::handRemoveDominated((const dph::i64*)maps3, n, (dph::i64)query_len, kept);
}
void dph_hand_trim_indices(int upto, const int32_t* match_a, const int64_t* off, int n_matches, int min_match, int length, int* out2) {
    std::vector<dph::SeedMatch> store((size_t)n_matches);
    std::vector<dph::SeedMatch*> ms;
    for (int i = 0; i < n_matches; i++) {
        store[(size_t)i].MatchA.assign(match_a + off[i], match_a + off[i + 1]);
        ms.push_back(&store[(size_t)i]);
    }
    dph::trimBestIndices(upto, ms, min_match, length, &out2[0], &out2[1]);
}

int dph_hand_add_seeds(const char* bases, int64_t len, int k, int num_seeds, const double* values, uint32_t* seed_map, int cap) {
    dph::SeedIndex ix(k);
    ix.addSeeds(bases, (dph::i64)len, num_seeds, dph::ValueView(values));
    if ((int)ix.seedMap.size() > cap) return -1;
    for (size_t i = 0; i < ix.seedMap.size(); i++) seed_map[i] = (uint32_t)ix.seedMap[i];
    return (int)ix.seedMap.size();
}
int dph_hand_bases_covered(const int32_t* a_seg, int a_n, const int32_t* b_seg, int b_n, const int32_t* match_a, const int32_t* match_b, int n,
                           int k, int64_t* out2) {
    dph::SeedSeq A, B;
    A.seg = a_seg;
    A.n = a_n;
    B.seg = b_seg;
    B.n = b_n;
    dph::SeedMatch m;
    m.SeqA = &A;
    m.SeqB = &B;
    m.MatchA.assign(match_a, match_a + n);
    m.MatchB.assign(match_b, match_b + n);
    dph::i64 a = 0, b = 0;
    bool panic = false;
    dph::matchBasesCovered(m, k, &a, &b, &panic);
    out2[0] = a;
    out2[1] = b;
    return panic ? 1 : 0;
}
int dph_hand_consensus(const int32_t* segs, const int64_t* off, int n_seqs, int k, int32_t* cons_out, int64_t cons_cap, int64_t* cons_n,
                       int* kept, int64_t* out_counts, int32_t* out_a, int32_t* out_b, int64_t cap, int64_t* n_matches) {
    dph::Arena ar;
    std::vector<dph::SeedSeq> store((size_t)n_seqs);
    std::vector<dph::SeedSeq*> seqs;
    for (int i = 0; i < n_seqs; i++) {
        store[(size_t)i].seg = segs + off[i];
        store[(size_t)i].n = (int)(off[i + 1] - off[i]);
        store[(size_t)i].id = i;
        seqs.push_back(&store[(size_t)i]);
    }
    std::vector<dph::SeedMatch*> ms;
    dph::SeedSeq* cons = dph::multiAlignerConsensus(ar, seqs, k, ms);
    if (!cons || cons->n > cons_cap) return -1;
    for (int i = 0; i < cons->n; i++) cons_out[i] = cons->seg[i];
    *cons_n = cons->n;
    *n_matches = (int64_t)ms.size();
    int64_t at = 0;
    for (size_t j = 0; j < ms.size(); j++) {
        kept[j] = ms[j]->SeqB ? ms[j]->SeqB->id : -1;
        out_counts[j] = (int64_t)ms[j]->MatchA.size();
        for (size_t x = 0; x < ms[j]->MatchA.size(); x++) {
            if (at >= cap) return -1;
            out_a[at] = ms[j]->MatchA[x];
            out_b[at] = ms[j]->MatchB[x];
            at++;
        }
    }
    return 0;
}
void dph_pack_bases(const char* bases, int64_t n, uint8_t* out, int scalar_only) { dph::packBases(bases, (size_t)n, out, scalar_only != 0); }
long dph_test_coroutines(int n_tasks, int yields) { return dph::coroSelfTest(n_tasks, yields); }

void dph_values_from_counts(uint64_t* counts, int k, double* out) {
    std::vector<uint64_t> c(counts, counts + ((size_t)1 << (2 * k)));
    std::vector<double> v = kmerValuesFromCounts(c, k);
    memcpy(out, v.data(), v.size() * sizeof(double));
    memcpy(counts, c.data(), c.size() * sizeof(uint64_t));
}
// discards the executor pipeline's in-flight rounds (they are re-executed): a timed region then starts from an empty pipeline
void dph_overlap_drain(void* hh) { ((OverlapH*)hh)->run.drain(); }

// ---- round-parallel mode: a rank executes ONE round speculatively and serialises the result; every rank then commits
// the gathered results in round order with the speculation check (OverlapRun::commitResults).
static void putv(std::string& b, const void* p, size_t n) { b.append((const char*)p, n); }
// text (may be null): the record's PAF text goes there instead of into the record (hdr[17] = 1, hdr[7] still says how long it is):
// the control part - a few KB per round - is all-gathered, the text travels to the printing rank alone (dp_gather_blobs)
static void serialise(RoundResult& res, std::string& blob, std::string* text = nullptr) {
    res.takeText();  // (the PAF text may still be with a formatter thread)
    int64_t hdr[18] = {res.round, res.empty ? 1 : 0, res.firstIn, res.firstOut, res.numQuerySeqs, (int64_t)res.ignores.size(),
                       (int64_t)res.indexedReads.size(), (int64_t)res.paf.size(), res.fs.badBack, res.fs.emptyMatch,
                       (int64_t)res.fs.lines, (int64_t)res.fs.hits, (int64_t)res.fs.qHits, 0, res.snapshot,
                       (int64_t)res.queryReads.size(), res.planRound, text ? 1 : 0};
    int64_t total = (int64_t)(sizeof hdr + sizeof(RoundStats) + res.ignores.size() * sizeof(int) + res.indexedReads.size() * 4 +
                              (text ? 0 : res.paf.size()) + res.queryReads.size() * 4);
    hdr[13] = total;
    putv(blob, hdr, sizeof hdr);
    putv(blob, &res.st, sizeof(RoundStats));
    putv(blob, res.ignores.data(), res.ignores.size() * sizeof(int));
    putv(blob, res.indexedReads.data(), res.indexedReads.size() * 4);
    if (text)
        text->append(res.paf);
    else
        putv(blob, res.paf.data(), res.paf.size());
    putv(blob, res.queryReads.data(), res.queryReads.size() * 4);
}
// Executes rounds first, first+1, ... (one per executor slot of this process, concurrently) and returns their
// serialised results back to back (each record carries its own length in hdr[13]).
const uint8_t* dph_overlap_exec_round(void* hh, int64_t first, uint64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    static thread_local std::string blob;
    std::vector<i64> rounds;
    for (size_t i = 0; i < h->run.slots.size(); i++) rounds.push_back(first + (i64)i);
    std::vector<RoundResult> outs;
    int rc = h->run.executeRounds(rounds, outs);
    if (rc < 0) {
        h->err = h->run.error;
        *n = 0;
        return nullptr;
    }
    blob.clear();
    for (RoundResult& r : outs) serialise(r, blob);
    *n = blob.size();
    return (const uint8_t*)blob.data();
}
int dph_overlap_slots(void* hh) { return (int)((OverlapH*)hh)->run.slots.size(); }
// blobs: concatenation; sizes[i] bytes each.  Returns the number of rounds committed (0..count) or <0.
// textSizes (may be null): per rank, the bytes of text its records announce but do not carry (hdr[17] = 1); the records' own
// lengths go to textLen in the order the records appear
static void deserialise(const uint8_t* blobs, const uint64_t* sizes, int count, std::vector<RoundResult>& rs, bool keepText = true,
                        std::vector<uint64_t>* textSizes = nullptr, std::vector<uint64_t>* textLen = nullptr) {
    uint64_t totalBytes = 0;
    for (int i = 0; i < count; i++) totalBytes += sizes[i];
    const uint8_t* q = blobs;
    const uint8_t* end = blobs + totalBytes;
    if (textSizes) textSizes->assign((size_t)count, 0);
    int rankOf = 0;
    uint64_t rankEnd = count ? sizes[0] : 0;
    while (q < end) {
        while (rankOf + 1 < count && (uint64_t)(q - blobs) >= rankEnd) rankEnd += sizes[++rankOf];
        const uint8_t* rec = q;
        int64_t hdr[18];
        memcpy(hdr, q, sizeof hdr);
        q += sizeof hdr;
        RoundResult r;
        r.round = hdr[0];
        r.empty = hdr[1] != 0;
        r.firstIn = hdr[2];
        r.firstOut = hdr[3];
        r.numQuerySeqs = hdr[4];
        memcpy(&r.st, q, sizeof(RoundStats));
        q += sizeof(RoundStats);
        r.ignores.assign((const int*)q, (const int*)q + hdr[5]);
        q += hdr[5] * sizeof(int);
        r.indexedReads.assign((const uint32_t*)q, (const uint32_t*)q + hdr[6]);
        q += hdr[6] * 4;
        if (hdr[17]) {  // the text travels on its own
            if (textSizes) (*textSizes)[(size_t)rankOf] += (uint64_t)hdr[7];
            if (textLen) textLen->push_back((uint64_t)hdr[7]);
        } else {
            if (keepText) r.paf.assign((const char*)q, (size_t)hdr[7]);  // (a rank that prints nothing keeps the counts, not the text)
            q += hdr[7];
            if (textLen) textLen->push_back(0);
        }
        r.snapshot = hdr[14];
        r.planRound = hdr[16];
        r.queryReads.assign((const uint32_t*)q, (const uint32_t*)q + hdr[15]);
        r.fs.badBack = hdr[8];
        r.fs.emptyMatch = hdr[9];
        r.fs.lines = (uint64_t)hdr[10];
        r.fs.hits = (uint64_t)hdr[11];
        r.fs.qHits = (uint64_t)hdr[12];
        rs.push_back(std::move(r));
        q = rec + hdr[13];
    }
}

// ---- pipelined round-parallel mode: rank `rank` of `world` owns the rounds r % world == rank
void dph_overlap_set_ranks(void* hh, int rank, int world) { ((OverlapH*)hh)->run.setRanks(rank, world); }
// serialised results of this rank's contribution to the current superstep: its next owned round (blocks until its executor
// pipeline has it) and up to max_rounds - 1 finished owned rounds after it, back to back (each record carries its length)
const uint8_t* dph_overlap_wait_owned_many(void* hh, int max_rounds, uint64_t* n) {
    OverlapH* h = (OverlapH*)hh;
    static thread_local std::string blob;
    std::vector<RoundResult> res;
    int rc = h->run.waitOwned(res, max_rounds);
    if (rc != 0) {
        h->err = h->run.error;
        *n = 0;
        return nullptr;
    }
    blob.clear();
    for (RoundResult& r : res) serialise(r, blob);
    *n = blob.size();
    return (const uint8_t*)blob.data();
}
const uint8_t* dph_overlap_wait_owned(void* hh, uint64_t* n) { return dph_overlap_wait_owned_many(hh, 1, n); }
int dph_overlap_commit_gathered(void* hh, const uint8_t* blobs, const uint64_t* sizes, int count) {
    OverlapH* h = (OverlapH*)hh;
    std::vector<RoundResult> rs;
    deserialise(blobs, sizes, count, rs, h->keepText);
    int c = h->run.commitGathered(rs);
    if (c >= 0) h->addPaf(h->run.paf);
    return c;
}

// One superstep of the round-parallel layout with the exchange inside the library (dp_allgather_blobs on the communicator of
// dph_overlap_comm_init / _comm_init_local): this rank's next owned round and up to max_rounds - 1 finished ones after it,
// all-gathered, committed in round order on every rank.  Collective.  Returns the rounds committed (0 = the superstep's first
// round was rejected and is executed again; call dph_overlap_done to learn whether the command is finished), < 0 on error.
int dph_overlap_superstep(void* hh, int max_rounds) {
    OverlapH* h = (OverlapH*)hh;
    if (!h->comm) {
        h->err = "dph_overlap_superstep without a communicator (dph_overlap_comm_init)";
        return -1;
    }
    if (!h->xctx && dp_ctx_create_shared(h->ctx, &h->xctx) != 0) {
        h->err = dp_last_error(nullptr);
        return -1;
    }
    std::vector<RoundResult> res;
    int rc = h->run.waitOwned(res, max_rounds);
    if (rc != 0) {
        h->err = h->run.error;
        dp_comm_abort(h->comm);  // (the peers must not wait for this rank's contribution)
        return rc < 0 ? rc : -1;
    }
    static thread_local std::string blob, text;
    blob.clear();
    text.clear();
    const bool split = h->textRoot >= 0;
    for (RoundResult& r : res) serialise(r, blob, split ? &text : nullptr);
    const uint8_t* all = nullptr;
    const uint64_t* sizes = nullptr;
    rc = dp_allgather_blobs(h->comm, h->xctx, (const uint8_t*)blob.data(), blob.size(), &all, &sizes);
    if (rc != 0) {
        h->err = dp_last_error(h->xctx);
        return rc;
    }
    std::vector<RoundResult> rs;
    std::vector<uint64_t> textSizes, textLen;
    deserialise(all, sizes, dp_comm_size(h->comm), rs, h->keepText, &textSizes, &textLen);
    if (split) {
        // the rounds' text: to the printing rank alone.  Every rank knows every rank's text length from the control records.
        const uint8_t* allText = nullptr;
        rc = dp_gather_blobs(h->comm, h->xctx, (const uint8_t*)text.data(), text.size(), textSizes.data(), h->textRoot, &allText);
        if (rc != 0) {
            h->err = dp_last_error(h->xctx);
            return rc;
        }
        if (allText && h->keepText) {
            const char* t = (const char*)allText;
            for (size_t i = 0; i < rs.size(); i++) {
                rs[i].paf.assign(t, (size_t)textLen[i]);
                t += textLen[i];
            }
        }
    }
    const int c = h->run.commitGathered(rs);
    if (c >= 0) h->addPaf(h->run.paf);
    return c;
}

int dph_overlap_commit_blobs(void* hh, const uint8_t* blobs, const uint64_t* sizes, int count) {
    OverlapH* h = (OverlapH*)hh;
    std::vector<RoundResult> rs;
    deserialise(blobs, sizes, count, rs, h->keepText);
    std::sort(rs.begin(), rs.end(), [](const RoundResult& a, const RoundResult& b) { return a.round < b.round; });
    int c = h->run.commitResults(rs);
    if (c >= 0) h->addPaf(h->run.paf);
    return c;
}
int dph_overlap_done(void* hh) { return ((OverlapH*)hh)->run.done ? 1 : 0; }
// Multi-rank runs: a rank that does not print the PAF (every rank but the one that writes the output) need not keep the other
// ranks' text - the gathered rounds are committed with their counts, flags and read lists, the text is dropped as it arrives
// (at 8 ranks every rank would otherwise copy all 235 MB of a config-2 job's PAF three times).  keep != 0 (default): as before.
void dph_overlap_keep_text(void* hh, int keep) { ((OverlapH*)hh)->keepText = keep != 0; }
// dph_overlap_superstep: root >= 0 = the rounds' PAF text is gathered to rank `root` alone (dp_gather_blobs) and only their control
// records - a few KB per round - are all-gathered; every rank of the job must say the same.  -1 (default): text and control records
// travel together to every rank, as before (what a caller needs that reads the PAF on every rank).
void dph_overlap_text_root(void* hh, int root) { ((OverlapH*)hh)->textRoot = root; }

}  // extern "C"

// ---- host-logic test hook (no GPU needed): ignore flags that arrive while the planner thread holds a finished plan in
// its hands (set DPH_TEST_PLAN_DELAY_US).  A commit flags one read below the next plan's firstIn and one of that plan's own
// query reads; the plan must be thrown away and recomputed.  Returns 0 = the plan handed out is current, 1 = stale plan,
// <0 = the scenario could not be set up.
extern "C" int dph_selftest_planner_flags(void* readsH, int k, int64_t seedBatchSize, const double* values) {
    ReadSet& reads = ((ReadsH*)readsH)->set;
    OverlapParams p;
    p.k = k;
    p.seedBatchSize = seedBatchSize;
    Planner pl(reads, p, values, true, nullptr);
    std::shared_ptr<const RoundPlan> p0 = pl.get(0);  // the thread goes on with plan 1 (prefetch depth)
    if (!p0 || p0->empty || p0->firstOut <= 1 || p0->firstOut >= (i64)reads.size()) return -1;
    const long delay = dph::dph_tune("plan_delay_us", 0);
    if (delay <= 0) return -2;
    usleep((useconds_t)(delay / 2));  // plan 1 is computed by now and sits in the hook's sleep
    const int small = 0, big = (int)p0->firstOut;  // big = first query read of plan 1
    pl.applyIgnores({small, big}, 0);
    pl.dropBefore(1, p0->firstOut);
    std::shared_ptr<const RoundPlan> p1 = pl.get(1);
    if (!p1 || p1->firstIn != p0->firstOut) return -3;
    for (const auto& w : p1->windows)
        if ((int)w.read == big || (int)w.read == small) return 1;
    return 0;
}

// ---- host-logic test hook (no GPU needed): the whole plan chain of a read set three ways - the general PrepareQueries path
// (no window cache), the window-cache path with one planner lane, and with `lanes` lanes that start plans from guesses of
// where their predecessors end - with the same commits in all three (every `flagEvery`-th round flags two reads ahead of the
// chain, as a final check would).  The window cache's producer selects on the host here (WindowCache with ctx == nullptr).
// Returns 0 = all chains equal, r + 1 = they differ in round r, < 0 = set-up problem; *nRounds = plans of the chain.
extern "C" int dph_selftest_planner_lanes(void* readsH, int k, int64_t seedBatchSize, const double* values, int lanes, int flagEvery,
                                          int64_t* nRounds) {
    ReadSet& reads = ((ReadsH*)readsH)->set;
    OverlapParams p;
    p.k = k;
    p.seedBatchSize = seedBatchSize;
    struct Rec {
        i64 firstIn, firstOut;
        std::vector<Overlapper::Window> windows;
        std::vector<uint32_t> seedMap;
        bool empty;
    };
    auto chain = [&](bool useCache, int nLanes, std::vector<Rec>& out) -> int {
        std::fill(reads.ignore.begin(), reads.ignore.end(), 0);
        std::unique_ptr<WindowCache> wc;
        if (useCache) wc.reset(new WindowCache(nullptr, reads, p.overlapSize, k, p.numSeeds, ValueView(values)));
        Planner pl(reads, p, ValueView(values), true, nullptr, wc.get());
        pl.setLanes(nLanes);
        for (i64 r = 0;; r++) {
            std::shared_ptr<const RoundPlan> pp = pl.get(r);
            if (!pp) return -1;
            if (pp->failed) return -2;
            out.push_back({pp->firstIn, pp->firstOut, pp->windows, pp->seedMap, pp->empty});
            if (pp->empty) break;
            if (r > 100000) return -3;
            std::vector<int> flags;
            if (flagEvery > 0 && r % flagEvery == flagEvery - 1) {  // one read a little ahead of the chain, one far ahead
                const i64 a = pp->firstOut + 3, b = pp->firstOut + 40;
                if (a < (i64)reads.size()) flags.push_back((int)a);
                if (b < (i64)reads.size()) flags.push_back((int)b);
            }
            pl.applyIgnores(flags, r);
            pl.dropBefore(r + 1, pp->firstOut);
        }
        return 0;
    };
    std::vector<Rec> a, b, c;
    if (int rc = chain(false, 1, a)) return rc;
    if (int rc = chain(true, 1, b)) return rc - 10;
    if (int rc = chain(true, lanes, c)) return rc - 20;
    std::fill(reads.ignore.begin(), reads.ignore.end(), 0);
    if (nRounds) *nRounds = (int64_t)a.size();
    auto same = [](const Rec& x, const Rec& y) {
        if (x.firstIn != y.firstIn || x.firstOut != y.firstOut || x.empty != y.empty || x.seedMap != y.seedMap || x.windows.size() != y.windows.size()) return false;
        for (size_t i = 0; i < x.windows.size(); i++)
            if (x.windows[i].read != y.windows[i].read || x.windows[i].start != y.windows[i].start || x.windows[i].len != y.windows[i].len) return false;
        return true;
    };
    for (size_t r = 0; r < std::max(a.size(), std::max(b.size(), c.size())); r++) {
        if (r >= a.size() || r >= b.size() || r >= c.size()) return (int)r + 1;
        if (!same(a[r], b[r]) || !same(a[r], c[r])) return (int)r + 1;
    }
    return 0;
}

// ---- test hook: SeedIndex::touchesSeed on evaluated k-mers, every instruction-set variant the CPU has against the scalar one.
// Fills res[w] (w < nWindows; windows of `stride` k-mers) with the scalar answers; returns the number of disagreements, and
// sets *isaMask to the variants that ran (bit 1 AVX2, bit 2 AVX-512).
extern "C" int dph_selftest_touch(int k, const uint32_t* seeds, int64_t nSeeds, const uint32_t* kmers, int64_t nWindows, int64_t stride,
                                  uint8_t* res, int* isaMask) {
    SeedIndex ix(k, 21);
    for (int64_t i = 0; i < nSeeds; i++) ix.addSeedKmer(seeds[i]);
    int bad = 0, mask = 0;
    if (__builtin_cpu_supports("avx2")) mask |= 2;
    if (__builtin_cpu_supports("avx512f")) mask |= 4;
    for (int64_t w = 0; w < nWindows; w++) {
        const uint32_t* km = kmers + w * stride;
        const bool a = ix.touchesSeedWith(0, km, (uint32_t)stride);
        res[w] = a ? 1 : 0;
        if ((mask & 2) && ix.touchesSeedWith(1, km, (uint32_t)stride) != a) bad++;
        if ((mask & 4) && ix.touchesSeedWith(2, km, (uint32_t)stride) != a) bad++;
        if (ix.touchesSeed(km, (uint32_t)stride) != a) bad++;
    }
    if (isaMask) *isaMask = mask;
    return bad;
}

// ---- host-logic test hook (no GPU needed): finalCheckWorker over externally supplied matches ----------------------
// Queries and indexed sequences come as flat arrays in the reference's segment layout; matches as (query index,
// target index, MatchA, MatchB).  Returns the PAF text of the round and applies SetIgnore to `reads`.
extern "C" const char* dph_finalcheck(void* readsH, int k, int64_t overlapSize, const uint32_t* seedKmers, int64_t nSeeds,
                                      const int32_t* qSegs, const int64_t* qOff, const int64_t* qId, const int64_t* qSeqId,
                                      const int64_t* qLen, const int64_t* qOffset, const int64_t* qInset, int64_t nQ,
                                      const int32_t* iSegs, const int64_t* iOff, const int64_t* iId, const int64_t* iLen,
                                      const int64_t* iOffset, const int64_t* iInset, int64_t nI, const int64_t* mQuery,
                                      const int64_t* mTarget, const int64_t* mOff, const int32_t* mA, const int32_t* mB,
                                      int64_t nM, int64_t numQuerySeqs, int64_t* outLen, int64_t* outStats) {
    static thread_local std::string paf;
    ReadSet& reads = ((ReadsH*)readsH)->set;
    SeedIndex index(k);
    for (int64_t i = 0; i < nSeeds; i++) index.addSeedKmer(seedKmers[i]);
    index.buildRcTable();
    Arena& ar = index.arena;
    std::vector<SeedSeq*> qs, is;
    for (int64_t q = 0; q < nQ; q++) {
        SeedSeq* s = ar.make();
        s->seg = qSegs + qOff[q];
        s->n = (int)(qOff[q + 1] - qOff[q]);
        s->id = (int)qSeqId[q];
        s->length = qLen[q];
        s->offset = qOffset[q];
        s->inset = qInset[q];
        s->rc = (q & 1) != 0;
        qs.push_back(s);
    }
    for (int64_t q = 0; q + 1 < nQ; q += 2) qs[(size_t)q + 1]->reverseComplement = qs[(size_t)q];  // rc query -> fwd (sequence.go:157)
    for (int64_t i = 0; i < nI; i++) {
        SeedSeq* root = ar.make();
        root->length = reads.length((size_t)iId[i]);
        root->id = (int)iId[i];
        SeedSeq* s = ar.make();
        s->seg = iSegs + iOff[i];
        s->n = (int)(iOff[i + 1] - iOff[i]);
        s->id = (int)iId[i];
        s->length = iLen[i];
        s->offset = iOffset[i];
        s->inset = iInset[i];
        s->parent = root;
        is.push_back(s);
    }
    std::vector<std::unique_ptr<SeedMatch>> matches;
    for (int64_t m = 0; m < nM; m++) {
        std::unique_ptr<SeedMatch> sm(new SeedMatch());
        sm->MatchA.assign(mA + mOff[m], mA + mOff[m + 1]);
        sm->MatchB.assign(mB + mOff[m], mB + mOff[m + 1]);
        sm->SeqA = qs[(size_t)mQuery[m]];
        sm->SeqB = is[(size_t)mTarget[m]];
        sm->QueryID = (int)qId[mQuery[m]];
        sm->ReverseComplementQuery = (mQuery[m] & 1) != 0;
        matches.push_back(std::move(sm));
    }
    FinalCheckStats fs;
    paf.clear();
    finalCheck(ar, index, reads, matches, numQuerySeqs, overlapSize, paf, fs);
    if (outStats) {
        outStats[0] = fs.badBack;
        outStats[1] = fs.emptyMatch;
        outStats[2] = (int64_t)fs.lines;
    }
    *outLen = (int64_t)paf.size();
    return paf.data();
}

// ---- `downpore map` ------------------------------------------------------------------------------------------------
// params: circular,k,querySize,minLength,chunkSize,seedRate.  Returns a handle holding paf/err text, or NULL.
namespace {
struct MapH {
    std::string paf, err;
    MapStats st;
};
}  // namespace
extern "C" void* dph_map_run(void* refH, void* readsH, const int64_t* params, int device) {
    MapH* h = new MapH();
    MapParams p;
    p.circular = params[0] != 0;
    p.k = (int)params[1];
    p.querySize = params[2];
    p.minLength = params[3];
    p.chunkSize = params[4];
    p.seedRate = params[5];
    std::string error;
    int rc = runMap(((ReadsH*)refH)->set, ((ReadsH*)readsH)->set, p, device, h->paf, h->err, &h->st, error);
    if (rc != 0) {
        g_err = error;
        delete h;
        return nullptr;
    }
    return h;
}
extern "C" void dph_map_free(void* h) { delete (MapH*)h; }
extern "C" const char* dph_map_paf(void* h, int64_t* n) {
    *n = (int64_t)((MapH*)h)->paf.size();
    return ((MapH*)h)->paf.data();
}
extern "C" const char* dph_map_errtext(void* h, int64_t* n) {
    *n = (int64_t)((MapH*)h)->err.size();
    return ((MapH*)h)->err.data();
}
// out: n_chunks,n_seeds,n_windows,n_chains,n_batches,k_scan_ms,k_map_ms
extern "C" void dph_map_stats(void* h, double* out) {
    const MapStats& s = ((MapH*)h)->st;
    double v[] = {(double)s.n_chunks, (double)s.n_seeds, (double)s.n_windows, (double)s.n_chains, (double)s.n_batches, s.k_scan_ms,
                  s.k_map_ms, s.t_setup_s, s.t_scan_s, s.t_chain_s, s.t_host_s, s.map_bytes, s.scan_bytes};
    memcpy(out, v, sizeof v);
}

// ---- host-logic test hook (no GPU needed): the plan chain of a round-parallel run of `world` ranks whose planners each compute
// only the plans of their own rounds (Planner::setOwnership) and guess where the rounds in between end.  Every simulated rank has
// its own copy of the read set (its own flags), its own window cache (host producer) and planner; the driver plays the commit:
// round r's plan comes from its owner, is accepted iff it starts at the committed firstSequence (what OverlapRun::resultValid
// checks) - otherwise every rank is told the truth (dropBefore) and the owner is asked again - and is then committed on every
// rank (flags of every `flagEvery`-th round + dropBefore).  The accepted chain must equal the one a single dense planner walks.
// Returns 0 = equal, r + 1 = differs in round r, < 0 = set-up problem / a round that never became valid; *nRounds = plans of the
// chain, *nRedone = plans that had to be asked for again because their guess did not hold.
extern "C" int dph_selftest_planner_sparse(void* readsH, int k, int64_t seedBatchSize, const double* values, int world, int flagEvery,
                                           int64_t* nRounds, int64_t* nRedone) {
    ReadSet& reads = ((ReadsH*)readsH)->set;
    OverlapParams p;
    p.k = k;
    p.seedBatchSize = seedBatchSize;
    struct Rec {
        i64 firstIn, firstOut;
        std::vector<Overlapper::Window> windows;
        std::vector<uint32_t> seedMap;
        bool empty;
    };
    auto flagsOf = [&](i64 r, i64 firstOut, size_t n) {
        std::vector<int> flags;
        if (flagEvery > 0 && r % flagEvery == flagEvery - 1) {
            const i64 a = firstOut + 3, b = firstOut + 40;
            if (a < (i64)n) flags.push_back((int)a);
            if (b < (i64)n) flags.push_back((int)b);
        }
        return flags;
    };
    std::vector<Rec> ref, got;
    {
        std::fill(reads.ignore.begin(), reads.ignore.end(), 0);
        WindowCache wc(nullptr, reads, p.overlapSize, k, p.numSeeds, ValueView(values));
        Planner pl(reads, p, ValueView(values), true, nullptr, &wc);
        for (i64 r = 0;; r++) {
            std::shared_ptr<const RoundPlan> pp = pl.get(r);
            if (!pp || pp->failed) return -1;
            ref.push_back({pp->firstIn, pp->firstOut, pp->windows, pp->seedMap, pp->empty});
            if (pp->empty) break;
            if (r > 100000) return -3;
            pl.applyIgnores(flagsOf(r, pp->firstOut, reads.size()), r);
            pl.dropBefore(r + 1, pp->firstOut);
        }
        std::fill(reads.ignore.begin(), reads.ignore.end(), 0);
    }
    i64 redone = 0;
    {
        std::vector<std::unique_ptr<ReadSet>> rs;
        std::vector<std::unique_ptr<WindowCache>> wcs;
        std::vector<std::unique_ptr<Planner>> pls;
        for (int w = 0; w < world; w++) {
            rs.emplace_back(new ReadSet(reads));
            wcs.emplace_back(new WindowCache(nullptr, *rs[w], p.overlapSize, k, p.numSeeds, ValueView(values)));
            pls.emplace_back(new Planner(*rs[w], p, ValueView(values), true, nullptr, wcs[w].get()));
            pls[w]->setLanes(3);
            pls[w]->setOwnership(w, world);
            if (!pls[w]->sparse()) return -4;
        }
        i64 firstSequence = 0;
        // (the executor slots of a rank ask for the plans of rounds well ahead of the commit point - that is what makes them
        // speculative: here every round's plan is fetched from its owner 2 * world rounds before its turn)
        std::map<i64, std::pair<std::shared_ptr<const RoundPlan>, i64>> ahead;  // plan, commit point when it was fetched
        std::vector<i64> flagRound(reads.size(), -1);                            // round whose commit flagged the read
        bool ended = false;
        for (i64 r = 0;; r++) {
            Planner& owner = *pls[(size_t)(r % world)];
            for (i64 j = r; j <= r + 2 * world && !ended; j++) {
                if (ahead.count(j)) continue;
                std::shared_ptr<const RoundPlan> q = pls[(size_t)(j % world)]->get(j);
                if (!q || q->failed) return -5;
                ahead[j] = {q, r};
                if (q->empty) break;  // (what lies behind an "end of input" - real or guessed - is asked for once it is settled)
            }
            std::shared_ptr<const RoundPlan> pp = ahead.count(r) ? ahead[r].first : owner.get(r);
            i64 snap = ahead.count(r) ? ahead[r].second : r;
            ahead.erase(r);
            for (int attempt = 0;; attempt++) {
                if (!pp || pp->failed) return -5;
                // OverlapRun::resultValid / emptyResultValid: the plan starts at the committed firstSequence and none of its query
                // reads was flagged by a round committed since it was made
                bool valid = pp->empty ? (pp->round == r && pp->firstIn == firstSequence) : pp->firstIn == firstSequence;
                for (size_t i = 0; valid && i < pp->windows.size(); i++)
                    if (flagRound[pp->windows[i].read] >= snap) valid = false;
                if (valid) break;
                snap = r;
                if (attempt >= 3) return -6;
                redone++;
                // the commit point has reached r and the plan does not start there: every rank knows where r starts (dropBefore
                // of the last commit), the owner plans the round again
                pp = owner.get(r);
            }
            ended = pp->empty;
            got.push_back({pp->firstIn, pp->firstOut, pp->windows, pp->seedMap, pp->empty});
            if (pp->empty) break;
            if (r > 100000) return -3;
            firstSequence = pp->firstOut;
            const std::vector<int> flags = flagsOf(r, pp->firstOut, reads.size());
            for (int id : flags)
                if (flagRound[(size_t)id] < 0) flagRound[(size_t)id] = r;
            for (auto& q : pls) {
                q->applyIgnores(flags, r);
                q->dropBefore(r + 1, firstSequence);
            }
        }
    }
    if (nRounds) *nRounds = (int64_t)ref.size();
    if (nRedone) *nRedone = redone;
    auto same = [](const Rec& x, const Rec& y) {
        if (x.firstIn != y.firstIn || x.firstOut != y.firstOut || x.empty != y.empty || x.seedMap != y.seedMap || x.windows.size() != y.windows.size()) return false;
        for (size_t i = 0; i < x.windows.size(); i++)
            if (x.windows[i].read != y.windows[i].read || x.windows[i].start != y.windows[i].start || x.windows[i].len != y.windows[i].len) return false;
        return true;
    };
    for (size_t r = 0; r < std::max(ref.size(), got.size()); r++) {
        if (r >= ref.size() || r >= got.size()) return (int)r + 1;
        if (!same(ref[r], got[r])) return (int)r + 1;
    }
    return 0;
}
