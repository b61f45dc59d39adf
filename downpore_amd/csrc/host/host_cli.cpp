// `downpore` command-line front end of the product: same commands, flag names, defaults, aliases, stderr lines and
// PAF columns as the reference for the path in scope (downpore.go:34-92; commands/overlap.go:22-29; commands/map.go:17-22).
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "host_util.hpp"

using namespace dph;

static bool parseBool(const std::string& a) { return a == "1" || (!a.empty() && (a[0] == 'T' || a[0] == 't')); }
static i64 parseInt(const std::string& a, bool& ok) {
    char* e = nullptr;
    long long v = strtoll(a.c_str(), &e, 10);
    if (a.empty() || *e) {
        fprintf(stderr, "Invalid integer argument value:%s\n", a.c_str());
        ok = false;
    }
    return v;
}

static int runOverlap(ArgTable& t) {
    bool ok = true;
    OverlapParams p;
    p.overlapSize = parseInt(t.args["overlap_size"], ok);
    p.numSeeds = (int)parseInt(t.args["num_seeds"], ok);
    p.seedBatchSize = parseInt(t.args["seed_batch_size"], ok);
    p.queryBatchSize = parseInt(t.args["query_batch_size"], ok);
    p.chunkSize = parseInt(t.args["chunk_size"], ok);
    p.numWorkers = (int)parseInt(t.args["num_workers"], ok);
    p.k = (int)parseInt(t.args["k"], ok);
    p.minHits = atof(t.args["min_hits"].c_str());
    p.himem = parseBool(t.args["himem"]);
    if (!ok) return 1;
    if (!t.args["seed_values"].empty()) {
        fprintf(stderr, "seed_values files are not supported by this build (out of scope, SURVEY #16)\n");
        return 1;
    }
    ReadSet reads;
    std::string err;
    const bool prof = getenv("DPH_PROFILE") != nullptr;
    double tm = now();
    auto mark = [&](const char* what) {
        if (!prof) return;
        const double tn = now();
        fprintf(stderr, "[cli] %-24s %.1f ms\n", what, 1e3 * (tn - tm));
        tm = tn;
    };
    if (!ReadSet::fromFile(t.args["input"], p.overlapSize, p.himem, reads, err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    mark("read input");
    dp_ctx* ctx = nullptr;
    if (dp_ctx_create(0, &ctx) != 0) {
        fprintf(stderr, "downpore: %s\n", dp_last_error(nullptr));
        return 2;
    }
    int rc = dp_reads_upload(ctx, (const uint8_t*)reads.bases.data(), reads.off.data(), (uint32_t)reads.size());
    OverlapRun run;
    const char* ns = getenv("DP_EXEC_SLOTS");
    if (rc == 0) rc = run.init(ctx, &reads, p, nullptr, ns ? atoi(ns) : 8);
    if (rc != 0) {
        fprintf(stderr, "downpore: %s\n", run.error.empty() ? dp_last_error(ctx) : run.error.c_str());
        return 2;
    }
    mark("set-up (device)");
    size_t shown = 0;
    for (;;) {
        rc = run.step();
        if (rc == 0) break;
        if (rc < 0) {
            fprintf(stderr, "downpore: %s\n", run.error.c_str());
            return 2;
        }
        fwrite(run.paf.data(), 1, run.paf.size(), stdout);
        fwrite(run.errText.data() + shown, 1, run.errText.size() - shown, stderr);
        shown = run.errText.size();
    }
    fwrite(run.errText.data() + shown, 1, run.errText.size() - shown, stderr);
    mark("rounds + output");
    fprintf(stderr, "[downpore_amd] rounds=%lld bad_back_suppressed=%lld empty_match_panics_avoided=%lld\n", (long long)run.round,
            (long long)run.badBack, (long long)run.emptyMatch);
    run.shutdown();
    dp_ctx_destroy(ctx);
    return 0;
}

int main(int argc, char** argv) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // one hardware queue per executor slot's stream (the runtime's default is 4)
    ArgTable ov, mp;
    ov.make({"overlap_size", "k", "num_seeds", "seed_batch_size", "chunk_size", "query_batch_size", "min_hits", "num_workers",
             "input", "seed_values", "himem"},
            {"1000", "10", "15", "10000", "10000", "20000", "0.25", "4", "", "", "true"},
            {"Size of overlap to search for in bases", "Number of bases in each seed",
             "Minimum number of seeds to generate for each overlap query", "Maximum total unique seeds to use in each query batch",
             "Size to chop long reads into for querying against, in bases",
             "Maximum number of queries per batch (if max seeds not reached)", "Minimum proportion of seeds that must match each query",
             "Number of worker threads to spawn", "Fasta/fastq input file", "File containing values to use during seed selection.",
             "Whether to cache all reads in memory"});
    mp.make({"input", "reference", "circular", "k", "query_size", "min_length", "chunk_size", "seed_rate", "num_workers"},
            {"", "", "true", "11", "1000", "500", "10000", "40", "4"},
            {"Fasta/fastq input file", "A fasta file containing a reference sequence to align against",
             "Whether the reference genome is circular", "Length of seeds in bases", "The number of bases to query at a time",
             "The minimum sequence size to generate queries from", "The number of bases for reference index chunks",
             "The maximum number of bases between seeds in the reference", "The number of worker process to use for mapping"});
    if (argc == 1) {
        printf("Available commands:\n help <command> Describe the command and its arguments\n overlap\n map\n");
        return 0;
    }
    std::string cmd = argv[1];
    if (cmd == "help") {
        ArgTable* t = argc > 2 && !strcmp(argv[2], "overlap") ? &ov : argc > 2 && !strcmp(argv[2], "map") ? &mp : nullptr;
        if (!t) {
            printf("Usage: downpore help <command>\nTo see a list of available commands just run downpore\n");
            return 0;
        }
        for (size_t i = 0; i < t->names.size(); i++) {
            auto a = t->alias.find(t->names[i]);
            printf("-%s  %s  %s  (default:%s)\n", t->names[i].c_str(), a != t->alias.end() ? ("-" + a->second).c_str() : "",
                   t->descriptions[i].c_str(), t->defaults[i].c_str());
        }
        return 0;
    }
    std::string err;
    if (cmd == "overlap") {
        if (!ov.parse(argc, argv, err)) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        return runOverlap(ov);
    }
    if (cmd == "map") {
        if (!mp.parse(argc, argv, err)) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        bool ok = true;
        MapParams p;
        p.k = (int)parseInt(mp.args["k"], ok);
        p.numWorkers = (int)parseInt(mp.args["num_workers"], ok);
        p.minLength = parseInt(mp.args["min_length"], ok);
        p.circular = parseBool(mp.args["circular"]);
        p.querySize = parseInt(mp.args["query_size"], ok);
        p.chunkSize = parseInt(mp.args["chunk_size"], ok);
        p.seedRate = parseInt(mp.args["seed_rate"], ok);
        if (!ok) return 1;
        ReadSet ref, reads;
        if (!ReadSet::fromFile(mp.args["reference"], 0, false, ref, err) || !ReadSet::fromFile(mp.args["input"], p.minLength, false, reads, err)) {
            fprintf(stderr, "%s\n", err.c_str());
            return 1;
        }
        std::string paf, errText, error;
        int rc = runMap(ref, reads, p, 0, paf, errText, nullptr, error);
        if (rc != 0) {
            fprintf(stderr, "downpore: %s\n", error.c_str());
            return 2;
        }
        fwrite(paf.data(), 1, paf.size(), stdout);
        fwrite(errText.data(), 1, errText.size(), stderr);
        return 0;
    }
    printf("Available commands:\n help <command> Describe the command and its arguments\n");
    return 0;
}
