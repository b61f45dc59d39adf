// ORACLE — test infrastructure only (see oracle.hpp).  Command-line front end with the reference's
// flag table (commands/overlap.go:24-25, commands/map.go:19-20; "-name value" pairs, downpore.go:34-51).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>

#include "oracle.hpp"

using namespace dpo;

static bool parseBool(const std::string& a) { return a == "1" || (!a.empty() && (a[0] == 'T' || a[0] == 't')); }

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "usage: dp_oracle overlap|map [-arg value]...\n");
        return 2;
    }
    std::string cmd = argv[1];
    std::map<std::string, std::string> args;
    if (cmd == "overlap")
        args = {{"overlap_size", "1000"}, {"k", "10"}, {"num_seeds", "15"}, {"seed_batch_size", "10000"},
                {"chunk_size", "10000"}, {"query_batch_size", "20000"}, {"min_hits", "0.25"}, {"num_workers", "4"},
                {"input", ""}, {"seed_values", ""}, {"himem", "true"}, {"max_rounds", "-1"}};
    else if (cmd == "map")
        args = {{"input", ""}, {"reference", ""}, {"circular", "true"}, {"k", "11"}, {"query_size", "1000"},
                {"min_length", "500"}, {"chunk_size", "10000"}, {"seed_rate", "40"}, {"num_workers", "4"}};
    else {
        fprintf(stderr, "unknown command %s\n", cmd.c_str());
        return 2;
    }
    for (int i = 2; i + 1 < argc; i += 2) {
        std::string name = argv[i];
        while (!name.empty() && name[0] == '-') name.erase(0, 1);
        if (!args.count(name)) {
            // unique-prefix aliases (commands/command.go:26-54) — accept any unambiguous prefix
            std::string hit;
            int n = 0;
            for (auto& kv : args)
                if (kv.first.compare(0, name.size(), name) == 0) {
                    hit = kv.first;
                    n++;
                }
            if (n != 1) {
                fprintf(stderr, "Unrecognised argument:%s\n", name.c_str());
                return 1;
            }
            name = hit;
        }
        args[name] = argv[i + 1];
    }
    try {
        if (cmd == "overlap") {
            OverlapParams p;
            p.overlapSize = atoll(args["overlap_size"].c_str());
            p.k = atoi(args["k"].c_str());
            p.numSeeds = atoll(args["num_seeds"].c_str());
            p.seedBatchSize = atoll(args["seed_batch_size"].c_str());
            p.chunkSize = atoll(args["chunk_size"].c_str());
            p.queryBatchSize = atoll(args["query_batch_size"].c_str());
            p.minHits = atof(args["min_hits"].c_str());
            p.himem = parseBool(args["himem"]);
            FastaSet set = FastaSet::fromFile(args["input"], p.overlapSize, p.himem);
            if (!set.error.empty()) {
                fprintf(stderr, "%s\n", set.error.c_str());
                return 1;
            }
            OverlapResult r = runOverlap(set, p, nullptr, atoll(args["max_rounds"].c_str()), false);
            fputs(r.err.c_str(), stderr);
            fwrite(r.paf.data(), 1, r.paf.size(), stdout);
        } else {
            MapParams p;
            p.circular = parseBool(args["circular"]);
            p.k = atoi(args["k"].c_str());
            p.querySize = atoll(args["query_size"].c_str());
            p.minLength = atoll(args["min_length"].c_str());
            p.chunkSize = atoll(args["chunk_size"].c_str());
            p.seedRate = atoll(args["seed_rate"].c_str());
            FastaSet ref = FastaSet::fromFile(args["reference"], 0, false);
            FastaSet reads = FastaSet::fromFile(args["input"], p.minLength, false);
            if (!ref.error.empty() || !reads.error.empty()) {
                fprintf(stderr, "%s\n", (ref.error + reads.error).c_str());
                return 1;
            }
            MapResult r = runMap(ref, reads, p);
            fputs(r.err.c_str(), stderr);
            fwrite(r.paf.data(), 1, r.paf.size(), stdout);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 3;
    }
    return 0;
}
