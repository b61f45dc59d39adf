// Seeded synthetic long-read generator.  Neutral tooling: used by tests/, bench.py and
// the CLI front ends to create inputs; it is neither part of the oracle nor of the product path.
//
//   PRNG   : xoshiro256** whose 4 state words are successive splitmix64(seed) outputs
//   genome : G i.i.d. uniform bases, base = "ACGT"[next() >> 62]
//   read r : len = L (fixed) or floor(L * (0.5 + u)), u = (next() >> 11) * 2^-53      (variable)
//            start = next() % (G - len + 1); strand = next() >> 63 (1 = reverse complement)
//            per template base, with probability e an error: 1/2 substitution (to one of the 3 other
//            bases), 1/4 insertion (random base before the base), 1/4 deletion
//   names  : r%07d
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Rng {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) {
        for (int i = 0; i < 4; i++) s[i] = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};
const char BASES[4] = {'A', 'C', 'G', 'T'};
inline int code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3; }
}  // namespace

extern "C" {

// Fills genome (G bytes, ASCII).  Same stream prefix as dps_reads uses.
void dps_genome(uint64_t seed, int64_t G, char* genome) {
    Rng r(seed);
    for (int64_t i = 0; i < G; i++) genome[i] = BASES[r.next() >> 62];
}

// Generates N reads.  bases must hold at least N*(maxLen*(1+e*...)) bytes: pass capacity; returns the
// number of bytes written or -1 if capacity is too small.  off has N+1 entries.  starts/strands (N each)
// may be NULL.  variable != 0 selects the L*U[0.5,1.5] length model.
int64_t dps_reads(uint64_t seed, int64_t G, int64_t N, int64_t L, double e, int variable, char* bases, int64_t cap,
                  int64_t* off, int64_t* starts, uint8_t* strands) {
    Rng r(seed);
    std::vector<char> genome((size_t)G);
    for (int64_t i = 0; i < G; i++) genome[(size_t)i] = BASES[r.next() >> 62];
    int64_t pos = 0;
    std::vector<char> tmpl;
    for (int64_t n = 0; n < N; n++) {
        int64_t len = L;
        if (variable) len = (int64_t)((double)L * (0.5 + r.uniform()));
        if (len > G) len = G;
        int64_t start = (int64_t)(r.next() % (uint64_t)(G - len + 1));
        int strand = (int)(r.next() >> 63);
        tmpl.assign(genome.begin() + start, genome.begin() + start + len);
        if (strand) {
            for (int64_t i = 0, j = len - 1; i < j; i++, j--) std::swap(tmpl[(size_t)i], tmpl[(size_t)j]);
            for (auto& c : tmpl) c = BASES[3 - code(c)];
        }
        off[n] = pos;
        if (starts) starts[n] = start;
        if (strands) strands[n] = (uint8_t)strand;
        for (int64_t i = 0; i < len; i++) {
            if (pos + 2 > cap) return -1;
            char b = tmpl[(size_t)i];
            if (e > 0.0 && r.uniform() < e) {
                double u = r.uniform();
                if (u < 0.5) {
                    bases[pos++] = BASES[(code(b) + 1 + (int)(r.next() % 3)) & 3];
                } else if (u < 0.75) {
                    bases[pos++] = BASES[r.next() >> 62];
                    bases[pos++] = b;
                }  // else deletion
            } else {
                bases[pos++] = b;
            }
        }
    }
    off[N] = pos;
    return pos;
}

}  // extern "C"

#ifdef DPS_MAIN
// dp_synth reads <seed> <G> <N> <L> <e> <variable> > reads.fa   |   dp_synth genome <seed> <G> <name> > ref.fa
int main(int argc, char** argv) {
    if (argc >= 5 && !strcmp(argv[1], "genome")) {
        int64_t G = atoll(argv[3]);
        std::vector<char> g((size_t)G);
        dps_genome(strtoull(argv[2], 0, 10), G, g.data());
        printf(">%s\n", argv[4]);
        fwrite(g.data(), 1, (size_t)G, stdout);
        printf("\n");
        return 0;
    }
    if (argc >= 8 && !strcmp(argv[1], "reads")) {
        uint64_t seed = strtoull(argv[2], 0, 10);
        int64_t G = atoll(argv[3]), N = atoll(argv[4]), L = atoll(argv[5]);
        double e = atof(argv[6]);
        int variable = atoi(argv[7]);
        int64_t cap = N * (L * 2 + 16);
        std::vector<char> bases((size_t)cap);
        std::vector<int64_t> off((size_t)N + 1);
        if (dps_reads(seed, G, N, L, e, variable, bases.data(), cap, off.data(), 0, 0) < 0) return 1;
        for (int64_t n = 0; n < N; n++) {
            printf(">r%07lld\n", (long long)n);
            fwrite(bases.data() + off[(size_t)n], 1, (size_t)(off[(size_t)n + 1] - off[(size_t)n]), stdout);
            printf("\n");
        }
        return 0;
    }
    fprintf(stderr, "usage: dp_synth reads <seed> <G> <N> <L> <e> <variable> | dp_synth genome <seed> <G> <name>\n");
    return 2;
}
#endif
