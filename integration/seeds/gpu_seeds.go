// gpu_seeds.go — goes into package seeds (github.com/jteutenberg/downpore/seeds) next to seeds.go / sequence.go.
//
// SeedSequence and SeedIndex keep their fields unexported, so the little a GPU-backed Overlapper / Mapper needs from
// outside the package is added here: building a SeedSequence from segments the device produced (what NewSeedSequence,
// seeds.go:33-50, builds from CountKmers + WriteSegments), reading the seed list (seedMap), and registering an indexed
// sequence whose seed set lives on the device (AddSequence, seeds.go:272-290, without the host-side IntSet).
package seeds

import "github.com/jteutenberg/downpore/sequence"

// NewSeedSequenceFromSegments wraps segments = [gap, seed, gap, ..., gap] (seed ids, not k-mers) exactly as
// SeedIndex.NewSeedSequence would have for a sequence with this id / name / Len() / GetOffset() / GetInset().
func NewSeedSequenceFromSegments(segments []int, id int, name string, length, offset, inset int) *SeedSequence {
	nm := name
	return &SeedSequence{segments: segments, length: length, id: id, name: &nm, offset: offset, inset: inset, rc: false}
}

// NewSeedSequenceLike is the same for a sequence.Sequence view the caller still holds (query windows).
func NewSeedSequenceLike(segments []int, seq sequence.Sequence) *SeedSequence {
	return NewSeedSequenceFromSegments(segments, seq.GetID(), seq.GetName(), seq.Len(), seq.GetOffset(), seq.GetInset())
}

// SeedKmers returns seedMap[:size]: the k-mer of every seed id, which is what dp_round_begin installs on the device.
func (g *SeedIndex) SeedKmers() []uint32 {
	out := make([]uint32, g.size)
	for i := 0; i < g.size; i++ {
		out[i] = uint32(g.seedMap[i])
	}
	return out
}

// ReverseComplementSeeds returns kmerMap[ReverseComplement(seedMap[s], k)] for every seed s (sequence.go:125-159): the
// table dp_consensus_paf needs to reverse-complement target chunks on the device.
func (g *SeedIndex) ReverseComplementSeeds() []int32 {
	out := make([]int32, g.size)
	for i := 0; i < g.size; i++ {
		out[i] = g.kmerMap[ReverseComplement(uint(g.seedMap[i]), g.seedSize)]
	}
	return out
}

// AddDeviceSequence registers an indexed sequence whose seed bitset and posting-list entries are built on the device
// (dp_index_build): only the sequence list grows, so that GetSeedSequence(index) keeps working for the consensus stage.
// Returns the sequence's index.
func (g *SeedIndex) AddDeviceSequence(seq *SeedSequence) int {
	g.lock.Lock()
	g.sequences = append(g.sequences, seq)
	i := len(g.sequences) - 1
	g.lock.Unlock()
	return i
}
