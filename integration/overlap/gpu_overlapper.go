// gpu_overlapper.go — goes into package overlap (github.com/jteutenberg/downpore/overlap) next to overlap.go.
//
// An overlap.Overlapper whose scan, index and chaining run on an MI355X through libdownpore_hip.so (package gpu wraps
// its C ABI).  commands/overlap.go changes in two places (Run, :96-195):
//
//	gpuReads := overlap.NewGPUReads(seqSet, uint(k), 0)   // once, after getKmerValues: the reads become resident in HBM
//	...
//	overlapper = overlap.NewGPUOverlapper(gpuReads, seedIndex, chunkSize, numWorkers, overlapSize, numSeeds, hitFraction)
//
// and prints the same PAF.  What runs where:
//   PrepareQueries  unchanged (the embedded CPU overlapper: AddSeeds is sequential by nature, the windows are 1000 bases);
//   AddSequences    dp_round_begin + dp_scan_reads (every read whose id arrives on the channel; the library compacts the
//                   reads with >= minSeeds hits on the device), chunkWorker's slicing here, dp_index_build;
//   FindOverlaps    dp_find_overlaps: Matches -> GetSharedIDs -> CountIntersectionTo -> PairwiseAlignments -> ratchet for all
//                   queries at once; matches are sent in the canonical single-worker order.
package overlap

import (
	"log"

	"github.com/jteutenberg/downpore/gpu"
	"github.com/jteutenberg/downpore/seeds"
	"github.com/jteutenberg/downpore/sequence"
)

// GPUReads is the read set resident on one GPU for the whole command (the seed index and the overlapper are per round).
type GPUReads struct {
	ctx      *gpu.Context
	numReads int
	lengths  []int    // Len() of every read as it was uploaded, by id
	names    []string
	himem    bool     // later passes serve cached views (seqio.go:115), not top-level sequences
	ignore   []byte   // scratch: 1 = the read did not arrive on AddSequences' channel this round
	epoch    uint64
}

// NewGPUReads uploads every sequence of `set` once (2-bit packed on the device, resident for all rounds) and prepares the
// rounds' scans for seed length k (the resident k-mer position index from 1 Gbase up).  device = HIP device ordinal.
func NewGPUReads(set sequence.SequenceSet, k uint, device int) *GPUReads {
	ctx, err := gpu.NewContext(device)
	if err != nil {
		log.Fatal(err)
	}
	g := &GPUReads{ctx: ctx, himem: true}
	bases := make([]byte, 0, 1<<30)
	off := []int64{0}
	for s := range set.GetSequences() {
		if s == nil {
			continue
		}
		for s.GetID() > len(g.lengths) { // ids are file order; reads ignored already leave holes
			g.lengths = append(g.lengths, 0)
			g.names = append(g.names, "")
			off = append(off, int64(len(bases)))
		}
		bases = append(bases, s.String()...)
		off = append(off, int64(len(bases)))
		g.lengths = append(g.lengths, s.Len())
		g.names = append(g.names, s.GetName())
	}
	g.numReads = len(g.lengths)
	if err := ctx.UploadReads(bases, off); err != nil {
		log.Fatal(err)
	}
	if err := ctx.ScanPrepare(int(k)); err != nil {
		log.Fatal(err)
	}
	g.ignore = make([]byte, g.numReads)
	return g
}

// Close releases the device context (the reads, the k-mer index, every per-round buffer).
func (g *GPUReads) Close() { g.ctx.Close() }

type gpuOverlapper struct {
	*overlapper                       // PrepareQueries, SetOverlapSize and the parameters
	*GPUReads
	chunks []*seeds.SeedSequence      // indexed sequences of the round, index == device sequence index
}

// NewGPUOverlapper has NewOverlapper's signature (overlap.go:40) plus the resident reads.
func NewGPUOverlapper(reads *GPUReads, index *seeds.SeedIndex, chunkSize uint, numWorkers int, overlapSize int, minSeeds int, hitFraction float64) Overlapper {
	return &gpuOverlapper{overlapper: &overlapper{index, chunkSize, numWorkers, overlapSize, hitFraction, minSeeds}, GPUReads: reads}
}

// addChunks is chunkWorker (overlap.go:253-318) for one seed sequence; `segBase` is where its segments start in the
// device-resident scan output, so every chunk is a view (offset, seed count) into it.
func (lap *gpuOverlapper) addChunks(s *seeds.SeedSequence, segBase uint64, refs *[]gpu.SeqRef) {
	k := int(lap.index.GetSeedLength())
	add := func(c *seeds.SeedSequence, firstSeed int) {
		lap.chunks = append(lap.chunks, c)
		lap.index.AddDeviceSequence(c)
		*refs = append(*refs, gpu.SeqRef{SegOff: segBase + uint64(2*firstSeed), NSeeds: uint32(c.GetNumSeeds())})
	}
	numChunks := s.Len()/int(lap.chunkSize) + 1
	if numChunks == 1 || s.GetNumSeeds() < lap.minSeeds*3 {
		if s.GetNumSeeds() >= lap.minSeeds {
			add(s, 0)
		}
		return
	}
	prevSeedIndex := 0
	totalOffset := s.GetSeedOffset(0, k)
	lengthInBases := 0
	for {
		seedCount := 0
		if prevSeedIndex >= s.GetNumSeeds()-150 {
			if prevSeedIndex == 0 {
				add(s, 0)
			} else {
				newFirstGap := s.GetNextSeedOffset(prevSeedIndex-1, k) - k
				lengthInBases += s.GetSeedOffsetFromEnd(prevSeedIndex, k) + k + newFirstGap
				add(s.SubSequence(prevSeedIndex, s.GetNumSeeds()-1, lengthInBases, totalOffset-newFirstGap, 0), prevSeedIndex)
			}
			break
		}
		for ; lengthInBases < int(lap.chunkSize) && seedCount < 100 && prevSeedIndex+seedCount < s.GetNumSeeds(); seedCount++ {
			lengthInBases += s.GetNextSeedOffset(prevSeedIndex+seedCount, k)
		}
		if seedCount >= lap.minSeeds {
			newFirstGap := s.GetNextSeedOffset(prevSeedIndex-1, k) - k
			lengthInBases += newFirstGap
			add(s.SubSequence(prevSeedIndex, prevSeedIndex+seedCount-1, lengthInBases, totalOffset-newFirstGap, s.GetLength()-totalOffset-lengthInBases+newFirstGap), prevSeedIndex)
			totalOffset += lengthInBases - newFirstGap
			lengthInBases = 0
			prevSeedIndex += seedCount
			if prevSeedIndex >= s.GetNumSeeds() {
				break
			}
			for seedCount = 0; seedCount < 5 && lengthInBases < lap.overlap/2 && prevSeedIndex > 0; seedCount++ {
				prevSeedIndex--
				step := s.GetNextSeedOffset(prevSeedIndex, k)
				lengthInBases += step
				totalOffset -= step
			}
			lengthInBases = 0
		} else {
			prevSeedIndex += seedCount
			for seedCount = 0; lengthInBases < lap.overlap/2 && prevSeedIndex > 0; seedCount++ {
				prevSeedIndex--
				step := s.GetNextSeedOffset(prevSeedIndex, k)
				lengthInBases += step
				totalOffset -= step
			}
			lengthInBases = 0
		}
	}
}

// AddSequences (overlap.go:217-250): the channel only tells which reads take part (the set skips ignored ones); the bases
// are resident on the device already.
func (lap *gpuOverlapper) AddSequences(seqs <-chan sequence.Sequence) {
	for i := range lap.ignore {
		lap.ignore[i] = 1
	}
	for s := range seqs {
		if s != nil && s.GetID() < lap.numReads {
			lap.ignore[s.GetID()] = 0
		}
	}
	lap.epoch++
	k := int(lap.index.GetSeedLength())
	if err := lap.ctx.RoundBegin(k, lap.index.SeedKmers()); err != nil {
		log.Fatal(err)
	}
	surv, err := lap.ctx.ScanReads(lap.ignore, lap.epoch, 0, lap.numReads, !lap.himem, lap.minSeeds, nil)
	if err != nil {
		log.Fatal(err)
	}
	lap.chunks = lap.chunks[:0]
	refs := make([]gpu.SeqRef, 0, len(surv.Read)*2)
	inset := 0
	if lap.himem {
		inset = 1 // a cached view is SubSequence(0, Len()): its inset is one too large (sequence.go:365)
	}
	for i, r := range surv.Read {
		s := seeds.NewSeedSequenceFromSegments(surv.Segments[i], int(r), lap.names[r], lap.lengths[r], 0, inset)
		lap.addChunks(s, surv.SegOff[i], &refs)
	}
	if err := lap.ctx.IndexBuild(refs); err != nil {
		log.Fatal(err)
	}
}

// FindOverlaps (overlap.go:320-345 + matchWorker :346-387) for all queries in one device call.
func (lap *gpuOverlapper) FindOverlaps(queries []*SeedQuery) <-chan *seeds.SeedMatch {
	output := make(chan *seeds.SeedMatch, lap.numWorkers*2)
	qSegs := make([]int32, 0, len(queries)*64)
	qOff := make([]uint64, 1, len(queries)+1)
	for _, q := range queries {
		for _, v := range q.Query.GetSegments() {
			qSegs = append(qSegs, int32(v))
		}
		qOff = append(qOff, uint64(len(qSegs)))
	}
	matches, err := lap.ctx.FindOverlaps(qSegs, qOff, lap.hitFraction, int(lap.index.GetSeedLength()), lap.overlap/2)
	if err != nil {
		log.Fatal(err)
	}
	go func() {
		for _, m := range matches {
			q := queries[m.Query]
			output <- &seeds.SeedMatch{MatchA: m.MatchA, MatchB: m.MatchB, SeqA: q.Query, SeqB: lap.chunks[m.Target], QueryID: q.ID, ReverseComplementQuery: q.ReverseComplement}
		}
		close(output)
	}()
	return output
}

// KmerValues replaces getKmerValues (commands/overlap.go:39-94) when no seed_values file is given: KmerOccurrences, the
// value per k-mer, the forward + reverse-complement merge and the top-1 % cut, all on the device; bit-identical table.
func (g *GPUReads) KmerValues(k int) []float64 {
	v, err := g.ctx.KmerValues(k)
	if err != nil {
		log.Fatal(err)
	}
	return v
}
