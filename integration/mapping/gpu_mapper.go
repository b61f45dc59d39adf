// gpu_mapper.go — goes into package mapping (github.com/jteutenberg/downpore/mapping) next to mapping.go.
//
// A mapping.Mapper whose window scans, index queries and SeedSequence.Match chaining run on an MI355X
// (libdownpore_hip.so through package gpu).  commands/map.go changes one line (:73-85):
//
//	mapper := mapping.NewGPUMapper(reference, seqSet, circular, uint(k), values, seedRate, querySize, chunkSize, numWorkers, 0)
//
// Map's control flow (mapEnds / mapNext / findSplitPoint / matchPairs / removeDominated, mapping.go:112-487) is unchanged: it
// is data dependent per read and stays in Go.  Only performMapping (:489-611) is replaced: a window that Map wants mapped
// is handed to a batcher goroutine, which scans the windows of all reads currently waiting in ONE dp_scan and chains them in
// ONE dp_map_windows, so the GPU sees thousands of windows per call while every Map goroutine still sees a blocking call.
// Start as many MapWorker goroutines as reads should be in flight (a few thousand: they only park on a channel).
package mapping

import (
	"log"
	"sort"
	"sync"

	"github.com/jteutenberg/downpore/gpu"
	"github.com/jteutenberg/downpore/seeds"
	"github.com/jteutenberg/downpore/sequence"
)

type windowRequest struct {
	query sequence.Sequence // a SubSequence view of an uploaded read (GetID() = read id, GetOffset() = start)
	reply chan []*Mapping
}

type gpuMapper struct {
	*mapper                            // reference, index (CPU side: seeds + chunk SeedSequences), edgeSize, circular
	ctx      *gpu.Context
	readBase int                       // device read id of query read 0 (the reference is device read 0)
	requests chan windowRequest
	active   sync.WaitGroup
	waiting  chan int                  // +1 a Map goroutine parked on a request, -1 it left (see batcher)
	workers  int
}

// NewGPUMapper has NewMapper's arguments (mapping.go:67) plus the query read set and the device.  The reference and every
// query read (forward and, produced on the device, reverse complement) become resident in HBM; the reference chunks are
// scanned and indexed on the device in the order NewMapper indexes them.
func NewGPUMapper(reference sequence.Sequence, set sequence.SequenceSet, circular bool, k uint, kmerValues []float64, seedRate int, edgeSize int, chunkSize int, numWorkers int, device int) Mapper {
	m := &mapper{index: seeds.NewSeedIndex(k), reference: reference, edgeSize: edgeSize, circular: circular}
	m.index.AddSingleSeeds(reference, seedRate, kmerValues) // sequential by nature, one pass over the reference
	ctx, err := gpu.NewContext(device)
	if err != nil {
		log.Fatal(err)
	}
	g := &gpuMapper{mapper: m, ctx: ctx, readBase: 1, requests: make(chan windowRequest, 1<<16), waiting: make(chan int, 1<<16), workers: numWorkers}
	m.perform = g.performMapping
	// device read 0 = the reference (with its first edgeSize bases appended when circular: the join chunk is then a plain view),
	// device reads 1 + 2i / 2 + 2i = query read i forward / reverse complement
	bases := []byte(reference.String())
	if circular {
		bases = append(bases, reference.SubSequence(0, edgeSize).String()...)
	}
	off := []int64{0, int64(len(bases))}
	for s := range set.GetSequences() {
		if s == nil {
			continue
		}
		bases = append(bases, s.String()...)
		off = append(off, int64(len(bases)))
	}
	if err := ctx.UploadReadsRC(bases, off, 1); err != nil {
		log.Fatal(err)
	}
	if err := ctx.RoundBegin(int(k), m.index.SeedKmers()); err != nil {
		log.Fatal(err)
	}
	// chunk schedule of NewMapper (:81-97); a chunk is a top-level SubSequence view of the reference
	var items []gpu.ScanItem
	var views []sequence.Sequence
	kk := int(k)
	addChunk := func(start, end int) {
		n := end - start - kk + 1
		if n < 0 {
			n = 0
		}
		items = append(items, gpu.ScanItem{Read: 0, Start: uint32(start), NKmers: uint32(n)})
	}
	for j := 0; j < 10; j++ {
		start := j * chunkSize
		step := chunkSize*10 - edgeSize
		for i := start; i < reference.Len()-chunkSize/2; i += step {
			end := i + chunkSize
			if end > reference.Len() {
				end = reference.Len()
			}
			addChunk(i, end)
			views = append(views, reference.SubSequence(i, end))
		}
	}
	if circular {
		addChunk(reference.Len()-edgeSize, reference.Len()+edgeSize)
		views = append(views, reference.SubSequence(reference.Len()-edgeSize, reference.Len()).Append(0, reference.SubSequence(0, edgeSize), nil))
	}
	segs, segOff, err := ctx.Scan(items)
	if err != nil {
		log.Fatal(err)
	}
	refs := make([]gpu.SeqRef, len(items))
	for i := range items {
		seq := seeds.NewSeedSequenceLike(segs[i], views[i])
		seq.SetID(i)
		m.index.AddDeviceSequence(seq)
		refs[i] = gpu.SeqRef{SegOff: segOff[i], NSeeds: uint32(seq.GetNumSeeds())}
	}
	if err := ctx.IndexBuild(refs); err != nil {
		log.Fatal(err)
	}
	go g.batcher()
	return g
}

// batcher gathers window requests until every Map goroutine that is still running is parked on one, then serves them all.
func (g *gpuMapper) batcher() {
	k := int(g.index.GetSeedLength())
	var pending []windowRequest
	running, parked := 0, 0
	flush := func() {
		if len(pending) == 0 {
			return
		}
		// items 2i / 2i+1: the window on the forward read and the same window on the device's reverse-complement copy
		items := make([]gpu.ScanItem, 0, 2*len(pending))
		for _, r := range pending {
			q := r.query
			id, start, n := q.GetID(), q.GetOffset(), q.Len()-k+1
			if n < 0 {
				n = 0
			}
			full := q.Len() + q.GetOffset() + q.GetInset() // length of the read this window is a view of
			items = append(items, gpu.ScanItem{Read: uint32(g.readBase + 2*id), Start: uint32(start), NKmers: uint32(n)},
				gpu.ScanItem{Read: uint32(g.readBase + 2*id + 1), Start: uint32(full - start - q.Len()), NKmers: uint32(n)})
		}
		segs, _, err := g.ctx.Scan(items)
		if err != nil {
			log.Fatal(err)
		}
		wSegs := make([]int32, 0, 64*len(items))
		wOff := make([]uint64, 1, len(items)+1)
		wLen := make([]uint32, 0, len(items))
		for i := range items {
			for _, v := range segs[i] {
				wSegs = append(wSegs, int32(v))
			}
			wOff = append(wOff, uint64(len(wSegs)))
			wLen = append(wLen, uint32(pending[i/2].query.Len()))
		}
		chains, err := g.ctx.MapWindows(wSegs, wOff, wLen, k)
		if err != nil {
			log.Fatal(err)
		}
		out := make([][]*Mapping, len(pending))
		for _, c := range chains {
			w := c.Window / 2
			q := pending[w].query
			var sq *seeds.SeedSequence
			if c.Window%2 == 0 {
				sq = seeds.NewSeedSequenceLike(segs[c.Window], q)
			} else {
				sq = seeds.NewSeedSequenceLike(segs[c.Window], q.ReverseComplement())
			}
			out[w] = append(out[w], g.mappingFromChain(sq, c, k))
		}
		for i, r := range pending {
			r.reply <- dedupMappings(out[i])
		}
		parked -= len(pending)
		pending = pending[:0]
	}
	for {
		select {
		case d := <-g.waiting:
			running += d
		case r, ok := <-g.requests:
			if !ok {
				flush()
				return
			}
			pending = append(pending, r)
			parked++
		}
		if parked > 0 && parked >= running && len(g.requests) == 0 && len(g.waiting) == 0 {
			flush()
		}
	}
}

// mappingFromChain turns one kept chain into the Mapping performMapping builds for it (mapping.go:527-548, 566-587); the
// 2/3-flank test and both minMatches ratchets were applied on the device (only kept chains come back, in append order).
func (g *gpuMapper) mappingFromChain(sq *seeds.SeedSequence, c gpu.Chain, k int) *Mapping {
	match := g.index.GetSeedSequence(uint(c.Target))
	sm := &seeds.SeedMatch{MatchA: c.MatchA, MatchB: c.MatchB, SeqA: sq, SeqB: match}
	start := match.GetOffset() + match.GetSeedOffset(c.MatchB[0], k)
	end := g.reference.Len() - match.GetInset() - match.GetSeedOffsetFromEnd(c.MatchB[len(c.MatchB)-1], k)
	if g.circular && start > g.reference.Len() {
		start -= g.reference.Len()
	}
	_, ids := sm.GetBasesCovered(k)
	if c.Window%2 == 0 {
		qOffset := sq.GetSeedOffset(c.MatchA[0], k) + sq.GetOffset()
		qInset := sq.GetSeedOffsetFromEnd(c.MatchA[len(c.MatchA)-1], k) + sq.GetInset()
		return &Mapping{Start: start, End: end, QueryOffset: qOffset, QueryInset: qInset, RC: false, match: sm, ids: ids}
	}
	qInset := sq.GetSeedOffset(c.MatchA[0], k) + sq.GetOffset()
	qOffset := sq.GetSeedOffsetFromEnd(c.MatchA[len(c.MatchA)-1], k) + sq.GetInset()
	return &Mapping{Start: start, End: end, QueryOffset: qOffset, QueryInset: qInset, RC: true, match: sm, ids: ids}
}

// dedupMappings is the tail of performMapping (:590-608): sort by Start, of two overlapping same-strand neighbours keep the longer.
func dedupMappings(results []*Mapping) []*Mapping {
	if len(results) > 1 {
		sort.Sort(mappingsByPos(results))
		for i := len(results) - 1; i > 0; i-- {
			ra, rb := results[i-1], results[i]
			if ra.RC == rb.RC && rb.Start < ra.End {
				if ra.End-ra.Start > rb.End-rb.Start {
					results[i] = results[len(results)-1]
					results = results[:len(results)-1]
				} else {
					results[i-1] = results[i]
					results[i] = results[len(results)-1]
					results = results[:len(results)-1]
				}
			}
		}
	}
	return results
}

// performMapping (:489) as Map sees it: a blocking call.
func (g *gpuMapper) performMapping(query sequence.Sequence, aligner seeds.Aligner) []*Mapping {
	r := windowRequest{query: query, reply: make(chan []*Mapping, 1)}
	g.requests <- r
	return <-r.reply
}

// Map (:430) keeps the embedded mapper's control flow (mapEnds / mapNext / findSplitPoint / matchPairs / removeDominated).
// Those methods reach performMapping through the `perform` field this integration adds to `mapper` (INTEGRATION.md:
// `perform func(sequence.Sequence, seeds.Aligner) []*Mapping`, set to m.performMapping by NewMapper and to the batched
// device call below by NewGPUMapper; the ten call sites become m.perform(...)).
func (g *gpuMapper) Map(query sequence.Sequence, aligner seeds.Aligner) []*Mapping {
	g.waiting <- 1
	defer func() { g.waiting <- -1 }()
	return g.mapper.Map(query, aligner)
}

func (g *gpuMapper) MapWorker(queries <-chan sequence.Sequence, results chan<- []*Mapping, done chan<- bool) {
	aligner := seeds.NewSeedAligner(g.edgeSize)
	for query := range queries {
		results <- g.Map(query, aligner)
	}
	done <- true
}
