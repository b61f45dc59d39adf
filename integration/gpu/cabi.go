// Package gpu is the cgo face of libdownpore_hip.so (include/downpore_hip.h): thin Go wrappers around the C ABI that the
// GPU-backed overlap.Overlapper (overlap/gpu_overlapper.go) and mapping.Mapper (mapping/gpu_mapper.go) are written against.
// Every slice handed to C is borrowed for the duration of the call only (the library copies), every result is copied out of
// the library's pinned buffers before the call returns, so no Go pointer is retained by C and no C pointer by Go.
//
// Build: CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/downpore_amd/lib -ldownpore_hip -Wl,-rpath,<repo>/downpore_amd/lib"
package gpu

/*
#include <stdlib.h>
#include "downpore_hip.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"
)

// Context owns one HIP stream on one device (dp_ctx).  One goroutine at a time.
type Context struct {
	h      *C.dp_ctx
	parent *Context // a Shared() context borrows its parent's resident reads: the parent stays reachable as long as the child is
	pinned runtime.Pinner // the bases of an UploadReadsRCBegin whose reads still travel (unpinned by WaitReads(-1) and Close)
}

func fail(h *C.dp_ctx, what string, rc C.int) error {
	return errors.New(what + ": " + C.GoString(C.dp_last_error(h)) + " (" + itoa(int(rc)) + ")")
}

func itoa(v int) string {
	if v == 0 {
		return "0"
	}
	neg := v < 0
	if neg {
		v = -v
	}
	var b [24]byte
	i := len(b)
	for v > 0 {
		i--
		b[i] = byte('0' + v%10)
		v /= 10
	}
	if neg {
		i--
		b[i] = '-'
	}
	return string(b[i:])
}

// NewContext creates a context on HIP device `device`.  There is no CPU fallback: without a GPU this fails.
func NewContext(device int) (*Context, error) {
	var h *C.dp_ctx
	if rc := C.dp_ctx_create(C.int(device), &h); rc != 0 {
		return nil, fail(nil, "dp_ctx_create", rc)
	}
	c := &Context{h: h}
	runtime.SetFinalizer(c, func(c *Context) { c.Close() })
	return c, nil
}

// Shared returns a second context on the same device that borrows this one's resident reads (own stream, own per-round
// state): one per goroutine that drives rounds concurrently.
func (c *Context) Shared() (*Context, error) {
	var h *C.dp_ctx
	if rc := C.dp_ctx_create_shared(c.h, &h); rc != 0 {
		return nil, fail(nil, "dp_ctx_create_shared", rc)
	}
	// (the child keeps its parent alive for the garbage collector; should the parent be closed first all the same, the
	// library defers freeing the resident reads until the last borrower has gone - dp_ctx_destroy)
	ch := &Context{h: h, parent: c}
	runtime.SetFinalizer(ch, func(c *Context) { c.Close() })
	return ch, nil
}

// SetStreamWait chooses how calls of the library wait for their stream (process-wide): spinning wakes up at once and occupies
// a core per waiting goroutine, polling costs nothing.  A caller with one goroutine per context and cores to spare wants true.
func SetStreamWait(spin bool) {
	v := C.int(0)
	if spin {
		v = 1
	}
	C.dp_set_stream_wait(v)
}

// SetKernelTiming: each context brackets the kernels of every n-th of its rounds with timing events (0 = never; the kernel
// times of the other rounds read 0).  Every event is a packet of its own for the GPU's command processor: with several
// rounds in flight, timing every round costs 7 % of a job.  Process-wide; the library's default is 8.
func SetKernelTiming(every int) { C.dp_set_kernel_timing(C.int(every)) }

// ReleaseDeviceCaches gives the device blocks and pinned host buffers that destroyed contexts left parked in the library's cache
// back to the driver (they are kept - up to DP_DEV_CACHE_MB / DP_PIN_CACHE_MB - so that the next context does not pay for them
// again).  Returns the bytes released.  gpuhost.ReleaseCaches() calls it too.
func ReleaseDeviceCaches() int64 { return int64(C.dp_release_device_caches()) }

func (c *Context) Close() {
	if c.h != nil {
		C.dp_ctx_destroy(c.h) // (joins an upload thread that is still running)
		c.h = nil
		c.pinned.Unpin()
	}
}

// UploadReads makes the read set resident (2-bit packed on the device): bases = concatenated ASCII, read r = bases[off[r]:off[r+1]].
func (c *Context) UploadReads(bases []byte, off []int64) error {
	if len(off) < 1 {
		return errors.New("UploadReads: empty offset table")
	}
	var bp *C.uint8_t
	if len(bases) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&bases[0]))
	}
	rc := C.dp_reads_upload(c.h, bp, (*C.int64_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(off)-1))
	if rc != 0 {
		return fail(c.h, "dp_reads_upload", rc)
	}
	return nil
}

// UploadReadsRC stores every read r >= firstPaired twice: forward, then reverse-complemented (`map` scans both strands).
func (c *Context) UploadReadsRC(bases []byte, off []int64, firstPaired int) error {
	var bp *C.uint8_t
	if len(bases) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&bases[0]))
	}
	rc := C.dp_reads_upload_rc(c.h, bp, (*C.int64_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(off)-1), C.uint32_t(firstPaired))
	if rc != 0 {
		return fail(c.h, "dp_reads_upload_rc", rc)
	}
	return nil
}

// UploadReadsPackedRC is UploadReadsRC for reads held the way sequence.packedSequence holds them (2 bits per base, first base in a
// byte's top bits): packed = the reads' bytes one after another, every read on a 16-byte boundary (PackedLayout gives the offsets);
// a quarter of UploadReadsRC's bytes cross PCIe.
func (c *Context) UploadReadsPackedRC(packed []byte, lens []uint32, firstPaired int) error {
	if len(lens) == 0 {
		return errors.New("UploadReadsPackedRC: no reads")
	}
	var bp *C.uint8_t
	if len(packed) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&packed[0]))
	}
	rc := C.dp_reads_upload_packed_rc(c.h, bp, (*C.uint32_t)(unsafe.Pointer(&lens[0])), C.uint32_t(len(lens)), C.uint32_t(firstPaired))
	if rc != 0 {
		return fail(c.h, "dp_reads_upload_packed_rc", rc)
	}
	return nil
}

// PackedLayout returns where read r starts in the buffer UploadReadsPackedRC takes (off[len(lens)] = its size).
func PackedLayout(lens []uint32) []int64 {
	off := make([]int64, len(lens)+1)
	for r, l := range lens {
		off[r+1] = off[r] + (int64(l+3)/4+15)&^15
	}
	return off
}

// UploadReadsRCBegin is UploadReadsRC that returns while the reads still travel: the read set's tables are resident and host reads
// [0, readyFirst) packed when it returns, a thread of the library sends the rest on.  Kernels may only be given
// reads a WaitReads has covered.  The library's thread reads `bases` after this call has returned: the binding pins the slice's array
// (runtime.Pinner, Go >= 1.21) until WaitReads(-1) or Close, so that the caller only has to keep the slice unmodified.
func (c *Context) UploadReadsRCBegin(bases []byte, off []int64, firstPaired, readyFirst int) error {
	var bp *C.uint8_t
	if len(bases) > 0 {
		bp = (*C.uint8_t)(unsafe.Pointer(&bases[0]))
		c.pinned.Pin(&bases[0])
	}
	rc := C.dp_reads_upload_rc_begin(c.h, bp, (*C.int64_t)(unsafe.Pointer(&off[0])), C.uint32_t(len(off)-1), C.uint32_t(firstPaired), C.uint32_t(readyFirst))
	if rc != 0 {
		c.pinned.Unpin()
		return fail(c.h, "dp_reads_upload_rc_begin", rc)
	}
	return nil
}

// WaitReads blocks until host reads [0, upTo) of an UploadReadsRCBegin are packed on the device; upTo < 0: the whole set (ends the
// library's upload thread and returns its error, if it had one - call it on the context that owns the reads).
func (c *Context) WaitReads(upTo int) error {
	hi := C.uint32_t(0xffffffff)
	if upTo >= 0 {
		hi = C.uint32_t(upTo)
	}
	rc := C.dp_reads_upload_wait(c.h, hi)
	if upTo < 0 {
		c.pinned.Unpin() // (the upload thread has ended, with or without an error)
	}
	if rc != 0 {
		return fail(c.h, "dp_reads_upload_wait", rc)
	}
	return nil
}

// KmerValues replaces the "Counting all k-mers ... Counting complete" block of the commands: the value table (4^k float64).
func (c *Context) KmerValues(k int) ([]float64, error) {
	out := make([]float64, 1<<uint(2*k))
	if rc := C.dp_kmer_values(c.h, C.int(k), (*C.double)(unsafe.Pointer(&out[0]))); rc != 0 {
		return nil, fail(c.h, "dp_kmer_values", rc)
	}
	return out, nil
}

// ScanPrepare does the one-off work of the rounds' scans (the resident k-mer position index from 1 Gbase up).
func (c *Context) ScanPrepare(k int) error {
	if rc := C.dp_scan_prepare(c.h, C.int(k)); rc != 0 {
		return fail(c.h, "dp_scan_prepare", rc)
	}
	return nil
}

// RoundBegin installs the round's seed set: seed id = position in seedKmers (SeedIndex.seedMap).
func (c *Context) RoundBegin(k int, seedKmers []uint32) error {
	var p *C.uint32_t
	if len(seedKmers) > 0 {
		p = (*C.uint32_t)(unsafe.Pointer(&seedKmers[0]))
	}
	if rc := C.dp_round_begin(c.h, C.int(k), p, C.uint32_t(len(seedKmers))); rc != 0 {
		return fail(c.h, "dp_round_begin", rc)
	}
	return nil
}

// ScanItem is one view the reference's scan examines: k-mer start positions [Start, Start+NKmers) of read Read.
type ScanItem struct {
	Read, Start, NKmers, MinSeeds uint32
}

// Survivors is the outcome of ScanReads: the reads of [lo, hi) with at least minSeeds hits (ascending ids) and the extra
// items, each with its [gap, seed, ..., gap] segments as ints; SegOff are offsets into the DEVICE-resident scan output, which
// IndexBuild refers to.
type Survivors struct {
	Read       []uint32
	Segments   [][]int
	SegOff     []uint64
	Extra      [][]int
	ExtraOff   []uint64
	BasesScanned uint64
}

func cItems(items []ScanItem) *C.dp_scan_item {
	if len(items) == 0 {
		return nil
	}
	return (*C.dp_scan_item)(unsafe.Pointer(&items[0])) // same layout: four uint32
}

func segsOf(segs *C.int32_t, off uint64, nSeeds uint32) []int {
	n := 2*int(nSeeds) + 1
	src := unsafe.Slice((*int32)(unsafe.Pointer(segs)), int(off)+n)[int(off):]
	out := make([]int, n)
	for i := range out {
		out[i] = int(src[i])
	}
	return out
}

// ScanReads is AddSequences' scan (overlap.go:217-250) for every read of [lo, hi) whose ignore byte is 0, plus `extra`.
func (c *Context) ScanReads(ignore []byte, epoch uint64, lo, hi int, topLevel bool, minSeeds int, extra []ScanItem) (*Survivors, error) {
	var b C.dp_survivor_batch
	tl := C.int(0)
	if topLevel {
		tl = 1
	}
	rc := C.dp_scan_reads(c.h, (*C.uint8_t)(unsafe.Pointer(&ignore[0])), C.uint64_t(epoch), C.uint32_t(lo), C.uint32_t(hi), tl,
		C.uint32_t(minSeeds), cItems(extra), C.uint32_t(len(extra)), &b)
	if rc != 0 {
		return nil, fail(c.h, "dp_scan_reads", rc)
	}
	s := &Survivors{BasesScanned: uint64(b.bases_scanned)}
	n := int(b.n_survivors)
	if n > 0 {
		reads := unsafe.Slice((*uint32)(unsafe.Pointer(b.read)), n)
		ns := unsafe.Slice((*uint32)(unsafe.Pointer(b.n_seeds)), n)
		so := unsafe.Slice((*uint64)(unsafe.Pointer(b.seg_off)), n)
		s.Read = append(s.Read, reads...)
		s.SegOff = append(s.SegOff, so...)
		for i := 0; i < n; i++ {
			s.Segments = append(s.Segments, segsOf(b.segs, so[i], ns[i]))
		}
	}
	ne := int(b.n_extra)
	if ne > 0 {
		ns := unsafe.Slice((*uint32)(unsafe.Pointer(b.extra_n_seeds)), ne)
		so := unsafe.Slice((*uint64)(unsafe.Pointer(b.extra_seg_off)), ne)
		s.ExtraOff = append(s.ExtraOff, so...)
		for i := 0; i < ne; i++ {
			s.Extra = append(s.Extra, segsOf(b.segs, so[i], ns[i]))
		}
	}
	return s, nil
}

// Scan is NewSeedSequence for a batch of arbitrary views (the `map` windows and reference chunks).
func (c *Context) Scan(items []ScanItem) (segments [][]int, segOff []uint64, err error) {
	var b C.dp_seedseq_batch
	if rc := C.dp_scan(c.h, cItems(items), C.uint32_t(len(items)), &b); rc != 0 {
		return nil, nil, fail(c.h, "dp_scan", rc)
	}
	n := int(b.n_items)
	if n == 0 {
		return nil, nil, nil
	}
	ns := unsafe.Slice((*uint32)(unsafe.Pointer(b.n_seeds)), n)
	so := unsafe.Slice((*uint64)(unsafe.Pointer(b.seg_off)), n+1)
	for i := 0; i < n; i++ {
		if so[i+1] == so[i] { // below its min_seeds: nothing written
			segments = append(segments, nil)
		} else {
			segments = append(segments, segsOf(b.segs, so[i], ns[i]))
		}
		segOff = append(segOff, so[i])
	}
	return segments, segOff, nil
}

// SeqRef is an indexed sequence: a view into the device-resident scan output.
type SeqRef struct {
	SegOff   uint64
	NSeeds   uint32
	Reserved uint32
}

// IndexBuild is AddSequence + IndexSequences (seeds.go:272-305,372-384) for the chunks of a round.
func (c *Context) IndexBuild(refs []SeqRef) error {
	var p *C.dp_seq_ref
	if len(refs) > 0 {
		p = (*C.dp_seq_ref)(unsafe.Pointer(&refs[0]))
	}
	if rc := C.dp_index_build(c.h, p, C.uint32_t(len(refs))); rc != 0 {
		return fail(c.h, "dp_index_build", rc)
	}
	return nil
}

// Match is one chain: query index, indexed-sequence index, and the matched seed indices on both sides.
type Match struct {
	Query, Target  int
	MatchA, MatchB []int
}

func ints32(p *C.int32_t, from, to uint64) []int {
	src := unsafe.Slice((*int32)(unsafe.Pointer(p)), int(to))[int(from):]
	out := make([]int, len(src))
	for i, v := range src {
		out[i] = int(v)
	}
	return out
}

// FindOverlaps is matchWorker (overlap.go:346-387) for all queries of a round; matches come back in canonical order
// (queries ascending, candidates ascending).  qOff has len(queries)+1 entries into qSegs.
func (c *Context) FindOverlaps(qSegs []int32, qOff []uint64, hitFraction float64, k, maxQueryLen int) ([]Match, error) {
	var b C.dp_match_batch
	var sp *C.int32_t
	if len(qSegs) > 0 {
		sp = (*C.int32_t)(unsafe.Pointer(&qSegs[0]))
	}
	rc := C.dp_find_overlaps(c.h, sp, (*C.uint64_t)(unsafe.Pointer(&qOff[0])), C.uint32_t(len(qOff)-1), C.double(hitFraction), C.int(k),
		C.uint32_t(maxQueryLen), 0, &b)
	if rc != 0 {
		return nil, fail(c.h, "dp_find_overlaps", rc)
	}
	n := int(b.n_matches)
	out := make([]Match, 0, n)
	if n == 0 {
		return out, nil
	}
	q := unsafe.Slice((*uint32)(unsafe.Pointer(b.query)), n)
	t := unsafe.Slice((*uint32)(unsafe.Pointer(b.target)), n)
	off := unsafe.Slice((*uint64)(unsafe.Pointer(b.off)), n+1)
	for i := 0; i < n; i++ {
		out = append(out, Match{Query: int(q[i]), Target: int(t[i]), MatchA: ints32(b.match_a, off[i], off[i+1]), MatchB: ints32(b.match_b, off[i], off[i+1])})
	}
	return out, nil
}

// Chain is one kept chain of performMapping: window index (2i forward, 2i+1 reverse complement), reference chunk, indices.
type Chain struct {
	Window, Target int
	MatchA, MatchB []int
}

// MapWindows is the core of performMapping (mapping.go:489-589) for a batch of (forward, reverse-complement) window pairs.
func (c *Context) MapWindows(wSegs []int32, wOff []uint64, wLen []uint32, k int) ([]Chain, error) {
	var b C.dp_chain_batch
	var sp *C.int32_t
	if len(wSegs) > 0 {
		sp = (*C.int32_t)(unsafe.Pointer(&wSegs[0]))
	}
	rc := C.dp_map_windows(c.h, sp, (*C.uint64_t)(unsafe.Pointer(&wOff[0])), (*C.uint32_t)(unsafe.Pointer(&wLen[0])), C.uint32_t(len(wLen)), C.int(k), &b)
	if rc != 0 {
		return nil, fail(c.h, "dp_map_windows", rc)
	}
	n := int(b.n_chains)
	out := make([]Chain, 0, n)
	if n == 0 {
		return out, nil
	}
	w := unsafe.Slice((*uint32)(unsafe.Pointer(b.window)), n)
	t := unsafe.Slice((*uint32)(unsafe.Pointer(b.target)), n)
	off := unsafe.Slice((*uint64)(unsafe.Pointer(b.off)), n+1)
	for i := 0; i < n; i++ {
		out = append(out, Chain{Window: int(w[i]), Target: int(t[i]), MatchA: ints32(b.match_a, off[i], off[i+1]), MatchB: ints32(b.match_b, off[i], off[i+1])})
	}
	return out, nil
}

// ---- the reference index of `map` held in shards (a 3 Gb reference does not fit one GPU) --------------------------------
// Every shard context has had RoundBegin with the same seeds and IndexBuild on its own contiguous range of reference chunks
// (first chunk id a multiple of 64).

// IndexMeta returns the shard's {count, first word, last word, last+1} per seed (local word numbers).
func (c *Context) IndexMeta(nSeeds int) ([]uint32, error) {
	out := make([]uint32, 4*nSeeds)
	if nSeeds == 0 {
		return out, nil
	}
	if rc := C.dp_index_meta(c.h, (*C.uint32_t)(unsafe.Pointer(&out[0])), C.uint32_t(nSeeds)); rc != 0 {
		return nil, fail(c.h, "dp_index_meta", rc)
	}
	return out, nil
}

// CombineIndexMeta joins the shards' rows into the rows of the whole sets: counts add up, windows in global word numbers
// (wordBase[s] = first chunk of shard s / 64); an empty set keeps NewIntSet()'s start 1 / end 0.
func CombineIndexMeta(perShard [][]uint32, wordBase []uint32, nSeeds int) []uint32 {
	g := make([]uint32, 4*nSeeds)
	for i := 0; i < nSeeds; i++ {
		g[4*i+1], g[4*i+3] = 1, 1
	}
	for s, m := range perShard {
		for i := 0; i < nSeeds; i++ {
			cnt := m[4*i]
			if cnt == 0 {
				continue
			}
			st, en := m[4*i+1]+wordBase[s], m[4*i+2]+wordBase[s]
			if g[4*i] == 0 || st < g[4*i+1] {
				g[4*i+1] = st
			}
			if g[4*i] == 0 || en > g[4*i+2] {
				g[4*i+2] = en
			}
			g[4*i] += cnt
			g[4*i+3] = g[4*i+2] + 1
		}
	}
	return g
}

// IndexSetGlobal installs the whole sets' rows in a shard: its index query then follows util.GetSharedIDs on the global
// windows (early return, drops, 16-ladder gather order) and fills in the candidate words of its own range.
func (c *Context) IndexSetGlobal(metaGlobal []uint32, wordBase, nChunksGlobal int) error {
	if len(metaGlobal) == 0 {
		return nil
	}
	if rc := C.dp_index_set_global(c.h, (*C.uint32_t)(unsafe.Pointer(&metaGlobal[0])), C.uint32_t(len(metaGlobal)/4), C.uint32_t(wordBase), C.uint32_t(nChunksGlobal)); rc != 0 {
		return fail(c.h, "dp_index_set_global", rc)
	}
	return nil
}

// MapWindowsShard is MapWindows for one strand (phase 0 = forward windows, 1 = reverse complements) against one shard.
// thr (len = number of windows, -1 = "the window's own minMatches") carries performMapping's ratchets (mapping.go:543-549,
// 583-586) from shard to shard: call phase 0 on the shards in ascending chunk order, then phase 1 in the same order.
// Chain.Target is the chunk's index inside the shard.
func (c *Context) MapWindowsShard(wSegs []int32, wOff []uint64, wLen []uint32, k, phase int, thr []int32) ([]Chain, error) {
	var b C.dp_chain_batch
	var sp *C.int32_t
	if len(wSegs) > 0 {
		sp = (*C.int32_t)(unsafe.Pointer(&wSegs[0]))
	}
	rc := C.dp_map_windows_shard(c.h, sp, (*C.uint64_t)(unsafe.Pointer(&wOff[0])), (*C.uint32_t)(unsafe.Pointer(&wLen[0])), C.uint32_t(len(wLen)), C.int(k), C.int(phase),
		(*C.int32_t)(unsafe.Pointer(&thr[0])), &b)
	if rc != 0 {
		return nil, fail(c.h, "dp_map_windows_shard", rc)
	}
	n := int(b.n_chains)
	out := make([]Chain, 0, n)
	if n == 0 {
		return out, nil
	}
	w := unsafe.Slice((*uint32)(unsafe.Pointer(b.window)), n)
	t := unsafe.Slice((*uint32)(unsafe.Pointer(b.target)), n)
	off := unsafe.Slice((*uint64)(unsafe.Pointer(b.off)), n+1)
	for i := 0; i < n; i++ {
		out = append(out, Chain{Window: int(w[i]), Target: int(t[i]), MatchA: ints32(b.match_a, off[i], off[i+1]), MatchB: ints32(b.match_b, off[i], off[i+1])})
	}
	return out, nil
}

// Comm is one rank of a multi-GPU job (dp_comm): RCCL across processes, or in-process peers.
type Comm struct {
	h *C.dp_comm
}

// NewLocalComms wires the contexts of ONE process (one per GPU, each driven by its own goroutine) into a communicator.
func NewLocalComms(ctxs []*Context) ([]*Comm, error) {
	if len(ctxs) == 0 {
		return nil, errors.New("NewLocalComms: no contexts")
	}
	hs := make([]*C.dp_ctx, len(ctxs))
	for i, c := range ctxs {
		hs[i] = c.h
	}
	out := make([]*C.dp_comm, len(ctxs))
	if rc := C.dp_comm_init_local((**C.dp_ctx)(unsafe.Pointer(&hs[0])), C.int(len(ctxs)), (**C.dp_comm)(unsafe.Pointer(&out[0]))); rc != 0 {
		return nil, errors.New("dp_comm_init_local failed")
	}
	comms := make([]*Comm, len(ctxs))
	for i := range out {
		comms[i] = &Comm{out[i]}
	}
	return comms, nil
}

func (m *Comm) Close() {
	if m.h != nil {
		C.dp_comm_destroy(m.h)
		m.h = nil
	}
}

// ScanReadsSharded is ScanReads on this rank's read range followed by dp_allgather_survivors: the result describes the
// survivors of ALL ranks in rank order = file order, and the context's device-resident scan output holds exactly them, so
// the IndexBuild / FindOverlaps that follow see what a single GPU scanning every read would.  Collective: every rank calls it.
func (c *Context) ScanReadsSharded(m *Comm, ignore []byte, epoch uint64, lo, hi int, topLevel bool, minSeeds int, extra []ScanItem) (*Survivors, error) {
	var b, g C.dp_survivor_batch
	tl := C.int(0)
	if topLevel {
		tl = 1
	}
	rc := C.dp_scan_reads(c.h, (*C.uint8_t)(unsafe.Pointer(&ignore[0])), C.uint64_t(epoch), C.uint32_t(lo), C.uint32_t(hi), tl,
		C.uint32_t(minSeeds), cItems(extra), C.uint32_t(len(extra)), &b)
	if rc != 0 {
		return nil, fail(c.h, "dp_scan_reads", rc)
	}
	if rc = C.dp_allgather_survivors(m.h, c.h, &b, &g); rc != 0 {
		return nil, fail(c.h, "dp_allgather_survivors", rc)
	}
	s := &Survivors{BasesScanned: uint64(g.bases_scanned)}
	n := int(g.n_survivors)
	if n > 0 {
		reads := unsafe.Slice((*uint32)(unsafe.Pointer(g.read)), n)
		ns := unsafe.Slice((*uint32)(unsafe.Pointer(g.n_seeds)), n)
		so := unsafe.Slice((*uint64)(unsafe.Pointer(g.seg_off)), n)
		s.Read = append(s.Read, reads...)
		s.SegOff = append(s.SegOff, so...)
		for i := 0; i < n; i++ {
			s.Segments = append(s.Segments, segsOf(g.segs, so[i], ns[i]))
		}
	}
	ne := int(g.n_extra)
	if ne > 0 {
		ns := unsafe.Slice((*uint32)(unsafe.Pointer(g.extra_n_seeds)), ne)
		so := unsafe.Slice((*uint64)(unsafe.Pointer(g.extra_seg_off)), ne)
		s.ExtraOff = append(s.ExtraOff, so...)
		for i := 0; i < ne; i++ {
			s.Extra = append(s.Extra, segsOf(g.segs, so[i], ns[i]))
		}
	}
	return s, nil
}

// ---- the entry points a pipelined host needs beyond one synchronous round at a time ------------------------------------------

// QualityUpload makes the reads' FASTQ quality bytes (phred - 33, at the offsets of UploadReads; hasQual[r] = 0: the record had
// no usable quality line) resident: SelectWindows then weights every k-mer's value as AddSeeds does (seeds/seeds.go:99-101).
// Once per read set, before any Shared() context exists.
func (c *Context) QualityUpload(qual []byte, off []int64, hasQual []byte) error {
	rc := C.dp_quality_upload(c.h, (*C.uint8_t)(unsafe.Pointer(&qual[0])), (*C.int64_t)(unsafe.Pointer(&off[0])),
		(*C.uint8_t)(unsafe.Pointer(&hasQual[0])), C.uint32_t(len(off)-1))
	if rc != 0 {
		return fail(c.h, "dp_quality_upload", rc)
	}
	return nil
}

// KmerValuesResident computes the value table on the device and leaves it there (SelectWindows reads it) without copying the
// 4^k doubles to the host; ValueCodes then fetches it at 2 bytes per k-mer.
func (c *Context) KmerValuesResident(k int) error {
	if rc := C.dp_kmer_values(c.h, C.int(k), nil); rc != 0 {
		return fail(c.h, "dp_kmer_values", rc)
	}
	return nil
}

// ValueCodes returns the resident value table as one count code per k-mer (0 = value 0) and the total count: value(i) =
// f(codes[i], total) with overlap.go:73-88's expression.  overflow: a valued k-mer was seen more than 65535 times - fetch the
// doubles with KmerValues instead.
func (c *Context) ValueCodes(k int) (codes []uint16, total uint64, overflow bool, err error) {
	n := 1 << uint(2*k)
	codes = make([]uint16, n)
	var tot C.uint64_t
	var ovf C.int
	if rc := C.dp_values_download_codes(c.h, (*C.uint16_t)(unsafe.Pointer(&codes[0])), C.uint64_t(n), &tot, &ovf); rc != 0 {
		return nil, 0, false, fail(c.h, "dp_values_download_codes", rc)
	}
	return codes, uint64(tot), ovf != 0, nil
}

// ValueCodes8 is ValueCodes with one byte per k-mer (1..254 = the count); overflow: a valued k-mer counts 255 or more - take
// ValueCodes instead (dp_values_download_codes8).
func (c *Context) ValueCodes8(k int) (codes []uint8, total uint64, overflow bool, err error) {
	n := 1 << uint(2*k)
	codes = make([]uint8, n)
	var tot C.uint64_t
	var ovf C.int
	if rc := C.dp_values_download_codes8(c.h, (*C.uint8_t)(unsafe.Pointer(&codes[0])), C.uint64_t(n), &tot, &ovf); rc != 0 {
		return nil, 0, false, fail(c.h, "dp_values_download_codes8", rc)
	}
	return codes, uint64(tot), ovf != 0, nil
}

// SingleSeedCandidates is the parallel part of SeedIndex.AddSingleSeeds (seeds/seeds.go:160-200) for resident read `read`: per window
// of seedRate bases its best-valued k-mer and the k-mers of its count region that are the best of any window.  The caller walks the
// windows in order: a window none of whose candidates is a seed yet adds its best k-mer (dp_single_seed_candidates).
func (c *Context) SingleSeedCandidates(read uint32, k int, seedRate int64) (best []uint32, candOff []uint32, cand []uint32, err error) {
	var b C.dp_single_seed_batch
	if rc := C.dp_single_seed_candidates(c.h, C.uint32_t(read), C.int(k), C.int64_t(seedRate), &b); rc != 0 {
		return nil, nil, nil, fail(c.h, "dp_single_seed_candidates", rc)
	}
	n := int(b.n_windows)
	if n == 0 {
		return nil, []uint32{0}, nil, nil
	}
	best = append([]uint32(nil), unsafe.Slice((*uint32)(unsafe.Pointer(b.best)), n)...)
	candOff = append([]uint32(nil), unsafe.Slice((*uint32)(unsafe.Pointer(b.cand_off)), n+1)...)
	cand = append([]uint32(nil), unsafe.Slice((*uint32)(unsafe.Pointer(b.cand)), int(candOff[n]))...)
	return best, candOff, cand, nil
}

// SelectWindows is the selection half of AddSeeds (seeds/seeds.go:62-129) for many query windows at once, assuming no
// evaluated k-mer is a seed yet: top[w*numSeeds:] = the window's list (untouched slots hold k-mer 0), evaluated[w*stride:] =
// every k-mer the walk evaluated (0xffffffff = unused): what the caller probes against its committed seeds to learn whether
// the assumption held (a window it did not hold for is re-selected on the host, exactly as AddSeeds would).
func (c *Context) SelectWindows(win []ScanItem, k, numSeeds, stride int) (top []uint32, evaluated []uint32, err error) {
	if len(win) == 0 {
		return nil, nil, nil
	}
	top = make([]uint32, len(win)*numSeeds)
	var ev *C.uint32_t
	rc := C.dp_select_windows(c.h, cItems(win), C.uint32_t(len(win)), C.int(k), C.int(numSeeds), (*C.uint32_t)(unsafe.Pointer(&top[0])),
		&ev, C.uint32_t(stride))
	if rc != 0 {
		return nil, nil, fail(c.h, "dp_select_windows", rc)
	}
	evaluated = append(evaluated, unsafe.Slice((*uint32)(unsafe.Pointer(ev)), len(win)*stride)...)
	return top, evaluated, nil
}

// ScanFetchMode(true): ScanReads leaves the surviving reads' segments on the device (IndexBuildChunked chunks them there) and
// brings only the extra items' - the query windows' - to the host.
func (c *Context) ScanFetchMode(extrasOnly bool) error {
	v := C.int(0)
	if extrasOnly {
		v = 1
	}
	if rc := C.dp_scan_fetch_mode(c.h, v); rc != 0 {
		return fail(c.h, "dp_scan_fetch_mode", rc)
	}
	return nil
}

// IndexBuildChunked is chunkWorker (overlap/overlap.go:253-318) + AddSequence + IndexSequences for the first nSurvivors
// survivors of the last ScanReads, all on the device.  Returns an upper bound of the number of chunks.
func (c *Context) IndexBuildChunked(chunkSize, overlap int64, minSeeds, inset, nSurvivors int) (int, error) {
	var cap C.uint32_t
	rc := C.dp_index_build_chunked(c.h, C.int64_t(chunkSize), C.int64_t(overlap), C.uint32_t(minSeeds), C.int32_t(inset), C.uint32_t(nSurvivors), &cap)
	if rc != 0 {
		return 0, fail(c.h, "dp_index_build_chunked", rc)
	}
	return int(cap), nil
}

// IndexPrechain announces the IndexBuildChunked call that follows the next ScanReads: an index-mode scan then launches the chunk
// stage itself, behind its own kernels, and returns as soon as its own output has arrived (dp_index_prechain).
func (c *Context) IndexPrechain(chunkSize, overlap int64, minSeeds, inset int) error {
	if rc := C.dp_index_prechain(c.h, C.int64_t(chunkSize), C.int64_t(overlap), C.uint32_t(minSeeds), C.int32_t(inset)); rc != 0 {
		return fail(c.h, "dp_index_prechain", rc)
	}
	return nil
}

// IndexPrechained reports whether the last ScanReads launched the chunk stage (dp_index_prechained).
func (c *Context) IndexPrechained() bool { return C.dp_index_prechained(c.h) != 0 }

// FindOverlapsOnDevice is matchWorker (overlap.go:346-387) for all queries of the round with the matches left on the device and
// the stage left pending: the next call on the context must be ConsensusPAF, which finishes it in the wait it needs anyway.
func (c *Context) FindOverlapsOnDevice(qSegs []int32, qOff []uint64, hitFraction float64, k, maxQueryLen int) error {
	var b C.dp_match_batch
	rc := C.dp_find_overlaps(c.h, (*C.int32_t)(unsafe.Pointer(&qSegs[0])), (*C.uint64_t)(unsafe.Pointer(&qOff[0])), C.uint32_t(len(qOff)-1),
		C.double(hitFraction), C.int(k), C.uint32_t(maxQueryLen), 6, &b)
	if rc != 0 {
		return fail(c.h, "dp_find_overlaps", rc)
	}
	return nil
}

// QueryPrestage announces the queries of the round's coming FindOverlapsOnDevice before IndexBuildChunked is called: the
// index build's first launch then carries them to the device and the query stage starts with its kernel.  Optional; the
// later FindOverlaps* call must be given slices of the same content.
func (c *Context) QueryPrestage(qSegs []int32, qOff []uint64, hitFraction float64) error {
	if len(qOff) < 2 || len(qSegs) == 0 {
		return nil
	}
	rc := C.dp_query_prestage(c.h, (*C.int32_t)(unsafe.Pointer(&qSegs[0])), (*C.uint64_t)(unsafe.Pointer(&qOff[0])), C.uint32_t(len(qOff)-1),
		C.double(hitFraction))
	if rc != 0 {
		return fail(c.h, "dp_query_prestage", rc)
	}
	return nil
}

// PAFRecord is one PAF line without its names (commands/overlap.go:223-228); Group one query window's lines and SetIgnore ids.
type PAFRecord struct {
	QRead, TRead         uint32
	QLen, QStart, QEnd   int32
	TLen, TStart, TEnd   int32
	Ident                int32
	Minus                uint32
}
type Group struct {
	Lines   []PAFRecord
	Ignore  []uint32
	Flagged bool // the window does not fit the device layout: run BuildConsensus on the host for it (FetchOverlaps)
	Matches int
}

// ConsensusPAF is finalCheckWorker + BuildConsensus + multiAligner.Consensus + trimToBestSeed (commands/overlap.go:197-233,
// overlap/combine.go:21-193, seeds/alignment.go:52-247) for every query window of the round, on the device.  rcOf[s] = seed id of
// seed s's reverse complement; the chunks' fields come from IndexBuildChunked (metas == nil).
func (c *Context) ConsensusPAF(rcOf []int32, k, overlapSize int) ([]Group, int, error) {
	var b C.dp_paf_batch
	rc := C.dp_consensus_paf(c.h, nil, 0, (*C.int32_t)(unsafe.Pointer(&rcOf[0])), C.uint32_t(len(rcOf)), C.int(k), C.int(overlapSize), &b)
	if rc != 0 {
		return nil, 0, fail(c.h, "dp_consensus_paf", rc)
	}
	ng := int(b.n_groups)
	out := make([]Group, ng)
	if ng == 0 {
		return out, int(b.n_indexed), nil
	}
	gm := unsafe.Slice((*C.dp_group_meta)(unsafe.Pointer(b.groups)), ng)
	for g := 0; g < ng; g++ {
		m := gm[g]
		out[g].Flagged = m.flag != 0
		out[g].Matches = int(m.n_matches)
		if m.n_lines > 0 {
			recs := unsafe.Slice((*PAFRecord)(unsafe.Pointer(b.paf)), int(m.slot)+int(m.n_lines))[int(m.slot):] // same layout: ten 32-bit fields
			out[g].Lines = append(out[g].Lines, recs...)
		}
		if m.n_ignore > 0 {
			ids := unsafe.Slice((*uint32)(unsafe.Pointer(b.ignore_ids)), int(m.slot)+int(m.n_ignore))[int(m.slot):]
			out[g].Ignore = append(out[g].Ignore, ids...)
		}
	}
	return out, int(b.n_indexed), nil
}

// ---- multi-GPU, one process per GPU ------------------------------------------------------------------------------------------

// CommUniqueID makes the 128-byte id of an RCCL communicator on rank 0; the caller hands it to the other ranks.
func CommUniqueID() ([]byte, error) {
	id := make([]byte, 128)
	if rc := C.dp_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		return nil, errors.New("dp_comm_unique_id failed (librccl not loadable?)")
	}
	return id, nil
}

// NewComm joins this rank's context to the communicator `id` (RCCL over xGMI; collective: every rank calls it).
func NewComm(c *Context, nRanks, rank int, id []byte) (*Comm, error) {
	if len(id) != 128 {
		return nil, errors.New("NewComm: the id has 128 bytes")
	}
	var h *C.dp_comm
	if rc := C.dp_comm_init(c.h, C.int(nRanks), C.int(rank), (*C.uint8_t)(unsafe.Pointer(&id[0])), &h); rc != 0 {
		return nil, fail(c.h, "dp_comm_init", rc)
	}
	return &Comm{h}, nil
}

// Abort: this rank gives up; its peers' exchanges return an error instead of waiting for it.
func (m *Comm) Abort() {
	if m.h != nil {
		C.dp_comm_abort(m.h)
	}
}

// SetPriority gives the context's stream the device's highest (or default) scheduling priority: for the goroutine that runs the
// PrepareQueries chain and waits for SelectWindows while other contexts keep the GPU busy with whole rounds.
func (c *Context) SetPriority(high bool) error {
	v := C.int(0)
	if high {
		v = 1
	}
	if rc := C.dp_ctx_set_priority(c.h, v); rc != 0 {
		return fail(c.h, "dp_ctx_set_priority", rc)
	}
	return nil
}

// ScanRelease frees what a job left resident on the context that owns the reads (k-mer position index, histogram).
func (c *Context) ScanRelease() error {
	if rc := C.dp_scan_release(c.h); rc != 0 {
		return fail(c.h, "dp_scan_release", rc)
	}
	return nil
}

// ValuesUpload installs a value table computed elsewhere (the -seed_values path) for SelectWindows.
func (c *Context) ValuesUpload(values []float64) error {
	if rc := C.dp_values_upload(c.h, (*C.double)(unsafe.Pointer(&values[0])), C.uint64_t(len(values))); rc != 0 {
		return fail(c.h, "dp_values_upload", rc)
	}
	return nil
}
