// Package gpuhost is the cgo face of libdownpore_host.so (include/downpore_host.h): the whole `downpore overlap` /
// `downpore map` pipeline behind four calls - open, init, step, paf.  This is the path bench.py measures (BENCH_r*.json): the
// planner lanes, the window cache, the executor slots, the consensus and the PAF numbers on the device all live behind
// dph_overlap_step; a command loop written against package gpu (one synchronous round at a time) gets the kernels but not that
// rate.  commands/gpu_overlap.go and commands/gpu_map.go are the two commands written against this package.
//
// Build: CGO_CFLAGS="-I<repo>/include"
//        CGO_LDFLAGS="-L<repo>/downpore_amd/lib -ldownpore_host -ldownpore_hip -Wl,-rpath,<repo>/downpore_amd/lib"
//
// Every slice handed to C is borrowed for the duration of the call (the library copies); text returned by the library is copied
// into Go strings / byte slices before the call returns.  One goroutine per handle.
package gpuhost

/*
#include <stdlib.h>
#include "downpore_host.h"
*/
import "C"

import (
	"errors"
	"runtime"
	"unsafe"
)

func lastError(h unsafe.Pointer, what string) error {
	return errors.New(what + ": " + C.GoString(C.dph_last_error(h)))
}

// Reads is a read set in the library's memory (sequence.NewFastaSequenceSet's rules: one-line FASTA / FASTQ records, ids in
// file order, minLen as the commands pass it).
type Reads struct {
	h unsafe.Pointer
}

// ReadsFromFile reads a FASTA / FASTQ file (sequence/seqio.go:106-276).  himem = the -himem flag of `overlap`; `map` opens both
// of its files with false.
func ReadsFromFile(path string, minLen int, himem bool) (*Reads, error) {
	cp := C.CString(path)
	defer C.free(unsafe.Pointer(cp))
	hm := C.int(0)
	if himem {
		hm = 1
	}
	h := C.dph_reads_from_fasta(cp, C.int64_t(minLen), hm)
	if h == nil {
		return nil, lastError(nil, "dph_reads_from_fasta")
	}
	r := &Reads{h}
	runtime.SetFinalizer(r, func(r *Reads) { r.Close() })
	return r, nil
}

// ReadsFromBases wraps sequences the caller already holds (a sequence.SequenceSet drained into one byte slice): read i =
// bases[off[i]:off[i+1]]; quals (may be nil) = raw FASTQ quality characters at the same offsets.
func ReadsFromBases(bases []byte, quals []byte, off []int64, minLen int, himem bool) (*Reads, error) {
	if len(off) < 1 {
		return nil, errors.New("ReadsFromBases: empty offset table")
	}
	hm := C.int(0)
	if himem {
		hm = 1
	}
	var bp, qp *C.char
	if len(bases) > 0 {
		bp = (*C.char)(unsafe.Pointer(&bases[0]))
	}
	var h unsafe.Pointer
	if quals != nil && len(quals) == len(bases) && len(quals) > 0 {
		qp = (*C.char)(unsafe.Pointer(&quals[0]))
		h = C.dph_reads_from_arrays_q(bp, qp, (*C.int64_t)(unsafe.Pointer(&off[0])), C.int64_t(len(off)-1), C.int64_t(minLen), hm)
	} else {
		h = C.dph_reads_from_arrays(bp, (*C.int64_t)(unsafe.Pointer(&off[0])), C.int64_t(len(off)-1), C.int64_t(minLen), hm)
	}
	if h == nil {
		return nil, lastError(nil, "dph_reads_from_arrays")
	}
	r := &Reads{h}
	runtime.SetFinalizer(r, func(r *Reads) { r.Close() })
	return r, nil
}

func (r *Reads) Close() {
	if r.h != nil {
		C.dph_reads_free(r.h)
		r.h = nil
	}
}
func (r *Reads) Size() int       { return int(C.dph_reads_count(r.h)) }
func (r *Reads) TotalBases() int { return int(C.dph_reads_total_bases(r.h)) }

// Ignored returns the SetIgnore flags (sequence/seqio.go:375) as the last job left them.
func (r *Reads) Ignored() []bool {
	n := r.Size()
	raw := make([]byte, n)
	if n > 0 {
		C.dph_reads_get_ignore(r.h, (*C.uint8_t)(unsafe.Pointer(&raw[0])))
	}
	out := make([]bool, n)
	for i, b := range raw {
		out[i] = b != 0
	}
	return out
}

// OverlapParams is the flag table of `downpore overlap` (commands/overlap.go:24-25).
type OverlapParams struct {
	OverlapSize, K, NumSeeds, SeedBatchSize, ChunkSize, QueryBatchSize int
	MinHits                                                              float64
	Himem                                                                bool
	QueryType                                                            int // overlap.QueryEdges (1) for the overlap command
	Slots                                                                int // rounds in flight on the GPU (8 = what bench.py uses)
}

// Overlap is one `downpore overlap` command on one GPU: reads resident in HBM, any number of jobs on them.
type Overlap struct {
	h     unsafe.Pointer
	reads *Reads // (keeps the read set alive: the handle points into it)
}

// OpenOverlap creates the device context and uploads + packs the reads (FASTQ qualities travel with them).
func OpenOverlap(reads *Reads, device int) (*Overlap, error) {
	h := C.dph_overlap_open(reads.h, C.int(device))
	if h == nil {
		return nil, lastError(nil, "dph_overlap_open")
	}
	o := &Overlap{h, reads}
	runtime.SetFinalizer(o, func(o *Overlap) { o.Close() })
	return o, nil
}

// Init does everything the command does between "Counting all k-mers" and its first round: k-mer position index, k-mer value
// table (computed on the device unless values - 4^k entries, the -seed_values table - is given), executor slots, planner.
func (o *Overlap) Init(p OverlapParams, values []float64) error {
	hm := int64(0)
	if p.Himem {
		hm = 1
	}
	qt := p.QueryType
	if qt == 0 {
		qt = 1
	}
	slots := p.Slots
	if slots < 1 {
		slots = 8
	}
	params := [8]C.int64_t{C.int64_t(p.OverlapSize), C.int64_t(p.K), C.int64_t(p.NumSeeds), C.int64_t(p.SeedBatchSize), C.int64_t(p.ChunkSize),
		C.int64_t(p.QueryBatchSize), C.int64_t(hm | int64(qt)<<8), C.int64_t(slots)}
	var vp *C.double
	if len(values) > 0 {
		vp = (*C.double)(unsafe.Pointer(&values[0]))
	}
	if rc := C.dph_overlap_init(o.h, &params[0], C.double(p.MinHits), vp); rc != 0 {
		return lastError(o.h, "dph_overlap_init")
	}
	return nil
}

// Step commits the next finished round(s) in round order and returns how many (0: the command is finished).
func (o *Overlap) Step() (int, error) {
	rc := C.dph_overlap_step(o.h)
	if rc < 0 {
		return 0, lastError(o.h, "dph_overlap_step")
	}
	return int(rc), nil
}

func text(p *C.char, n C.int64_t) []byte {
	if n == 0 {
		return nil
	}
	return C.GoBytes(unsafe.Pointer(p), C.int(n))
}

// RoundPAF returns the PAF lines of the rounds the last Step committed (commands/overlap.go:225's lines, query order).
func (o *Overlap) RoundPAF() []byte {
	var n C.int64_t
	p := C.dph_overlap_round_paf(o.h, &n)
	return text(p, n)
}

// ErrText returns the reference's stderr progress lines accumulated so far.
func (o *Overlap) ErrText() string {
	var n C.int64_t
	p := C.dph_overlap_errtext(o.h, &n)
	return string(text(p, n))
}

func (o *Overlap) Done() bool   { return C.dph_overlap_done(o.h) != 0 }

// KeepText(false) on a rank that does not print the PAF: gathered rounds are committed without the other ranks' text.
func (o *Overlap) KeepText(keep bool) {
	v := C.int(0)
	if keep {
		v = 1
	}
	C.dph_overlap_keep_text(o.h, v)
}
func (o *Overlap) Rounds() int  { return int(C.dph_overlap_round(o.h)) }
func (o *Overlap) StepLines() int { return int(C.dph_overlap_step_lines(o.h)) }

// Values returns the job's k-mer value table (commands/overlap.go:55-93), e.g. to write a -seed_values file.
func (o *Overlap) Values() []float64 {
	var n C.int64_t
	p := C.dph_overlap_values(o.h, &n)
	out := make([]float64, int(n))
	if n > 0 {
		copy(out, unsafe.Slice((*float64)(unsafe.Pointer(p)), int(n)))
	}
	return out
}

// Reset ends the job and keeps the handle (reads resident, executor contexts kept): Init starts the next one.
func (o *Overlap) Reset() error {
	if rc := C.dph_overlap_reset(o.h); rc != 0 {
		return lastError(o.h, "dph_overlap_reset")
	}
	return nil
}

func (o *Overlap) Close() {
	if o.h != nil {
		C.dph_overlap_destroy(o.h)
		o.h = nil
	}
}

// ReleaseCaches gives back the host buffers the library keeps between jobs (PAF text, window-cache chunks, the map command's
// staging block: up to ~0.9 GB after a config-2 overlap job and a config-3 map job) and, since round 5, the device blocks and
// pinned buffers that destroyed contexts left parked in the device library (dp_release_device_caches).  Returns the bytes released.
func ReleaseCaches() int64 { return int64(C.dph_release_caches()) }

// ---- multi-GPU, one process per GPU: the north-star layout (reads partitioned, survivors' seed index all-gathered) ----------

// CommUniqueID makes the 128-byte id of an RCCL communicator (rank 0 calls it once per executor slot and hands the bytes to
// every rank by whatever means the host has - MPI, a file, a socket).
func CommUniqueID() ([]byte, error) {
	id := make([]byte, 128)
	if rc := C.dph_comm_unique_id((*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		return nil, errors.New("dph_comm_unique_id failed (librccl not loadable?)")
	}
	return id, nil
}

// InitComms joins this rank to one communicator per executor slot (ids = slots x 128 bytes, the same on every rank) and gives
// it its read range [lo, hi) - ascending contiguous ranges in rank order.  Call before Init.
func (o *Overlap) InitComms(nRanks, rank int, ids []byte, slots, lo, hi int) error {
	if len(ids) < 128*slots {
		return errors.New("InitComms: need 128 bytes of id per executor slot")
	}
	if rc := C.dph_overlap_comm_init_slots(o.h, C.int(nRanks), C.int(rank), (*C.uint8_t)(unsafe.Pointer(&ids[0])), C.int(slots)); rc != 0 {
		return lastError(o.h, "dph_overlap_comm_init_slots")
	}
	C.dph_overlap_set_shard(o.h, C.int64_t(lo), C.int64_t(hi))
	return nil
}

// StepSharded runs one round per executor slot of this rank, concurrently, each exchanging its survivors on its own
// communicator, and commits them in order (collective: every rank calls it).  Returns the rounds committed, 0 = finished.
func (o *Overlap) StepSharded() (int, error) {
	rc := C.dph_overlap_rounds_sharded(o.h)
	if rc < 0 {
		return 0, lastError(o.h, "dph_overlap_rounds_sharded")
	}
	return int(rc), nil
}

// ---- multi-GPU, one process per GPU: query batches dealt to the ranks (what bench.py --gpus N leads with) ---------------------
//
// Every rank holds the reads and the k-mer position index; rank r plans and runs the rounds r, r + world, ...; a superstep
// all-gathers the ranks' finished rounds over RCCL inside the library and commits them in round order on every rank (a round
// whose speculation on the ignore flags failed is rejected by all ranks alike and runs again).  HISTORY.md 7.2.

// InitComm joins this rank to ONE communicator (id = the 128 bytes of CommUniqueID, the same on every rank): the result
// exchange of Superstep runs on it.  Call before Init.
func (o *Overlap) InitComm(nRanks, rank int, id []byte) error {
	if len(id) < 128 {
		return errors.New("InitComm: need the 128 bytes of CommUniqueID")
	}
	if rc := C.dph_overlap_comm_init(o.h, C.int(nRanks), C.int(rank), (*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		return lastError(o.h, "dph_overlap_comm_init")
	}
	return nil
}

// TextRoot(root): supersteps gather the rounds' PAF text to rank root alone and all-gather only their control records (every rank
// must say the same, before the first Superstep); the default (-1) sends text and control to every rank.
func (o *Overlap) TextRoot(root int) { C.dph_overlap_text_root(o.h, C.int(root)) }

// SetRanks deals the rounds to the ranks (round r belongs to rank r % world).  Call after Init, before the first Superstep.
func (o *Overlap) SetRanks(rank, world int) {
	C.dph_overlap_set_ranks(o.h, C.int(rank), C.int(world))
}

// Superstep contributes up to maxRounds of this rank's finished rounds, exchanges with the other ranks and commits (collective:
// every rank calls it).  Returns the rounds committed; 0 with Done() false = the superstep's first round was rejected and is
// being executed again - call again.
func (o *Overlap) Superstep(maxRounds int) (int, error) {
	rc := C.dph_overlap_superstep(o.h, C.int(maxRounds))
	if rc < 0 {
		return 0, lastError(o.h, "dph_overlap_superstep")
	}
	return int(rc), nil
}

// RunRoundParallel is the whole round loop of one rank: supersteps until the command is finished, emit(RoundPAF()) after each
// one that committed something (only on a rank that called KeepText(true), i.e. the printing rank).
func (o *Overlap) RunRoundParallel(slots int, emit func(paf []byte)) error {
	if slots < 1 {
		slots = 1
	}
	for !o.Done() {
		n, err := o.Superstep(slots)
		if err != nil {
			return err
		}
		if n > 0 && emit != nil {
			emit(o.RoundPAF())
		}
	}
	return nil
}

// ---- `downpore map` ---------------------------------------------------------------------------------------------------------

// MapParams is the flag table of `downpore map` (commands/map.go:19-20).
type MapParams struct {
	Circular                                           bool
	K, QuerySize, MinLength, ChunkSize, SeedRate int
}

// MapResult holds what the command prints: the PAF lines (read order) and the stderr summary.
type MapResult struct {
	PAF     []byte
	ErrText string
}

// RunMap runs the whole command on HIP device `device`: reference = first sequence of ref (opened with himem = false, minLen 0),
// reads opened with minLen = -min_length.
func RunMap(ref, reads *Reads, p MapParams, device int) (*MapResult, error) {
	circ := C.int64_t(0)
	if p.Circular {
		circ = 1
	}
	params := [6]C.int64_t{circ, C.int64_t(p.K), C.int64_t(p.QuerySize), C.int64_t(p.MinLength), C.int64_t(p.ChunkSize), C.int64_t(p.SeedRate)}
	h := C.dph_map_run(ref.h, reads.h, &params[0], C.int(device))
	if h == nil {
		return nil, lastError(nil, "dph_map_run")
	}
	defer C.dph_map_free(h)
	var n C.int64_t
	pp := C.dph_map_paf(h, &n)
	res := &MapResult{PAF: text(pp, n)}
	pe := C.dph_map_errtext(h, &n)
	res.ErrText = string(text(pe, n))
	return res, nil
}
