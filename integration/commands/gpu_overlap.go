package commands

// `downpore overlap` on the GPU at the measured rate: the same flag table, the same stderr lines, the same PAF lines on stdout
// as commands/overlap.go - with the round loop of Run (:119-195) and everything under it (PrepareQueries, AddSequences,
// FindOverlaps, the collation, finalCheckWorker / BuildConsensus) running behind dph_overlap_step in libdownpore_host.so
// (include/downpore_host.h; package gpuhost).  Goes to commands/gpu_overlap.go of the reference tree; downpore.go registers
// NewGPUOverlapCommand() in place of NewOverlapCommand() (or next to it, under another name).
//
// num_workers only serves the -seed_values path (the GPU path is batch-parallel).  seed_values: the table comes from the
// reference's own getKmerValues (file + the counts of this input); without it the library counts k-mers and builds the table
// on the device (commands/overlap.go:39-94, bit-identical).

import (
	"bufio"
	"errors"
	"log"
	"os"
	"time"

	"github.com/jteutenberg/downpore/gpuhost"
	"github.com/jteutenberg/downpore/sequence"
)

type gpuOverlapCommand struct {
	args  map[string]string
	alias map[string]string
	desc  map[string]string
}

func NewGPUOverlapCommand() Command {
	args, alias, desc := MakeArgs(
		[]string{"overlap_size", "k", "num_seeds", "seed_batch_size", "chunk_size", "query_batch_size", "min_hits", "num_workers", "input", "seed_values", "himem", "gpu", "slots", "ranks", "rank", "comm_id"},
		[]string{"1000", "10", "15", "10000", "10000", "20000", "0.25", "4", "", "", "true", "0", "8", "1", "0", ""},
		[]string{"Size of overlap to search for in bases", "Number of bases in each seed", "Minimum number of seeds to generate for each overlap query", "Maximum total unique seeds to use in each query batch", "Size to chop long reads into for querying against, in bases", "Maximum number of queries per batch (if max seeds not reached)", "Minimum proportion of seeds that must match each query", "Number of worker threads to spawn", "Fasta/fastq input file", "File containing values to use during seed selection.", "Whether to cache all reads in memory", "HIP device to run on", "Rounds in flight on the GPU", "Processes of this job (one per GPU)", "This process's rank", "File through which rank 0 hands the 128-byte RCCL id to the other ranks"})
	ov := gpuOverlapCommand{args: args, alias: alias, desc: desc}
	return &ov
}

func (com *gpuOverlapCommand) GetName() string {
	return "overlap"
}

func (com *gpuOverlapCommand) GetArgs() (map[string]string, map[string]string, map[string]string) {
	return com.args, com.alias, com.desc
}

func (com *gpuOverlapCommand) Run(args map[string]string) {
	p := gpuhost.OverlapParams{
		OverlapSize:    ParseInt(args["overlap_size"]),
		K:              ParseInt(args["k"]),
		NumSeeds:       ParseInt(args["num_seeds"]),
		SeedBatchSize:  ParseInt(args["seed_batch_size"]),
		ChunkSize:      ParseInt(args["chunk_size"]),
		QueryBatchSize: ParseInt(args["query_batch_size"]),
		MinHits:        ParseFloat(args["min_hits"]),
		Himem:          ParseBool(args["himem"]),
		QueryType:      1, // overlap.QueryEdges
		Slots:          ParseInt(args["slots"]),
	}
	// sequence.NewFastaSequenceSet(args["input"], overlapSize, numWorkers, himem, false) (:108): the library's reader follows the
	// same rules (a line is a sequence iff it starts in ['A','T'], kept iff len(line) >= overlapSize)
	reads, err := gpuhost.ReadsFromFile(args["input"], p.OverlapSize, p.Himem)
	if err != nil {
		log.Fatal(err)
	}
	defer reads.Close()
	var values []float64
	if args["seed_values"] != "" {
		// a table from a file still needs the k-mer counts of THIS input (entries of k-mers seen fewer than 3 times and of the
		// top 1 % become 0, commands/overlap.go:73-93): the reference's own getKmerValues does all of that on the CPU path
		seqSet := sequence.NewFastaSequenceSet(args["input"], p.OverlapSize, ParseInt(args["num_workers"]), p.Himem, false)
		values = getKmerValues(args["seed_values"], p.K, ParseInt(args["num_workers"]), seqSet)
		if values == nil {
			return
		}
	}
	ov, err := gpuhost.OpenOverlap(reads, ParseInt(args["gpu"]))
	if err != nil {
		log.Fatal(err)
	}
	defer ov.Close()
	ranks, rank := ParseInt(args["ranks"]), ParseInt(args["rank"])
	if ranks > 1 {
		// one process per GPU: the communicator has to exist before Init
		id, err := exchangeCommID(args["comm_id"], rank)
		if err != nil {
			log.Fatal(err)
		}
		if err := ov.InitComm(ranks, rank, id); err != nil {
			log.Fatal(err)
		}
	}
	if err := ov.Init(p, values); err != nil {
		log.Fatal(err)
	}
	out := bufio.NewWriterSize(os.Stdout, 1<<20)
	defer out.Flush()
	if ranks > 1 {
		// query batches dealt to the ranks: rank r plans and runs the rounds r, r + ranks, ...; every superstep all-gathers the
		// finished rounds over RCCL and commits them in order on every rank; rank 0 prints (the others drop the text as it arrives)
		ov.SetRanks(rank, ranks)
		ov.KeepText(rank == 0)
		if os.Getenv("DP_TEXT_ROOT") == "1" {
			// opt-in until a two-GPU run has shown parity (ncclSend / ncclRecv have not run with a peer yet): the text goes to the
			// printing rank alone, the control records (a few KB per round) to everybody
			ov.TextRoot(0)
		}
		err := ov.RunRoundParallel(p.Slots, func(paf []byte) {
			if rank == 0 {
				out.Write(paf)
			}
		})
		if err != nil {
			log.Fatal(err)
		}
		if rank == 0 {
			os.Stderr.WriteString(ov.ErrText())
		}
		return
	}
	printed := 0
	for {
		n, err := ov.Step()
		if err != nil {
			log.Fatal(err)
		}
		// stderr: "Counting all k-mers ...", "Using query set ...", "Total ... hits across ... overlaps." - as they accumulate
		if e := ov.ErrText(); len(e) > printed {
			os.Stderr.WriteString(e[printed:])
			printed = len(e)
		}
		if n == 0 {
			break
		}
		out.Write(ov.RoundPAF()) // fmt.Print(s) of finalCheckWorker (:225-228), query order
	}
}

// exchangeCommID: rank 0 makes the RCCL id and publishes it through a file (written under another name and renamed, so a reader
// never sees half of it); the other ranks wait for the file.  Any other transport of 128 bytes does as well.
func exchangeCommID(path string, rank int) ([]byte, error) {
	if path == "" {
		return nil, errors.New("-ranks > 1 needs -comm_id <file>")
	}
	if rank == 0 {
		id, err := gpuhost.CommUniqueID()
		if err != nil {
			return nil, err
		}
		if err := os.WriteFile(path+".tmp", id, 0600); err != nil {
			return nil, err
		}
		return id, os.Rename(path+".tmp", path)
	}
	for i := 0; i < 6000; i++ {
		if id, err := os.ReadFile(path); err == nil && len(id) == 128 {
			return id, nil
		}
		time.Sleep(10 * time.Millisecond)
	}
	return nil, errors.New("no RCCL id in " + path + " after 60 s")
}
