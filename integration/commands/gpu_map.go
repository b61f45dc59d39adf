package commands

// `downpore map` on the GPU: the same flag table, stdout and stderr as commands/map.go, with NewMapper (mapping/mapping.go:67-109),
// the MapWorker pool and performMapping (:489-611) behind dph_map_run (include/downpore_host.h; package gpuhost).  Goes to
// commands/gpu_map.go of the reference tree; downpore.go registers NewGPUMapCommand() in place of NewMapCommand().
// DP_MAP_SHARDS / DP_MAP_DEVICES in the environment spread the reference index over several GPUs (a 3 Gb reference).

import (
	"log"
	"os"

	"github.com/jteutenberg/downpore/gpuhost"
)

type gpuMapCommand struct {
	args  map[string]string
	alias map[string]string
	desc  map[string]string
}

func NewGPUMapCommand() Command {
	args, alias, desc := MakeArgs(
		[]string{"input", "reference", "circular", "k", "query_size", "min_length", "chunk_size", "seed_rate", "num_workers", "gpu"},
		[]string{"", "", "true", "11", "1000", "500", "10000", "40", "4", "0"},
		[]string{"Fasta/fastq input file", "A fasta file containing a reference sequence to align against", "Whether the reference genome is circular", "Length of seeds in bases", "The number of bases to query at a time", "The minimum sequence size to generate queries from", "The number of bases for reference index chunks", "The maximum number of bases between seeds in the reference", "The number of worker process to use for mapping", "HIP device to run on"})
	cons := gpuMapCommand{args: args, alias: alias, desc: desc}
	return &cons
}

func (com *gpuMapCommand) GetName() string {
	return "map"
}

func (com *gpuMapCommand) GetArgs() (map[string]string, map[string]string, map[string]string) {
	return com.args, com.alias, com.desc
}

func (com *gpuMapCommand) Run(args map[string]string) {
	// sequence.NewFastaSequenceSet(args["reference"], 0, 1, false, false) / (args["input"], minLength, 1, false, false) (:34, :75)
	ref, err := gpuhost.ReadsFromFile(args["reference"], 0, false)
	if err != nil {
		log.Fatal(err)
	}
	defer ref.Close()
	minLength := ParseInt(args["min_length"])
	reads, err := gpuhost.ReadsFromFile(args["input"], minLength, false)
	if err != nil {
		log.Fatal(err)
	}
	defer reads.Close()
	p := gpuhost.MapParams{
		Circular:  ParseBool(args["circular"]),
		K:         ParseInt(args["k"]),
		QuerySize: ParseInt(args["query_size"]),
		MinLength: minLength,
		ChunkSize: ParseInt(args["chunk_size"]),
		SeedRate:  ParseInt(args["seed_rate"]),
	}
	res, err := gpuhost.RunMap(ref, reads, p, ParseInt(args["gpu"]))
	if err != nil {
		log.Fatal(err)
	}
	os.Stdout.Write(res.PAF)          // fmt.Println(mapper.AsString(m)) per mapping (:92), read order
	os.Stderr.WriteString(res.ErrText) // "K-mer counting complete ...", "Uniquely mapped: ..." (:69, :112-115)
}
