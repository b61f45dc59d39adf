import sys, json, time, hashlib
sys.path.insert(0, ".")
import numpy as np
from tools.synth import gen_genome, gen_reads
from downpore_amd.mapping import map_reads
from downpore_amd.overlap import Reads
g = json.load(open("tests/golden_full/config3_map.json"))
genome = np.frombuffer(gen_genome(3, 4600000), dtype=np.uint8); goff = np.array([0, 4600000], dtype=np.int64)
bases, off = gen_reads(3, 4600000, 50000, 8000, 0.1, False)
ref = Reads(genome, goff, min_len=0, himem=False); reads = Reads(bases, off, min_len=500, himem=False)
best = 1e9
for i in range(4):
    t0 = time.perf_counter(); paf, err, st = map_reads(ref, reads, circular=True, k=11); dt = time.perf_counter() - t0
    best = min(best, dt)
print(round(50000/best), "reads/s", round(best,3), hashlib.sha256(paf.encode()).hexdigest() == g["paf_sha256"], {k: round(v,3) for k,v in st.items() if k.startswith("t_")}, st["n_batches"])
