"""CPU test of the product's FASTA reader (ReadSet::fromFile, mmap + memchr) against the oracle's restatement of
readFasta (sequence/seqio.go:106-200): same reads, same names, same 2-bit content, on well-formed and awkward files."""
import ctypes as C

import pytest

from tests import oracle_lib as O

CASES = {
    "plain": b">r1\nACGTACGTAC\n>r2\nTTTTGGGGCC\n",
    "no_trailing_newline": b">r1\nACGTACGTAC\n>r2\nTTTTGGGGCC",
    "blank_lines": b">r1\n\nACGTACGTAC\n\n>r2\nTTTTGGGGCC\n\n",
    "multi_line_record": b">r1 some description\nACGTACGTAC\nGGGGGGGGGG\nTT\n>r2\nCCCCCCCCCCCC\n",
    "lowercase_and_n": b">r1\nacgtacgtac\nNNNNACGTNN\n>r2\nAnCgT\n",
    "crlf": b">r1\r\nACGTACGTAC\r\n>r2\r\nTTTTGGGGCC\r\n",
    "no_header_first": b"ACGTACGTAC\n>r2\nTTTTGGGGCC\n",
    "only_headers": b">r1\n>r2\n>r3\n",
    "single_line_no_newline": b">r1",
    "empty": b"",
    "header_with_spaces": b">  r1  extra \nACGTACGTACGT\n>\tr2\t\nGGGGCCCCAAAA\n",
    "short_and_long": b">a\nACG\n>b\nACGTACGTACGTACGTACGTACGTACGTACGT\n>c\nA\n>d\nTTTTTTTTTTTTTTTTTTTTTTTTT\n",
    "fastq_like": b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nTTTTGGGGCC\n+\nFFFFFFFFFF\n",
    # FASTQ (sequence/seqio.go:208-267): '+' and quality lines are consumed with their record, kept read or not; a quality
    # line counts only when it is exactly one byte longer than the sequence; the bytes are stored minus 33 (modulo 256)
    "fastq_last_quality_without_newline": b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nTTTTGGGGCC\n+\nFFFFFFFFFF",
    "fastq_quality_length_mismatch": b"@r1\nACGTACGTAC\n+\nIIII\n@r2\nTTTTGGGGCC\n+r2 again\nFFFFFFFFFFFF\n@r3\nGGGGGGGGGGGG\n+\n!!!!!!!!!!!!\n",
    "fastq_quality_starts_like_other_lines": b"@r1\nACGTACGTAC\n+\n@AT+>AT@+>\n@r2\nTTTTGGGGCC\n+\nACGTACGTAC\n",
    "fastq_short_reads_skipped": b"@a\nACG\n+\nIII\n@b\nACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIII\n@c\nAC\n+\nII\n",
    "fastq_low_bytes_wrap": b"@r1\nACGTACGTAC\n+\n \x01\x02!\"#$%&~\n",
    "fastq_after_fasta_records": b">r1\nACGTACGTAC\n@r2\nTTTTGGGGCC\n+\nFFFFFFFFFF\n",
    "fastq_truncated_after_sequence": b"@r1\nACGTACGTAC\n+\nIIIIIIIIII\n@r2\nTTTTGGGGCC\n",
    "fastq_bad_plus_line": b"@r1\nACGTACGTAC\nIIIIIIIIII\n@r2\nTTTTGGGGCC\n+\nFFFFFFFFFF\n",
}


def _dump(fn, h):
    n = C.c_int64(0)
    fn.restype = C.POINTER(C.c_char)
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    p = fn(h, C.byref(n))
    return C.string_at(p, n.value)


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("min_len", [0, 5, 12])
def test_fasta_reader_matches_oracle(tmp_path, name, min_len):
    from downpore_amd.overlap import Reads, load_host
    H = load_host()
    path = str(tmp_path / (name + ".fa"))
    with open(path, "wb") as f:
        f.write(CASES[name])
    try:
        want_set = O.ReadSet(fasta=path, min_len=min_len)
    except RuntimeError as e:  # the reference calls log.Fatal("Invalid fastq format ...")
        assert "Invalid fastq" in str(e)
        with pytest.raises(Exception, match="Invalid fastq"):
            Reads(fasta=path, min_len=min_len)
        return
    got_set = Reads(fasta=path, min_len=min_len)
    want = _dump(O.lib().dpo_reads_dump, want_set.h)
    got = _dump(H.dph_reads_dump, got_set.h)
    assert got == want
    assert len(got_set) == want.count(b"\n")


def test_fasta_reader_missing_file(tmp_path):
    from downpore_amd.overlap import Reads
    with pytest.raises(Exception):
        Reads(fasta=str(tmp_path / "nope.fa"), min_len=0)
