"""The reference's own python test-suite (/root/reference/tests/python_test.py:82-227) restated against this build's
`mappy_rs` module on the bundled fixtures (copied as data into tests/golden/): same inputs, same assertions."""
import copy
import os
from itertools import repeat

import pytest

pytestmark = pytest.mark.gpu

from test_oracle_golden import BACILLUS, ENTERO, read_fasta


@pytest.fixture
def al(built, golden_dir):
    import mappy_rs
    return mappy_rs.Aligner(os.path.join(golden_dir, "test.mmi"))


@pytest.fixture
def fasta_list(golden_dir):
    seqs = [s for _, s in read_fasta(os.path.join(golden_dir, "test.fa"))]
    return [{"id": i, "seq": seq} for i, seq in enumerate(copy.copy(s) for _ in range(10) for s in seqs)]


def test_properties(al):
    assert al and al.k == 15 and al.n_seq == 4 and al.w == 10
    names = al.seq_names
    names.sort()
    assert names == ["Bacillus_subtilis", "Enterococcus_faecalis", "Escherichia_coli_1", "Escherichia_coli_2"]
    assert al.seq("Bacillus_subtilis") == BACILLUS


def test_map_one(al):
    mappings = al.map(ENTERO, cs=True)
    assert len(mappings) == 1
    m = mappings[0]
    assert m.target_start == 0 and m.target_end == 400
    # derived record
    assert (m.q_st, m.q_en, m.strand, m.ctg, m.ctg_len, m.mlen, m.blen, m.NM, m.mapq, m.is_primary) == (0, 400, 1, "Enterococcus_faecalis", 400, 400, 400, 0, 60, True)
    assert m.cigar == [(400, 0)] and m.cigar_str == "400M" and m.cs == ":400" and m.MD is None
    assert str(m) == "0\t400\t+\tEnterococcus_faecalis\t400\t0\t400\t400\t400\t60\ttp:A:P\tcg:Z:400M"


@pytest.mark.parametrize("kind", ["iter", "list", "tuple", "generator"])
def test_map_batch(al, fasta_list, kind):
    al.enable_threading(2)
    src = {"iter": iter(fasta_list), "list": fasta_list, "tuple": tuple(fasta_list), "generator": (x for x in fasta_list)}[kind]
    n = 0
    for maps, item in al.map_batch(src):
        assert len(maps) == 1 and maps[0].cs == ":400" and maps[0].MD is None and "seq" in item
        n += 1
    assert n == 40


def test_map_batch_100000(al, fasta_list):
    al.enable_threading(4)
    n = sum(1 for _ in al.map_batch(repeat(fasta_list[0], 100000), back_off=True))
    assert n == 100000


def test_map_batch_100000_no_backoff(al, fasta_list):
    al.enable_threading(4)
    with pytest.raises(RuntimeError) as excinfo:
        for _ in al.map_batch(repeat(fasta_list[0], 100000), back_off=False):
            pass
    assert "Internal error adding data to work queue, without backoff" in str(excinfo)
    assert "Is your fastq batch larger than 50000? Perhaps try `map_batch` with back_off=True?" in str(excinfo)


def test_map_batch_failures(al, fasta_list):
    al.enable_threading(2)
    with pytest.raises(TypeError, match="Unsupported batch type, pass a list, iter, generator or tuple"):
        al.map_batch(fasta_list[0])
    with pytest.raises(TypeError, match="Unsupported batch type, pass a list, iter, generator or tuple"):
        al.map_batch({i: d for i, d in enumerate(fasta_list)})
    with pytest.raises(TypeError, match="Element in iterable is not a dictionary"):
        al.map_batch([d["seq"] for d in fasta_list])
    with pytest.raises(KeyError) as e:
        al.map_batch([{"SEQ": d["seq"]} for d in fasta_list])
    assert "AHHH Key 🗝️  not found in iterated dictionary" in str(e)
    with pytest.raises(ValueError, match="`seq` must be a string"):
        al.map_batch([{"seq": d["seq"].encode()} for d in fasta_list])
    it = iter(fasta_list)
    _ = list(it)
    assert len(list(al.map_batch(it))) == 0
