"""SURVEY 8(f)-3 on the GPU: ParseBarcodedFastqs with the pair order, the 2-bit packing and the PQVec encoder on the device
(dfk_pbf_run, the program's default path) -- byte for byte against the fixture the reference's own binary wrote
(tests/golden/pbf/), against that binary where it is present (oracle/_ref/ParseBarcodedFastqs travels with the repository),
and against the program's HOST=True path on seeded inputs: ragged lengths, duplicates (ties in the per-barcode order), reads
with N, READS_PER_BC, bucket counts from 1 to 256, and a set larger than MAX_MEM_GB done in groups of buckets."""
import gzip
import os

import pytest

from tests.fastq_synth import make_fastq
from tests.test_parse_barcoded_fastqs import OURS, REF, run, same_files

pytestmark = pytest.mark.gpu


def test_device_path_matches_the_reference_fixture(tmp_path, golden_dir):
    g = os.path.join(golden_dir, "pbf")
    r = run(OURS, g + "/r_1.fq.gz", g + "/r_2.fq.gz", f"{tmp_path}/o/reads", "NUM_BUCKETS=4", "NUM_THREADS=3", device=True)
    assert r.returncode == 0, r.stderr
    assert "device: pair order" in r.stderr
    same_files(f"{tmp_path}/o/reads", g + "/reads")


@pytest.mark.parametrize("pairs,seed,n_bc,ragged,extra", [
    (2000, 11, 40, True, ("NUM_BUCKETS=5",)), (1500, 12, 300, False, ()), (800, 13, 3, True, ("NUM_BUCKETS=256",)),
    (1200, 14, 25, True, ("NUM_BUCKETS=2",)), (1500, 15, 30, False, ("NUM_BUCKETS=7", "READS_PER_BC=90")), (900, 16, 12, True, ("NUM_BUCKETS=1",))])
def test_device_path_matches_host_path_and_reference_binary(tmp_path, pairs, seed, n_bc, ragged, extra):
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, pairs, seed, n_bc=n_bc, ragged=ragged)
    d = run(OURS, fq1, fq2, f"{tmp_path}/dev/reads", "NUM_THREADS=4", *extra, device=True)
    assert d.returncode == 0, d.stderr
    h = run(OURS, fq1, fq2, f"{tmp_path}/host/reads", "NUM_THREADS=4", *extra)
    assert h.returncode == 0, h.stderr
    same_files(f"{tmp_path}/dev/reads", f"{tmp_path}/host/reads")
    if os.path.exists(REF) and "NUM_BUCKETS=1" not in extra:                 # (the reference aborts on one bucket)
        r = run(REF, fq1, fq2, f"{tmp_path}/ref/reads", "NUM_THREADS=1", *extra)
        assert r.returncode == 0, r.stdout + r.stderr
        same_files(f"{tmp_path}/dev/reads", f"{tmp_path}/ref/reads")


def test_two_hundred_thousand_pairs_and_groups_of_buckets(tmp_path):
    """A seeded 200 k-pair set (ragged lengths, 3000 barcodes): the device path whole, the device path with a MAX_MEM_GB that
    forces several groups of buckets (each group a dfk_pbf_run of its own), and the host path -- three identical file sets."""
    fq1, fq2 = f"{tmp_path}/a_1.fq.gz", f"{tmp_path}/a_2.fq.gz"
    make_fastq(fq1, fq2, 200_000, 77, n_bc=3000, ragged=True)
    whole = run(OURS, fq1, fq2, f"{tmp_path}/whole/reads", "NUM_THREADS=8", "NUM_BUCKETS=16", device=True)
    assert whole.returncode == 0 and "more passes" not in whole.stderr, whole.stderr
    grouped = run(OURS, fq1, fq2, f"{tmp_path}/groups/reads", "NUM_THREADS=8", "NUM_BUCKETS=16", "MAX_MEM_GB=0.04", device=True)
    assert grouped.returncode == 0, grouped.stderr
    assert int(grouped.stderr.split("allowed: ")[1].split()[0]) >= 3, grouped.stderr
    host = run(OURS, fq1, fq2, f"{tmp_path}/host/reads", "NUM_THREADS=8", "NUM_BUCKETS=16")
    assert host.returncode == 0, host.stderr
    same_files(f"{tmp_path}/whole/reads", f"{tmp_path}/host/reads")
    same_files(f"{tmp_path}/groups/reads", f"{tmp_path}/host/reads")


def test_device_path_refuses_what_the_reference_refuses(tmp_path):
    """A quality above 63 (PQVecEncoder: "Your input reads are funny", feudal/PQVec.cc:30-35) and a character that is no base."""
    def write(p, recs):
        with gzip.open(p, "wt") as f:
            for name, s, q in recs: f.write(f"@{name}\n{s}\n+\n{q}\n")
    ok = [("r0#1_2_3/1", "ACGTACGTAC", "IIIIIIIIII"), ("r1#1_2_3/1", "ACGTTCGTAC", "IIIIIIIIII")]
    write(f"{tmp_path}/a_2.fq.gz", [(n.replace("/1", "/2"), s, q) for n, s, q in ok])
    write(f"{tmp_path}/q_1.fq.gz", [ok[0], ("r1#1_2_3/1", "ACGTTCGTAC", "IIII" + chr(33 + 64) + "IIIII")])
    r = run(OURS, f"{tmp_path}/q_1.fq.gz", f"{tmp_path}/a_2.fq.gz", f"{tmp_path}/o1/reads", device=True)
    assert r.returncode != 0 and "funny" in r.stderr and not os.path.exists(f"{tmp_path}/o1/reads.fastb")
    write(f"{tmp_path}/b_1.fq.gz", [ok[0], ("r1#1_2_3/1", "ACGTXCGTAC", "IIIIIIIIII")])
    r = run(OURS, f"{tmp_path}/b_1.fq.gz", f"{tmp_path}/a_2.fq.gz", f"{tmp_path}/o2/reads", device=True)
    assert r.returncode != 0 and "unexpected base" in r.stderr and not os.path.exists(f"{tmp_path}/o2/reads.fastb")
