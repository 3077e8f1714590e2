"""The `DF` front-end (seam B2).  Ingest is host-only, so these run on the CPU: DF ... EXIT_LOAD=True
must reproduce the files the reference's LoadData/FirstLoadData write (tests/golden/side/* were written
by the reference's own BinaryWriter and containers through oracle/_ref/refdrv)."""
import os
import subprocess

import numpy as np
import pytest

from superplus_amd import feudal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DF = os.path.join(ROOT, "superplus_amd", "DF")


def run_df(*args):
    return subprocess.run([DF, *args], capture_output=True, text=True)


def test_ingest_matches_reference_side_files(tmp_path, golden_dir):
    r = run_df(f"LR={golden_dir}/reads.fastb", f"OUT_DIR={tmp_path}/w", "EXIT_LOAD=True", "PIPELINE=cs", "NUM_THREADS=4")
    assert r.returncode == 0, r.stdout + r.stderr
    rd = lambda p: open(p, "rb").read()
    for ext in ("fastb", "qualp", "bci"):          # one LR input already in LoadData order: copied verbatim
        assert rd(f"{tmp_path}/w/data/frag_reads_orig.{ext}") == rd(f"{golden_dir}/reads.{ext}"), ext
    # .1000.*: the 500-pair sample LoadData writes with the reference's own random stream (reproduced in df_main.cc)
    for name in ("frag_reads_orig.lens", "frag_reads_orig.qhist", "frag_reads_orig.1000.fastb", "frag_reads_orig.1000.qualp"):
        assert rd(f"{tmp_path}/w/data/{name}") == rd(f"{golden_dir}/side/{name}"), name
    # .dti: 16-byte records {u8 dt, 7 pad bytes (undefined in the reference), i64 start} -- compare fields
    dt = np.dtype([("dt", "u1"), ("pad", "V7"), ("start", "<i8")])
    a = np.frombuffer(rd(f"{tmp_path}/w/data/frag_reads_orig.dti")[16:], dt)
    b = np.frombuffer(rd(f"{golden_dir}/side/frag_reads_orig.dti")[16:], dt)
    assert list(a["dt"]) == list(b["dt"]) == [2, 3] and list(a["start"]) == list(b["start"])
    assert rd(f"{tmp_path}/w/subsam.starts") == rd(f"{golden_dir}/side/subsam.starts")
    # FeudalString writes size+1 bytes; the byte after the text is whatever follows it in the reference
    assert rd(f"{tmp_path}/w/subsam.names")[:-1] == rd(f"{golden_dir}/side/subsam.names")[:-1]
    assert "the_command" in os.listdir(f"{tmp_path}/w")


def test_two_inputs_reordered_like_loaddata(tmp_path, golden_dir):
    """LoadData (10X/DfTools.cc:99-160): all inputs' unbarcoded pairs first, then barcoded pairs input by input,
    barcode by barcode, with the barcode index rebuilt."""
    packed, boff, rlen = feudal.read_fastb(f"{golden_dir}/reads.fastb")
    pq, qoff = feudal.read_qualp(f"{golden_dir}/reads.qualp")
    bci = feudal.read_bci(f"{golden_dir}/reads.bci")
    # second input = a slice of the first: unbarcoded [0,100) + barcodes 1..3
    cut = int(bci[4])
    def sub(lo, hi):
        return (packed[int(boff[lo]):int(boff[hi])], boff[lo:hi + 1] - boff[lo], rlen[lo:hi],
                pq[int(qoff[lo]):int(qoff[hi])], qoff[lo:hi + 1] - qoff[lo])
    u = sub(0, 100); b = sub(int(bci[1]), cut)
    p2 = np.concatenate([u[0], b[0]]); bo2 = np.concatenate([u[1], b[1][1:] + u[1][-1]]); l2 = np.concatenate([u[2], b[2]])
    q2 = np.concatenate([u[3], b[3]]); qo2 = np.concatenate([u[4], b[4][1:] + u[4][-1]])
    bci2 = np.concatenate([[0], bci[1:5] - bci[1] + 100]).astype(np.int64)
    feudal.write_fastb(f"{tmp_path}/b.fastb", p2, bo2, l2); feudal.write_qualp(f"{tmp_path}/b.qualp", q2, qo2)
    feudal.write_bci(f"{tmp_path}/b.bci", bci2)
    r = run_df("LR={" + f"{golden_dir}/reads.fastb,{tmp_path}/b.fastb" + "}", f"ROOT={tmp_path}", "EXIT_LOAD=True")
    assert r.returncode == 0, r.stdout + r.stderr
    w = f"{tmp_path}/GapToy/1/data/frag_reads_orig"
    _, obo, ol = feudal.read_fastb(w + ".fastb")
    obci = feudal.read_bci(w + ".bci")
    n1, nu1 = len(rlen), int(bci[1])
    exp_len = np.concatenate([rlen[:nu1], l2[:100], rlen[nu1:], l2[100:]])
    assert np.array_equal(ol, exp_len)
    exp_bci = np.concatenate([[0], bci[1:-1] + 100, (bci2[1:-1] - 100) + n1 + 100, [n1 + len(l2)]])
    assert np.array_equal(obci, exp_bci)
    assert "UNBAR_10X starts at 0" in r.stdout and f"BAR_10X starts at {nu1 + 100}" in r.stdout


def test_missing_input_gives_up_like_the_reference(tmp_path):
    r = run_df(f"LR={tmp_path}/nope.fastb", f"OUT_DIR={tmp_path}/w")
    assert r.returncode == 1 and "Can't file your LR input files" in r.stdout
    assert run_df("K=47", "LR=x").returncode == 1


def test_lr_select_frac_follows_the_reference_random_stream(tmp_path, golden_dir):
    """LR_SELECT_FRAC=0.7: which pairs stay is decided by the reference's global random stream (one draw per pair,
    DfTools.cc:115-117); tests/golden/side_frac07/* were written by refdrv with the reference's randomx() and
    feudal writers."""
    r = run_df(f"LR={golden_dir}/reads.fastb", f"OUT_DIR={tmp_path}/w", "EXIT_LOAD=True", "LR_SELECT_FRAC=0.7")
    assert r.returncode == 0, r.stdout + r.stderr
    rd = lambda p: open(p, "rb").read()
    for name in ("frag_reads_orig.fastb", "frag_reads_orig.qualp", "frag_reads_orig.bci", "frag_reads_orig.lens",
                 "frag_reads_orig.qhist", "frag_reads_orig.1000.fastb", "frag_reads_orig.1000.qualp"):
        assert rd(f"{tmp_path}/w/data/{name}") == rd(f"{golden_dir}/side_frac07/{name}"), name


def test_runall_command_line_parses_without_a_gpu(tmp_path, golden_dir):
    """runall.sh:127 passes MAX_MEM_GB=640 -- a HOST memory cap (system/System.cc:1073-1078), not a device budget --
    beside PIPELINE/ALIGN/NUM_THREADS; the ingest half must run with exactly those arguments."""
    r = run_df(f"ROOT={tmp_path}", f"LR={golden_dir}/reads.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=8", "MAX_MEM_GB=640",
               "EXIT_LOAD=True")
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(f"{tmp_path}/GapToy/1/data/frag_reads_orig.fastb")


def test_malformed_feudal_offsets_are_refused(tmp_path, golden_dir):
    raw = bytearray(open(f"{golden_dir}/reads.qualp", "rb").read())
    var_tab = int.from_bytes(raw[8:16], "little")
    raw[var_tab + 8 * 5: var_tab + 8 * 6] = (var_tab + 1000).to_bytes(8, "little")      # an offset past the var data
    for ext in ("fastb", "bci"):
        open(f"{tmp_path}/bad.{ext}", "wb").write(open(f"{golden_dir}/reads.{ext}", "rb").read())
    open(f"{tmp_path}/bad.qualp", "wb").write(raw)
    r = run_df(f"LR={tmp_path}/bad.fastb", f"OUT_DIR={tmp_path}/w", "EXIT_LOAD=True")
    assert r.returncode == 1 and "offset table" in r.stderr


def _kvec(path):
    kv = open(path, "rb").read()
    assert kv[:8] == b"BINWRITE"
    n = int.from_bytes(kv[8:16], "little")
    assert len(kv) == 16 + 32 * n
    from superplus_amd.dfk import ENTRY_DTYPE
    return kv, np.frombuffer(kv, ENTRY_DTYPE, count=n, offset=16)


@pytest.mark.gpu
def test_df_end_to_end_on_gpu(tmp_path, golden_dir, oracle):
    from tests import util
    r = run_df(f"LR={golden_dir}/reads.fastb", f"OUT_DIR={tmp_path}/w", "K=48", "KVEC=True")
    assert r.returncode == 0, r.stdout + r.stderr
    exp = np.load(f"{golden_dir}/expect_k48.npz")
    assert open(f"{tmp_path}/w/stats/histogram_kmer_count.json").read() == oracle.spectrum_json(exp["spectrum"])
    # kmers.kvec holds the dictionary in device order (the reference's is in thread-arrival order): same entries
    _, e = _kvec(f"{tmp_path}/w/kmers.kvec")
    util.assert_same_solid(e[np.lexsort((e["w1"], e["w0"]))], exp["solid_post"], "kmers.kvec, sorted here")
    assert f"dictionary covers {len(exp['solid_post']):,}".replace(",", "") in r.stdout.replace(",", "")
    assert "DF_TIMING {" in r.stdout and '"fast_path": true' in r.stdout
    # ... and byte for byte the sorted dictionary on request
    r = run_df(f"LR={golden_dir}/reads.fastb", f"OUT_DIR={tmp_path}/w2", "K=48", "KVEC_SORTED=True")
    assert r.returncode == 0, r.stdout + r.stderr
    kv, _ = _kvec(f"{tmp_path}/w2/kmers.kvec")
    assert kv[16:] == exp["solid_post"].tobytes()
    rd = lambda p: open(p, "rb").read()
    for ext in ("fastb", "qualp", "bci"):
        assert rd(f"{tmp_path}/w/data/frag_reads_orig.{ext}") == rd(f"{golden_dir}/reads.{ext}"), ext
    # a.48/: the graph files WriteAssemblyFiles writes, against the reference-written fixture
    for f in ("a.k", "a.fastb", "a.hbv", "a.hbx", "a.kmers", "a.inv", "a.to_left", "a.to_right", "a.paths"):
        assert rd(f"{tmp_path}/w/a.48/{f}") == rd(f"{golden_dir}/graph_k48/{f}"), f


@pytest.mark.gpu
def test_df_writes_paths_index_and_dups(tmp_path, golden_dir):
    """DF on the fragmented fixture: a.48/ holds the graph, the read paths, the inverted paths index, the counts and the
    duplicate marks -- what 10X/DF.cc:541-561 leaves behind -- byte for byte as the reference's classes wrote them."""
    r = run_df(f"LR={golden_dir}/frag.fastb", f"OUT_DIR={tmp_path}/w", "K=48", "HBM_GB=8")
    assert r.returncode == 0, r.stdout + r.stderr
    assert not os.path.exists(f"{tmp_path}/w/kmers.kvec")              # transient in the reference (BuildReadQGraph48.cc:303): not left behind unless asked for
    for f in ("a.fastb", "a.hbv", "a.inv", "a.paths", "a.paths.inv", "a.countsb", "a.dup"):
        assert open(f"{tmp_path}/w/a.48/{f}", "rb").read() == open(f"{golden_dir}/graph_frag_k48/{f}", "rb").read(), f
    assert "% of pairs appear to be duplicates" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("reserve,set_name", [("0", "frag"), ("7", "frag"), ("18", "pathy2"), ("400", "frag"), ("20", "pathy2"), ("0", "pathy2"), ("19.3", "pathy2")])
def test_df_paths_file_is_the_same_however_much_was_reserved_for_it(tmp_path, golden_dir, reserve, set_name):
    """The pages of a.paths and a.paths.inv are made ahead of the pathing (PATHS_RESERVE bytes a read, default 20) and the
    writers do not truncate what they find; what the guess was -- nothing, too little, a little short, far too much (cut at
    the end) -- and stale larger files in the way change nothing in the bytes."""
    os.makedirs(f"{tmp_path}/w/a.48")
    for f in ("a.paths", "a.paths.inv"):
        open(f"{tmp_path}/w/a.48/{f}", "wb").write(b"\xee" * (3 << 20))          # (left by an earlier run)
    r = run_df(f"LR={golden_dir}/{set_name}.fastb", f"OUT_DIR={tmp_path}/w", "K=48", "HBM_GB=8", f"PATHS_RESERVE={reserve}")
    assert r.returncode == 0, r.stdout + r.stderr
    for f in ("a.paths", "a.paths.inv", "a.dup"):
        assert open(f"{tmp_path}/w/a.48/{f}", "rb").read() == open(f"{golden_dir}/graph_{set_name}_k48/{f}", "rb").read(), f


@pytest.mark.gpu
def test_df_leaves_what_the_reference_resumes_from(tmp_path, golden_dir):
    """Seam B2 as a pipeline: `superplus_amd/DF ROOT=... LR=...` then the reference's `DF ROOT=... START=patch ...` on the same
    ROOT.  That branch of the reference loads exactly these files (10X/DF.cc:304-341 for START != "",
    :569-594 for START=patch): data/frag_reads_orig.{qualp,fastb,bci,lens,qhist,dti} and a.<K>/{a.hbv,a.hbx,a.inv,a.paths,a.dup}
    + a.paths.inv (handed to StagePatch by name).  Every one must exist and parse as the reader the reference uses on it would
    read it, with sizes that agree with each other."""
    import struct
    r = run_df(f"ROOT={tmp_path}", f"LR={golden_dir}/frag.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=4", "MAX_MEM_GB=640", "HBM_GB=8")
    assert r.returncode == 0, r.stdout + r.stderr
    w = f"{tmp_path}/GapToy/1"
    rd = lambda p: open(p, "rb").read()
    head = f"{w}/data/frag_reads_orig"
    # bases.ReadAll(.fastb), quals_om.newFile(.qualp): feudal files with one element per read
    packed, base_off, read_len = feudal.read_fastb(head + ".fastb")
    pq, pq_off = feudal.read_qualp(head + ".qualp")
    n = len(read_len)
    assert n == 6000 and len(pq_off) == n + 1 and int(base_off[-1]) == len(packed) and int(pq_off[-1]) == len(pq)
    # BinaryReader::readFile of vec<int64_t> (.bci), vec<int16_t> (.lens), vec<vec<vec<int64_t>>> (.qhist), vec<DataSet> (.dti)
    bci = feudal.read_bci(head + ".bci")
    assert bci[0] == 0 and bci[-1] == n and all(a <= b for a, b in zip(bci, bci[1:]))
    b = rd(head + ".lens")
    assert b[:8] == b"BINWRITE" and struct.unpack_from("<Q", b, 8)[0] == n and len(b) == 16 + 2 * n
    assert list(np.frombuffer(b, "<i2", n, 16)) == [int(x) for x in read_len]
    b = rd(head + ".qhist")
    assert b[:8] == b"BINWRITE" and struct.unpack_from("<Q", b, 8)[0] == 2
    at, total = 16, 0
    for par in range(2):
        (n_pos,) = struct.unpack_from("<Q", b, at); at += 8
        assert n_pos == int(read_len.max())
        for pos in range(n_pos):
            (n_q,) = struct.unpack_from("<Q", b, at); at += 8
            total += int(np.frombuffer(b, "<i8", n_q, at).sum()); at += 8 * n_q
    assert at == len(b) and total == int(read_len.astype(np.int64).sum())      # every quality of every read counted once
    b = rd(head + ".dti")
    dt = np.frombuffer(b[16:], np.dtype([("dt", "u1"), ("pad", "V7"), ("start", "<i8")]))
    assert struct.unpack_from("<Q", b, 8)[0] == len(dt) == 2 and list(dt["dt"]) == [2, 3] and dt["start"][0] == 0 and dt["start"][1] == bci[1]
    # the graph and the paths: a.hbv / a.hbx / a.inv (BinaryReader), a.paths (ReadPathVec::ReadAll), a.dup (vec<Bool>), a.paths.inv
    d = f"{w}/a.48"
    for f in ("a.hbv", "a.hbx", "a.inv", "a.paths", "a.dup", "a.paths.inv", "a.countsb"):
        assert rd(f"{d}/{f}") == rd(f"{golden_dir}/graph_frag_k48/{f}"), f      # (byte for byte what the reference's writers wrote)
    inv = rd(f"{d}/a.inv")
    n_edges = struct.unpack_from("<Q", inv, 8)[0]
    hbv = rd(f"{d}/a.hbv")
    assert hbv[:8] == b"BINWRITE" and struct.unpack_from("<i", hbv, 8)[0] == 48
    hbx = rd(f"{d}/a.hbx")
    assert hbx[:8] == b"BINWRITE" and struct.unpack_from("<i", hbx, 8)[0] == 48
    paths = rd(f"{d}/a.paths")
    assert struct.unpack_from("<I", paths, 0)[0] == n and paths[4:8] == bytes([1, 0, 24, 4])          # feudal control block of ReadPath
    assert struct.unpack_from("<Q", paths, 16)[0] == len(paths)                                         # no fixed data behind the offset table
    dup = rd(f"{d}/a.dup")
    assert dup[:8] == b"BINWRITE" and struct.unpack_from("<Q", dup, 8)[0] == n // 2 == len(dup) - 16
    pinv = rd(f"{d}/a.paths.inv")
    assert struct.unpack_from("<I", pinv, 0)[0] == n_edges and pinv[4:8] == bytes([1, 0, 16, 8])
    # ... and nothing else the branch reads is missing: the set, as a set
    need = {"data/frag_reads_orig." + e for e in ("fastb", "qualp", "bci", "lens", "qhist", "dti")} | {"a.48/" + f for f in ("a.hbv", "a.hbx", "a.inv", "a.paths", "a.dup", "a.paths.inv")}
    assert all(os.path.isfile(f"{w}/{f}") for f in need), [f for f in need if not os.path.isfile(f"{w}/{f}")]
    assert "DF_DIGESTS " in r.stdout


@pytest.mark.gpu
def test_runall_command_line_on_gpu(tmp_path, golden_dir, oracle):
    """The full command line of runall.sh:127, unchanged: MAX_MEM_GB=640 must not become a 640 GiB HBM plan.  The
    transfers are forced through the many-chunk staged path (a few KB per chunk, three host threads)."""
    from tests import util
    env = dict(os.environ, DFK_XFER_CHUNK="4096", DFK_HOST_THREADS="3")
    r = subprocess.run([DF, f"ROOT={tmp_path}", f"LR={golden_dir}/reads.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=8",
                        "MAX_MEM_GB=640", "KVEC=True"], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = np.load(f"{golden_dir}/expect_k48.npz")
    w = f"{tmp_path}/GapToy/1"
    assert open(f"{w}/stats/histogram_kmer_count.json").read() == oracle.spectrum_json(exp["spectrum"])
    _, e = _kvec(f"{w}/kmers.kvec")
    util.assert_same_solid(e[np.lexsort((e["w1"], e["w0"]))], exp["solid_post"], "kmers.kvec through the staged transfers")


@pytest.mark.gpu
def test_df_general_path_on_gpu(tmp_path, golden_dir, oracle):
    """Two inputs (LoadData reorders): the gathered arrays go through the same count."""
    from tests import util
    from oracle import pyoracle
    r = run_df("LR={" + f"{golden_dir}/reads.fastb,{golden_dir}/reads.fastb" + "}", f"OUT_DIR={tmp_path}/w", "K=48", "HBM_GB=8", "KVEC=True")
    assert r.returncode == 0, r.stdout + r.stderr
    assert '"fast_path": false' in r.stdout
    w = f"{tmp_path}/w/data/frag_reads_orig"
    packed, boff, rlen = feudal.read_fastb(w + ".fastb")
    pq, qoff = feudal.read_qualp(w + ".qualp")
    bci = feudal.read_bci(w + ".bci")
    bc = np.zeros(len(rlen), np.int32)
    for b in range(len(bci) - 1):
        bc[int(bci[b]):int(bci[b + 1])] = b
    ref = pyoracle.run(packed, boff, rlen, pq, qoff, bc, K=48)
    _, e = _kvec(f"{tmp_path}/w/kmers.kvec")
    util.assert_same_solid(e[np.lexsort((e["w1"], e["w0"]))], ref["solid"], "two-input run")
    assert open(f"{tmp_path}/w/stats/histogram_kmer_count.json").read() == oracle.spectrum_json(ref["hist"])


def _sorted_kvec(path):
    _, e = _kvec(path)
    return e[np.lexsort((e["w1"], e["w0"]))]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rccl1", "loopback2", "loopback4"])
def test_df_num_gpus_cpp_host(tmp_path, golden_dir, oracle, mode):
    """`DF NUM_GPUS=N`: the C++ host (df_shard.h) -- one rank per GPU over RCCL, spawned before any HIP call; on this
    one-GPU box: a single RCCL rank through the sharded path, and two / four ranks as threads sharing the GPU over
    the loopback transport (RCCL refuses two ranks on one device).  Same dictionary and spectrum as the oracle."""
    from tests import util
    env = dict(os.environ)
    if mode == "rccl1":
        args, env["DF_FORCE_SHARDED"] = ["NUM_GPUS=1"], "1"
    else:
        args, env["DF_TRANSPORT"] = [f"NUM_GPUS={mode[-1]}"], "loopback"
        env["DFK_A2A_PIECE_BYTES"] = "4096"                               # several rounds per pair of ranks
    r = subprocess.run([DF, f"ROOT={tmp_path}", f"LR={golden_dir}/reads.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=8", "MAX_MEM_GB=640",
                        "HBM_GB=8", "KVEC=True", *args], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    exp = np.load(f"{golden_dir}/expect_k48.npz")
    w = f"{tmp_path}/GapToy/1"
    assert open(f"{w}/stats/histogram_kmer_count.json").read() == oracle.spectrum_json(exp["spectrum"])
    util.assert_same_solid(_sorted_kvec(f"{w}/kmers.kvec"), exp["solid_post"], "kmers.kvec written by the ranks")
    rd = lambda p: open(p, "rb").read()
    for ext in ("fastb", "qualp", "bci"):
        assert rd(f"{w}/data/frag_reads_orig.{ext}") == rd(f"{golden_dir}/reads.{ext}"), ext
    assert "DF_TIMING {" in r.stdout and f"dictionary covers {len(exp['solid_post'])}" in r.stdout.replace(",", "")
    # GRAPH defaults to True: every rank receives the whole dictionary and builds the graph; the reads stay sharded -- each rank paths
    # its pair range and writes its part of a.paths -- and the files are the single-GPU run's, byte for byte
    for f in ("a.k", "a.fastb", "a.hbv", "a.hbx", "a.kmers", "a.inv", "a.to_left", "a.to_right", "a.paths"):
        assert rd(f"{w}/a.48/{f}") == rd(f"{golden_dir}/graph_k48/{f}"), f


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rccl1", "loopback2", "loopback4"])
def test_df_num_gpus_paths_index_and_dups_stay_sharded(tmp_path, golden_dir, mode):
    """Rows f-2 and f-4 of `DF NUM_GPUS=N` on the fragmented fixture (1816 edges, PCR-duplicate pairs): every rank paths its own
    pair range and writes its part of a.paths; the (edge, read) pairs travel to the owners of the edge ranges, each of which
    writes its range of a.paths.inv; the duplicate keys travel to the owners of their hash and the marks come back -- all twelve
    files as the reference's own writers wrote them, and the digests the ranks add up are the single-GPU run's."""
    env = dict(os.environ)
    if mode == "rccl1":
        args, env["DF_FORCE_SHARDED"] = ["NUM_GPUS=1"], "1"
    else:
        args, env["DF_TRANSPORT"] = [f"NUM_GPUS={mode[-1]}"], "loopback"
        env["DFK_A2A_PIECE_BYTES"] = "4096"
    r = subprocess.run([DF, f"ROOT={tmp_path}/s", f"LR={golden_dir}/frag.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=8", "HBM_GB=8", *args],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    rd = lambda p: open(p, "rb").read()
    for f in sorted(os.listdir(f"{golden_dir}/graph_frag_k48")):
        if f.startswith("a."):
            assert rd(f"{tmp_path}/s/GapToy/1/a.48/{f}") == rd(f"{golden_dir}/graph_frag_k48/{f}"), f
    one = subprocess.run([DF, f"ROOT={tmp_path}/o", f"LR={golden_dir}/frag.fastb", "PIPELINE=cs", "ALIGN=False", "NUM_THREADS=8", "HBM_GB=8"],
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout + one.stderr
    dig = lambda out: [l for l in out.splitlines() if l.startswith("DF_DIGESTS ")]
    assert dig(r.stdout) and dig(r.stdout) == dig(one.stdout)
    assert '"path_reads_s"' in r.stdout                                   # rank 0's DF_TIMING carries the sharded phases


def test_df_stops_every_rank_when_one_dies(tmp_path, golden_dir):
    """`DF NUM_GPUS=4` with rank 2 dying at once and the others waiting for ever (as ranks do inside a collective whose peer
    is gone): DF must notice, end the others and exit non-zero -- not wait for the first pid in order.  CPU only: the ranks
    stop before they would touch a GPU."""
    import time
    t0 = time.time()
    r = subprocess.run([DF, f"ROOT={tmp_path}", f"LR={golden_dir}/reads.fastb", "NUM_GPUS=4", "NUM_THREADS=2"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, DF_TEST_RANK_FATE="die:2"))
    assert r.returncode != 0, r.stdout + r.stderr
    assert time.time() - t0 < 60
    assert "a rank process ended with status" in r.stderr
    out = subprocess.run(["pgrep", "-f", f"ROOT={tmp_path}"], capture_output=True, text=True).stdout.split()
    assert not out, f"rank processes left behind: {out}"


def test_df_sharded_refuses_what_it_cannot_shard(tmp_path, golden_dir):
    """NUM_GPUS > 1 reads the one input by pair range: a subsampled or multi-input run is refused before any rank is spawned."""
    r = run_df(f"ROOT={tmp_path}", f"LR={golden_dir}/reads.fastb", "NUM_GPUS=2", "LR_SELECT_FRAC=0.5")
    assert r.returncode != 0 and "LR_SELECT_FRAC" in (r.stdout + r.stderr)
    r = run_df(f"ROOT={tmp_path}", "LR={" + f"{golden_dir}/reads.fastb,{golden_dir}/reads.fastb" + "}", "NUM_GPUS=2")
    assert r.returncode != 0 and "one LR input" in (r.stdout + r.stderr)
    # EXIT_LOAD stops after the ingest whatever NUM_GPUS says: no rank is spawned, the files are there
    r = run_df(f"ROOT={tmp_path}", f"LR={golden_dir}/reads.fastb", "NUM_GPUS=2", "EXIT_LOAD=True")
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(f"{tmp_path}/GapToy/1/data/frag_reads_orig.fastb")


@pytest.mark.gpu
def test_df_num_gpus_larger_set_equals_single_gpu(tmp_path, oracle):
    """80 k pairs, four ranks (loopback) against one GPU through the same binary: same spectrum text, same entries."""
    from superplus_amd import synth
    from tests import util
    g = synth.make_genome(400000, 41)
    rs = synth.make_reads(g, 80000, 42).numpy()
    feudal.write_fastb(f"{tmp_path}/r.fastb", rs["packed"], rs["base_off"], rs["read_len"])
    feudal.write_qualp(f"{tmp_path}/r.qualp", rs["pq_bytes"], rs["pq_off"])
    feudal.write_bci(f"{tmp_path}/r.bci", rs["bci"])
    a = subprocess.run([DF, f"OUT_DIR={tmp_path}/one", f"LR={tmp_path}/r.fastb", "HBM_GB=16", "KVEC=True"], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run([DF, f"OUT_DIR={tmp_path}/four", f"LR={tmp_path}/r.fastb", "NUM_GPUS=4", "HBM_GB=4", "KVEC=True"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, DF_TRANSPORT="loopback"))
    assert b.returncode == 0, b.stdout + b.stderr
    assert open(f"{tmp_path}/one/stats/histogram_kmer_count.json").read() == open(f"{tmp_path}/four/stats/histogram_kmer_count.json").read()
    util.assert_same_solid(_sorted_kvec(f"{tmp_path}/four/kmers.kvec"), _sorted_kvec(f"{tmp_path}/one/kmers.kvec"), "four ranks against one GPU")
    ref = oracle.run(rs["packed"], rs["base_off"], rs["read_len"], rs["pq_bytes"], rs["pq_off"], rs["bc"], K=48)
    util.assert_same_solid(_sorted_kvec(f"{tmp_path}/four/kmers.kvec"), ref["solid"], "four ranks against the oracle")
    # ... and a.48/: graph, read paths (each rank its pair range), paths index (each rank its edge range), duplicate marks
    files = sorted(f for f in os.listdir(f"{tmp_path}/one/a.48") if f.startswith("a."))
    assert {"a.hbv", "a.paths", "a.paths.inv", "a.countsb", "a.dup"} <= set(files)
    for f in files:
        assert open(f"{tmp_path}/one/a.48/{f}", "rb").read() == open(f"{tmp_path}/four/a.48/{f}", "rb").read(), f
    dig = lambda out: [l for l in out.splitlines() if l.startswith("DF_DIGESTS ")]
    assert dig(a.stdout) and dig(a.stdout) == dig(b.stdout)
    for f in ("a.k", "a.fastb", "a.hbv", "a.hbx", "a.kmers", "a.inv", "a.to_left", "a.to_right", "a.paths"):      # gathered on rank 0 = built on one GPU
        assert open(f"{tmp_path}/four/a.48/{f}", "rb").read() == open(f"{tmp_path}/one/a.48/{f}", "rb").read(), f
