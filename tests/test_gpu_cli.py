"""End-to-end drop-in surface on the GPU: `hyper-gen sketch` on a directory of FASTA files, the
.sketch it writes, `hyper-gen dist`, and the ANI TSV -- against a model assembled from the oracle
(src/sketch.rs:12-69, src/dist.rs:11-63, src/utils.rs:260-308)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def write_fasta(path, seq, name, width=80, two_records=False):
    s = bytes(seq).decode()
    with open(path, "w") as f:
        if two_records:
            half = len(s) // 2
            parts = [(name + "_a", s[:half]), (name + "_b extra words", s[half:])]
        else:
            parts = [(name, s)]
        for n, body in parts:
            f.write(">%s\n" % n)
            for i in range(0, len(body), width):
                f.write(body[i:i + width] + "\n")


def model_tsv(orc, hg, files, merged, params, ani_th, sym_with=None):
    recs = []
    for f, m in zip(files, merged):
        hv, n2, nh = orc.sketch_genome(m, ksize=params["k"], scaled=params["s"], norm=orc.NORM_U2T)
        recs.append((f, hv, n2))
    hvs = np.stack([r[1] for r in recs])
    n2s = np.array([r[2] for r in recs], np.int32)
    ani = orc.ani_matrix(hvs, n2s, hvs, n2s, params["k"])
    pairs = [(i, j) for i in range(len(files)) for j in range(i + 1, len(files))]  # symmetric: i < j
    order = sorted(range(len(pairs)), key=lambda t: ani[pairs[t]])
    order.reverse()
    lines = []
    for t in order:
        i, j = pairs[t]
        if ani[i, j] >= ani_th:
            lines.append((files[i], files[j], float(ani[i, j])))
    return recs, lines


def test_cli_sketch_dist_roundtrip(tmp_path, orc):
    import hypergen_amd as hg
    d = tmp_path / "fa"
    d.mkdir()
    L = 300_000
    gen = {"g000.fna": 0, "g010.fna": 10, "g030.fna": 30, "g100.fa": 100, "g101.fasta": 101}
    for name, g in gen.items():
        seq = orc.synth_genome(g, L)[1:]
        write_fasta(str(d / name), seq, name, two_records=(g == 10))
    files = sorted(str(d / n) for n in gen if n.endswith(".fna")) + [str(d / "g100.fa"), str(d / "g101.fasta")]
    merged = [hg.read_merge_seq(f) for f in files]
    out = str(tmp_path / "all.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out, "-s", "100", "-t", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Sketching 5 files took" in r.stdout and "Dump sketch file to" in r.stdout

    recs_model, lines_model = model_tsv(orc, hg, files, merged, {"k": 21, "s": 100}, 85.0)
    recs = hg.read_sketch_file(out)
    assert [x["file_str"] for x in recs] == files  # glob order: *.fna, *.fa, *.fasta (src/utils.rs:208-221)
    for x, (f, hv, n2) in zip(recs, recs_model):
        assert (x["ksize"], x["scaled"], x["seed"], x["canonical"], x["hv_d"]) == (21, 100, 123, True, 4096)
        assert x["hv_norm_2"] == n2
        q, packed = orc.pack_hv(hv)
        assert x["hv_quant_bits"] == q and (x["hv"].view(np.uint8) == packed).all()
        assert (hg.hv_unpack(x["hv"].view(np.uint8), 4096, q) == hv).all()

    tsv = str(tmp_path / "ani.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out, "-o", tsv, "-a", "85"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [l.split("\t") for l in open(tsv).read().splitlines()]
    assert len(got) == len(lines_model) >= 3
    for (a, b, v), (ma, mb, mv) in zip(got, lines_model):
        assert (a, b) == (ma, mb)
        assert abs(float(v) - mv) <= 1e-3 + 1e-4  # 3 printed decimals
    # ref != query path: all R x Q pairs, including self pairs at 100.000
    out2 = str(tmp_path / "copy.sketch")
    hg.write_sketch_file(out2, recs)
    tsv2 = str(tmp_path / "ani2.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out2, "-o", tsv2, "-a", "99.5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = [l.split("\t") for l in open(tsv2).read().splitlines()]
    assert len(rows) >= 5 and all(x[2] == "100.000" for x in rows[:5])


def test_cli_reference_fixture(tmp_path):
    """BASELINE config 1: the reference's test/test.fna (40 bases) -> empty hash set, zero HV, 6-bit payload."""
    import hypergen_amd as hg
    d = tmp_path / "t"
    d.mkdir()
    (d / "test.fna").write_text(">test_seq\nAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt")
    out = str(tmp_path / "t.sketch")
    r = subprocess.run([hg.CLI_PATH, "sketch", "-p", str(d), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rec = hg.read_sketch_file(out)
    assert len(rec) == 1 and rec[0]["hv_norm_2"] == 0 and rec[0]["hv_quant_bits"] == 6 and rec[0]["hv"].size == 1536
    assert (hg.hv_unpack(rec[0]["hv"].view(np.uint8), 4096, 6) == 0).all()
    tsv = str(tmp_path / "t.tsv")
    r = subprocess.run([hg.CLI_PATH, "dist", "-r", out, "-q", out, "-o", tsv], capture_output=True, text=True)
    assert r.returncode == 0 and open(tsv).read() == ""  # one record, symmetric -> zero pairs
