"""byte compatibility with a `.sketch` written by the reference binary.  The file cannot be produced in
this image (no cargo); tools/make_ref_sketch.md has the exact commands.  Skipped until tests/golden/ref_v0.2.2.sketch
exists -- then this is the test that turns "parity unpinned" for BitPacker8x / bincode into "pinned"."""
import os
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN

REF = os.path.join(GOLDEN, "ref_v0.2.2.sketch")
REF_TSV = os.path.join(GOLDEN, "ref_v0.2.2.tsv")
INPUTS = [("a_test.fna", None, None), ("b_g0.fna", 0, 200_000), ("c_g20.fna", 20, 200_000)]

pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="no reference-written .sketch (tools/make_ref_sketch.md)")


def _fasta_text(name, g, L, orc):
    if g is None:
        return ">test_seq\nAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt"  # the reference's test/test.fna
    s = bytes(orc.synth_genome(g, L)[1:]).decode()
    return ">g%d\n" % g + "".join(s[i:i + 80] + "\n" for i in range(0, len(s), 80))


def test_reference_written_sketch_is_reproduced_byte_for_byte(orc):
    import hypergen_amd as hg
    recs = hg.read_sketch_file(REF)
    assert len(recs) == 3
    mine = []
    with tempfile.TemporaryDirectory() as td:
        for (name, g, L), ref in zip(INPUTS, recs):
            assert os.path.basename(ref["file_str"]) == name
            assert (ref["ksize"], ref["scaled"], ref["seed"], ref["canonical"], ref["hv_d"]) == (21, 100, 123, True, 4096)
            merged = orc.read_needletail(_fasta_text(name, g, L, orc).encode())
            # a host with AVX2 writes the AVX2 dimension order (src/sketch.rs:40-44) in BitPacker8x blocks (src/hd.rs:138-157);
            # one without writes the scalar order (src/hd.rs:94-112) in the naive bit stream (src/hd.rs:158-166): the
            # payload's length says which kind of host wrote the file
            lay = hg.hv_payload_layout(4096, ref["hv_quant_bits"], ref["hv"].size * 2)
            assert lay in (hg.PAYLOAD_BITPACKER8X, hg.PAYLOAD_NAIVE), name
            naive = lay == hg.PAYLOAD_NAIVE
            hv, n2, _ = orc.sketch_genome(merged, scaled=100, norm=orc.NORM_U2T, layout=orc.LAYOUT_SCALAR if naive else orc.LAYOUT_AVX2)
            q, packed = (hg.hv_pack_naive if naive else hg.hv_pack)(hv)
            packed = packed[: packed.size // 2 * 2]  # (the i16 view the file stores)
            assert ref["hv_norm_2"] == n2 and ref["hv_quant_bits"] == q, name
            diff = np.nonzero(ref["hv"].view(np.uint8) != packed)[0]
            assert diff.size == 0, "%s: first differing payload byte %d (%s)" % (
                name, diff[0], "naive stream" if naive else "256-block %d" % (diff[0] // (32 * q)))
            mine.append(dict(ref, hv=packed.view(np.int16), hv_quant_bits=q, hv_norm_2=n2))
        out = os.path.join(td, "mine.sketch")
        hg.write_sketch_file(out, mine)
        a, b = open(REF, "rb").read(), open(out, "rb").read()
        first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
        assert len(a) == len(b) and first is None, "container differs at byte %s (lengths %d / %d)" % (first, len(a), len(b))


@pytest.mark.skipif(not os.path.exists(REF_TSV), reason="no reference-written TSV")
def test_reference_written_tsv_matches_the_oracle_ani(orc):
    import hypergen_amd as hg
    recs = hg.read_sketch_file(REF)
    hv = np.stack([(hg.hv_unpack_naive if hg.hv_payload_layout(r["hv_d"], r["hv_quant_bits"], r["hv"].size * 2) == hg.PAYLOAD_NAIVE
                    else hg.hv_unpack)(r["hv"].view(np.uint8), r["hv_d"], r["hv_quant_bits"]) for r in recs])
    n2 = np.array([r["hv_norm_2"] for r in recs], np.int32)
    want = orc.ani_matrix(hv, n2, hv, n2, 21)
    names = [r["file_str"] for r in recs]
    for line in open(REF_TSV).read().splitlines():
        a, b, v = line.split("\t")
        assert abs(float(v) - want[names.index(a), names.index(b)]) <= 1e-3 + 1e-4
