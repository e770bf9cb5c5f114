#!/usr/bin/env python3
"""Writes the CPU-side golden fixtures under tests/golden/.

Sources of the numbers (none is produced by the code under test):
  * kat_t1ha2.json  -- upstream t1ha's self-check table for t1ha2_atonce, all 81 entries (t1ha_refval_2atonce: test
    pattern prefixes of 1..63 bytes, unaligned 57..63-byte inputs, 128..247-byte inputs -- 47 of them longer than 32 bytes,
    the loop src/cuda_kernel.cu omits), typed in here; + DNA strings of every length 33..255 hashed by a Python-integer
    t1ha2 written in this script, which must reproduce the 81 upstream values first.  The oracle and the kernels merely have
    to reproduce both.
  * kat_wyrng.json  -- NOT written here: tools/gen_golden_wyrng.py (the published wyrng definition in Python integers).
  * g1_test_fna.json -- SURVEY.md 8c "G1": the reference's own kernel source
    (src/cuda_kernel.cu) executed on the reference fixture test/test.fna.
  * g4_ani.json     -- src/dist.rs:153-160 evaluated with numpy float32 scalars (the i32 denominator wraps like the
    reference's release build, Cargo.toml:63-65); this script is not the oracle and shares no code with it.  The
    logarithm is glibc's logf algorithm (f32::ln = the C library's logf; sysdeps/ieee754/flt-32/e_logf.c) written out
    in Python doubles below -- NOT numpy's own float32 log, whose last bit differs from glibc's for ~0.5 % of the
    inputs -- so the fixture's values are the floats the reference computes and the tests assert equality.
"""
import json
import os

import numpy as np

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

PATTERN = [0, 1, 2, 3, 4, 5, 6, 7, 0xFF, 0x7F, 0x3F, 0x1F, 0xF, 8, 16, 32, 64, 0x80, 0xFE,
           0xFC, 0xF8, 0xF0, 0xE0, 0xC0, 0xFD, 0xFB, 0xF7, 0xEF, 0xDF, 0xBF, 0x55, 0xAA, 11,
           17, 19, 23, 29, 37, 42, 43] + list(b"abcdefghijklmnopqrstuvwx")

# t1ha_refval_2atonce[81]: upstream's self-check table for t1ha2_atonce, in the order of its t1ha_selfcheck():
#   [0] empty, seed 0   [1] empty, seed ~0   [2] pattern[0:64], seed 0
#   [3..65]  pattern[0:i], seed 1 << (i - 1), i = 1..63        (33..63 bytes: one round of the long-input loop + a tail)
#   [66..72] pattern[i:64], seed ~0 << i, i = 1..7              (unaligned starts, 57..63 bytes)
#   [73..80] long[i : i + 128 + 17 i], seed ~0 << 7, i = 0..7   (long[j] = j & 255: 128..247 bytes, 4..7 rounds)
REFVAL_2ATONCE = [
    0x0000000000000000, 0x772C7311BE32FF42, 0x444753D23F207E03, 0x71F6DF5DA3B4F532, 0x555859635365F660,
    0xE98808F1CD39C626, 0x2EB18FAF2163BB09, 0x7B9DD892C8019C87, 0xE2B1431C4DA4D15A, 0x1984E718A5477F70,
    0x08DD17B266484F79, 0x4C83A05D766AD550, 0x92DCEBB131D1907D, 0xD67BC6FC881B8549, 0xF6A9886555FBF66B,
    0x6E31616D7F33E25E, 0x36E31B7426E3049D, 0x4F8E4FAF46A13F5F, 0x03EB0CB3253F819F, 0x636A7769905770D2,
    0x3ADF3781D16D1148, 0x92D19CB1818BC9C2, 0x283E68F4D459C533, 0xFA83A8A88DECAA04, 0x8C6F00368EAC538C,
    0x7B66B0CF3797B322, 0x5131E122FDABA3FF, 0x6E59FF515C08C7A9, 0xBA2C5269B2C377B0, 0xA9D24FD368FE8A2B,
    0x22DB13D32E33E891, 0x7B97DFC804B876E5, 0xC598BDFCD0E834F9, 0xB256163D3687F5A7, 0x66D7A73C6AEF50B3,
    0x25A7201C85D9E2A3, 0x911573EDA15299AA, 0x5C0062B669E18E4C, 0x17734ADE08D54E28, 0xFFF036E33883F43B,
    0xFE0756E7777DF11E, 0x37972472D023F129, 0x6CFCE201B55C7F57, 0xE019D1D89F02B3E1, 0xAE5CC580FA1BB7E6,
    0x295695FB7E59FC3A, 0x76B6C820A40DD35E, 0xB1680A1768462B17, 0x2FB6AF279137DADA, 0x28FB6B4366C78535,
    0xEC278E53924541B1, 0x164F8AAB8A2A28B5, 0xB6C330AEAC4578AD, 0x7F6F371070085084, 0x94DEAD60C0F448D3,
    0x99737AC232C559EF, 0x6F54A6F9CA8EDD57, 0x979B01E926BFCE0C, 0xF7D20BC85439C5B4, 0x64EDB27CD8087C12,
    0x11488DE5F79C0BE2, 0x25541DDD1680B5A4, 0x8B633D33BE9D1973, 0x404A3113ACF7F6C6, 0xC59DBDEF8550CD56,
    0x039D23C68F4F992C, 0x5BBB48E4BDD6FD86, 0x41E312248780DF5A, 0xD34791CE75D4E94F, 0xED523E5D04DCDCFF,
    0x7A6BCE0B6182D879, 0x21FB37483CAC28D8, 0x19A1B66E8DA878AD, 0x6F804C5295B09ABE, 0x2A4BE5014115BA81,
    0xA678ECC5FC924BE0, 0x50F7A54A99A36F59, 0x0FD7E63A39A66452, 0x5AB1B213DD29C4E4, 0xF3ED80D9DF6534C5,
    0xC736B12EF90615FD]
M64 = 2**64 - 1


def selfcheck_inputs():
    """(data, seed, value, name) in upstream's order"""
    pat, long_ = bytes(PATTERN), bytes(j & 255 for j in range(512))
    out = [(b"", 0, "empty-zero"), (b"", M64, "empty-all1"), (pat[:64], 0, "bin64-zero")]
    out += [(pat[:i], 1 << (i - 1), "bin%02d-1p%02d" % (i, i - 1)) for i in range(1, 64)]
    out += [(pat[i:64], (M64 << i) & M64, "align%d_F%d" % (i, i)) for i in range(1, 8)]
    out += [(long_[i:i + 128 + 17 * i], (M64 << 7) & M64, "long-%05d" % (128 + 17 * i)) for i in range(8)]
    return [(d, s, v, nm) for (d, s, nm), v in zip(out, REFVAL_2ATONCE)]


# ---- t1ha2_atonce in Python integers, from the published description (little-endian, unaligned reads) ------------------
# Written for this script only -- it shares no code with oracle/hg_oracle.c or the kernels -- and checked against all 81
# upstream values above before anything it produces is written.  It then generates the vectors upstream has none for:
# DNA strings of every length 33..255 (the k-mer lengths `-k` can take beyond the reference CUDA kernel's 32).
T1HA_PRIMES = (0xEC99BF0D8372CAAB, 0x82434FE90EDCEF39, 0xD4F06DB99D67BE4B, 0xBD9CACC22C6E9571, 0x9C06FAF4D023E3AB,
               0xC060724A8424F345, 0xCB5AF53AE3AAAC31)


def py_t1ha2_atonce(data, seed):
    rot = lambda v, s: ((v >> s) | (v << (64 - s))) & M64
    le = lambda off, n: int.from_bytes(data[off:off + n], "little")
    p = T1HA_PRIMES
    n, pos = len(data), 0
    st = {"a": seed, "b": n}

    def mixup(x, y, v, prime):  # st[x] ^= lo, st[y] += hi of (st[y] + v) * prime
        m = ((st[y] + v) & M64) * prime
        st[x] ^= m & M64
        st[y] = (st[y] + (m >> 64)) & M64

    if n > 32:
        c = (rot(n, 23) + (~seed & M64)) & M64
        d = ((~n & M64) + rot(seed, 19)) & M64
        a, b = st["a"], st["b"]
        while True:
            w = [le(pos + 8 * t, 8) for t in range(4)]
            pos += 32
            d02 = (w[0] + rot((w[2] + d) & M64, 56)) & M64
            c13 = (w[1] + rot((w[3] + c) & M64, 19)) & M64
            d ^= (b + rot(w[1], 38)) & M64
            c ^= (a + rot(w[0], 57)) & M64
            b ^= (p[6] * ((c13 + w[2]) & M64)) & M64
            a ^= (p[5] * ((d02 + w[3]) & M64)) & M64
            if pos >= n - 31:
                break
        a ^= (p[6] * ((c + rot(d, 23)) & M64)) & M64
        b ^= (p[5] * ((rot(c, 19) + d) & M64)) & M64
        st["a"], st["b"] = a, b
    rem = n - pos if n > 32 else n
    rem &= 31 if n > 32 else M64
    # the tail: up to four words, primes 4, 3, 2, 1, alternating which of (a, b) takes the low half
    plan = [("a", "b", 4, 24), ("b", "a", 3, 16), ("a", "b", 2, 8)]
    for x, y, pi, above in plan:
        if rem > above:
            mixup(x, y, le(pos, 8), p[pi])
            pos, rem = pos + 8, rem - 8
    if rem > 0:
        mixup("b", "a", le(pos, rem), p[1])
    a, b = st["a"], st["b"]
    x = (((a + rot(b, 41)) & M64) * p[0]) & M64
    y = (((rot(a, 23) + b) & M64) * p[6]) & M64
    m = (x ^ y) * p[5]
    return (m & M64) ^ (m >> 64)


G1_FASTA = ">test_seq\nAGCTCTTANNAGCCCNTTacgttacagccctgaaaacttt"  # reference test/test.fna
G1_K21_S1 = ["908794018d1f0246", "967bde3c7bcdbcba", "c0bd0cee44a5f3e0", "e003c78b7d4bace3"]
G1_K5_S1 = """05154f11423d6f2d 1087f45976b8c17d 3736bdc2862be4cf 37490079ec20f055
383d31b8dfc57fad 3f2c505a26d57370 3ffe1a360a93b3aa 440aab0326f64422 45816a9685dc411a
4aafaa6d9d57e55a 4b1e0f3197bbc2f9 4f3dc66c3f45e403 57c55bdcac318125 654aaae8425fb2b4
7a89512fda0db251 928883e5f36a9997 9347173f0ea6d47a 95eee8d29b40209c 998418d696683628
a66dcaadba250048 e02a59e9b5121e75 ec4ebc51bcb9a28f edb41fdf139cc8f5 ff1b1c5541e296b1""".split()


LOGF_TAB = [float.fromhex(v) for v in """
0x1.661ec79f8f3bep+0 -0x1.57bf7808caadep-2 0x1.571ed4aaf883dp+0 -0x1.2bef0a7c06ddbp-2 0x1.49539f0f010bp+0 -0x1.01eae7f513a67p-2
0x1.3c995b0b80385p+0 -0x1.b31d8a68224e9p-3 0x1.30d190c8864a5p+0 -0x1.6574f0ac07758p-3 0x1.25e227b0b8eap+0 -0x1.1aa2bc79c81p-3
0x1.1bb4a4a1a343fp+0 -0x1.a4e76ce8c0e5ep-4 0x1.12358f08ae5bap+0 -0x1.1973c5a611cccp-4 0x1.0953f419900a7p+0 -0x1.252f438e10c1ep-5
0x1p+0 0x0p+0 0x1.e608cfd9a47acp-1 0x1.aa5aa5df25984p-5 0x1.ca4b31f026aap-1 0x1.c5e53aa362eb4p-4 0x1.b2036576afce6p-1
0x1.526e57720db08p-3 0x1.9c2d163a1aa2dp-1 0x1.bc2860d22477p-3 0x1.886e6037841edp-1 0x1.1058bc8a07ee1p-2 0x1.767dcf5534862p-1
0x1.4043057b6ee09p-2""".split()]
LOGF_LN2 = float.fromhex("0x1.62e42fefa39efp-1")
LOGF_A = [float.fromhex(v) for v in ("-0x1.00ea348b88334p-2", "0x1.5575b0be00b6ap-2", "-0x1.ffffef20a4123p-2")]


def glibc_logf(x):
    """glibc >= 2.27 logf(float32) -> float32: table of 16 {1/c, log c}, log(x) = log1p(z/c - 1) + log c + k ln 2, cubic in
    double (Python floats are IEEE doubles; every product rounded on its own -- the -mfma build of the same file, which fuses
    them, returns the same float for all 2^32 inputs: oracle/hg_oracle.c orc_logf_sweep)"""
    f = np.float32
    x = f(x)
    ix = int(x.view(np.uint32))
    if ix == 0x3F800000:
        return f(0.0)
    if (ix - 0x00800000) % 2**32 >= 0x7F800000 - 0x00800000:
        if (ix * 2) % 2**32 == 0:
            return f(-np.inf)
        if ix == 0x7F800000:
            return x
        if (ix & 0x80000000) or (ix * 2) % 2**32 >= 0xFF000000:
            return f(np.nan)
        ix = int(f(x * f(2.0**23)).view(np.uint32)) - (23 << 23)
    tmp = (ix - 0x3F330000) % 2**32
    i = (tmp >> 19) & 15
    k = (tmp >> 23) - (512 if tmp & 0x80000000 else 0)  # arithmetic shift of the int32
    iz = (ix - (tmp & 0xFF800000)) % 2**32
    z = float(np.uint32(iz).view(np.float32))
    invc, logc = LOGF_TAB[2 * i], LOGF_TAB[2 * i + 1]
    r = z * invc - 1.0
    y0 = logc + k * LOGF_LN2
    r2 = r * r
    y = LOGF_A[1] * r + LOGF_A[2]
    y = LOGF_A[0] * r2 + y
    y = y * r2 + (y0 + r)
    return f(y)


def ani_f32(dot, nr, nq, k):
    f = np.float32
    with np.errstate(all="ignore"):
        den = (int(nr) + int(nq) - int(dot) + 2**31) % 2**32 - 2**31  # i32 wrapping sum (release build: no overflow check)
        den = np.int32(den)
        j = f(dot) / f(den)
        ani = f(1.0) + glibc_logf(f(2.0) / (f(1.0) / j + f(1.0))) / f(k)
    if np.isnan(ani):
        return 0.0
    return float(np.maximum(np.minimum(ani, f(1.0)), f(0.0)) * f(100.0))


def main():
    os.makedirs(OUT, exist_ok=True)
    checks = selfcheck_inputs()
    for data, seed, value, name in checks:
        assert py_t1ha2_atonce(data, seed) == value, name  # the Python restatement reproduces all 81 upstream values
    rng_k = np.random.default_rng(2133)
    dna = []
    for n in range(33, 256):  # one DNA string per k-mer length beyond the reference kernel's 32
        sq = bytes(rng_k.choice(np.frombuffer(b"ACGT", np.uint8), n))
        rc = sq.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]  # the other strand: canonical = the smaller of the two strings
        dna.append({"seq": sq.decode(), "seed": "123", "hash": "%016x" % py_t1ha2_atonce(sq, 123),
                    "hash_revcomp": "%016x" % py_t1ha2_atonce(rc, 123)})
    json.dump({"pattern": PATTERN,
               # (kept from round 1: lengths 0..18 and the 64-byte entry, by length)
               "cases": [{"len": 0, "seed": "0", "hash": "%016x" % REFVAL_2ATONCE[0]},
                         {"len": 0, "seed": str(M64), "hash": "%016x" % REFVAL_2ATONCE[1]},
                         {"len": 64, "seed": "0", "hash": "%016x" % REFVAL_2ATONCE[2]}] +
                        [{"len": i, "seed": str(1 << (i - 1)), "hash": "%016x" % REFVAL_2ATONCE[2 + i]} for i in range(1, 19)],
               # the whole upstream table t1ha_refval_2atonce[81], inputs spelled out (hex)
               "upstream_selfcheck": [{"name": nm, "data": d.hex(), "seed": str(sd), "hash": "%016x" % v} for d, sd, v, nm in checks],
               # DNA strings of length 33..255 hashed by this script's own Python-integer t1ha2 (seed 123, the sketch default)
               "dna_33_255": dna},
              open(os.path.join(OUT, "kat_t1ha2.json"), "w"), indent=1)
    json.dump({"fasta": G1_FASTA, "seed": 123, "canonical": True,
               "k21_scaled1": G1_K21_S1, "k5_scaled1": G1_K5_S1, "k21_scaled1500": []},
              open(os.path.join(OUT, "g1_test_fna.json"), "w"), indent=1)
    cases = []
    rng = np.random.default_rng(7)
    tuples = [(13650000, 13650000, 13650000, 21), (0, 100, 100, 21), (-5, 100, 100, 21),
              (50, 100, 100, 21), (1, 1, 1, 31), (9000000, 13000000, 14000000, 21),
              (100, 0, 0, 21), (0, 0, 0, 21), (12000000, 13650000, 13700000, 16),
              # dot = nr = nq (J = 1) at both ends of the range; den = 0 with dot != 0 (+-inf) and dot = 0 (NaN -> 0)
              (1, 1, 1, 21), (2**31 - 1, 2**31 - 1, 2**31 - 1, 21), (7, 7, 7, 1), (5, 2, 3, 21), (-5, -2, -3, 21),
              (0, 5, -5, 21),
              # negative dot products: J < 0 -> ln of a negative number or of a value < e^-k
              (-1, 100, 100, 21), (-13650000, 13650000, 13650000, 21), (-2**31, 1, 1, 21), (-90, 100, 100, 21),
              # J in (0, tiny]: the clamp at 0
              (1, 2**30, 2**30, 21), (1000, 2**30, 2**30, 5),
              # i32-wrapping denominators (nr + nq - dot leaves the i32 range: the reference wraps silently)
              (5, 2**31 - 1, 2**31 - 1, 21), (-10, 2**31 - 1, 5, 21), (2**30, 2**31 - 1, 2**31 - 1, 21),
              (-2**31, 2**31 - 1, 2**31 - 1, 21), (100, -2**31, -2**31, 21), (2**31 - 1, -2**31, 0, 21),
              # norms that wrapped negative in compute_hv_l2_norm (src/dist.rs:132-137) with an ordinary dot
              (12000000, -2000000000, 13650000, 21), (12000000, -2000000000, -2000000000, 21)]
    for _ in range(40):
        nr, nq = (int(x) for x in rng.integers(1_000_000, 30_000_000, 2))
        dot = int(rng.integers(-200_000, min(nr, nq)))
        tuples.append((dot, nr, nq, int(rng.choice([15, 21, 31]))))
    rng2 = np.random.default_rng(70)  # (a second stream: the first 69 cases keep their order)
    for _ in range(600):  # sketch-like: norms ~ D n, dot products spread up to the Cauchy-Schwarz bound
        nr, nq = (int(x) for x in rng2.integers(8_000_000, 20_000_000, 2))
        dot = int((nr * nq) ** 0.5 * rng2.random() ** 0.5)
        tuples.append((dot, nr, nq, int(rng2.choice([16, 21, 31]))))
    for dot, nr, nq, k in tuples:
        cases.append({"dot": dot, "nr": nr, "nq": nq, "k": k, "ani": ani_f32(dot, nr, nq, k)})
    json.dump(cases, open(os.path.join(OUT, "g4_ani.json"), "w"), indent=1)
    print("wrote fixtures to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
