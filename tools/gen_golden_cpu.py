#!/usr/bin/env python3
"""Writes the CPU-side golden fixtures under tests/golden/.

Sources of the numbers (none is produced by the code under test):
  * kat_t1ha2.json  -- upstream t1ha self-check table for t1ha2_atonce (test pattern,
    seed = 1 << (len-1)), lengths 0..18 and the 64-byte/seed-0 entry.  The values are
    typed in here; the oracle merely has to reproduce them.
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

T1HA2 = [  # (len, seed, value)
    (0, 0, 0x0),
    (0, 2**64 - 1, 0x772C7311BE32FF42),
    (64, 0, 0x444753D23F207E03),
] + [(i + 1, 1 << i, v) for i, v in enumerate([
    0x71F6DF5DA3B4F532, 0x555859635365F660, 0xE98808F1CD39C626, 0x2EB18FAF2163BB09,
    0x7B9DD892C8019C87, 0xE2B1431C4DA4D15A, 0x1984E718A5477F70, 0x08DD17B266484F79,
    0x4C83A05D766AD550, 0x92DCEBB131D1907D, 0xD67BC6FC881B8549, 0xF6A9886555FBF66B,
    0x6E31616D7F33E25E, 0x36E31B7426E3049D, 0x4F8E4FAF46A13F5F, 0x03EB0CB3253F819F,
    0x636A7769905770D2, 0x3ADF3781D16D1148])]

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
    json.dump({"pattern": PATTERN,
               "cases": [{"len": l, "seed": str(s), "hash": "%016x" % v} for l, s, v in T1HA2]},
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
