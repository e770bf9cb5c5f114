"""ctypes loader for oracle/libhg_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (hyper-gen_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhg_oracle.so")

NORM_ACGT, NORM_U2T = 0, 1
LAYOUT_SCALAR, LAYOUT_AVX2 = 0, 1


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
        os.path.join(_HERE, "hg_oracle.c")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libhg_oracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        u8p, u64p, i16p, i32p, f32p = (
            C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int16),
            C.POINTER(C.c_int32), C.POINTER(C.c_float))
        L.orc_t1ha2_atonce.restype = C.c_uint64
        L.orc_t1ha2_atonce.argtypes = [u8p, C.c_size_t, C.c_uint64]
        L.orc_wyrng_next.restype = C.c_uint64
        L.orc_wyrng_next.argtypes = [u64p]
        L.orc_kmer_hash_sample.restype = C.c_size_t
        L.orc_kmer_hash_sample.argtypes = [u8p, C.c_size_t, C.c_uint, C.c_uint64, C.c_uint64,
                                           C.c_int, C.c_int, u64p, C.c_size_t]
        L.orc_sort_unique_u64.restype = C.c_size_t
        L.orc_sort_unique_u64.argtypes = [u64p, C.c_size_t]
        L.orc_read_merge_seq.restype = C.c_size_t
        L.orc_read_merge_seq.argtypes = [u8p, C.c_size_t, u8p]
        L.orc_encode_hv.restype = None
        L.orc_encode_hv.argtypes = [u64p, C.c_size_t, C.c_size_t, C.c_int, i16p]
        L.orc_encode_hv_avx2_emulated.restype = None
        L.orc_encode_hv_avx2_emulated.argtypes = [u64p, C.c_size_t, C.c_size_t, i16p]
        L.orc_hv_norm2.restype = C.c_int32
        L.orc_hv_norm2.argtypes = [i16p, C.c_size_t]
        L.orc_quant_bits.restype = C.c_uint
        L.orc_quant_bits.argtypes = [i16p, C.c_size_t]
        L.orc_pack_hv.restype = None
        L.orc_pack_hv.argtypes = [i16p, C.c_size_t, C.c_uint, u8p]
        L.orc_unpack_hv.restype = None
        L.orc_unpack_hv.argtypes = [u8p, C.c_size_t, C.c_uint, i16p]
        L.orc_packed_words_naive.restype = C.c_size_t
        L.orc_packed_words_naive.argtypes = [C.c_size_t, C.c_uint]
        L.orc_pack_hv_naive.restype = None
        L.orc_pack_hv_naive.argtypes = [i16p, C.c_size_t, C.c_uint, i16p]
        L.orc_unpack_hv_naive.restype = None
        L.orc_unpack_hv_naive.argtypes = [i16p, C.c_size_t, C.c_uint, i16p]
        L.orc_hv_dot.restype = C.c_int32
        L.orc_hv_dot.argtypes = [i16p, i16p, C.c_size_t]
        L.orc_ani_from_dot.restype = C.c_float
        L.orc_ani_from_dot.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_uint]
        L.orc_ani_from_dots.restype = None
        L.orc_ani_from_dots.argtypes = [i32p, i32p, i32p, C.c_size_t, C.c_uint, f32p]
        L.orc_logf_glibc.restype = C.c_float
        L.orc_logf_glibc.argtypes = [C.c_float, C.c_int]
        L.orc_logf_sweep.restype = C.c_uint64
        L.orc_logf_sweep.argtypes = [C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_uint32)]
        L.orc_logf_array.restype = None
        L.orc_logf_array.argtypes = [f32p, C.c_size_t, f32p, C.c_int]
        L.orc_ani_matrix.restype = None
        L.orc_ani_matrix.argtypes = [i16p, i32p, C.c_size_t, i16p, i32p, C.c_size_t,
                                     C.c_size_t, C.c_uint, f32p]
        L.orc_binarize.restype = None
        L.orc_binarize.argtypes = [i16p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint32)]
        L.orc_hamming_matrix.restype = None
        L.orc_hamming_matrix.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint32), C.c_size_t,
                                         C.c_size_t, C.POINTER(C.c_uint32)]
        L.orc_set_threads.restype = None
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_sketch_genome.restype = C.c_int
        L.orc_sketch_genome.argtypes = [u8p, C.c_size_t, C.c_uint, C.c_uint64, C.c_uint64,
                                        C.c_int, C.c_int, C.c_size_t, C.c_int, i16p, i32p,
                                        C.POINTER(C.c_uint32)]
        L.orc_sketch_batch_mt.restype = C.c_int
        L.orc_sketch_batch_mt.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_uint,
                                          C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int,
                                          i16p, i32p, C.POINTER(C.c_uint32)]
        L.orc_synth_genomes_mt.restype = None
        L.orc_synth_genomes_mt.argtypes = [C.c_uint64, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint32, C.c_int, u8p]
        L.orc_synth_genome.restype = None
        L.orc_synth_genome.argtypes = [C.c_uint64, C.c_size_t, C.c_uint, C.c_uint32, u8p]
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def _bytes_arr(b):
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(0, np.uint8)


def t1ha2_atonce(data, seed):
    a = _bytes_arr(data)
    buf = a if a.size else np.zeros(1, np.uint8)
    return int(lib().orc_t1ha2_atonce(_p(buf, C.c_uint8), a.size, C.c_uint64(seed)))


def wyrng_stream(seed, n):
    st = C.c_uint64(seed)
    return [int(lib().orc_wyrng_next(C.byref(st))) for _ in range(n)]


def kmer_hash_sample(seq, ksize=21, scaled=1500, seed=123, canonical=True, norm=NORM_ACGT,
                     threshold=None, unique=True):
    """Sampled hashes of a merged sequence buffer; sorted-unique by default."""
    a = _bytes_arr(seq)
    thr = (2**64 - 1) // scaled if threshold is None else threshold
    cap = max(1024, a.size // max(1, scaled) * 2 + 1024)
    while True:
        out = np.zeros(cap, np.uint64)
        buf = a if a.size else np.zeros(1, np.uint8)
        n = lib().orc_kmer_hash_sample(_p(buf, C.c_uint8), a.size, ksize, C.c_uint64(thr),
                                       C.c_uint64(seed), int(canonical), norm,
                                       _p(out, C.c_uint64), cap)
        if n <= cap:
            break
        cap = n
    out = out[:n]
    if unique:
        out = np.unique(out)
    return out


def read_merge_seq(text):
    a = _bytes_arr(text)
    out = np.zeros(max(1, a.size), np.uint8)
    buf = a if a.size else np.zeros(1, np.uint8)
    n = lib().orc_read_merge_seq(_p(buf, C.c_uint8), a.size, _p(out, C.c_uint8))
    return out[:n].copy()


def read_needletail(text):
    """What the reference's CPU path feeds its k-mer walk (src/sketch.rs:76-87) for one file, in the
    read_merge_seq layout ('N' per record start): needletail 0.5.1 `parse_fastx_file` picks FASTA or FASTQ from
    the first byte; a FASTA record's sequence is every line up to the next '>' line, a FASTQ record is 4 lines;
    `normalize(false)` drops blanks / tabs / CR / LF inside the sequence (upper-casing, u/U -> T and the
    non-ACGT breaks are the k-mer walk's NORM_U2T mode).  Restated from the crate's published behaviour:
    needletail is an un-vendored dependency (Cargo.toml: needletail = "0.5.1"), parity unpinned."""
    data = bytes(_bytes_arr(text))
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    strip = lambda ln: bytes(c for c in ln if c not in b" \t\r\n")
    out = bytearray()
    if data[:1] == b"@":
        for i, ln in enumerate(lines):
            if i % 4 == 0:
                out += b"N"
            elif i % 4 == 1:
                out += strip(ln)
    else:
        for ln in lines:
            out += b"N" if ln[:1] == b">" else strip(ln)
    return np.frombuffer(bytes(out), np.uint8).copy() if out else np.zeros(0, np.uint8)


def encode_hv(hashes, hv_d=4096, layout=LAYOUT_AVX2):
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    hv = np.zeros(hv_d, np.int16)
    buf = h if h.size else np.zeros(1, np.uint64)
    lib().orc_encode_hv(_p(buf, C.c_uint64), h.size, hv_d, layout, _p(hv, C.c_int16))
    return hv


def encode_hv_avx2_emulated(hashes, hv_d=4096):
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    hv = np.zeros(hv_d, np.int16)
    buf = h if h.size else np.zeros(1, np.uint64)
    lib().orc_encode_hv_avx2_emulated(_p(buf, C.c_uint64), h.size, hv_d, _p(hv, C.c_int16))
    return hv


def has_avx2_build():
    """True when libhg_oracle_avx2.so exists and this CPU executes AVX2."""
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " avx2" in flags and os.path.exists(os.path.join(_HERE, "libhg_oracle_avx2.so"))


def encode_hv_avx2_intrinsics(hashes, hv_d=4096):
    """src/hd.rs:14-92 executed with the real AVX2 intrinsics (oracle/hg_oracle_avx2.c)."""
    lib()
    L = C.CDLL(os.path.join(_HERE, "libhg_oracle_avx2.so"))
    L.orc_encode_hv_avx2_intrinsics.restype = None
    L.orc_encode_hv_avx2_intrinsics.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.POINTER(C.c_int16)]
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    hv = np.zeros(hv_d, np.int16)
    buf = h if h.size else np.zeros(1, np.uint64)
    L.orc_encode_hv_avx2_intrinsics(_p(buf, C.c_uint64), h.size, hv_d, _p(hv, C.c_int16))
    return hv


def hv_norm2(hv):
    hv = np.ascontiguousarray(hv, dtype=np.int16)
    return int(lib().orc_hv_norm2(_p(hv, C.c_int16), hv.size))


def quant_bits(hv):
    hv = np.ascontiguousarray(hv, dtype=np.int16)
    return int(lib().orc_quant_bits(_p(hv, C.c_int16), hv.size))


def pack_hv(hv, q=None):
    hv = np.ascontiguousarray(hv, dtype=np.int16)
    q = quant_bits(hv) if q is None else q
    out = np.zeros(q * (hv.size >> 3), np.uint8)  # src/hd.rs:146
    lib().orc_pack_hv(_p(hv, C.c_int16), hv.size, q, _p(out, C.c_uint8))
    return q, out


def unpack_hv(packed, hv_d, q):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    hv = np.zeros(hv_d, np.int16)
    lib().orc_unpack_hv(_p(packed, C.c_uint8), hv_d, q, _p(hv, C.c_int16))
    return hv


def pack_hv_naive(hv, q=None):
    """the non-AVX2 layout (src/hd.rs:158-166): returns (q, i16 words)"""
    hv = np.ascontiguousarray(hv, dtype=np.int16)
    q = quant_bits(hv) if q is None else q
    out = np.zeros(int(lib().orc_packed_words_naive(hv.size, q)), np.int16)
    lib().orc_pack_hv_naive(_p(hv, C.c_int16), hv.size, q, _p(out, C.c_int16))
    return q, out


def unpack_hv_naive(packed, hv_d, q):
    packed = np.ascontiguousarray(packed, dtype=np.int16)
    hv = np.zeros(hv_d, np.int16)
    lib().orc_unpack_hv_naive(_p(packed, C.c_int16), hv_d, q, _p(hv, C.c_int16))
    return hv


def ani_from_dot(dot, nr, nq, ksize=21):
    return float(lib().orc_ani_from_dot(int(dot), int(nr), int(nq), ksize))


def set_threads(n):
    lib().orc_set_threads(int(n))


def ani_from_dots(dot, nr, nq, ksize=21):
    d, a, b = (np.ascontiguousarray(v, np.int32) for v in (dot, nr, nq))
    out = np.empty(d.size, np.float32)
    lib().orc_ani_from_dots(_p(d, C.c_int32), _p(a, C.c_int32), _p(b, C.c_int32), d.size, ksize, _p(out, C.c_float))
    return out


def logf_sweep(first_bits, n, fused=1):
    """(mismatches, first bad bit pattern) of the HOST's logf against the restated glibc algorithm on n consecutive bit patterns"""
    fb = C.c_uint32()
    bad = lib().orc_logf_sweep(first_bits, n, fused, C.byref(fb))
    return int(bad), int(fb.value)


def logf_array(x, form=-1):
    """form -1: the host's logf (what orc_ani_from_dot calls); 0 / 1: glibc's algorithm restated, unfused / fused"""
    a = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(a)
    lib().orc_logf_array(_p(a, C.c_float), a.size, _p(out, C.c_float), form)
    return out


def ani_matrix(ref_hv, ref_n2, qry_hv, qry_n2, ksize=21):
    r = np.ascontiguousarray(ref_hv, dtype=np.int16)
    q = np.ascontiguousarray(qry_hv, dtype=np.int16)
    rn = np.ascontiguousarray(ref_n2, dtype=np.int32)
    qn = np.ascontiguousarray(qry_n2, dtype=np.int32)
    out = np.zeros((r.shape[0], q.shape[0]), np.float32)
    lib().orc_ani_matrix(_p(r, C.c_int16), _p(rn, C.c_int32), r.shape[0], _p(q, C.c_int16),
                         _p(qn, C.c_int32), q.shape[0], r.shape[1], ksize, _p(out, C.c_float))
    return out


def sketch_genome(seq, ksize=21, scaled=1500, seed=123, canonical=True, norm=NORM_ACGT,
                  hv_d=4096, layout=LAYOUT_AVX2):
    a = _bytes_arr(seq)
    hv = np.zeros(hv_d, np.int16)
    n2, nh = C.c_int32(0), C.c_uint32(0)
    buf = a if a.size else np.zeros(1, np.uint8)
    rc = lib().orc_sketch_genome(_p(buf, C.c_uint8), a.size, ksize, C.c_uint64(scaled),
                                 C.c_uint64(seed), int(canonical), norm, hv_d, layout,
                                 _p(hv, C.c_int16), C.byref(n2), C.byref(nh))
    if rc != 0:
        raise MemoryError("orc_sketch_genome failed")
    return hv, int(n2.value), int(nh.value)


def synth_genome(g, L, cluster_size=100, sub_ppm_per_member=1000):
    out = np.zeros(L + 1, np.uint8)
    lib().orc_synth_genome(C.c_uint64(g), L, cluster_size, sub_ppm_per_member,
                           _p(out, C.c_uint8))
    return out


def synth_genomes_mt(first, n, L, threads, cluster_size=100, sub_ppm_per_member=1000):
    out = np.zeros((n, L + 1), np.uint8)
    lib().orc_synth_genomes_mt(C.c_uint64(first), n, L, cluster_size, sub_ppm_per_member, threads,
                               _p(out, C.c_uint8))
    return out


def sketch_batch_mt(seqs, threads, ksize=21, scaled=1500, seed=123, canonical=True, norm=NORM_ACGT,
                    hv_d=4096, layout=LAYOUT_AVX2):
    """seqs: 2-D uint8 array (n x len) or list of 1-D arrays.  OpenMP over genomes."""
    arrs = [np.ascontiguousarray(s, np.uint8) for s in seqs]
    n = len(arrs)
    ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
    lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
    hv = np.zeros((n, hv_d), np.int16)
    n2 = np.zeros(n, np.int32)
    nh = np.zeros(n, np.uint32)
    rc = lib().orc_sketch_batch_mt(ptrs, lens, n, ksize, C.c_uint64(scaled), C.c_uint64(seed), int(canonical),
                                   norm, hv_d, layout, threads, _p(hv, C.c_int16), _p(n2, C.c_int32),
                                   _p(nh, C.c_uint32))
    if rc != 0:
        raise MemoryError("orc_sketch_batch_mt failed")
    return hv, n2, nh


def binarize(hv):
    hv = np.ascontiguousarray(hv, np.int16)
    n, d = hv.shape
    bits = np.zeros((n, (d + 31) // 32), np.uint32)
    lib().orc_binarize(_p(hv, C.c_int16), n, d, _p(bits, C.c_uint32))
    return bits


def hamming_matrix(ref_bits, qry_bits):
    r = np.ascontiguousarray(ref_bits, np.uint32)
    q = np.ascontiguousarray(qry_bits, np.uint32)
    out = np.zeros((r.shape[0], q.shape[0]), np.uint32)
    lib().orc_hamming_matrix(_p(r, C.c_uint32), r.shape[0], _p(q, C.c_uint32), q.shape[0], r.shape[1],
                             _p(out, C.c_uint32))
    return out
