"""hyper-gen_amd -- ctypes binding of libhypergen_hip.so (include/hypergen.h).

This is a harness-side mirror of the C ABI: the product is the shared library (HIP kernels +
C ABI) and the C++ `hyper-gen` CLI next to it.  There is no CPU fallback here: if the
library is missing, or no HIP device is usable, calls raise.

Import it as `hypergen_amd` (see the shim at the repo root; the directory name carries a
hyphen because it is the reference's crate name).

Note for processes that also use PyTorch-ROCm: import torch BEFORE this module loads the library
(torch bundles its own copy of the HIP runtime; loaded second it finds no devices).
"""
import ctypes as C
import os
import threading
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = (os.environ.get("HYPERGEN_LIB") or None) or os.path.join(_HERE, "libhypergen_hip.so")  # override: development builds
CLI_PATH = os.path.join(_HERE, "hyper-gen")

LAYOUT_SCALAR, LAYOUT_AVX2 = 0, 1
GATHER_PEER, GATHER_RCCL = 0, 1
NORM_ACGT, NORM_U2T = 0, 1
(OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_CAPACITY, ERR_UNSUPPORTED, ERR_IO,
 ERR_INEXACT) = range(9)


T_NAMES = ("kmer", "sort", "encode", "dist_prep", "dist")
T_COUNT = len(T_NAMES)


class HgError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("hypergen status %d: %s" % (status, msg))
        self.status = status


class SketchParams(C.Structure):
    _fields_ = [("ksize", C.c_uint32), ("canonical", C.c_uint32), ("scaled", C.c_uint64),
                ("seed", C.c_uint64), ("hv_d", C.c_uint32), ("hv_layout", C.c_uint32),
                ("norm_mode", C.c_uint32), ("reserved", C.c_uint32)]


class AniHit(C.Structure):
    _fields_ = [("ref_idx", C.c_uint32), ("qry_idx", C.c_uint32), ("ani", C.c_float)]


ANI_HIT_DTYPE = np.dtype([("ref_idx", "<u4"), ("qry_idx", "<u4"), ("ani", "<f4")])
HAM_HIT_DTYPE = np.dtype([("ref_idx", "<u4"), ("qry_idx", "<u4"), ("dist", "<u4")])


class FileSketch(C.Structure):
    _fields_ = [("ksize", C.c_uint8), ("canonical", C.c_uint8), ("hv_quant_bits", C.c_uint8),
                ("pad", C.c_uint8), ("hv_norm_2", C.c_int32), ("scaled", C.c_uint64),
                ("seed", C.c_uint64), ("hv_d", C.c_uint64), ("file_str", C.c_char_p),
                ("hv", C.POINTER(C.c_int16)), ("hv_len", C.c_uint64)]


EXPORTS = [
    "hg_status_str", "hg_last_error", "hg_version", "hg_ctx_create", "hg_ctx_destroy",
    "hg_ctx_set_stream", "hg_ctx_reset_stream", "hg_ctx_sync", "hg_ctx_sketch_step_counts", "hg_sketch_plan_describe", "hg_logf_dev", "hg_ani_from_dots_dev", "hg_device_count", "hg_dev_alloc", "hg_dev_free",
    "hg_copy_h2d", "hg_copy_d2h", "hg_sketch_params_default", "hg_kmer_hash_sample",
    "hg_hv_encode", "hg_sketch_batch_dev", "hg_sketch_batch", "hg_dist_full", "hg_dist_full_dev",
    "hg_dist", "hg_dist_dev", "hg_sort_ani_hits", "hg_hv_quant_bits", "hg_hv_pack", "hg_hv_packed_bytes",
    "hg_hv_unpack", "hg_sketch_file_write", "hg_sketch_file_read", "hg_sketch_file_count",
    "hg_sketch_file_get", "hg_sketch_file_free", "hg_read_merge_seq", "hg_read_merge_seq_into", "hg_free",
    "hg_synth_genomes_dev", "hg_ctx_enable_timing", "hg_ctx_timings", "hg_ctx_last_kernel",
    "hg_hv_binarize_dev", "hg_hamming_full_dev", "hg_hamming_search_dev",
    "hg_ctx_set_debug", "hg_read_fastx_into", "hg_dist_block_dev", "hg_hamming_search_block_dev",
    "hg_multi_create", "hg_multi_destroy", "hg_multi_size", "hg_multi_ctx", "hg_multi_last_error", "hg_shard_range",
    "hg_multi_peer_report", "hg_multi_set_gather", "hg_multi_gather_mode", "hg_multi_gather_report",
    "hg_sketch_batch_multi", "hg_dist_multi", "hg_dist_multi_dev", "hg_hamming_search_multi",
    "hg_sort_ani_hits_dev", "hg_sort_ani_hits_staged", "hg_topk_per_query_dev", "hg_ctx_last_dist_path",
    "hg_ctx_last_hamming_path", "hg_read_fastx_pinned", "hg_pinned_free",
    "hg_sketch_stream_open", "hg_sketch_stream_push", "hg_sketch_stream_pop", "hg_sketch_stream_finish",
    "hg_sketch_stream_last_error", "hg_sketch_stream_close", "hg_sketch_stream_stats", "hg_device_numa_node", "hg_bind_thread_to_numa_node", "hg_dist_tile_order",
    "hg_sketch_stream_push_packed", "hg_pack2_size", "hg_pack2", "hg_unpack2_dev",
    "hg_sketch_stream_try_push", "hg_sketch_stream_max_pending",
    "hg_sketch_batch_dev_packed", "hg_pack2_batch_dev", "hg_pack2_dev",
    "hg_pack2s_size", "hg_pack2s", "hg_sketch_stream_push_packed_sparse",
    "hg_dist_ops_row_bytes", "hg_dist_ops_meta_bytes", "hg_dist_ops_padded_rows", "hg_dist_prep_ops_dev", "hg_dist_block_ops_dev",
    "hg_hv_packed_bytes_naive", "hg_hv_pack_naive", "hg_hv_unpack_naive", "hg_hv_payload_layout", "hg_hv_unpack_batch_dev",
    "hg_sketch_file_read_image", "hg_sketch_file_image", "hg_sketch_file_payload_offset",
]


def source_stamp():
    """sha256 (first 16 hex digits) over the library's sources (csrc/* but the command-line tool hg_cli.cpp, which is not
    part of libhypergen_hip.so; include/hypergen.h), names and contents in sorted order.  tools/summarize_prof.py writes it into every profile summary and bench.py quotes a summary's counters only
    when the stamp equals that of the tree it runs from -- a kernel change without a re-profile then reports null
    instead of stale counters."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(_HERE, "csrc")
    files = sorted(f for f in os.listdir(src) if (f.endswith((".hip", ".h", ".cpp")) or f == "Makefile") and f != "hg_cli.cpp")
    for f in [os.path.join(src, x) for x in files] + [os.path.join(_HERE, "..", "include", "hypergen.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cli_source_stamp():
    """source_stamp() extended by the command-line tool's source: what a measurement of the `hyper-gen` binary belongs to
    (bench.py's `cli` object and the profiles/rNN_cli_* files carry it)."""
    import hashlib
    h = hashlib.sha256(source_stamp().encode())
    h.update(open(os.path.join(_HERE, "csrc", "hg_cli.cpp"), "rb").read())
    return h.hexdigest()[:16]


def build(force=False):
    """Compile the library and CLI in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(_HERE, "csrc"), "all"])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libhypergen_hip.so is not built: run `python -c 'import __graft_entry__ as g; "
                          "g.build()'` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, sz = C.c_void_p, C.c_size_t
    u8p, u64p, i16p, i32p, u32p, f32p = (C.POINTER(t) for t in (
        C.c_uint8, C.c_uint64, C.c_int16, C.c_int32, C.c_uint32, C.c_float))
    sig = {
        "hg_status_str": (C.c_char_p, [C.c_int]),
        "hg_last_error": (C.c_char_p, [vp]),
        "hg_version": (C.c_char_p, []),
        "hg_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "hg_ctx_destroy": (None, [vp]),
        "hg_ctx_set_stream": (C.c_int, [vp, vp]),
        "hg_ctx_reset_stream": (C.c_int, [vp]),
        "hg_ctx_sync": (C.c_int, [vp]),
        "hg_sketch_plan_describe": (C.c_int, [vp, vp, sz, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), vp, sz]),
        "hg_logf_dev": (C.c_int, [vp, vp, C.c_uint32, sz, vp]),
        "hg_ani_from_dots_dev": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint32, vp]),
        "hg_ctx_sketch_step_counts": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "hg_device_count": (C.c_int, []),
        "hg_dev_alloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
        "hg_dev_free": (C.c_int, [vp, vp]),
        "hg_copy_h2d": (C.c_int, [vp, vp, vp, sz]),
        "hg_copy_d2h": (C.c_int, [vp, vp, vp, sz]),
        "hg_sketch_params_default": (None, [C.POINTER(SketchParams)]),
        "hg_kmer_hash_sample": (C.c_int, [vp, vp, sz, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int,
                                          C.c_uint32, vp, sz, C.POINTER(sz)]),
        "hg_hv_encode": (C.c_int, [vp, vp, sz, C.c_uint32, C.c_uint32, vp, C.POINTER(C.c_int32)]),
        "hg_sketch_batch_dev": (C.c_int, [vp, vp, vp, vp, sz, C.POINTER(SketchParams), vp, vp, vp]),
        "hg_sketch_batch": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(SketchParams),
                                      vp, vp, vp]),
        "hg_dist_full": (C.c_int, [vp, vp, vp, sz, vp, vp, sz, C.c_uint32, C.c_uint32, vp]),
        "hg_dist_full_dev": (C.c_int, [vp, vp, vp, sz, vp, vp, sz, C.c_uint32, C.c_uint32, vp]),
        "hg_dist": (C.c_int, [vp, vp, vp, sz, vp, vp, sz, C.c_uint32, C.c_uint32, C.c_int, C.c_float,
                              vp, sz, C.POINTER(sz)]),
        "hg_dist_dev": (C.c_int, [vp, vp, vp, sz, vp, vp, sz, C.c_uint32, C.c_uint32, C.c_int,
                                  C.c_float, vp, sz, C.POINTER(sz)]),
        "hg_sort_ani_hits": (None, [vp, sz, sz, C.c_int]),
        "hg_hv_quant_bits": (C.c_uint32, [vp, C.c_uint32]),
        "hg_hv_pack": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "hg_hv_packed_bytes": (sz, [C.c_uint32, C.c_uint32]),
        "hg_hv_unpack": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "hg_sketch_file_write": (C.c_int, [C.c_char_p, C.POINTER(FileSketch), sz]),
        "hg_sketch_file_read": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "hg_sketch_file_count": (sz, [vp]),
        "hg_sketch_file_get": (C.POINTER(FileSketch), [vp, sz]),
        "hg_sketch_file_free": (None, [vp]),
        "hg_read_merge_seq": (C.c_int, [C.c_char_p, C.POINTER(vp), C.POINTER(sz)]),
        "hg_read_merge_seq_into": (C.c_int, [C.c_char_p, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]),
        "hg_read_fastx_into": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]),
        "hg_ctx_set_debug": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
        "hg_free": (None, [vp]),
        "hg_read_fastx_pinned": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]),
        "hg_pinned_free": (None, [vp]),
        "hg_device_numa_node": (C.c_int, [C.c_int]),
        "hg_bind_thread_to_numa_node": (C.c_int, [C.c_int, C.c_uint]),
        "hg_dist_tile_order": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
                                         C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
        "hg_sketch_stream_open": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(SketchParams), C.POINTER(vp)]),
        "hg_sketch_stream_push": (C.c_int, [vp, vp, sz, C.c_uint64]),
        "hg_sketch_stream_pop": (C.c_int, [vp, C.POINTER(C.c_uint64), vp, C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                                           C.POINTER(C.c_int)]),
        "hg_sketch_stream_push_packed": (C.c_int, [vp, vp, sz, C.c_uint64]),
        "hg_sketch_stream_try_push": (C.c_int, [vp, vp, sz, C.c_uint64, C.c_int]),
        "hg_sketch_stream_max_pending": (sz, [vp]),
        "hg_pack2_size": (sz, [sz]),
        "hg_pack2": (C.c_int, [vp, sz, C.c_uint32, vp]),
        "hg_unpack2_dev": (C.c_int, [vp, vp, sz, vp]),
        "hg_sketch_batch_dev_packed": (C.c_int, [vp, vp, vp, vp, sz, C.POINTER(SketchParams), vp, vp, vp]),
        "hg_pack2_batch_dev": (C.c_int, [vp, vp, vp, vp, sz, C.c_uint32, vp, vp]),
        "hg_pack2_dev": (C.c_int, [vp, vp, sz, C.c_uint32, vp]),
        "hg_dist_ops_row_bytes": (sz, [C.c_uint32]),
        "hg_dist_ops_meta_bytes": (sz, []),
        "hg_dist_ops_padded_rows": (sz, [sz]),
        "hg_dist_prep_ops_dev": (C.c_int, [vp, vp, sz, C.c_uint32, vp, vp, vp]),
        "hg_dist_block_ops_dev": (C.c_int, [vp, vp, vp, vp, sz, sz, vp, vp, sz, vp, vp, sz, sz, C.c_uint32, C.c_uint32, C.c_int,
                                            C.c_float, vp, sz, C.POINTER(sz)]),
        "hg_pack2s_size": (sz, [sz, sz]),
        "hg_pack2s": (C.c_int, [vp, sz, C.c_uint32, vp, sz, C.POINTER(sz)]),
        "hg_sketch_stream_push_packed_sparse": (C.c_int, [vp, vp, sz, sz, C.c_uint64]),
        "hg_sketch_stream_finish": (C.c_int, [vp]),
        "hg_sketch_stream_last_error": (C.c_char_p, [vp]),
        "hg_sketch_stream_close": (None, [vp]),
        "hg_sketch_stream_stats": (C.c_int, [vp, C.c_int, C.POINTER(C.c_double)]),
        "hg_ctx_last_dist_path": (C.c_int, [vp]),
        "hg_ctx_last_hamming_path": (C.c_int, [vp]),
        "hg_sort_ani_hits_dev": (C.c_int, [vp, vp, sz, sz]),
        "hg_sort_ani_hits_staged": (C.c_int, [vp, vp, sz, sz]),
        "hg_topk_per_query_dev": (C.c_int, [vp, vp, sz, sz, C.c_uint32, vp, vp]),
        "hg_dist_block_dev": (C.c_int, [vp, vp, vp, sz, sz, vp, vp, sz, sz, C.c_uint32, C.c_uint32, C.c_int,
                                        C.c_float, vp, sz, C.POINTER(sz)]),
        "hg_hamming_search_block_dev": (C.c_int, [vp, vp, sz, sz, vp, sz, sz, C.c_uint32, C.c_uint32, vp, sz,
                                                  C.POINTER(sz)]),
        "hg_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
        "hg_multi_destroy": (None, [vp]),
        "hg_multi_size": (C.c_int, [vp]),
        "hg_multi_ctx": (vp, [vp, C.c_int]),
        "hg_multi_last_error": (C.c_char_p, [vp]),
        "hg_multi_peer_report": (C.c_char_p, [vp]),
        "hg_multi_gather_report": (C.c_char_p, [vp]),
        "hg_multi_set_gather": (C.c_int, [vp, C.c_int]),
        "hg_multi_gather_mode": (C.c_int, [vp]),
        "hg_shard_range": (None, [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
        "hg_sketch_batch_multi": (C.c_int, [vp, C.POINTER(vp), C.POINTER(sz), sz, C.POINTER(SketchParams),
                                            vp, vp, vp]),
        "hg_dist_multi": (C.c_int, [vp, vp, vp, sz, vp, vp, sz, C.c_uint32, C.c_uint32, C.c_int, C.c_float,
                                    vp, sz, C.POINTER(sz)]),
        "hg_dist_multi_dev": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(sz), C.POINTER(vp),
                                        C.POINTER(vp), C.POINTER(sz), C.c_uint32, C.c_uint32, C.c_int,
                                        C.c_float, vp, sz, C.POINTER(sz)]),
        "hg_hamming_search_multi": (C.c_int, [vp, vp, sz, vp, sz, C.c_uint32, C.c_uint32, vp, sz, C.POINTER(sz)]),
        "hg_hv_binarize_dev": (C.c_int, [vp, vp, sz, C.c_uint32, vp]),
        "hg_hamming_full_dev": (C.c_int, [vp, vp, sz, vp, sz, C.c_uint32, vp]),
        "hg_hamming_search_dev": (C.c_int, [vp, vp, sz, vp, sz, C.c_uint32, C.c_uint32, vp, sz, C.POINTER(sz)]),
        "hg_ctx_enable_timing": (C.c_int, [vp, C.c_int]),
        "hg_ctx_last_kernel": (C.c_char_p, [vp, C.c_int]),
        "hg_ctx_timings": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
        "hg_synth_genomes_dev": (C.c_int, [vp, C.c_uint64, sz, C.c_uint64, C.c_uint32, C.c_uint32,
                                           C.c_uint64, vp]),
        "hg_hv_packed_bytes_naive": (sz, [C.c_uint32, C.c_uint32]),
        "hg_hv_pack_naive": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "hg_hv_unpack_naive": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp]),
        "hg_hv_payload_layout": (C.c_int, [C.c_uint32, C.c_uint32, sz]),
        "hg_hv_unpack_batch_dev": (C.c_int, [vp, vp, sz, vp, vp, vp, sz, C.c_uint32, vp]),
        "hg_sketch_file_read_image": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "hg_sketch_file_image": (vp, [vp, C.POINTER(sz)]),
        "hg_sketch_file_payload_offset": (C.c_uint64, [vp, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == the ABI lost a symbol
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if isinstance(a, np.ndarray) else C.c_void_p(int(a))


def dist_tile_order(tiles_m, tiles_n, tile_rows=256, tile_cols=320, diagonal_first=False, symmetric=False, ref_off=0, qry_off=0):
    """hg_dist_tile_order: the slot -> tile table of a launch as a uint32 array (tm | tn << 16, 0xFFFFFFFF = no tile)"""
    n = C.c_size_t(0)
    st = lib().hg_dist_tile_order(tiles_m, tiles_n, tile_rows, tile_cols, int(diagonal_first), int(symmetric), ref_off, qry_off,
                                  None, 0, C.byref(n))
    if st not in (OK, ERR_CAPACITY):
        raise HgError(st, "hg_dist_tile_order")
    out = np.zeros(max(n.value, 1), np.uint32)
    st = lib().hg_dist_tile_order(tiles_m, tiles_n, tile_rows, tile_cols, int(diagonal_first), int(symmetric), ref_off, qry_off,
                                  C.c_void_p(out.ctypes.data), out.size, C.byref(n))
    if st != OK:
        raise HgError(st, "hg_dist_tile_order")
    return out[: n.value]


def default_params(**kw):
    p = SketchParams()
    lib().hg_sketch_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


# Harness-side defaults for hg_ctx_set_debug, applied to every Context created while they are set (the test-suite's
# "run this module in both input forms" switch: tests/conftest.py sets {"kmer_input": "packed"}).  The library itself
# reads neither this nor the environment.
default_debug = {}


class Context:
    """One device + stream + workspaces (hg_ctx).  Not thread-safe."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = lib().hg_ctx_create(device, C.byref(self._h))
        if st != OK:
            raise HgError(st, lib().hg_last_error(None).decode())
        for k, v in default_debug.items():
            self.set_debug(k, v)

    def close(self):
        if self._h:
            lib().hg_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st, allow=()):
        if st != OK and st not in allow:
            raise HgError(st, lib().hg_last_error(self._h).decode())
        return st

    def set_stream(self, stream_handle):
        """Run on the given hipStream_t handle; 0 / None is HIP's default stream (torch's default current stream)."""
        self._ck(lib().hg_ctx_set_stream(self._h, C.c_void_p(stream_handle or 0)))

    def set_debug(self, key, value):
        """Development / test hook: force an internal code path of this ctx (hg_ctx_set_debug)."""
        self._ck(lib().hg_ctx_set_debug(self._h, key.encode(), str(value).encode()))

    def reset_stream(self):
        self._ck(lib().hg_ctx_reset_stream(self._h))

    def sync(self):
        """hg_ctx_sync: reads a queued sketch step's check word (re-running the step if it asks for it), then waits for the stream."""
        self._ck(lib().hg_ctx_sync(self._h))

    def logf_dev(self, d_out, n, d_x=None, first_bits=0):
        """hg_logf_dev: the library's logf of n floats (d_x), or of the bit patterns first_bits .. first_bits + n - 1"""
        self._ck(lib().hg_logf_dev(self._h, C.c_void_p(d_x or 0), first_bits, n, C.c_void_p(d_out)))

    def ani_from_dots_dev(self, d_dot, d_nr, d_nq, n, ksize, d_ani):
        self._ck(lib().hg_ani_from_dots_dev(self._h, C.c_void_p(d_dot), C.c_void_p(d_nr), C.c_void_p(d_nq), n, ksize, C.c_void_p(d_ani)))

    def sketch_step_counts(self):
        """(sync-free, synchronous, re-run) sketch steps of this ctx so far (hg_ctx_sketch_step_counts)."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._ck(lib().hg_ctx_sketch_step_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def last_kernel(self, cls):
        """Name of the kernel the last call launched for a timing class ("kmer", "dist"), as rocprofv3 prints it."""
        return lib().hg_ctx_last_kernel(self._h, T_NAMES.index(cls)).decode()

    def enable_timing(self, on=True):
        self._ck(lib().hg_ctx_enable_timing(self._h, int(on)))

    def timings(self):
        """{kernel class: (sum of launch durations in ms, launches)} since the last call."""
        ms = (C.c_float * T_COUNT)()
        n = (C.c_uint32 * T_COUNT)()
        self._ck(lib().hg_ctx_timings(self._h, ms, n))
        return {name: (float(ms[i]), int(n[i])) for i, name in enumerate(T_NAMES)}

    # ---- host-buffer entry points -----------------------------------------------------------
    def kmer_hash_sample(self, seq, ksize=21, scaled=1500, seed=123, canonical=True,
                         norm=NORM_ACGT, threshold=None, cap=None):
        a = np.ascontiguousarray(np.frombuffer(bytes(seq), np.uint8) if not isinstance(seq, np.ndarray) else seq,
                                 dtype=np.uint8)
        thr = (2**64 - 1) // scaled if threshold is None else threshold
        cap = cap if cap is not None else max(1024, a.size // max(1, scaled) * 2 + 1024)
        while True:
            out = np.zeros(max(cap, 1), np.uint64)
            n = C.c_size_t(0)
            st = lib().hg_kmer_hash_sample(self._h, _ptr(a) if a.size else None, a.size, ksize,
                                           C.c_uint64(thr), C.c_uint64(seed), int(canonical), norm,
                                           _ptr(out), cap, C.byref(n))
            if st == ERR_CAPACITY:
                cap = n.value
                continue
            self._ck(st)
            return out[: n.value].copy()

    def unpack2_dev(self, d_blob, n_bps, d_out):
        """hg_unpack2_dev on device pointers (ints)"""
        self._ck(lib().hg_unpack2_dev(self._h, C.c_void_p(d_blob), n_bps, C.c_void_p(d_out)))

    def hv_encode(self, hashes, hv_d=4096, layout=LAYOUT_AVX2):
        h = np.ascontiguousarray(hashes, dtype=np.uint64)
        hv = np.zeros(hv_d, np.int16)
        n2 = C.c_int32(0)
        self._ck(lib().hg_hv_encode(self._h, _ptr(h) if h.size else None, h.size, hv_d, layout,
                                    _ptr(hv), C.byref(n2)))
        return hv, int(n2.value)

    def sketch_batch(self, seqs, params=None):
        p = params or default_params()
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) if isinstance(s, np.ndarray)
                else np.frombuffer(bytes(s), np.uint8) for s in seqs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
        hv = np.zeros((n, p.hv_d), np.int16)
        n2 = np.zeros(n, np.int32)
        nh = np.zeros(n, np.uint32)
        self._ck(lib().hg_sketch_batch(self._h, ptrs, lens, n, C.byref(p), _ptr(hv), _ptr(n2), _ptr(nh)))
        return hv, n2, nh

    def dist_full(self, ref_hv, ref_n2, qry_hv, qry_n2, ksize=21):
        r = np.ascontiguousarray(ref_hv, np.int16)
        q = np.ascontiguousarray(qry_hv, np.int16)
        rn = np.ascontiguousarray(ref_n2, np.int32)
        qn = np.ascontiguousarray(qry_n2, np.int32)
        out = np.zeros((r.shape[0], q.shape[0]), np.float32)
        self._ck(lib().hg_dist_full(self._h, _ptr(r), _ptr(rn), r.shape[0], _ptr(q), _ptr(qn),
                                    q.shape[0], r.shape[1], ksize, _ptr(out)))
        return out

    def dist(self, ref_hv, ref_n2, qry_hv, qry_n2, ksize=21, symmetric=False, ani_th=85.0, cap=None):
        r = np.ascontiguousarray(ref_hv, np.int16)
        q = np.ascontiguousarray(qry_hv, np.int16)
        rn = np.ascontiguousarray(ref_n2, np.int32)
        qn = np.ascontiguousarray(qry_n2, np.int32)
        cap = cap if cap is not None else max(1024, r.shape[0] * q.shape[0] // 8)
        while True:
            out = np.zeros(cap, ANI_HIT_DTYPE)
            n = C.c_size_t(0)
            st = lib().hg_dist(self._h, _ptr(r), _ptr(rn), r.shape[0], _ptr(q), _ptr(qn), q.shape[0],
                               r.shape[1], ksize, int(symmetric), C.c_float(ani_th), _ptr(out), cap,
                               C.byref(n))
            if st == ERR_CAPACITY:
                cap = n.value
                continue
            self._ck(st)
            return out[: n.value].copy()

    # ---- device-resident entry points (pointers are ints, e.g. torch.Tensor.data_ptr()) -------
    def sketch_batch_dev(self, d_seq, offsets, lens, params, d_hv, d_norm2, d_nhash):
        off = np.ascontiguousarray(offsets, np.uint64)
        ln = np.ascontiguousarray(lens, np.uint64)
        self._ck(lib().hg_sketch_batch_dev(self._h, _ptr(d_seq), _ptr(off), _ptr(ln), off.size,
                                           C.byref(params), _ptr(d_hv), _ptr(d_norm2), _ptr(d_nhash)))

    def sketch_batch_dev_packed(self, d_blobs, offsets, n_bps, params, d_hv, d_norm2, d_nhash):
        """hg_sketch_batch_dev_packed: genome i is the hg_pack2 blob at d_blobs + offsets[i] (multiples of 16)"""
        off = np.ascontiguousarray(offsets, np.uint64)
        ln = np.ascontiguousarray(n_bps, np.uint64)
        self._ck(lib().hg_sketch_batch_dev_packed(self._h, _ptr(d_blobs), _ptr(off), _ptr(ln), off.size,
                                                  C.byref(params), _ptr(d_hv), _ptr(d_norm2), _ptr(d_nhash)))

    def pack2_batch_dev(self, d_seq, offsets, lens, d_blobs, blob_offsets, norm_mode=NORM_ACGT):
        """hg_pack2_batch_dev: ASCII genomes resident in HBM -> hg_pack2 blobs (device pointers are ints)"""
        off = np.ascontiguousarray(offsets, np.uint64)
        ln = np.ascontiguousarray(lens, np.uint64)
        bo = np.ascontiguousarray(blob_offsets, np.uint64)
        self._ck(lib().hg_pack2_batch_dev(self._h, _ptr(d_seq), _ptr(off), _ptr(ln), off.size, norm_mode, _ptr(d_blobs), _ptr(bo)))

    def pack2_dev(self, d_seq, n_bps, d_blob, norm_mode=NORM_ACGT):
        self._ck(lib().hg_pack2_dev(self._h, C.c_void_p(d_seq), n_bps, norm_mode, C.c_void_p(d_blob)))

    def synth_genomes_dev(self, first, n, L, stride, d_out, cluster_size=100, sub_ppm_per_member=1000):
        done = 0
        while done < n:  # grid.y limit
            m = min(n - done, 32768)
            self._ck(lib().hg_synth_genomes_dev(self._h, first + done, m, L, cluster_size, sub_ppm_per_member,
                                                stride, C.c_void_p(int(d_out) + done * stride)))
            done += m

    # ---- bit-packed extension (device pointers) ---------------------------------------------------
    def hv_binarize_dev(self, d_hv, n, hv_d, d_bits):
        self._ck(lib().hg_hv_binarize_dev(self._h, _ptr(d_hv), n, hv_d, _ptr(d_bits)))

    def hamming_full_dev(self, d_ref, R, d_qry, Q, hv_d, d_out):
        self._ck(lib().hg_hamming_full_dev(self._h, _ptr(d_ref), R, _ptr(d_qry), Q, hv_d, _ptr(d_out)))

    def hamming_search_dev(self, d_ref, R, d_qry, Q, hv_d, max_dist, d_out, cap):
        n = C.c_size_t(0)
        st = lib().hg_hamming_search_dev(self._h, _ptr(d_ref), R, _ptr(d_qry), Q, hv_d, max_dist, _ptr(d_out), cap,
                                         C.byref(n))
        self._ck(st, allow=(ERR_CAPACITY,))
        return n.value, st

    def dist_full_dev(self, d_ref, d_rn, R, d_qry, d_qn, Q, hv_d, ksize, d_out):
        self._ck(lib().hg_dist_full_dev(self._h, _ptr(d_ref), _ptr(d_rn), R, _ptr(d_qry), _ptr(d_qn), Q,
                                        hv_d, ksize, _ptr(d_out)))

    def dist_block_dev(self, d_ref, d_rn, R, ref_off, d_qry, d_qn, Q, qry_off, hv_d, ksize, symmetric, ani_th,
                       d_out, cap):
        n = C.c_size_t(0)
        st = lib().hg_dist_block_dev(self._h, _ptr(d_ref), _ptr(d_rn), R, ref_off, _ptr(d_qry), _ptr(d_qn), Q,
                                     qry_off, hv_d, ksize, int(symmetric), C.c_float(ani_th), _ptr(d_out), cap,
                                     C.byref(n))
        self._ck(st, allow=(ERR_CAPACITY,))
        return n.value, st

    def dist_prep_ops_dev(self, d_hv, rows, hv_d, d_ops, d_meta, d_flag):
        """hg_dist_prep_ops_dev: byte operands + control records of this rank's reference rows (device pointers)"""
        self._ck(lib().hg_dist_prep_ops_dev(self._h, _ptr(d_hv), rows, hv_d, _ptr(d_ops), _ptr(d_meta), _ptr(d_flag)))

    def dist_block_ops_dev(self, d_ref_ops, d_ref_meta, d_rn, R, ref_off, d_ref_index, d_flags, n_flags, d_qry, d_qn, Q, qry_off,
                           hv_d, ksize, symmetric, ani_th, d_out, cap):
        """hg_dist_block_ops_dev; returns (hits, status) -- status ERR_INEXACT: vetoed, fall back to dist_block_dev"""
        n = C.c_size_t(0)
        st = lib().hg_dist_block_ops_dev(self._h, _ptr(d_ref_ops), _ptr(d_ref_meta), _ptr(d_rn), R, ref_off,
                                         _ptr(d_ref_index) if d_ref_index else None, _ptr(d_flags) if d_flags else None, n_flags,
                                         _ptr(d_qry), _ptr(d_qn), Q, qry_off, hv_d, ksize, int(symmetric), C.c_float(ani_th),
                                         _ptr(d_out), cap, C.byref(n))
        self._ck(st, allow=(ERR_CAPACITY, ERR_INEXACT))
        return n.value, st

    def hamming_search_block_dev(self, d_ref, R, ref_off, d_qry, Q, qry_off, hv_d, max_dist, d_out, cap):
        n = C.c_size_t(0)
        st = lib().hg_hamming_search_block_dev(self._h, _ptr(d_ref), R, ref_off, _ptr(d_qry), Q, qry_off, hv_d,
                                               max_dist, _ptr(d_out), cap, C.byref(n))
        self._ck(st, allow=(ERR_CAPACITY,))
        return n.value, st

    def hv_unpack_batch_dev(self, d_payloads, payloads_bytes, offsets, quant_bits, layouts, hv_d, d_hv):
        """decompress_file_sketch on the device: payload i at d_payloads + offsets[i] -> row i of d_hv (n x hv_d int16)"""
        offsets = np.ascontiguousarray(offsets, np.uint64)
        quant_bits = np.ascontiguousarray(quant_bits, np.uint8)
        lay = None if layouts is None else np.ascontiguousarray(layouts, np.uint8)
        self._ck(lib().hg_hv_unpack_batch_dev(self._h, d_payloads, payloads_bytes, _ptr(offsets), _ptr(quant_bits),
                                              None if lay is None else _ptr(lay), offsets.size, hv_d, d_hv))

    def last_dist_path(self):
        """0 = f16 MFMA, 1 = i8 MFMA, 2 = integer VALU (all exact), -1 = no thresholded dist call yet."""
        return int(lib().hg_ctx_last_dist_path(self._h))

    def last_hamming_path(self):
        """0 = xor + popcount kernel, 1 = +-1 byte GEMM on the matrix pipe (same integers), -1 = none yet."""
        return int(lib().hg_ctx_last_hamming_path(self._h))

    def sort_ani_hits_dev(self, d_hits, n, Q):
        self._ck(lib().hg_sort_ani_hits_dev(self._h, _ptr(d_hits), n, Q))

    def sort_ani_hits_staged(self, hits, Q):
        hits = np.ascontiguousarray(hits, ANI_HIT_DTYPE).copy()
        self._ck(lib().hg_sort_ani_hits_staged(self._h, _ptr(hits), hits.size, Q))
        return hits

    def topk_per_query_dev(self, d_hits, n, Q, k, d_out, d_counts):
        self._ck(lib().hg_topk_per_query_dev(self._h, _ptr(d_hits), n, Q, k, _ptr(d_out), _ptr(d_counts)))

    def dist_dev(self, d_ref, d_rn, R, d_qry, d_qn, Q, hv_d, ksize, symmetric, ani_th, d_out, cap):
        n = C.c_size_t(0)
        st = lib().hg_dist_dev(self._h, _ptr(d_ref), _ptr(d_rn), R, _ptr(d_qry), _ptr(d_qn), Q, hv_d,
                               ksize, int(symmetric), C.c_float(ani_th), _ptr(d_out), cap, C.byref(n))
        self._ck(st, allow=(ERR_CAPACITY,))
        return n.value, st


def shard_range(n, shard, n_shards):
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    lib().hg_shard_range(n, shard, n_shards, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


class Multi:
    """Several GPUs in one process (hg_multi): one hg_ctx per entry of device_ids (ids may repeat)."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        st = lib().hg_multi_create(ids, len(device_ids), C.byref(self._h))
        if st != OK:
            raise HgError(st, lib().hg_last_error(None).decode())
        self.n = len(device_ids)

    def close(self):
        if self._h:
            lib().hg_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, st, allow=()):
        if st != OK and st not in allow:
            raise HgError(st, lib().hg_multi_last_error(self._h).decode())
        return st

    def ctx_handle(self, shard):
        return lib().hg_multi_ctx(self._h, shard)

    def peer_report(self):
        return lib().hg_multi_peer_report(self._h).decode()

    def set_gather(self, mode):
        """GATHER_PEER (direct hipMemcpyPeerAsync pulls) or GATHER_RCCL (ncclAllGather / grouped ncclBroadcast)."""
        self._ck(lib().hg_multi_set_gather(self._h, mode))

    def gather_mode(self):
        return lib().hg_multi_gather_mode(self._h)

    def gather_report(self):
        return lib().hg_multi_gather_report(self._h).decode()

    def sketch_batch(self, seqs, params=None):
        p = params or default_params()
        arrs = [np.ascontiguousarray(s, dtype=np.uint8) if isinstance(s, np.ndarray)
                else np.frombuffer(bytes(s), np.uint8) for s in seqs]
        n = len(arrs)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data if a.size else None for a in arrs])
        lens = (C.c_size_t * max(n, 1))(*[a.size for a in arrs])
        hv = np.zeros((n, p.hv_d), np.int16)
        n2 = np.zeros(n, np.int32)
        nh = np.zeros(n, np.uint32)
        self._ck(lib().hg_sketch_batch_multi(self._h, ptrs, lens, n, C.byref(p), _ptr(hv), _ptr(n2), _ptr(nh)))
        return hv, n2, nh

    def dist(self, ref_hv, ref_n2, qry_hv=None, qry_n2=None, ksize=21, symmetric=False, ani_th=85.0, cap=None):
        r = np.ascontiguousarray(ref_hv, np.int16)
        rn = np.ascontiguousarray(ref_n2, np.int32)
        q = r if qry_hv is None else np.ascontiguousarray(qry_hv, np.int16)
        qn = rn if qry_hv is None else np.ascontiguousarray(qry_n2, np.int32)
        cap = cap if cap is not None else max(1024, r.shape[0] * q.shape[0] // 8)
        while True:
            out = np.zeros(cap, ANI_HIT_DTYPE)
            n = C.c_size_t(0)
            st = lib().hg_dist_multi(self._h, _ptr(r), _ptr(rn), r.shape[0], _ptr(q), _ptr(qn), q.shape[0],
                                     r.shape[1], ksize, int(symmetric), C.c_float(ani_th), _ptr(out), cap,
                                     C.byref(n))
            if st == ERR_CAPACITY:
                cap = n.value
                continue
            self._ck(st)
            return out[: n.value].copy()

    def dist_dev(self, d_ref, d_rn, ref_rows, d_qry, d_qn, qry_rows, hv_d, ksize=21, symmetric=False, ani_th=85.0,
                 cap=1 << 20):
        """d_ref / d_rn (and d_qry / d_qn or None): per-shard device pointers (ints)."""
        arr = lambda xs: (C.c_void_p * self.n)(*[int(x) for x in xs])
        szs = lambda xs: (C.c_size_t * self.n)(*[int(x) for x in xs])
        while True:
            out = np.zeros(cap, ANI_HIT_DTYPE)
            n = C.c_size_t(0)
            st = lib().hg_dist_multi_dev(self._h, arr(d_ref), arr(d_rn), szs(ref_rows),
                                         arr(d_qry) if d_qry is not None else None,
                                         arr(d_qn) if d_qry is not None else None,
                                         szs(qry_rows) if d_qry is not None else None, hv_d, ksize, int(symmetric),
                                         C.c_float(ani_th), _ptr(out), cap, C.byref(n))
            if st == ERR_CAPACITY:
                cap = n.value
                continue
            self._ck(st)
            return out[: n.value].copy()

    def hamming_search(self, ref_bits, qry_bits, hv_d, max_dist, cap=1 << 20):
        r = np.ascontiguousarray(ref_bits, np.uint32)
        q = np.ascontiguousarray(qry_bits, np.uint32)
        while True:
            out = np.zeros(cap, HAM_HIT_DTYPE)
            n = C.c_size_t(0)
            st = lib().hg_hamming_search_multi(self._h, _ptr(r), r.shape[0], _ptr(q), q.shape[0], hv_d, max_dist,
                                               _ptr(out), cap, C.byref(n))
            if st == ERR_CAPACITY:
                cap = n.value
                continue
            self._ck(st)
            return out[: n.value].copy()


# ---- host-side formats (no device involved) ------------------------------------------------------
def hv_quant_bits(hv):
    hv = np.ascontiguousarray(hv, np.int16)
    return int(lib().hg_hv_quant_bits(_ptr(hv), hv.size))


def hv_pack(hv, q=None):
    hv = np.ascontiguousarray(hv, np.int16)
    q = hv_quant_bits(hv) if q is None else q
    out = np.zeros(lib().hg_hv_packed_bytes(hv.size, q), np.uint8)
    st = lib().hg_hv_pack(_ptr(hv), hv.size, q, _ptr(out))
    if st != OK:
        raise HgError(st, "hg_hv_pack")
    return q, out


def hv_unpack(packed, hv_d, q):
    packed = np.ascontiguousarray(packed, np.uint8)
    hv = np.zeros(hv_d, np.int16)
    st = lib().hg_hv_unpack(_ptr(packed), hv_d, q, _ptr(hv))
    if st != OK:
        raise HgError(st, "hg_hv_unpack")
    return hv


PAYLOAD_BITPACKER8X, PAYLOAD_NAIVE = 0, 1


def hv_pack_naive(hv, q=None):
    """the payload layout of hosts without AVX2 (src/hd.rs:158-166): (q, bytes)"""
    hv = np.ascontiguousarray(hv, np.int16)
    q = hv_quant_bits(hv) if q is None else q
    out = np.zeros(lib().hg_hv_packed_bytes_naive(hv.size, q), np.uint8)
    st = lib().hg_hv_pack_naive(_ptr(hv), hv.size, q, _ptr(out))
    if st != OK:
        raise HgError(st, "hg_hv_pack_naive")
    return q, out


def hv_unpack_naive(packed, hv_d, q):
    packed = np.ascontiguousarray(packed).view(np.uint8)
    hv = np.zeros(hv_d, np.int16)
    st = lib().hg_hv_unpack_naive(_ptr(packed), hv_d, q, _ptr(hv))
    if st != OK:
        raise HgError(st, "hg_hv_unpack_naive")
    return hv


def hv_payload_layout(hv_d, q, payload_bytes):
    return int(lib().hg_hv_payload_layout(hv_d, q, payload_bytes))


def read_sketch_file_image(path):
    """(image bytes as a numpy array, records) -- the records carry `payload_off` / `payload_bytes` instead of `hv`"""
    h = C.c_void_p()
    st = lib().hg_sketch_file_read_image(os.fsencode(path), C.byref(h))
    if st != OK:
        raise HgError(st, "hg_sketch_file_read_image(%s)" % path)
    try:
        nb = C.c_size_t(0)
        base = lib().hg_sketch_file_image(h, C.byref(nb))
        img = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(nb.value,)).copy() if nb.value else np.zeros(0, np.uint8)
        out = []
        for i in range(lib().hg_sketch_file_count(h)):
            r = lib().hg_sketch_file_get(h, i).contents
            assert not r.hv
            out.append(dict(ksize=r.ksize, scaled=r.scaled, canonical=bool(r.canonical), seed=r.seed,
                            hv_d=r.hv_d, hv_quant_bits=r.hv_quant_bits, hv_norm_2=r.hv_norm_2,
                            file_str=r.file_str.decode(), payload_off=int(lib().hg_sketch_file_payload_offset(h, i)),
                            payload_bytes=int(r.hv_len) * 2))
        return img, out
    finally:
        lib().hg_sketch_file_free(h)


def sort_ani_hits(hits, Q, symmetric=False):
    hits = np.ascontiguousarray(hits, ANI_HIT_DTYPE).copy()
    lib().hg_sort_ani_hits(_ptr(hits), hits.size, Q, int(symmetric))
    return hits


READ_MERGE, READ_NEEDLETAIL = 0, 1


def read_merge_seq(path, mode=READ_MERGE):
    p, n, cap = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
    st = lib().hg_read_fastx_into(os.fsencode(path), mode, C.byref(p), C.byref(cap), C.byref(n))
    if st != OK:
        lib().hg_free(p)
        raise HgError(st, "hg_read_fastx_into(%s)" % path)
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy() if n.value \
            else np.zeros(0, np.uint8)
    finally:
        lib().hg_free(p)


def pack2s(seq, norm_mode=0, cap=None):
    """hg_pack2s: the sparse link form (2-bit codes + a table of the not-a-base runs), or None if the table does not fit
    `cap` bytes (default: the size of the hg_pack2 blob -- beyond that the bitmap form is the smaller one)."""
    a = np.ascontiguousarray(seq, np.uint8)
    n = a.size
    cap = lib().hg_pack2_size(n) if cap is None else cap
    out = np.empty(max(cap, 16), np.uint8)
    size = C.c_size_t(0)
    st = lib().hg_pack2s(_ptr(a), n, norm_mode, _ptr(out), out.size, C.byref(size))
    if st == ERR_CAPACITY:
        return None
    if st != OK:
        raise HgError(st, "hg_pack2s")
    return out[:size.value]


def pack2(seq, norm_mode=0, in_place=False):
    """hg_pack2: uint8 blob of hg_pack2_size(len(seq)) bytes (2-bit codes + not-a-base bits)."""
    a = np.ascontiguousarray(seq, np.uint8)
    n = a.size
    size = lib().hg_pack2_size(n)
    if in_place:
        buf = np.zeros(max(n, size), np.uint8)
        buf[:n] = a
        st = lib().hg_pack2(_ptr(buf), n, norm_mode, _ptr(buf))
        out = buf[:size]
    else:
        out = np.empty(size, np.uint8)
        st = lib().hg_pack2(_ptr(a), n, norm_mode, _ptr(out))
    if st != OK:
        raise HgError(st, "hg_pack2")
    return out


class PinnedReader:
    """Page-locked read buffers (hg_read_fastx_pinned): read(path) returns a uint8 view that stays valid until
    the next read into the same slot or close(); what the CLI's reader threads use to feed hg_sketch_batch."""

    def __init__(self, slots=1):
        self._p = [C.c_void_p() for _ in range(slots)]
        self._cap = [C.c_size_t(0) for _ in range(slots)]

    def read(self, path, slot=0, mode=READ_MERGE):
        n = C.c_size_t(0)
        st = lib().hg_read_fastx_pinned(os.fsencode(path), mode, C.byref(self._p[slot]), C.byref(self._cap[slot]), C.byref(n))
        if st != OK:
            raise HgError(st, "hg_read_fastx_pinned(%s)" % path)
        if not n.value:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(C.cast(self._p[slot], C.POINTER(C.c_uint8)), shape=(n.value,))

    def close(self):
        for p in self._p:
            if p:
                lib().hg_pinned_free(p)
                p.value = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class SketchStream:
    """Continuous host-fed sketching (hg_sketch_stream_*): push(seq, tag) from any thread, pop() -> (tag, hv, norm2,
    nhash) in completion order (None once finish() was called and everything has been popped).  The arrays pushed
    are kept alive until their result is popped."""

    def __init__(self, device_ids=(0,), params=None):
        self.params = params or default_params()
        ids = (C.c_int * len(device_ids))(*device_ids)
        self._h = C.c_void_p()
        st = lib().hg_sketch_stream_open(ids, len(device_ids), C.byref(self.params), C.byref(self._h))
        if st != OK:
            raise HgError(st, "hg_sketch_stream_open: " + lib().hg_last_error(None).decode())
        self._keep = {}      # internal id -> (user tag, the array the device may still read)
        self._next = 0
        self._ready = []     # results popped on the pusher's behalf while the stream was full
        self._lock = threading.Lock()

    def _check(self, st, what):
        if st != OK:
            raise HgError(st, what + ": " + lib().hg_sketch_stream_last_error(self._h).decode())

    def _push(self, a, n_bps, tag, packed):
        # The C push blocks once max_pending results are outstanding; a caller that pushes everything before it pops
        # (the natural single-threaded use) would wait for ever.  So: never block here -- when the stream is full, take
        # one finished result out and hand it over on a later pop().  Arrays are kept by an internal id, not by the
        # user's tag (a reused tag must not drop the keep-alive of an array the device may still read).
        with self._lock:
            iid = self._next
            self._next += 1
            self._keep[iid] = (tag, a)
        try:
            while True:
                st = lib().hg_sketch_stream_try_push(self._h, _ptr(a), n_bps, iid, int(packed))
                if st != ERR_CAPACITY:
                    break
                r = self._pop_c()
                if r is not None:
                    with self._lock:
                        self._ready.append(r)
        except BaseException:  # a stream error surfaced by the pop: the genome was never queued, drop its keep-alive
            with self._lock:
                self._keep.pop(iid, None)
            raise
        if st != OK:
            with self._lock:
                self._keep.pop(iid, None)
        self._check(st, "hg_sketch_stream_try_push")

    def push(self, seq, tag):
        a = np.ascontiguousarray(seq, np.uint8)
        self._push(a, a.size, tag, 0)

    def push_packed(self, blob, n_bps, tag):
        a = np.ascontiguousarray(blob, np.uint8)
        assert a.size >= lib().hg_pack2_size(n_bps)
        self._push(a, n_bps, tag, 1)

    def push_packed_sparse(self, blob, n_bps, tag):
        """a hg_pack2s blob (pack2s()): codes + run table.  The buffer must hold the whole blob (checked here: the
        non-blocking C entry point takes the caller's word for the size; the library validates the table's contents)."""
        a = np.ascontiguousarray(blob, np.uint8)
        tab = (((n_bps + 3) // 4) + 15) & ~15
        if n_bps and (a.size < tab + 8 or a.size < lib().hg_pack2s_size(n_bps, int(a[tab: tab + 4].view("<u4")[0]))):
            raise HgError(ERR_INVALID, "push_packed_sparse: the buffer is shorter than the blob it claims to hold")
        self._push(a, n_bps, tag, 2)

    def finish(self):
        self._check(lib().hg_sketch_stream_finish(self._h), "hg_sketch_stream_finish")

    def _pop_c(self):
        iid, n2, nh, got = C.c_uint64(), C.c_int32(), C.c_uint32(), C.c_int()
        hv = np.empty(self.params.hv_d, np.int16)
        self._check(lib().hg_sketch_stream_pop(self._h, C.byref(iid), _ptr(hv), C.byref(n2), C.byref(nh), C.byref(got)),
                    "hg_sketch_stream_pop")
        if not got.value:
            return None
        with self._lock:
            tag, _ = self._keep.pop(iid.value)
        return tag, hv, n2.value, nh.value

    def pop(self):
        with self._lock:
            if self._ready:
                return self._ready.pop(0)
        return self._pop_c()

    def stats(self, engine=0):
        out = (C.c_double * 6)()
        self._check(lib().hg_sketch_stream_stats(self._h, engine, out), "hg_sketch_stream_stats")
        return dict(zip(("uploader_idle_s", "uploader_no_chunk_s", "uploader_copy_s", "compute_idle_s", "compute_busy_s",
                         "chunks"), out))

    def close(self):
        if self._h:
            lib().hg_sketch_stream_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def sketch_plan_describe(offsets, lens, ksize=21, scaled=1500):
    """hg_sketch_plan_describe: (dict of counts, group_first array or None) -- host arithmetic, needs no GPU"""
    off = np.ascontiguousarray(offsets, np.uint64)
    ln = np.ascontiguousarray(lens, np.uint64)
    counts = (C.c_uint64 * 6)()
    st = lib().hg_sketch_plan_describe(_ptr(off), _ptr(ln), off.size, ksize, scaled, counts, None, 0)
    if st != OK:
        raise HgError(st, "hg_sketch_plan_describe")
    d = dict(zip(("items", "workgroups", "hit_slots", "max_cap", "max_expect", "item_tiles"), (int(x) for x in counts)))
    gf = None
    if ksize <= 32 and d["items"]:
        gf = np.zeros(d["workgroups"] + 1, np.uint32)
        st = lib().hg_sketch_plan_describe(_ptr(off), _ptr(ln), off.size, ksize, scaled, counts, _ptr(gf), gf.size)
        if st != OK:
            raise HgError(st, "hg_sketch_plan_describe")
    return d, gf


def write_sketch_file(path, records):
    """records: dicts with ksize, scaled, canonical, seed, hv_d, hv_quant_bits, hv_norm_2, file_str, hv."""
    arr = (FileSketch * max(len(records), 1))()
    keep = []
    for i, r in enumerate(records):
        hv = np.ascontiguousarray(r["hv"], np.int16)
        name = r["file_str"].encode()
        keep += [hv, name]
        arr[i] = FileSketch(r["ksize"], int(bool(r["canonical"])), r["hv_quant_bits"], 0, r["hv_norm_2"],
                            r["scaled"], r["seed"], r["hv_d"], name,
                            hv.ctypes.data_as(C.POINTER(C.c_int16)), hv.size)
    st = lib().hg_sketch_file_write(os.fsencode(path), arr, len(records))
    if st != OK:
        raise HgError(st, "hg_sketch_file_write(%s)" % path)


def read_sketch_file(path):
    h = C.c_void_p()
    st = lib().hg_sketch_file_read(os.fsencode(path), C.byref(h))
    if st != OK:
        raise HgError(st, "hg_sketch_file_read(%s)" % path)
    try:
        out = []
        for i in range(lib().hg_sketch_file_count(h)):
            r = lib().hg_sketch_file_get(h, i).contents
            hv = np.ctypeslib.as_array(r.hv, shape=(r.hv_len,)).copy() if r.hv_len else np.zeros(0, np.int16)
            out.append(dict(ksize=r.ksize, scaled=r.scaled, canonical=bool(r.canonical), seed=r.seed,
                            hv_d=r.hv_d, hv_quant_bits=r.hv_quant_bits, hv_norm_2=r.hv_norm_2,
                            file_str=r.file_str.decode(), hv=hv))
        return out
    finally:
        lib().hg_sketch_file_free(h)
