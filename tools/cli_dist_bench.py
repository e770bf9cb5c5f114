#!/usr/bin/env python3
"""End-to-end `hyper-gen dist` / `hyper-gen search` on synthetic .sketch files of N clustered sketches (GPU).

Three runs of the CLI binary, as a user would start them (src/dist.rs:11-63, src/utils.rs:260-308):
  dist -r A -q A     the reference's path_r == path_q case: symmetric, i < j
  dist -r A -q B     two different files (B = other members of A's clusters)
  search -r A -q B   top-n references per query
Each run twice (the second from a warm page cache); the stage split comes from the CLI's own RUST_LOG=debug lines.
Prints a human-readable report and, last, ONE JSON line (`--json PATH` also writes it there): bench.py's `cli` object.
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hypergen_amd as hg  # noqa: E402
import bench  # noqa: E402

STAGES = (("load_ms", r"sketch files loaded in ([\d.]+) ms"),
          ("upload_unpack_ms", r"payloads uploaded and decompressed on the device\(s\) in ([\d.]+) ms"),
          ("devices_ms", r"devices opened in ([\d.]+) ms"),
          ("ani_matrix_ms", r"ANI matrix \(\d+ hits\) in ([\d.]+) ms"),
          ("ani_ordered_on_host_ms", r"ANI matrix \(\d+ hits\), ordered, on the host in ([\d.]+) ms"),
          ("dist_device_ms", r"dist on the device ([\d.]+) ms"),
          ("order_device_ms", r"dist on the device [\d.]+ ms, order ([\d.]+) ms"),
          ("download_ms", r"order [\d.]+ ms, download ([\d.]+) ms"),
          ("order_ms", r"hits ordered in ([\d.]+) ms"),
          ("topk_ms", r"top-\d+ per query in ([\d.]+) ms"),
          ("format_ms", r"TSV formatted \([\d.]+ MB\) in ([\d.]+) ms"),
          ("write_ms", r"TSV written in ([\d.]+) ms"),
          ("format_write_ms", r"TSV formatted and written \([\d.]+ MB\) in ([\d.]+) ms"))


def write_db(path, hv, first):
    n2 = (hv.astype(np.int64) ** 2).sum(1).astype(np.int32)
    recs = []
    for i in range(hv.shape[0]):
        q, packed = hg.hv_pack(hv[i])
        recs.append(dict(ksize=21, scaled=1500, canonical=True, seed=123, hv_d=hv.shape[1], hv_quant_bits=q,
                         hv_norm_2=int(n2[i]), file_str="/data/genomes/cluster%04d/genome_%06d.fna" % (i // 100, first + i),
                         hv=packed.view(np.int16)))
    hg.write_sketch_file(path, recs)


def run_cli(exe, args, out, threads, log):
    best = None
    for rep in range(2):
        t0 = time.time()
        o = subprocess.run([exe] + args + ["-o", out, "-t", str(threads)], check=True, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, env=dict(os.environ, RUST_LOG="debug")).stdout.decode()
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, o)
    dt, o = best
    res = {"wall_s": dt, "tsv_lines": sum(1 for _ in open(out)), "tsv_mb": os.path.getsize(out) / 1e6}
    for key, pat in STAGES:
        m = re.search(pat, o)
        if m:
            res[key] = float(m.group(1))
    m = re.search(r"took ([\d.]+)s", o)
    if m:
        res["reported_s"] = float(m.group(1))
    log("    " + "\n    ".join(l.split(" - ", 1)[1] for l in o.splitlines() if " - " in l))
    return res


def sketch_leg(exe, d, a, dev, log):
    """`hyper-gen sketch -p DIR -o X` over 80-column FASTA files of synthetic 5 Mbp genomes in the page cache (src/sketch.rs:12-69:
    glob, read, sketch, compress, dump): process start, HIP bring-up, reading and parsing included.  The directory holds
    --sketch-distinct distinct genomes (the library's own generator) and links to them up to --sketch-files names -- each name
    is opened, read and parsed on its own; the links only bound the page-cache footprint (320 MB instead of 40 GB)."""
    L, nd = 5_000_000, min(a.sketch_distinct, a.sketch_files)
    fdir = os.path.join(d, "fasta")
    os.mkdir(fdir)
    t0 = time.time()
    with hg.Context(0) as ctx:
        stride = (L + 1 + 15) // 16 * 16
        seq = torch.empty(nd * stride + 64, dtype=torch.uint8, device=dev)
        ctx.synth_genomes_dev(0, nd, L, stride, seq.data_ptr())
        torch.cuda.synchronize()
        host = seq[: nd * stride].view(nd, stride).cpu().numpy()
    nl = np.full((L // 80, 1), 10, np.uint8)
    for i in range(nd):
        with open(os.path.join(fdir, "g%05d.fna" % i), "wb") as f:
            f.write(b">g%d\n" % i)
            np.concatenate([host[i, 1:L + 1].reshape(L // 80, 80), nl], axis=1).tofile(f)  # (byte 0 is the merged form's separator)
    for i in range(nd, a.sketch_files):
        os.symlink("g%05d.fna" % (i % nd), os.path.join(fdir, "g%05d.fna" % i))
    log("wrote %d FASTA files (+ %d links) in %.1f s" % (nd, a.sketch_files - nd, time.time() - t0))
    sk = os.path.join(d, "out.sketch")
    best = None
    for rep in range(2):
        t0 = time.time()
        o = subprocess.run([exe, "sketch", "-p", fdir, "-o", sk, "-t", str(a.threads)], check=True, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, env=dict(os.environ, RUST_LOG="debug")).stdout.decode()
        dt = time.time() - t0
        m = re.search(r"Sketching (\d+) files took ([\d.]+)s", o)
        r = {"wall_s": dt, "files": int(m.group(1)), "reported_s": float(m.group(2)), "files_per_s": int(m.group(1)) / dt,
             "fasta_gb_per_s": int(m.group(1)) * (L * 81 // 80) / dt / 1e9, "sketch_file_mb": os.path.getsize(sk) / 1e6}
        m = re.search(r"device\(s\) opened in ([\d.]+) ms", o)
        if m:
            r["devices_ms"] = float(m.group(1))
        m = re.search(r"(\d+) of \d+ files sent 2-bit packed", o)
        if m:
            r["files_sent_packed"] = int(m.group(1))
        if best is None or dt < best["wall_s"]:
            best = r
    assert best["files"] == a.sketch_files
    best["distinct_genomes"] = nd
    log("hyper-gen sketch -p DIR (%d files of 5 Mbp, -t %d; the faster of two runs)" % (a.sketch_files, a.threads))
    log("  => %.2f s wall, %.0f files/s = %.1f GB/s of FASTA end to end" % (best["wall_s"], best["files_per_s"], best["fasta_gb_per_s"]))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--top", type=int, default=5)
    ap.add_argument("--json", default="")
    ap.add_argument("--sketch-files", type=int, default=8192, help="`hyper-gen sketch` leg: files in the directory (0 = skip)")
    ap.add_argument("--sketch-distinct", type=int, default=64, help="... of which this many are distinct 5 Mbp genomes (the others are links)")
    a = ap.parse_args()

    def log(m):
        print(m, flush=True)
    exe = os.environ.get("HYPERGEN_CLI") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hyper-gen_amd", "hyper-gen")
    d = tempfile.mkdtemp(prefix="hgdist_", dir="/tmp")
    out = {"n": a.n, "threads": a.threads}
    try:
        dev = torch.device("cuda:0")
        t0 = time.time()
        ska, skb = os.path.join(d, "a.sketch"), os.path.join(d, "b.sketch")
        write_db(ska, bench.clustered_hvs(a.n, 0, dev).cpu().numpy(), 0)
        write_db(skb, bench.clustered_hvs(a.n, 0, dev, salt=1).cpu().numpy(), a.n)
        out["sketch_file_mb"] = os.path.getsize(ska) / 1e6
        log("wrote 2 x %d sketches (%.1f MB each) in %.1f s" % (a.n, out["sketch_file_mb"], time.time() - t0))
        tsv = os.path.join(d, "ani.tsv")
        for name, args in (("dist_symmetric", ["dist", "-r", ska, "-q", ska]),
                           ("dist_two_files", ["dist", "-r", ska, "-q", skb]),
                           ("search_top%d" % a.top, ["search", "-r", ska, "-q", skb, "-n", str(a.top)])):
            log("hyper-gen %s  (%d x %d, -t %d; the faster of two runs)" % (" ".join(os.path.basename(x) for x in args), a.n, a.n, a.threads))
            r = run_cli(exe, args, tsv, a.threads, log)
            pairs = a.n * (a.n - 1) // 2 if name == "dist_symmetric" else a.n * a.n
            r["pairs"] = pairs
            r["m_pairs_per_s_end_to_end"] = pairs / r["wall_s"] / 1e6
            out[name] = r
            log("  => %.2f s wall, %d TSV lines (%.1f MB), %.0f M pairs/s end to end" % (r["wall_s"], r["tsv_lines"], r["tsv_mb"], r["m_pairs_per_s_end_to_end"]))
        if a.sketch_files:
            out["sketch"] = sketch_leg(exe, d, a, dev, log)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    line = json.dumps(out)
    if a.json:
        with open(a.json, "w") as f:
            f.write(line + "\n")
    print(line)


if __name__ == "__main__":
    main()
