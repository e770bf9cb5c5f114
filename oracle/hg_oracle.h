/*
 * hg_oracle.h -- CPU restatement of the HyperGen sketch + ANI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under hyper-gen_amd/ (the product) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and the cpu_baseline leg of bench.py.
 *
 * Every function cites the reference file:line (relative to the wh-xu/Hyper-Gen
 * checkout) whose behaviour it restates.  Pinning status is in oracle/README.md.
 */
#ifndef HG_ORACLE_H
#define HG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashing ------------------------------------------------------------ */

/* t1ha2_atonce (crate t1ha 0.1.0, called at src/sketch.rs:90; the <=32 byte
 * arithmetic is restated by the reference itself at src/cuda_kernel.cu:71-246).
 * Any length (the >32 byte loop follows the published t1ha2 algorithm). */
uint64_t orc_t1ha2_atonce(const uint8_t *data, size_t len, uint64_t seed);

/* WyRng::next_u64 (crate wyhash 0.5.0, used at src/hd.rs:44-51,100-103).
 * seed_from_u64(h) sets state = h. */
uint64_t orc_wyrng_next(uint64_t *state);

/* ---- FracMinHash k-mer sampling ------------------------------------------ */

#define ORC_NORM_ACGT 0 /* src/cuda_kernel.cu:277-296: only ACGTacgt are bases   */
#define ORC_NORM_U2T 1  /* needletail normalize(false): also u/U -> T            */

/* Walk every valid k-window of `seq` (a read_merge_seq buffer,
 * src/fastx_reader.rs:6-29, or one FASTA record), pick the canonical strand
 * (src/sketch.rs:89 / src/cuda_kernel.cu:306-311), hash its ASCII bytes and
 * keep h < threshold (strict, src/sketch.rs:92).  Hits are appended in walk
 * order WITH duplicates.  Returns the number of hits found (may exceed cap;
 * only the first cap are stored). */
size_t orc_kmer_hash_sample(const uint8_t *seq, size_t n_bps, unsigned ksize,
                            uint64_t threshold, uint64_t seed, int canonical,
                            int norm_mode, uint64_t *out, size_t cap);

/* sort ascending + unique in place (HashSet semantics, src/sketch.rs:93).
 * Returns the distinct count. */
size_t orc_sort_unique_u64(uint64_t *v, size_t n);

/* read_merge_seq (src/fastx_reader.rs:6-29) applied to an in-memory FASTA text:
 * sequence lines concatenated (trailing \n / \r stripped), one 'N' per header
 * line.  Returns bytes written (out must hold n_text bytes). */
size_t orc_read_merge_seq(const uint8_t *text, size_t n_text, uint8_t *out);

/* ---- hypervector encode --------------------------------------------------- */

#define ORC_LAYOUT_SCALAR 0 /* src/hd.rs:94-112  */
#define ORC_LAYOUT_AVX2 1   /* src/hd.rs:14-92 (what x86-64 hosts write to disk) */

/* hv[d] = -n + 2 * sum_h bit_d(h), i16 wrapping; hashes must be distinct. */
void orc_encode_hv(const uint64_t *hashes, size_t n, size_t hv_d, int layout,
                   int16_t *hv);

/* Step-by-step scalar emulation of the AVX2 intrinsics sequence of
 * src/hd.rs:14-92 (used only to pin the closed-form ORC_LAYOUT_AVX2 permutation). */
void orc_encode_hv_avx2_emulated(const uint64_t *hashes, size_t n, size_t hv_d,
                                 int16_t *hv);

/* compute_hv_l2_norm (src/dist.rs:132-137), i32 wrapping. */
int32_t orc_hv_norm2(const int16_t *hv, size_t hv_d);

/* ---- sketch compression ---------------------------------------------------- */

/* smallest lossless width 6..16 (src/hd.rs:119-136). */
unsigned orc_quant_bits(const int16_t *hv, size_t hv_d);

/* compress_hd_sketch AVX2 branch (src/hd.rs:138-157): BitPacker8x blocks of 256.
 * out must hold quant_bits*hv_d/8 bytes.  hv_d must be a multiple of 256. */
size_t orc_packed_bytes(size_t hv_d, unsigned quant_bits); /* q * (hv_d >> 3), src/hd.rs:146 */
void orc_pack_hv(const int16_t *hv, size_t hv_d, unsigned quant_bits,
                 uint8_t *out);
/* decompress_hd_sketch AVX2 branch (src/hd.rs:188-212). */
void orc_unpack_hv(const uint8_t *packed, size_t hv_d, unsigned quant_bits,
                   int16_t *hv);

/* the layout of hosts WITHOUT AVX2 (src/hd.rs:158-166, 213-231): (q*hv_d + 16) / 16 i16 words, the low q bits of each
 * value LSB first, no offset; the decode's strict `>` and its q = 16 shifts are the reference's (see hg_oracle.c). */
size_t orc_packed_words_naive(size_t hv_d, unsigned quant_bits);
void orc_pack_hv_naive(const int16_t *hv, size_t hv_d, unsigned quant_bits, int16_t *out);
void orc_unpack_hv_naive(const int16_t *packed, size_t hv_d, unsigned quant_bits, int16_t *hv);

/* ---- ANI ------------------------------------------------------------------- */

/* i32 dot (src/dist.rs:147-151), wrapping. */
int32_t orc_hv_dot(const int16_t *r, const int16_t *q, size_t hv_d);

/* compute_pairwise_ani tail (src/dist.rs:153-160) given the integer dot. */
float orc_ani_from_dot(int32_t dot, int32_t norm2_r, int32_t norm2_q,
                       unsigned ksize);

void orc_ani_from_dots(const int32_t *dot, const int32_t *nr, const int32_t *nq, size_t n, unsigned ksize, float *out);

/* glibc's logf algorithm (sysdeps/ieee754/flt-32/e_logf.c) restated: fused = 1 as the -mfma build evaluates it
 * (__logf_fma), 0 with every product rounded on its own (__logf_sse2).  orc_logf_sweep: number of bit patterns in
 * [first_bits, first_bits + n) on which the HOST's logf differs from the restatement (*first_bad: the first one).
 * orc_logf_array: form -1 = the host's logf, 0 / 1 = the restatements. */
float orc_logf_glibc(float x, int fused);
uint64_t orc_logf_sweep(uint32_t first_bits, uint64_t n, int fused, uint32_t *first_bad);
uint64_t orc_logf_forms_differ(uint32_t first_bits, uint64_t n, uint32_t *out, size_t cap);
void orc_logf_array(const float *x, size_t n, float *out, int form);

/* OpenMP team size for orc_ani_matrix */
void orc_set_threads(int n);

/* full R x Q ANI matrix, row-major [r][q]; OpenMP over rows when available. */
void orc_ani_matrix(const int16_t *ref_hv, const int32_t *ref_norm2, size_t R,
                    const int16_t *qry_hv, const int32_t *qry_norm2, size_t Q,
                    size_t hv_d, unsigned ksize, float *ani_out);

/* ---- bit-packed hypervectors: extension of BASELINE configs[4], defined by this repository
 * (the reference has no such path).  bit d = (hv[d] >= 0); uint32 word w holds dims 32w..32w+31,
 * LSB first; distance = popcount(xor). */
void orc_binarize(const int16_t *hv, size_t n, size_t hv_d, uint32_t *bits);
void orc_hamming_matrix(const uint32_t *ref_bits, size_t R, const uint32_t *qry_bits, size_t Q,
                        size_t words, uint32_t *dist_out);

/* ---- whole-genome sketch (what src/sketch.rs:35-56 does per file) ---------- */

/* seq = merged buffer.  Writes hv (layout as given), norm2, n distinct hashes.
 * Returns 0 on success. */
int orc_sketch_genome(const uint8_t *seq, size_t n_bps, unsigned ksize,
                      uint64_t scaled, uint64_t seed, int canonical,
                      int norm_mode, size_t hv_d, int layout, int16_t *hv,
                      int32_t *norm2, uint32_t *n_hash);

/* the same for n genomes, task-parallel over genomes with OpenMP (the rayon loop of
 * src/sketch.rs:35); hv is n x hv_d.  This is the timed CPU baseline of bench.py. */
int orc_sketch_batch_mt(const uint8_t *const *seqs, const size_t *lens, size_t n,
                        unsigned ksize, uint64_t scaled, uint64_t seed, int canonical,
                        int norm_mode, size_t hv_d, int layout, int n_threads,
                        int16_t *hv, int32_t *norm2, uint32_t *n_hash);

/* ---- synthetic inputs (repo-defined) ----------------------------- */

/* Genome g of a clustered set: cluster c = g / cluster_size is an iid-uniform
 * ACGT root; member m = g % cluster_size carries iid substitutions at rate
 * m * sub_ppm_per_member / 1e6.  Counter-based, so device and host generate the same
 * bytes independently.  Writes 'N' + L bases (the read_merge_seq layout of a
 * single-record FASTA): out must hold L + 1 bytes. */
void orc_synth_genome(uint64_t g, size_t L, unsigned cluster_size,
                      uint32_t sub_ppm_per_member, uint8_t *out);

void orc_synth_genomes_mt(uint64_t first, size_t n, size_t L, unsigned cluster_size,
                          uint32_t sub_ppm_per_member, int n_threads, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
