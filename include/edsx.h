/*
 * edsx.h — C ABI of the MI355X-native EDS transformation engine (libedsx.so).
 *
 * This is the drop-in boundary for EDSParser's transform hot path.  Every entry point is
 * `extern "C"`, takes plain pointers and sizes, returns an int status and never throws.
 * The host-buffer calls are exactly what the reference's Transforms layer would bind (one call
 * per public function of src/cpp/lib/transforms/): the `edsparser::` C++ shims in
 * edsparser_amd/host/ slurp the std::istream into a byte buffer and call these.
 *
 *   edsx_msa_transform   replaces parse_msa_to_eds_streaming  (msa_transforms.hpp:27,  .cpp:334-345)
 *                        and      parse_msa_to_leds_streaming (msa_transforms.hpp:37-39, .cpp:351-365)
 *   edsx_leds_merge      replaces eds_to_leds_linear          (eds_transforms.hpp:29-37, .cpp:313-373)
 *                        and      eds_to_leds_cartesian       (eds_transforms.hpp:45-51, .cpp:381-426)
 *   edsx_vcf_transform   replaces parse_vcf_to_eds_streaming  (vcf_transforms.hpp:59-62, .cpp:677-729)
 *                        and      parse_vcf_to_leds_streaming (vcf_transforms.hpp:75-79, .cpp:735-755)
 *
 * Status codes mirror the reference's (unused) ErrorCode enum, src/cpp/lib/common.hpp:37-45.
 * The *_device calls take HBM pointers on the context's GPU and a hipStream_t (passed as void*),
 * so pipelines and the benchmark can keep data resident; they are what the host-buffer calls
 * are built from.  There is NO CPU fallback: every call fails with EDSX_ERR_BUILD_FAILED when no
 * gfx950 device is usable.
 */
#ifndef EDSX_H
#define EDSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum edsx_status {
    EDSX_OK = 0,
    EDSX_ERR_FILE_NOT_FOUND = 1,
    EDSX_ERR_INVALID_FORMAT = 2,     /* std::runtime_error in the reference (format errors)   */
    EDSX_ERR_INVALID_PARAMETER = 3,  /* std::invalid_argument / std::out_of_range             */
    EDSX_ERR_BUILD_FAILED = 4,       /* device / HIP runtime failure, no GPU, out of memory   */
    EDSX_ERR_QUERY_FAILED = 5,
    EDSX_ERR_UNKNOWN = 99
};

typedef struct edsx_ctx edsx_ctx;

/* Host byte buffer allocated by the library; release with edsx_buf_free. */
typedef struct { uint8_t* data; size_t size; } edsx_buf;

/* Counters of vcf_transforms.hpp:24-35 (VCFStats). */
typedef struct {
    uint64_t total_variants, processed_variants, skipped_malformed,
             skipped_unsupported_sv, variant_groups;
} edsx_vcf_stats;

/* ---- context: one per (host thread, GPU) ---- */
int  edsx_ctx_create(int device, edsx_ctx** out);
void edsx_ctx_destroy(edsx_ctx* ctx);
/* Message of the last failing call on this context (valid until the next call). */
const char* edsx_last_error(const edsx_ctx* ctx);
void edsx_buf_free(edsx_buf* buf);
const char* edsx_version(void);

/* ---- host-buffer entry points (the drop-in boundary) ---- */

/* MSA (FASTA with '-' gaps) -> EDS (context_len == 0) or l-EDS (context_len > 0) + sEDS.
 * Output bytes are identical to the std::string pair the reference returns. */
int edsx_msa_transform(edsx_ctx* ctx, const uint8_t* msa, size_t msa_size, uint32_t context_len,
                       edsx_buf* eds, edsx_buf* seds);
/* The same in `batches` column batches, one after the other on the context's GPU: the device holds one batch (its
 * image, variant columns, records and tables) at a time, every batch's text goes to the host as it is written, and the
 * segments that cross a batch boundary are stitched as between the GPUs of edsx_msa_transform_multi.  Output is
 * byte-identical to edsx_msa_transform, which falls back to this (2, 4, 8 ... batches) when an alignment and its tables
 * do not fit the device in one piece.  An input that cannot be cut (not a plain uniform alignment, batches narrower than
 * 4 * context_len columns, or - context_len > 0 - a batch without a common run of context_len columns near both ends)
 * is transformed in one piece.  *batches_used (may be NULL) receives the number of batches taken (1: one piece). */
int edsx_msa_transform_batched(edsx_ctx* ctx, const uint8_t* msa, size_t msa_size, uint32_t context_len, int batches,
                               edsx_buf* eds, edsx_buf* seds, int* batches_used);
/* column batches the last successful edsx_msa_transform / _batched of this context took (1: one piece; 0: none yet) */
int edsx_msa_last_batches(const edsx_ctx* ctx);

/* EDS (+ optional sEDS) -> l-EDS.  seds == NULL => CARTESIAN, else LINEAR.
 * compact != 0 => COMPACT brackets (the CLI default), else FULL.  Outputs end in '\n' like
 * EDS::save / save_sources; seds_out->size == 0 when no sources were given. */
int edsx_leds_merge(edsx_ctx* ctx, const uint8_t* eds, size_t eds_size,
                    const uint8_t* seds, size_t seds_size, uint32_t context_len, int compact,
                    edsx_buf* leds, edsx_buf* seds_out);

/* 1 when the last edsx_leds_merge / _range call on this context tokenised its .eds/.seds text on the GPU (plain text:
 * no inner whitespace, no comma outside braces, well-formed), 0 when the host tokenisers took it (anything else,
 * and every input that ends in one of the reference's format errors). */
int edsx_leds_tokenised_on_device(const edsx_ctx* ctx);

/* VCF + FASTA -> EDS (context_len == 0) or l-EDS (> 0) + sEDS; stats may be NULL. */
int edsx_vcf_transform(edsx_ctx* ctx, const uint8_t* vcf, size_t vcf_size,
                       const uint8_t* fasta, size_t fasta_size, uint32_t context_len,
                       edsx_buf* eds, edsx_buf* seds, edsx_vcf_stats* stats);

/* 1 when the last edsx_vcf_transform / _range call on this context tokenised the VCF text on the GPU (plain files:
 * tab-separated lines without empty fields, POS all digits, no symbolic ALT other than <DEL>/<INS>, alleles "." or
 * digits, no '\r', no POS 0), 0 when the host tokeniser took the file (everything else). */
int edsx_vcf_tokenised_on_device(const edsx_ctx* ctx);

/* ---- multi-GPU VCF: partition by reference position ----
 * Groups of overlapping records (vcf_transforms.cpp:482-534) never span a cut placed at a group start, so
 * every GPU walks its own position range and the pieces concatenate to the reference's text with no repair.
 * What has to be global is the order of the records: the reference sorts the whole array with an unstable
 * std::sort (:715-718), so every rank indexes its byte range of the file, the (POS, REF length) arrays are
 * all-gathered, every rank derives the same order and cuts, and record lines that fall into another rank's
 * position range are exchanged (edsparser_amd/multigpu.py, VcfSharder). */

/* Index pass over VCF text: POS and REF length (uint64 each) and the line span (offset, length; uint64 each)
 * of every record the transform would accept, in file order; counters as edsx_vcf_transform (variant_groups
 * stays 0).  Genotypes are not parsed. */
int edsx_vcf_index(edsx_ctx* ctx, const uint8_t* vcf, size_t vcf_size, edsx_buf* pos, edsx_buf* reflen,
                   edsx_buf* line_off, edsx_buf* line_len, edsx_vcf_stats* stats);
/* order_out[k] = index (into pos) of the k-th record after the reference's std::sort of n records. */
int edsx_vcf_sort_order(const uint64_t* pos, size_t n, uint32_t* order_out);
/* edsx_vcf_transform (context_len 0) of one position range: `vcf` holds the range's record lines already in
 * their final order; the walk starts at reference position cur0 (pass the start of the range's first group
 * for every range but the first, 0 for the first) and the closing common text stops at next_start (0-based
 * start of the next range's first group; UINT64_MAX for the last range: flush to the end of the reference). */
int edsx_vcf_transform_range(edsx_ctx* ctx, const uint8_t* vcf, size_t vcf_size, const uint8_t* fasta,
                             size_t fasta_size, uint64_t cur0, uint64_t next_start, edsx_buf* eds,
                             edsx_buf* seds, edsx_vcf_stats* stats);

/* ---- multi-GPU merge: partition by symbol range ----
 * edsx_leds_merge of one symbol range of a larger EDS.  Neighbouring ranges overlap in one sentinel symbol (a single
 * string of at least context_len characters between two degenerate symbols; edsparser_amd/multigpu.py, MergeSharder,
 * finds them): head_sentinel / tail_sentinel say that the first / last symbol of this text is such a shared symbol.
 * The range that has it as its tail prints it, the one that has it as its head drops it; only a range without a tail
 * sentinel ends its outputs in '\n'.  *head_intact / *tail_intact come back 0 when the sentinel was drawn into a merge
 * (possible only when a LINEAR product collapses its neighbour to a single short string): the partition is then
 * invalid for this input and the caller must run edsx_leds_merge on the whole text. */
int edsx_leds_merge_range(edsx_ctx* ctx, const uint8_t* eds, size_t eds_size, const uint8_t* seds, size_t seds_size,
                          uint32_t context_len, int compact, int head_sentinel, int tail_sentinel,
                          edsx_buf* leds, edsx_buf* seds_out, int* head_intact, int* tail_intact);

/* ---- statistics and validation of an EDS / l-EDS (what edsparser-stats prints) ----
 * Replaces EDS::calculate_statistics / calculate_source_statistics (src/cpp/lib/formats/eds.cpp:361-470, :472-505,
 * struct EDS::Statistics eds.hpp:107-120) and is_leds (src/cpp/lib/transforms/eds_transforms.cpp:439-468): the text is
 * tokenised as for edsx_leds_merge and the numbers are device reductions over the per-symbol / per-string arrays.
 * seds == NULL: no sources (the three path fields stay 0).  Parse errors: as edsx_leds_merge. */
typedef struct {
    uint64_t n_symbols, n_chars, n_strings;            /* EDS::length(), size(), cardinality() */
    uint64_t num_degenerate_symbols, total_change_size, num_common_chars, num_empty_strings;
    uint64_t min_context_length, max_context_length, num_context_blocks;
    double   avg_context_length;                       /* num_common_chars / num_context_blocks (0 when there are none) */
    uint64_t has_sources, num_paths, max_paths_per_string, total_paths;
    double   avg_paths_per_string;                     /* total_paths / n_strings */
    int      is_leds;                                  /* is_leds(eds, context_len) */
} edsx_eds_statistics;
int edsx_eds_stats(edsx_ctx* ctx, const uint8_t* eds, size_t eds_size, const uint8_t* seds, size_t seds_size,
                   uint32_t context_len, edsx_eds_statistics* out);

/* ---- synthetic inputs: genrandomeds-shaped .eds + .seds generated in HBM ----
 * Flags and shape of src/cpp/tools/genrandomeds.cpp:383-405 / :221-352 (reference uniform over `alphabet`, single-position
 * variant sites with min_alt..max_alt alternatives, snp_ratio SNPs, else insertions of 1..var_len_max characters /
 * deletions, max(max_alt, 3) paths, {0} for common blocks, FULL brackets, no trailing newline).  Counter-based, not the
 * reference's mt19937 stream: the same (seed, flags) always give the same text, but not the reference's bytes; with
 * min_context == 0 the number of sites is binomial around total_bp * variability instead of exactly its floor. */
int edsx_genrandomeds(edsx_ctx* ctx, uint64_t total_bp, double variability, uint32_t min_alt, uint32_t max_alt,
                      uint32_t var_len_max, double snp_ratio, const char* alphabet, uint64_t min_context, uint64_t seed,
                      edsx_buf* eds, edsx_buf* seds, uint64_t* n_sites);

/* ---- device-resident MSA path (inputs/outputs stay in HBM) ---- */

/* Phase 1: index rows, scan columns, build the segment table and size the outputs.
 * d_msa must stay valid until the matching edsx_msa_emit_device returns.  Synchronises the
 * stream once to hand the sizes back. */
int edsx_msa_plan_device(edsx_ctx* ctx, const uint8_t* d_msa, size_t msa_size, uint32_t context_len,
                         void* stream, uint64_t* eds_bytes, uint64_t* seds_bytes);
/* Phase 2: write the .eds / .seds text into caller-provided HBM buffers of at least the planned
 * sizes.  Asynchronous on `stream`. */
int edsx_msa_emit_device(edsx_ctx* ctx, uint8_t* d_eds, uint8_t* d_seds, void* stream);

/* Geometry of the last planned alignment (for reporting). */
typedef struct {
    uint64_t n_rows, n_cols, line_width, n_variant_cols, n_segments, msa_bytes,
             n_slow_segments;   /* variant segments handled by the generic (slow) kernels */
} edsx_msa_info;
/* (EDSX_ERR_INVALID_PARAMETER when there is no plan, or the last transform ran in column batches: the numbers would be
 * those of its last batch) */
int edsx_msa_last_info(const edsx_ctx* ctx, edsx_msa_info* info);

/* ---- multi-GPU column slabs: what the boundary stitch needs from a planned+emitted slab ----
 * A run that crosses a slab boundary is stitched by the caller (edsparser_amd/multigpu.py): common
 * runs are joined textually, variant runs are recomputed from the raw columns of both sides. */
typedef struct {
    uint64_t n_segments;
    uint64_t first_is_variant, first_cols, first_eds_bytes, first_seds_bytes;
    uint64_t last_is_variant, last_cols, last_eds_bytes, last_seds_bytes;
} edsx_msa_edges;
int edsx_msa_edge_info(edsx_ctx* ctx, edsx_msa_edges* out);
/* l-EDS slabs (context length l > 0): the first and the last common segment of at least min_cols columns of the planned
 * alignment - the anchors between which a slab's text does not depend on its neighbours (msa_transforms.cpp:153).
 * found = 0: there is none.  last_*: the last anchor's first column and the text offsets in front of it; first_*: the
 * column behind the first anchor and the text offsets behind it. */
typedef struct {
    uint64_t n_segments, found, first_seg, last_seg;
    uint64_t last_col, last_eds_bytes, last_seds_bytes;
    uint64_t first_end, first_eds_end, first_seds_end;
} edsx_msa_anchors;
int edsx_msa_anchor_info(edsx_ctx* ctx, uint64_t min_cols, edsx_msa_anchors* out);
/* Alignment columns [col0, col0+ncols) of every row of the planned alignment, row-major
 * (n_rows * ncols bytes) into a host buffer. */
int edsx_msa_copy_columns(edsx_ctx* ctx, uint64_t col0, uint64_t ncols, uint8_t* host_out);
/* First segment of the planned alignment that starts at or after alignment column `col`: its index, start
 * column and the byte offsets of its text in the .eds / .seds outputs (index n_segments, column n_cols and
 * the output sizes when no segment starts there).  Lets a caller cut the device-resident outputs at segment
 * boundaries, e.g. to check sampled column windows of a 100 GB alignment against the CPU path. */
int edsx_msa_locate_segment(edsx_ctx* ctx, uint64_t col, uint64_t* seg, uint64_t* seg_col, uint64_t* eds_off,
                            uint64_t* seds_off);

/* ---- MSA -> EDS over several GPUs of one node, from C/C++ (replaces the single call of msa2eds.cpp:123-132 when the
 * alignment should be spread over N GPUs) ----
 * One host thread per GPU inside the call.  The alignment columns are range-partitioned into n column slabs; every
 * GPU transforms its slab, and the segments that cross a slab boundary are stitched with KB-sized all-gathers:
 * ncclAllGather over RCCL (use_rccl = 1: one distinct device per rank), or an in-process exchange between the rank
 * threads (use_rccl = 0: ranks may share a device - rehearsals and tests on a one-GPU box).  Output is byte-identical
 * to edsx_msa_transform.  With context_len > 0 every boundary is recomputed between the nearest common runs of at
 * least context_len columns on either side; files that are not plain uniform alignments, and l-EDS slabs without such
 * runs near their ends, are transformed by rank 0 alone. */
typedef struct edsx_multi edsx_multi;
int  edsx_multi_create(const int* device_ids, int n, int use_rccl, edsx_multi** out);
void edsx_multi_destroy(edsx_multi* m);
const char* edsx_multi_last_error(const edsx_multi* m);
int  edsx_msa_transform_multi(edsx_multi* m, const uint8_t* msa, size_t msa_size, uint32_t context_len,
                              edsx_buf* eds, edsx_buf* seds);
/* of the last edsx_msa_transform_multi: 1 if the columns were partitioned, the number of boundary chains stitched */
int  edsx_multi_last_partition(const edsx_multi* m, int* partitioned, int* chains);

/* Per-kernel device time, measured with HIP events on the stream each kernel is launched on and
 * accumulated over all plan/emit calls since edsx_set_timing(ctx, 1).  Arrays of capacity cap;
 * total_ms[i] / launches[i] is the average duration of kernel names[i].  Returns the entry count. */
void edsx_set_timing(edsx_ctx* ctx, int enabled);
int  edsx_get_timing(edsx_ctx* ctx, const char** names, float* total_ms, int* launches, int cap);

/* ---- synthetic VCF + FASTA of BASELINE configs[3]'s shape, generated in HBM ----
 * One FASTA record of ref_len uniform ACGT bases in 60-column lines; n_records record lines with strictly ascending
 * POS, 70 % SNP / 15 % insertion of 1..10 bases / 15 % deletion of 1..10 bases (REF spans them: a few per cent of the
 * records overlap the next one), n_samples diploid phased samples, each allele ALT with p = 0.3.  Counter-based: the
 * text depends on the parameters only.  Needs ref_len >= 64 and n_records <= ref_len / 16. */
int edsx_genvcf(edsx_ctx* ctx, uint64_t ref_len, uint64_t n_records, uint32_t n_samples, uint64_t seed,
                edsx_buf* vcf, edsx_buf* fasta);

/* ---- synthetic genrandomeds-shaped alignment, generated in HBM (bench / tests) ----
 * Rows 0..n_rows-1 of alignment columns [col0, col0+n_cols) of a virtual alignment, one line per
 * row, headers ">s<row>", trailing newline.  Bytes depend only on (seed, global column, row), so a
 * column slab generated on another GPU is bit-identical to the same columns of the whole. */
size_t edsx_msa_synth_size(uint32_t n_rows, uint64_t n_cols);
int edsx_msa_synth_device(edsx_ctx* ctx, uint8_t* d_out, size_t capacity, uint32_t n_rows,
                          uint64_t col0, uint64_t n_cols, double variant_fraction, uint64_t seed,
                          void* stream, size_t* written);
/* The same alignment with every header padded with blanks (">s<row>    ...") so that each row's first column lies a
 * multiple of row_align bytes (a power of two, e.g. 128) from the start of the image: the layout an upload that places
 * the rows for the column scan produces.  row_align <= 1: the plain image above.  Same cells, same outputs. */
size_t edsx_msa_synth_size_aligned(uint32_t n_rows, uint64_t n_cols, uint32_t row_align);
int edsx_msa_synth_device_aligned(edsx_ctx* ctx, uint8_t* d_out, size_t capacity, uint32_t n_rows, uint64_t col0, uint64_t n_cols,
                                  double variant_fraction, uint64_t seed, uint32_t row_align, void* stream, size_t* written);

#ifdef __cplusplus
}
#endif
#endif /* EDSX_H */
