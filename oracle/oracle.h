/*
 * oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT).
 *
 * The oracle is a CPU restatement of draessld/EDSParser's transform algorithms,
 * written from the reference's source as specification (each function cites the
 * reference file:line it follows).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; nothing under
 * edsparser_amd/ links, imports or calls it.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - MSA path:   pinned by the reference's own tests/cpp/test_msa.cpp vectors and
 *                 the KATs recorded in SURVEY.md §8 (the reference's msa_transforms.cpp
 *                 needs SDSL, which is absent here, so it is unbuildable in this image).
 *   - merge/VCF:  pinned by the reference's data/ goldens AND differentially against
 *                 oracle/_ref (the reference's own eds.cpp / eds_transforms.cpp /
 *                 vcf_transforms.cpp compiled in this container, see oracle/Makefile).
 */
#ifndef EDSX_ORACLE_H
#define EDSX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* All functions return 0 on success, non-zero on error (message in err, NUL-terminated).
 * Output buffers are malloc'ed; release with oracle_free. */

/* MSA -> EDS (l == 0) or l-EDS (l > 0).  msa_transforms.cpp:334-365. */
int oracle_msa(const uint8_t* file, size_t n, uint32_t l,
               char** eds, size_t* eds_n, char** seds, size_t* seds_n,
               char* err, size_t errcap);

/* Column-slab variant used by the multi-GPU stitch tests: transforms alignment columns
 * [c0, c1) of the same file as if they were a whole alignment (EDS mode only). */
int oracle_msa_slab(const uint8_t* file, size_t n, uint64_t c0, uint64_t c1,
                    char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                    char* err, size_t errcap);

/* EDS -> l-EDS.  seds == NULL => CARTESIAN (eds_transforms.cpp:381-426), else LINEAR
 * (eds_transforms.cpp:313-373).  compact != 0 => COMPACT output (eds.cpp:600-631). */
int oracle_merge(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n,
                 uint32_t l, int compact,
                 char** out, size_t* out_n, char** seds_out, size_t* seds_out_n,
                 char* err, size_t errcap);

typedef struct {
    uint64_t total_variants, processed_variants, skipped_malformed,
             skipped_unsupported_sv, variant_groups;
} oracle_vcf_stats;

/* VCF + FASTA -> EDS (l == 0) or l-EDS (l > 0).  vcf_transforms.cpp:677-755. */
int oracle_vcf(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n,
               uint32_t l, char** eds, size_t* eds_n, char** seds, size_t* seds_n,
               oracle_vcf_stats* stats, char* err, size_t errcap);

/* Position-range partition of the VCF path (multi-GPU tests, world_size-2 gloo): one range with its records
 * already in final order; index pass; the permutation of the reference's std::sort (:715-718). */
int oracle_vcf_range(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n,
                     uint64_t cur0, uint64_t next_start, char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                     oracle_vcf_stats* stats, char* err, size_t errcap);
int oracle_vcf_index(const uint8_t* vcf, size_t vcf_n, uint64_t* pos_out, uint64_t* reflen_out,
                     uint64_t* off_out, uint64_t* len_out, size_t cap, size_t* n_out, oracle_vcf_stats* stats);
void oracle_vcf_sort_order(const uint64_t* pos, size_t n, uint32_t* order_out);

/* Symbol-range partition of the merge (multi-GPU tests): see edsx_leds_merge_range in include/edsx.h. */
int oracle_merge_range(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n,
                       uint32_t l, int compact, int head_sentinel, int tail_sentinel,
                       char** out, size_t* out_n, char** seds_out, size_t* seds_out_n,
                       int* head_intact, int* tail_intact, char* err, size_t errcap);

/* EDS::calculate_statistics / calculate_source_statistics (eds.cpp:361-470, :472-505) and is_leds
 * (eds_transforms.cpp:439-468).  seds == NULL: no sources. */
typedef struct {
    uint64_t n_symbols, n_chars, n_strings, num_degenerate_symbols, total_change_size, num_common_chars,
             num_empty_strings, min_context_length, max_context_length, num_context_blocks;
    double avg_context_length;
    uint64_t has_sources, num_paths, max_paths_per_string, total_paths;
    double avg_paths_per_string;
    int is_leds;
} oracle_eds_statistics;
int oracle_eds_stats(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l,
                     oracle_eds_statistics* out, char* err, size_t errcap);

void oracle_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
