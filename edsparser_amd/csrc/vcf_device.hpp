// vcf_device.hpp — host driver of the VCF -> EDS overlay pipeline (see vcf_device.hip).
#pragma once

#include "msa_device.hpp"

#include <cstdlib>
#include <new>
#include <string>
#include <vector>

namespace edsx {

struct VcfCounters {   // vcf_transforms.hpp:24-35
    u64 total_variants = 0, processed_variants = 0, skipped_malformed = 0, skipped_unsupported_sv = 0, variant_groups = 0;
};

// One position range of a partitioned run: the records are handed over in their final order, the
// walk starts at cur0 (no common text in front of the first group when cur0 is its start) and the closing common
// text stops at the first group of the next range.
struct VcfRange {
    bool presorted = false;
    u64 cur0 = 0;
    u64 next_start = ~0ull;        // ~0: last (or only) range, flush to the end of the reference
};

// index pass of a partitioned run: positions, REF lengths and line spans of the accepted records, file order
void vcf_index(const uint8_t* vcf, size_t vcf_n, std::vector<u64>& pos, std::vector<u64>& reflen, std::vector<u64>& line_off,
               std::vector<u64>& line_len, VcfCounters& stats);
// permutation the reference's std::sort (vcf_transforms.cpp:715-718) gives an array with these positions
void vcf_sort_order(const u64* pos, size_t n, u32* order_out);

class VcfPipeline {
public:
    // host buffers in; eds/seds text out (FULL brackets, no trailing newline, like the reference)
    void run(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n, HostBytes& eds, HostBytes& seds,
             VcfCounters& stats, hipStream_t st, const VcfRange& range = VcfRange());

    bool tokenised_on_device() const { return tokenised_on_device_; }
    bool index_device(const uint8_t* vcf, size_t n, hipStream_t st, std::vector<u64>& pos, std::vector<u64>& reflen,
                      std::vector<u64>& line_off, std::vector<u64>& line_len, VcfCounters& stats);

private:
    bool tokenised_on_device_ = false;
    bool tokenize_device(const uint8_t* vcf, size_t n, bool presorted, hipStream_t st, u64& nrec, u64& max_samples,
                         VcfCounters& stats);
    DevBuf vt_raw_, vt_idx_, vt_lstart_, vt_pos_, vt_reflen_, vt_nalt_, vt_altc_, vt_ngt_, vt_nall_, vt_s1_, vt_s2_, vt_s3_,
           vt_s4_, vt_order_, vt_sorttmp_;
    DevBuf d_fasta_, refc_, blkpre_, scan_tmp_, ctl_, start_, reflen_, alt0_, altoff_, altchars_, pair0_, pa0_, alleles_,
           ends_, flag_, gidx_, grp_r0_, g_gs_, g_spanlen_, g_cs_, g_nraw_, g_rawchars_, g_ndist_, g_bitwords_, g_eds_,
           g_seds_, g_common_, g_cur_, raw0_, rawc0_, bit0_, rawlen_, rawoff_, canon_, hapchars_, carried_, d_eds_, d_seds_,
           d_one_;
};

} // namespace edsx
