// edsparser/transforms/vcf_transforms.hpp — VCF (+ reference FASTA) -> EDS / l-EDS.
// API of the reference's src/cpp/lib/transforms/vcf_transforms.hpp (:24-35 VCFStats, :59-62, :75-79),
// implemented over edsx_vcf_transform (include/edsx.h).
#ifndef EDSPARSER_TRANSFORMS_VCF_TRANSFORMS_HPP
#define EDSPARSER_TRANSFORMS_VCF_TRANSFORMS_HPP

#include "../common.hpp"
#include <iostream>
#include <string>
#include <utility>

namespace edsparser {

struct VCFStats {
    size_t total_variants = 0;          // data lines seen (headers excluded)
    size_t processed_variants = 0;
    size_t skipped_malformed = 0;
    size_t skipped_unsupported_sv = 0;
    size_t variant_groups = 0;          // after grouping overlapping records
    size_t total_skipped() const { return skipped_malformed + skipped_unsupported_sv; }
};

std::pair<std::string, std::string> parse_vcf_to_eds_streaming(std::istream& vcf_stream,
                                                               std::istream& fasta_stream,
                                                               VCFStats* stats = nullptr);

std::pair<std::string, std::string> parse_vcf_to_leds_streaming(std::istream& vcf_stream,
                                                                std::istream& fasta_stream,
                                                                size_t context_length,
                                                                VCFStats* stats = nullptr);

} // namespace edsparser

#endif
