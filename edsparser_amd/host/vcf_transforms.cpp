// edsparser::parse_vcf_to_{eds,leds}_streaming over the C ABI.
// Reference: src/cpp/lib/transforms/vcf_transforms.cpp:677-729, :735-755.
#include "edsparser/transforms/vcf_transforms.hpp"
#include "device.hpp"

namespace edsparser {

static std::pair<std::string, std::string> run_vcf(std::istream& vcf_stream, std::istream& fasta_stream,
                                                   size_t context_length, VCFStats* stats)
{
    if (context_length > 0xffffffffull) throw std::invalid_argument("context_length too large");
    std::string vcf = detail::slurp(vcf_stream), fasta = detail::slurp(fasta_stream);
    edsx_ctx* ctx = detail::context();
    detail::Buf eds, seds;
    edsx_vcf_stats st{};
    int rc = edsx_vcf_transform(ctx, reinterpret_cast<const uint8_t*>(vcf.data()), vcf.size(),
                                reinterpret_cast<const uint8_t*>(fasta.data()), fasta.size(),
                                static_cast<uint32_t>(context_length), &eds.b, &seds.b, &st);
    if (stats) {   // the reference updates the counters while parsing, i.e. also when it throws later
        stats->total_variants += st.total_variants;
        stats->processed_variants += st.processed_variants;
        stats->skipped_malformed += st.skipped_malformed;
        stats->skipped_unsupported_sv += st.skipped_unsupported_sv;
        if (rc == EDSX_OK) stats->variant_groups = st.variant_groups;
    }
    if (rc != EDSX_OK) detail::throw_status(rc, ctx);
    return {eds.str(), seds.str()};
}

std::pair<std::string, std::string> parse_vcf_to_eds_streaming(std::istream& vcf_stream, std::istream& fasta_stream,
                                                               VCFStats* stats)
{
    return run_vcf(vcf_stream, fasta_stream, 0, stats);
}

std::pair<std::string, std::string> parse_vcf_to_leds_streaming(std::istream& vcf_stream, std::istream& fasta_stream,
                                                                size_t context_length, VCFStats* stats)
{
    if (context_length == 0) throw std::invalid_argument("context_length must be > 0 for l-EDS transformation");
    return run_vcf(vcf_stream, fasta_stream, context_length, stats);
}

} // namespace edsparser
