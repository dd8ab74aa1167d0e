// edsparser::parse_msa_to_{eds,leds}_streaming over the C ABI.
// Reference: src/cpp/lib/transforms/msa_transforms.cpp:334-345, :351-365.
#include "edsparser/transforms/msa_transforms.hpp"
#include "device.hpp"

namespace edsparser {

static std::pair<std::string, std::string> run_msa(std::istream& msa_stream, size_t context_length)
{
    if (context_length > 0xffffffffull) throw std::invalid_argument("context_length too large");
    std::string msa = detail::slurp(msa_stream);
    edsx_ctx* ctx = detail::context();
    detail::Buf eds, seds;
    int rc = edsx_msa_transform(ctx, reinterpret_cast<const uint8_t*>(msa.data()), msa.size(),
                                static_cast<uint32_t>(context_length), &eds.b, &seds.b);
    if (rc != EDSX_OK) detail::throw_status(rc, ctx);
    return {eds.str(), seds.str()};
}

std::pair<std::string, std::string> parse_msa_to_eds_streaming(std::istream& msa_stream)
{
    return run_msa(msa_stream, 0);
}

std::pair<std::string, std::string> parse_msa_to_leds_streaming(std::istream& msa_stream, size_t context_length)
{
    return run_msa(msa_stream, context_length);
}

} // namespace edsparser
