// edsparser::eds_to_leds_{linear,cartesian} + is_leds over the C ABI.
// Reference: src/cpp/lib/transforms/eds_transforms.cpp:313-373, :381-426, :439-468.
#include "edsparser/transforms/eds_transforms.hpp"
#include "device.hpp"

namespace edsparser {

static void run_merge(std::istream& input, std::ostream& output, Length context_length, std::istream* phasing_input,
                      std::ostream* phasing_output, bool compact)
{
    if (context_length == 0)
        throw std::invalid_argument("context_length must be > 0 for l-EDS transformation");
    std::string eds = detail::slurp(input);
    std::string seds;
    if (phasing_input) seds = detail::slurp(*phasing_input);
    edsx_ctx* ctx = detail::context();
    detail::Buf out, sout;
    int rc = edsx_leds_merge(ctx, reinterpret_cast<const uint8_t*>(eds.data()), eds.size(),
                             phasing_input ? reinterpret_cast<const uint8_t*>(seds.data()) : nullptr, seds.size(),
                             context_length, compact ? 1 : 0, &out.b, &sout.b);
    if (rc != EDSX_OK) detail::throw_status(rc, ctx);
    output.write(reinterpret_cast<const char*>(out.b.data), static_cast<std::streamsize>(out.b.size));
    if (phasing_output && phasing_input)
        phasing_output->write(reinterpret_cast<const char*>(sout.b.data), static_cast<std::streamsize>(sout.b.size));
}

void eds_to_leds_linear(std::istream& input, std::ostream& output, Length context_length, std::istream* phasing_input,
                        std::ostream* phasing_output, size_t /*num_threads*/, bool compact)
{
    run_merge(input, output, context_length, phasing_input, phasing_output, compact);
}

void eds_to_leds_cartesian(std::istream& input, std::ostream& output, Length context_length, size_t /*num_threads*/,
                           bool compact)
{
    run_merge(input, output, context_length, nullptr, nullptr, compact);
}

bool is_leds(const EDS& eds, Length context_length)
{
    if (context_length == 0) return true;
    const auto& deg = eds.get_is_degenerate();
    const size_t n = eds.length();
    for (size_t i = 0; i < n; ++i) {
        if (!deg[i]) {
            Length len = eds.get_string_length(eds.get_metadata().cum_set_sizes[i]);
            if (i > 0 && i + 1 < n && len < context_length) return false;
        }
        if (i + 1 < n && deg[i] && deg[i + 1]) return false;
    }
    return true;
}

} // namespace edsparser
