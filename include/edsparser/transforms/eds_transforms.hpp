// edsparser/transforms/eds_transforms.hpp — EDS -> l-EDS (LINEAR with sources / CARTESIAN).
// API of the reference's src/cpp/lib/transforms/eds_transforms.hpp (:29-37, :45-51, :57),
// implemented over edsx_leds_merge (include/edsx.h).
#ifndef EDSPARSER_TRANSFORMS_EDS_TRANSFORMS_HPP
#define EDSPARSER_TRANSFORMS_EDS_TRANSFORMS_HPP

#include "../common.hpp"
#include "../formats/eds.hpp"
#include <iostream>

namespace edsparser {

// num_threads is accepted for compatibility; the merge runs on the GPU and its result does not
// depend on it (as in the reference).  context_length == 0 throws std::invalid_argument.
void eds_to_leds_linear(std::istream& input, std::ostream& output, Length context_length,
                        std::istream* phasing_input = nullptr, std::ostream* phasing_output = nullptr,
                        size_t num_threads = 1, bool compact = true);

void eds_to_leds_cartesian(std::istream& input, std::ostream& output, Length context_length,
                           size_t num_threads = 1, bool compact = true);

// true iff no internal common block is shorter than context_length and no two degenerate
// symbols are adjacent (host-side check on the container's metadata).
bool is_leds(const EDS& eds, Length context_length);

} // namespace edsparser

#endif
