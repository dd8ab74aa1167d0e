// edsparser/transforms/msa_transforms.hpp — MSA -> EDS / l-EDS.
// Same two entry points as the reference (src/cpp/lib/transforms/msa_transforms.hpp:27,37-39);
// here they slurp the stream and run the gfx950 pipeline through edsx_msa_transform (include/edsx.h).
#ifndef EDSPARSER_TRANSFORMS_MSA_TRANSFORMS_HPP
#define EDSPARSER_TRANSFORMS_MSA_TRANSFORMS_HPP

#include "../common.hpp"
#include <iostream>
#include <string>
#include <utility>

namespace edsparser {

// FASTA alignment ('-' = gap) -> (EDS text, sEDS text).  Throws std::runtime_error on malformed
// input or when no MI355X is available (there is no CPU fallback).
std::pair<std::string, std::string> parse_msa_to_eds_streaming(std::istream& msa_stream);

// Same, merging common runs shorter than context_length into their neighbours (l-EDS).
std::pair<std::string, std::string> parse_msa_to_leds_streaming(std::istream& msa_stream,
                                                                size_t context_length);

} // namespace edsparser

#endif
