// genrandomeds — random EDS with controlled variability (+ its .seds), generated on the GPU.
// Flags, defaults, validation messages and the progress lines on stderr follow the reference tool
// (src/cpp/tools/genrandomeds.cpp:383-405 options, :423-462 validation, :231-246 and :497-516 messages); the text
// itself comes from edsx_genrandomeds (counter-based generator in HBM: same shape, not the reference's mt19937 bytes).
#include "edsx.h"
#include "../cli_util.hpp"
#include "../device.hpp"
#include "tool_common.hpp"

#include <random>

using namespace edsparser;

int main(int argc, char** argv)
{
    Timer timer;
    timer.start();
    try {
        cli::Parser opts("Generate random EDS file with controlled variability");
        opts.add("help", 'h', false, false, "Show help message");
        opts.add("output", 'o', true, true, "Output EDS file (.eds or .leds)");
        opts.add("ref-size-mb", 0, true, true, "Reference size in megabytes (1 MB = 1,000,000 bp)");
        opts.add("variability", 'v', true, false, "Fraction of positions with variants (e.g., 0.10 = 10%)");
        opts.add("min-alternatives", 0, true, false, "Minimum number of strings per degenerate symbol");
        opts.add("max-alternatives", 0, true, false, "Maximum number of strings per degenerate symbol");
        opts.add("variant-length-max", 0, true, false, "Maximum length of indel variants in bp");
        opts.add("snp-ratio", 0, true, false, "Fraction of variants that are SNPs (rest are indels)");
        opts.add("alphabet", 0, true, false, "Character alphabet for sequence generation");
        opts.add("min-context", 0, true, false, "Minimum context length between variants (for l-EDS compliance, 0 = disabled)");
        opts.add("seed", 0, true, false, "Random seed for reproducibility");
        opts.parse(argc, argv);
        if (opts.has("help")) {
            std::cout << opts.usage() << "\n\nExample usage:\n"
                      << "  genrandomeds --ref-size-mb 100 --variability 0.10 -o random.eds\n"
                      << "  genrandomeds --ref-size-mb 50 --variability 0.05 --min-context 50 -o random.leds\n"
                      << "\nNote: Sources file (.seds) is automatically generated alongside the EDS file.\n";
            tool::print_performance(timer);
            return 0;
        }
        opts.notify();
        auto real = [&](const char* name, double def) {
            if (!opts.has(name)) return def;
            size_t used = 0;
            double v = 0;
            try { v = std::stod(opts.get(name), &used); } catch (...) { used = 0; }
            if (used != opts.get(name).size())
                throw std::runtime_error("the argument ('" + opts.get(name) + "') for option '--" + name + "' is invalid");
            return v;
        };
        const std::filesystem::path output_file = opts.get("output");
        const unsigned long ref_size_mb = opts.get_unsigned("ref-size-mb", 0);
        const double variability = real("variability", 0.10), snp_ratio = real("snp-ratio", 0.7);
        const unsigned long min_alt = opts.get_unsigned("min-alternatives", 2), max_alt = opts.get_unsigned("max-alternatives", 4);
        const unsigned long var_len_max = opts.get_unsigned("variant-length-max", 10), min_context = opts.get_unsigned("min-context", 0);
        const std::string alphabet = opts.get("alphabet", "ACGT");
        const unsigned long seed = opts.has("seed") ? opts.get_unsigned("seed", 0) : std::random_device{}();
        auto fail = [&](const char* msg) { std::cerr << "Error: " << msg << "\n"; tool::print_performance(timer); return 1; };
        if (ref_size_mb == 0) return fail("Reference size must be greater than 0 MB");
        if (!(variability >= 0.0 && variability <= 1.0)) return fail("Variability must be between 0.0 and 1.0");
        if (min_alt < 2) return fail("Minimum alternatives must be at least 2");
        // (range checks before the values are narrowed to the library's 32-bit parameters)
        if (min_alt > 0xffffffffUL || max_alt > 0xffffffffUL) return fail("Maximum alternatives above 16 are not supported by this build");
        if (var_len_max > 0xffffffffUL) return fail("Variant length max above 63 is not supported by this build");
        if (max_alt < min_alt) return fail("Maximum alternatives must be >= minimum alternatives");
        if (var_len_max == 0) return fail("Variant length max must be greater than 0");
        if (!(snp_ratio >= 0.0 && snp_ratio <= 1.0)) return fail("SNP ratio must be between 0.0 and 1.0");
        if (alphabet.empty()) return fail("Alphabet cannot be empty");

        const uint64_t total_bp = (uint64_t)ref_size_mb * 1000000ull;
        std::cerr << "Generating random EDS:\n";
        std::cerr << "  Reference size: " << ref_size_mb << " MB (" << total_bp << " bp)\n";
        std::cerr << "  Variability: " << (variability * 100) << "%\n";
        std::cerr << "  Alternatives per variant: [" << min_alt << ", " << max_alt << "]\n";
        std::cerr << "  Max variant length: " << var_len_max << " bp\n";
        std::cerr << "  SNP ratio: " << (snp_ratio * 100) << "%\n";
        std::cerr << "  Number of paths (samples): " << std::max<unsigned long>(max_alt, 3) << "\n";
        if (min_context > 0) std::cerr << "  Minimum context: " << min_context << " bp (l-EDS mode)\n";

        edsx_ctx* ctx = detail::context();
        detail::Buf eds, seds;
        uint64_t n_sites = 0;
        const int rc = edsx_genrandomeds(ctx, total_bp, variability, (uint32_t)min_alt, (uint32_t)max_alt, (uint32_t)var_len_max,
                                         snp_ratio, alphabet.c_str(), min_context, seed, &eds.b, &seds.b, &n_sites);
        if (rc != EDSX_OK) detail::throw_status(rc, ctx);
        std::cerr << "  Number of variant sites: " << n_sites << "\n";
        std::cerr << "EDS generation complete\n";

        std::cerr << "Writing to file: " << output_file << "\n";
        {
            std::ofstream out(output_file, std::ios::binary);
            if (!out) return fail(("Cannot open output file: \"" + output_file.string() + "\"").c_str());
            out.write(reinterpret_cast<const char*>(eds.b.data), (std::streamsize)eds.b.size);
        }
        std::filesystem::path seds_path = output_file;
        seds_path.replace_extension(".seds");                   // genrandomeds.cpp:508-509
        std::cerr << "Writing sources to file: " << seds_path << "\n";
        {
            std::ofstream out(seds_path, std::ios::binary);
            if (!out) throw std::runtime_error("Cannot open sources file: " + seds_path.string());
            out.write(reinterpret_cast<const char*>(seds.b.data), (std::streamsize)seds.b.size);
        }
        std::cerr << "Successfully generated random EDS with sources\n";
        std::cerr << "Output written to: " << output_file << "\n";
        std::cerr << "Sources written to: " << seds_path << "\n";
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
