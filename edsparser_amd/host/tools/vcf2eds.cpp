// vcf2eds — VCF + reference FASTA -> EDS / l-EDS + sources, on the GPU.
// Flags, validation, naming, statistics block: src/cpp/tools/vcf2eds.cpp:36-43, :102-114,
// :159-180, :204-216.
#include "edsparser/transforms/vcf_transforms.hpp"
#include "edsx.h"
#include "../cli_util.hpp"
#include "../device.hpp"
#include "tool_common.hpp"

using namespace edsparser;

int main(int argc, char** argv)
{
    Timer timer;
    timer.start();
    try {
        cli::Parser opts("Transform VCF (Variant Call Format) to EDS/l-EDS");
        opts.add("help", 'h', false, false, "Show help message");
        opts.add("input", 'i', true, true, "Input VCF file (.vcf)");
        opts.add("reference", 'r', true, true, "Reference FASTA file");
        opts.add("output", 'o', true, false, "Output EDS file (default: <input>.eds)");
        opts.add("sources", 's', true, false, "Output source file (default: <output>.seds)");
        opts.add("context-length", 'l', true, false, "Create l-EDS with minimum context length (0 = regular EDS)");
        opts.parse(argc, argv);
        if (opts.has("help")) {
            std::cout << "vcf2eds - Transform VCF (Variant Call Format) to EDS\n\n" << opts.usage() << "\n"
                      << "DESCRIPTION:\n"
                         "  Transforms a VCF file with a reference FASTA to an Elastic-Degenerate\n"
                         "  String (EDS) with sample-level source tracking. Each sample in the VCF\n"
                         "  is tracked as a separate path in the source file.\n\n"
                         "SUPPORTED VARIANTS:\n"
                         "  SNPs, small indels, <DEL>, <INS>, multi-allelic sites\n\n";
            tool::print_performance(timer);
            return 0;
        }
        opts.notify();
        const std::filesystem::path input_file = opts.get("input");
        const std::filesystem::path reference_file = opts.get("reference");
        const std::filesystem::path output_file = opts.get("output");
        const std::filesystem::path sources_file = opts.get("sources");
        const Length context_length = static_cast<Length>(opts.get_unsigned("context-length", 0));

        if (input_file.extension() != ".vcf") {
            std::cerr << "Error: Input file must be a VCF file (.vcf)\n";
            std::cerr << "Got: " << input_file << "\n";
            tool::print_performance(timer);
            return 1;
        }
        if (!std::filesystem::exists(reference_file)) {
            std::cerr << "Error: Reference FASTA file not found: " << reference_file << "\n";
            tool::print_performance(timer);
            return 1;
        }
        // both files are mapped and handed to the C ABI as they are (no ifstream -> std::string copies)
        tool::MappedFile vcf_in(input_file, "VCF"), fasta_in(reference_file, "reference FASTA");

        const bool create_leds = context_length > 0;
        if (create_leds) {
            std::cout << "VCF → l-EDS transformation (l=" << context_length << ")\n";
            std::cout << "  Using two-stage pipeline: VCF→EDS→l-EDS\n";
        } else {
            std::cout << "VCF → EDS transformation\n";
        }
        std::cout << "  Input: " << input_file << "\n";
        std::cout << "  Reference: " << reference_file << "\n";

        VCFStats stats;
        edsx_ctx* ctx = detail::context();
        detail::Buf eds_out, seds_out;
        edsx_vcf_stats cst{};
        const int rc = edsx_vcf_transform(ctx, vcf_in.data(), vcf_in.size(), fasta_in.data(), fasta_in.size(), context_length,
                                          &eds_out.b, &seds_out.b, &cst);
        stats.total_variants = cst.total_variants; stats.processed_variants = cst.processed_variants;
        stats.skipped_malformed = cst.skipped_malformed; stats.skipped_unsupported_sv = cst.skipped_unsupported_sv;
        stats.variant_groups = cst.variant_groups;
        if (rc != EDSX_OK) detail::throw_status(rc, ctx);

        std::filesystem::path eds_path, seds_path;
        if (create_leds) {
            const std::string base = input_file.stem().string(), suffix = "_l" + std::to_string(context_length);
            eds_path = output_file.empty() ? input_file.parent_path() / (base + suffix + ".leds") : output_file;
            seds_path = sources_file.empty() ? eds_path.parent_path() / (base + suffix + ".seds") : sources_file;
        } else {
            eds_path = output_file.empty() ? input_file.parent_path() / (input_file.stem().string() + ".eds") : output_file;
            seds_path = sources_file.empty() ? eds_path.parent_path() / (eds_path.stem().string() + ".seds") : sources_file;
        }
        tool::write_bytes(eds_path, eds_out.b.data, eds_out.b.size, "output");
        tool::write_bytes(seds_path, seds_out.b.data, seds_out.b.size, "sources");

        std::cout << "Transformation complete!\n";
        std::cout << "  Output: " << eds_path << "\n";
        std::cout << "  Sources: " << seds_path << "\n";
        std::cout << "\n";
        std::cout << "Variant Processing Statistics:\n";
        std::cout << "  Total variants read:        " << stats.total_variants << "\n";
        std::cout << "  Successfully processed:     " << stats.processed_variants << "\n";
        std::cout << "  Skipped (malformed):        " << stats.skipped_malformed << "\n";
        std::cout << "  Skipped (unsupported SV):   " << stats.skipped_unsupported_sv << "\n";
        std::cout << "  Total skipped:              " << stats.total_skipped() << "\n";
        std::cout << "  Variant groups created:     " << stats.variant_groups << "\n";
        if (stats.total_variants > 0) {
            const double rate = (100.0 * stats.processed_variants) / stats.total_variants;
            std::cout << "  Success rate:               " << std::fixed << std::setprecision(1) << rate << "%\n";
        }
        std::cout << "\n";
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
