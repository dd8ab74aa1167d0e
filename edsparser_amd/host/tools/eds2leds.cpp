// eds2leds — EDS -> l-EDS; LINEAR (phasing-aware) when --sources is given, else CARTESIAN.
// Flags, validation, naming and messages: src/cpp/tools/eds2leds.cpp:38-46, :98-124, :161-162, :176-196.
#include "edsparser/transforms/eds_transforms.hpp"
#include "../cli_util.hpp"
#include "tool_common.hpp"

#include <memory>

using namespace edsparser;

int main(int argc, char** argv)
{
    Timer timer;
    timer.start();
    try {
        cli::Parser opts("Transform EDS to l-EDS (length-constrained EDS)");
        opts.add("help", 'h', false, false, "Show help message");
        opts.add("input", 'i', true, true, "Input EDS file (.eds)");
        opts.add("output", 'o', true, false, "Output l-EDS file (default: <input>_l<N>.leds)");
        opts.add("context-length", 'l', true, true, "Minimum context length");
        opts.add("sources", 's', true, false, "Input source file (.seds) for linear (phasing-aware) merging");
        opts.add("full", 0, false, false, "Use full output format with brackets on all symbols (default: compact)");
        opts.add("threads", 't', true, false, "Number of threads for parallel processing");
        opts.parse(argc, argv);
        if (opts.has("help")) {
            std::cout << "eds2leds - Transform EDS to l-EDS (length-constrained EDS)\n\n" << opts.usage() << "\n"
                      << "MERGING METHODS (auto-detected):\n"
                         "  WITH sources (-s):  phasing-aware (linear) merging\n"
                         "  WITHOUT sources:    all-combinations (cartesian) merging\n\n"
                         "OUTPUT MODES:\n"
                         "  Default (compact): ACGT{A,ACA}CGT      --full: {ACGT}{A,ACA}{CGT}\n\n"
                         "OUTPUT FILES:\n"
                         "  <input_base>_l<N>.leds, and <input_base>_l<N>.seds with sources\n\n";
            tool::print_performance(timer);
            return 0;
        }
        opts.notify();
        const std::filesystem::path input_file = opts.get("input");
        std::filesystem::path output_file = opts.get("output");
        const std::filesystem::path sources_file = opts.get("sources");
        const Length context_length = static_cast<Length>(opts.get_unsigned("context-length", 0));
        const long num_threads = opts.get_int("threads", 1);
        const bool compact_mode = !opts.has("full");

        if (input_file.extension() != ".eds") {
            std::cerr << "Error: Input file must be an EDS file (.eds)\n";
            std::cerr << "Got: " << input_file << "\n";
            tool::print_performance(timer);
            return 1;
        }
        if (num_threads < 1) {
            std::cerr << "Error: Number of threads must be >= 1\n";
            tool::print_performance(timer);
            return 1;
        }
        if (context_length == 0) {
            std::cerr << "Error: Context length must be > 0\n";
            tool::print_performance(timer);
            return 1;
        }
        if (output_file.empty())
            output_file = input_file.parent_path() / (input_file.stem().string() + "_l" + std::to_string(context_length) + ".leds");

        std::cout << "EDS → l-EDS transformation\n";
        std::cout << "  Input: " << input_file << "\n";
        std::cout << "  Output: " << output_file << "\n";
        std::cout << "  Context length: " << context_length << "\n";
        if (!sources_file.empty()) std::cout << "  Sources: " << sources_file << "\n";
        std::cout << "  Output mode: " << (compact_mode ? "compact" : "full") << "\n";
        std::cout << "  Threads: " << num_threads << (num_threads == 1 ? " (sequential)" : " (parallel)") << "\n";

        std::ifstream input(input_file);
        if (!input) throw std::runtime_error("Cannot open input file: " + input_file.string());
        std::ofstream output(output_file);
        if (!output) throw std::runtime_error("Cannot open output file: " + output_file.string());

        std::unique_ptr<std::ifstream> sources_in;
        std::unique_ptr<std::ofstream> sources_out;
        if (!sources_file.empty()) {
            sources_in.reset(new std::ifstream(sources_file));
            if (!*sources_in) throw std::runtime_error("Cannot open sources file: " + sources_file.string());
            std::filesystem::path output_sources = output_file;
            output_sources.replace_extension(".seds");
            sources_out.reset(new std::ofstream(output_sources));
            if (!*sources_out) throw std::runtime_error("Cannot create output sources file: " + output_sources.string());
            std::cout << "  Output sources: " << output_sources << "\n";
        }
        if (sources_in)
            eds_to_leds_linear(input, output, context_length, sources_in.get(), sources_out.get(),
                               static_cast<size_t>(num_threads), compact_mode);
        else
            eds_to_leds_cartesian(input, output, context_length, static_cast<size_t>(num_threads), compact_mode);

        std::cout << "Transformation complete!\n";
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
