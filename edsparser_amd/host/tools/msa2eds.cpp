// msa2eds — MSA (.msa, FASTA with '-' gaps) -> EDS / l-EDS + sources, on the GPU.
// Flags, defaults, file naming and messages follow the reference tool
// (src/cpp/tools/msa2eds.cpp:35-41, :92-97, :139-160, :116-120, :178-180).
#include "edsparser/transforms/msa_transforms.hpp"
#include "../cli_util.hpp"
#include "tool_common.hpp"

using namespace edsparser;

int main(int argc, char** argv)
{
    Timer timer;
    timer.start();
    try {
        cli::Parser opts("Transform MSA (Multiple Sequence Alignment) to EDS/l-EDS");
        opts.add("help", 'h', false, false, "Show help message");
        opts.add("input", 'i', true, true, "Input MSA file (.msa) in FASTA format with gaps as '-'");
        opts.add("output", 'o', true, false, "Output EDS file (default: <input>.eds)");
        opts.add("sources", 's', true, false, "Output source file (default: <output>.seds)");
        opts.add("context-length", 'l', true, false, "Create l-EDS with minimum context length (0 = regular EDS)");
        opts.parse(argc, argv);
        if (opts.has("help")) {
            std::cout << "msa2eds - Transform MSA (Multiple Sequence Alignment) to EDS\n\n" << opts.usage() << "\n"
                      << "DESCRIPTION:\n"
                         "  Transforms a Multiple Sequence Alignment (MSA) in FASTA format to an\n"
                         "  Elastic-Degenerate String (EDS) with source tracking. Gaps in the MSA\n"
                         "  (represented as '-') are used to identify variant regions.\n\n"
                         "EXAMPLES:\n"
                         "  msa2eds -i alignment.msa            # alignment.eds + alignment.seds\n"
                         "  msa2eds -i alignment.msa -l 10      # alignment_l10.leds + alignment_l10.seds\n"
                         "  msa2eds -i alignment.msa -o output.eds -s output.seds\n\n"
                         "IMPLEMENTATION:\n"
                         "  The alignment is transformed on an AMD MI355X (gfx950) through libedsx.\n\n";
            tool::print_performance(timer);
            return 0;
        }
        opts.notify();
        const std::filesystem::path input_file = opts.get("input");
        const std::filesystem::path output_file = opts.get("output");
        const std::filesystem::path sources_file = opts.get("sources");
        const Length context_length = static_cast<Length>(opts.get_unsigned("context-length", 0));

        if (input_file.extension() != ".msa") {
            std::cerr << "Error: Input file must be an MSA file (.msa)\n";
            std::cerr << "Got: " << input_file << "\n";
            tool::print_performance(timer);
            return 1;
        }
        std::ifstream msa_in(input_file);
        if (!msa_in) throw std::runtime_error("Failed to open input file: " + input_file.string());

        const bool create_leds = context_length > 0;
        if (create_leds) std::cout << "MSA → l-EDS transformation (l=" << context_length << ")\n";
        else std::cout << "MSA → EDS transformation\n";
        std::cout << "  Input: " << input_file << "\n";

        auto result = create_leds ? parse_msa_to_leds_streaming(msa_in, context_length) : parse_msa_to_eds_streaming(msa_in);
        msa_in.close();

        std::filesystem::path eds_path, seds_path;
        if (create_leds) {
            const std::string base = input_file.stem().string(), suffix = "_l" + std::to_string(context_length);
            eds_path = output_file.empty() ? input_file.parent_path() / (base + suffix + ".leds") : output_file;
            seds_path = sources_file.empty() ? eds_path.parent_path() / (base + suffix + ".seds") : sources_file;
        } else {
            eds_path = output_file.empty() ? input_file.parent_path() / (input_file.stem().string() + ".eds") : output_file;
            seds_path = sources_file.empty() ? eds_path.parent_path() / (eds_path.stem().string() + ".seds") : sources_file;
        }
        tool::write_file(eds_path, result.first, "output");
        tool::write_file(seds_path, result.second, "sources");

        std::cout << "Transformation complete!\n";
        std::cout << "  Output: " << eds_path << "\n";
        std::cout << "  Sources: " << seds_path << "\n";
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
