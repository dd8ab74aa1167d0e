// msa2eds — MSA (.msa, FASTA with '-' gaps) -> EDS / l-EDS + sources, on the GPU.
// Flags, defaults, file naming and messages follow the reference tool
// (src/cpp/tools/msa2eds.cpp:35-41, :92-97, :139-160, :116-120, :178-180).
#include "edsparser/transforms/msa_transforms.hpp"
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
        cli::Parser opts("Transform MSA (Multiple Sequence Alignment) to EDS/l-EDS");
        opts.add("help", 'h', false, false, "Show help message");
        opts.add("input", 'i', true, true, "Input MSA file (.msa) in FASTA format with gaps as '-'");
        opts.add("output", 'o', true, false, "Output EDS file (default: <input>.eds)");
        opts.add("sources", 's', true, false, "Output source file (default: <output>.seds)");
        opts.add("context-length", 'l', true, false, "Create l-EDS with minimum context length (0 = regular EDS)");
        opts.add("gpus", 'g', true, false, "Spread the alignment's columns over this many GPUs of the node (RCCL boundary stitch; default 1)");
        opts.add("batches", 'b', true, false, "Transform in this many column batches, one after the other on the GPU (bounds device memory; default: one piece, batches only when the alignment does not fit)");
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
                         "  msa2eds -i alignment.msa -o output.eds -s output.seds\n"
                         "  msa2eds -i alignment.msa --gpus 8   # column slabs on GPUs 0..7, stitched over RCCL\n"
                         "  msa2eds -i alignment.msa --batches 4  # a quarter of the columns on the GPU at a time\n\n"
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
        // the file is mapped and handed to the C ABI as it is (what parse_msa_to_*_streaming does with a stream, minus
        // the copy into a std::string in front of the host-to-device copy)
        tool::MappedFile msa_in(input_file, "input");

        const bool create_leds = context_length > 0;
        if (create_leds) std::cout << "MSA → l-EDS transformation (l=" << context_length << ")\n";
        else std::cout << "MSA → EDS transformation\n";
        std::cout << "  Input: " << input_file << "\n";

        detail::Buf eds_out, seds_out;
        if (opts.has("gpus")) {
            // N rank threads inside the library, one per GPU (devices 0 .. N-1), ncclAllGather for the boundary stitch
            const unsigned long ngpu = opts.get_unsigned("gpus", 1);
            if (ngpu == 0 || ngpu > 64) throw std::runtime_error("--gpus must be between 1 and 64");
            std::vector<int> devs(ngpu);
            for (unsigned long i = 0; i < ngpu; i++) devs[i] = static_cast<int>(i);
            edsx_multi* mg = nullptr;
            if (edsx_multi_create(devs.data(), static_cast<int>(ngpu), 1, &mg) != EDSX_OK)
                throw std::runtime_error("cannot use " + std::to_string(ngpu) + " GPUs (gfx950 devices 0.." + std::to_string(ngpu - 1) + " with RCCL)");
            const int rc = edsx_msa_transform_multi(mg, msa_in.data(), msa_in.size(), context_length, &eds_out.b, &seds_out.b);
            const std::string what = rc != EDSX_OK ? edsx_multi_last_error(mg) : "";
            int parted = 0, chains = 0;
            edsx_multi_last_partition(mg, &parted, &chains);
            edsx_multi_destroy(mg);
            if (rc != EDSX_OK) throw std::runtime_error(what);
            std::cout << "  GPUs: " << ngpu << (parted ? " (column slabs, " + std::to_string(chains) + " boundary segments stitched)"
                                                         : std::string(ngpu > 1 ? " (not partitioned: one GPU transforms the file)" : "")) << "\n";
        } else if (opts.has("batches")) {
            const unsigned long nb = opts.get_unsigned("batches", 1);
            if (nb == 0 || nb > 4096) throw std::runtime_error("--batches must be between 1 and 4096");
            edsx_ctx* ctx = detail::context();
            int used = 0;
            const int rc = edsx_msa_transform_batched(ctx, msa_in.data(), msa_in.size(), context_length, static_cast<int>(nb),
                                                      &eds_out.b, &seds_out.b, &used);
            if (rc != EDSX_OK) detail::throw_status(rc, ctx);
            std::cout << "  Column batches: " << used << (used == 1 && nb > 1 ? " (not cut: one piece)" : "") << "\n";
        } else {
            edsx_ctx* ctx = detail::context();
            const int rc = edsx_msa_transform(ctx, msa_in.data(), msa_in.size(), context_length, &eds_out.b, &seds_out.b);
            if (rc != EDSX_OK) detail::throw_status(rc, ctx);
            if (edsx_msa_last_batches(ctx) > 1)
                std::cout << "  Column batches: " << edsx_msa_last_batches(ctx) << " (the alignment does not fit the device in one piece)\n";
        }

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
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
