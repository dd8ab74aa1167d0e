// edsparser-stats — statistics of an EDS / l-EDS file (+ optional .seds), computed on the GPU.
// Flags, layout of the text and JSON reports, memory estimates and recommendations follow the reference tool
// (src/cpp/tools/stats.cpp:13-73 helpers, :76-170 text, :173-235 JSON, :238-330 main).  The numbers come from
// edsx_eds_stats (device reductions over the tokenised text: EDS::calculate_statistics eds.cpp:361-470,
// calculate_source_statistics :472-505).  Both storage modes of the reference print the same statistics; the mode line
// and the memory section follow the flag as they do there.
#include "edsx.h"
#include "../cli_util.hpp"
#include "../device.hpp"
#include "tool_common.hpp"

#include <sstream>

using namespace edsparser;

namespace {

std::string format_number(size_t num)                       // stats.cpp:14-19
{
    std::stringstream ss;
    try { ss.imbue(std::locale("")); } catch (...) {}
    ss << std::fixed << num;
    return ss.str();
}

std::string format_size(uintmax_t bytes)                    // stats.cpp:22-35
{
    const char* units[] = {"B", "KB", "MB", "GB", "TB"};
    int unit_idx = 0;
    double size = static_cast<double>(bytes);
    while (size >= 1024.0 && unit_idx < 4) { size /= 1024.0; unit_idx++; }
    std::stringstream ss;
    ss << std::fixed << std::setprecision(1) << size << " " << units[unit_idx];
    return ss.str();
}

size_t estimate_full_mode_memory(size_t N, size_t m, size_t n)   // stats.cpp:38-51
{
    const size_t string_data = N, string_overhead = m * 32, vector_overhead = n * 24;
    const size_t bookkeeping = (string_data + string_overhead + vector_overhead) / 5;
    return string_data + string_overhead + vector_overhead + bookkeeping;
}

size_t estimate_metadata_memory(size_t m, size_t n)         // stats.cpp:54-73
{
    const size_t total = n * 8 + n * 4 + m * 4 + n * 4 + n * 1 + 64;
    return total + total / 10;
}

void print_standard(const edsx_eds_statistics& st, const std::filesystem::path& input_file, bool verbose, bool full_mode)
{
    const uintmax_t file_size = std::filesystem::file_size(input_file);
    const size_t metadata_mem = estimate_metadata_memory(st.n_strings, st.n_symbols);
    const size_t full_mem = estimate_full_mode_memory(st.n_chars, st.n_strings, st.n_symbols);
    const double reduction_factor = static_cast<double>(full_mem) / static_cast<double>(metadata_mem);
    std::cout << "========================================\n";
    std::cout << "EDS Statistics\n";
    std::cout << "========================================\n";
    std::cout << "File: " << input_file.filename().string() << "\n";
    std::cout << "Size: " << format_size(file_size) << "\n";
    std::cout << "Storage Mode: " << (!full_mode ? "METADATA_ONLY (memory-efficient)" : "FULL (all data in RAM)") << "\n\n";
    std::cout << "Structure:\n";
    std::cout << "  Number of symbols (n):        " << std::setw(12) << format_number(st.n_symbols) << "\n";
    std::cout << "  Total characters (N):         " << std::setw(12) << format_number(st.n_chars) << "\n";
    std::cout << "  Total strings (m):            " << std::setw(12) << format_number(st.n_strings) << "\n";
    std::cout << "  Degenerate symbols:           " << std::setw(12) << format_number(st.num_degenerate_symbols) << "\n";
    std::cout << "  Regular symbols:              " << std::setw(12) << format_number(st.n_symbols - st.num_degenerate_symbols) << "\n\n";
    std::cout << "Context Lengths (non-degenerate symbols):\n";
    std::cout << "  Minimum:                      " << std::setw(12) << st.min_context_length << "\n";
    std::cout << "  Maximum:                      " << std::setw(12) << st.max_context_length << "\n";
    std::cout << "  Average:                      " << std::setw(12) << std::fixed << std::setprecision(2) << st.avg_context_length << "\n\n";
    std::cout << "Variations:\n";
    std::cout << "  Total change size:            " << std::setw(12) << format_number(st.total_change_size) << "\n";
    std::cout << "  Common characters:            " << std::setw(12) << format_number(st.num_common_chars) << "\n";
    std::cout << "  Empty strings:                " << std::setw(12) << format_number(st.num_empty_strings) << "\n\n";
    if (verbose) {
        std::cout << "Detailed Metrics:\n";
        std::cout << "  Avg strings per symbol:       " << std::setw(12) << std::fixed << std::setprecision(2)
                  << (static_cast<double>(st.n_strings) / st.n_symbols) << "\n";
        std::cout << "  Avg chars per string:         " << std::setw(12) << std::fixed << std::setprecision(2)
                  << (static_cast<double>(st.n_chars) / st.n_strings) << "\n";
        std::cout << "  Degenerate ratio:             " << std::setw(12) << std::fixed << std::setprecision(2)
                  << (100.0 * st.num_degenerate_symbols / st.n_symbols) << " %\n\n";
    }
    if (st.has_sources) {
        std::cout << "Sources (pangenome paths):\n";
        std::cout << "  Strings with source info:     " << std::setw(12) << format_number(st.n_strings) << "\n";
        std::cout << "  Total paths (genomes):        " << std::setw(12) << format_number(st.num_paths) << "\n";
        std::cout << "  Max paths per string:         " << std::setw(12) << format_number(st.max_paths_per_string) << "\n";
        std::cout << "  Avg paths per string:         " << std::setw(12) << std::fixed << std::setprecision(2) << st.avg_paths_per_string << "\n\n";
    }
    std::cout << "Memory Usage:\n";
    std::cout << "  Current (" << (!full_mode ? "METADATA_ONLY" : "FULL") << "): " << std::setw(12)
              << format_size(!full_mode ? metadata_mem : full_mem) << "\n";
    if (!full_mode) {
        std::cout << "  Estimated FULL mode:          " << std::setw(12) << format_size(full_mem) << "\n";
        std::cout << "  Reduction factor:             " << std::setw(12) << std::fixed << std::setprecision(1) << reduction_factor << "x\n";
    }
    std::cout << "\nRecommendations:\n";
    if (st.min_context_length < 5) {
        std::cout << "  ⚠️  Minimum context length (" << st.min_context_length << ") < typical l-EDS threshold (5)\n";
        std::cout << "  → Transformation to l-EDS may require merging adjacent symbols\n";
        std::cout << "  → Suggested command:\n";
        std::cout << "      edsparser-transform -i " << input_file.filename().string() << " -l 5 --method linear\n";
    } else {
        std::cout << "  ✓ Minimum context length (" << st.min_context_length << ") ≥ 5\n";
        std::cout << "  → Ready for indexing with l ≤ " << st.min_context_length << "\n";
    }
    std::cout << "========================================\n";
}

void print_json(const edsx_eds_statistics& st, const std::filesystem::path& input_file, bool has_sources_file, bool full_mode)
{
    const uintmax_t file_size = std::filesystem::file_size(input_file);
    const size_t metadata_mem = estimate_metadata_memory(st.n_strings, st.n_symbols);
    const size_t full_mem = estimate_full_mode_memory(st.n_chars, st.n_strings, st.n_symbols);
    const double reduction_factor = static_cast<double>(full_mem) / static_cast<double>(metadata_mem);
    const size_t cur = !full_mode ? metadata_mem : full_mem;
    std::cout << "{\n  \"file\": {\n";
    std::cout << "    \"path\": \"" << input_file.string() << "\",\n";
    std::cout << "    \"size_bytes\": " << file_size << ",\n";
    std::cout << "    \"storage_mode\": \"" << (!full_mode ? "METADATA_ONLY" : "FULL") << "\"\n  },\n";
    std::cout << "  \"structure\": {\n";
    std::cout << "    \"n_symbols\": " << st.n_symbols << ",\n";
    std::cout << "    \"N_characters\": " << st.n_chars << ",\n";
    std::cout << "    \"m_strings\": " << st.n_strings << ",\n";
    std::cout << "    \"degenerate_symbols\": " << st.num_degenerate_symbols << ",\n";
    std::cout << "    \"regular_symbols\": " << (st.n_symbols - st.num_degenerate_symbols) << "\n  },\n";
    std::cout << "  \"context_lengths\": {\n";
    std::cout << "    \"min\": " << st.min_context_length << ",\n";
    std::cout << "    \"max\": " << st.max_context_length << ",\n";
    std::cout << "    \"avg\": " << std::fixed << std::setprecision(2) << st.avg_context_length << "\n  },\n";
    std::cout << "  \"variations\": {\n";
    std::cout << "    \"total_change_size\": " << st.total_change_size << ",\n";
    std::cout << "    \"common_characters\": " << st.num_common_chars << ",\n";
    std::cout << "    \"empty_strings\": " << st.num_empty_strings << "\n  },\n";
    std::cout << "  \"memory\": {\n";
    std::cout << "    \"current_bytes\": " << cur << ",\n";
    std::cout << "    \"current_mb\": " << std::fixed << std::setprecision(1) << (cur / 1024.0 / 1024.0) << ",\n";
    if (!full_mode) {
        std::cout << "    \"estimated_full_bytes\": " << full_mem << ",\n";
        std::cout << "    \"estimated_full_mb\": " << std::fixed << std::setprecision(1) << (full_mem / 1024.0 / 1024.0) << ",\n";
        std::cout << "    \"reduction_factor\": " << std::fixed << std::setprecision(1) << reduction_factor << "\n";
    } else {
        std::cout << "    \"mode\": \"FULL\"\n";
    }
    std::cout << "  },\n  \"sources\": {\n";
    std::cout << "    \"loaded\": " << (st.has_sources ? "true" : "false") << ",\n";
    std::cout << "    \"file_provided\": " << (has_sources_file ? "true" : "false") << ",\n";
    if (st.has_sources) {
        std::cout << "    \"num_paths\": " << st.num_paths << ",\n";
        std::cout << "    \"max_paths_per_string\": " << st.max_paths_per_string << ",\n";
        std::cout << "    \"avg_paths_per_string\": " << std::fixed << std::setprecision(2) << st.avg_paths_per_string << "\n";
    } else {
        std::cout << "    \"num_paths\": 0,\n    \"max_paths_per_string\": 0,\n    \"avg_paths_per_string\": 0.0\n";
    }
    std::cout << "  },\n  \"recommendations\": {\n";
    std::cout << "    \"needs_transformation\": " << (st.min_context_length < 5 ? "true" : "false") << ",\n";
    std::cout << "    \"ready_for_indexing\": " << (st.min_context_length >= 5 ? "true" : "false") << ",\n";
    std::cout << "    \"min_context_length\": " << st.min_context_length << ",\n";
    std::cout << "    \"suggested_command\": \"" << (st.min_context_length < 5
                  ? "edsparser-transform -i " + input_file.filename().string() + " -l 5" : std::string("ready for indexing")) << "\"\n";
    std::cout << "  }\n}\n";
}

std::string read_file(const std::filesystem::path& p)
{
    std::ifstream in(p, std::ios::binary);
    if (!in) throw std::runtime_error("Failed to open file: " + p.string());
    return detail::slurp(in);
}

} // namespace

int main(int argc, char** argv)
{
    Timer timer;
    timer.start();
    try {
        cli::Parser opts("Display statistics for EDS/l-EDS file");
        opts.add("help", 'h', false, false, "Show help message");
        opts.add("input", 'i', true, true, "Input EDS file");
        opts.add("sources", 's', true, false, "Source file (.seds) - optional");
        opts.add("full", 'f', false, false, "Use FULL mode (load all strings)");
        opts.add("json", 'j', false, false, "Output in JSON format");
        opts.add("verbose", 'v', false, false, "Show detailed statistics");
        opts.parse(argc, argv);
        if (opts.has("help")) {
            std::cout << "edsparser-stats - Display EDS statistics\n\n" << opts.usage() << "\n"
                      << "Examples:\n"
                         "  # Show statistics for EDS file (memory-efficient):\n  edsparser-stats -i data.eds\n\n"
                         "  # Show statistics with sources:\n  edsparser-stats -i data.eds -s data.seds\n\n"
                         "  # Show statistics in JSON format:\n  edsparser-stats -i data.eds --json\n\n"
                         "  # Use FULL mode (loads all strings, more memory):\n  edsparser-stats -i data.eds --full --verbose\n\n"
                         "Implementation:\n  The text is tokenised and reduced on an AMD MI355X (gfx950) through libedsx; both modes\n"
                         "  report the same numbers.\n";
            tool::print_performance(timer);
            return 0;
        }
        opts.notify();
        const std::filesystem::path input_file = opts.get("input"), sources_file = opts.get("sources");
        if (!std::filesystem::exists(input_file)) {
            std::cerr << "Error: Input file '" << input_file << "' not found\n";
            tool::print_performance(timer);
            return 1;
        }
        const bool with_sources = opts.has("sources");
        if (with_sources && !std::filesystem::exists(sources_file)) {
            std::cerr << "Error: Source file '" << sources_file << "' not found\n";
            tool::print_performance(timer);
            return 1;
        }
        const std::string eds = read_file(input_file), seds = with_sources ? read_file(sources_file) : std::string();
        edsx_ctx* ctx = detail::context();
        edsx_eds_statistics st;
        const int rc = edsx_eds_stats(ctx, reinterpret_cast<const uint8_t*>(eds.data()), eds.size(),
                                      with_sources ? reinterpret_cast<const uint8_t*>(seds.data()) : nullptr, seds.size(), 0, &st);
        if (rc != EDSX_OK) detail::throw_status(rc, ctx);
        if (opts.has("json")) print_json(st, input_file, with_sources, opts.has("full"));
        else print_standard(st, input_file, opts.has("verbose"), opts.has("full"));
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
