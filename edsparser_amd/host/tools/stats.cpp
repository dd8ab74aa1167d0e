// edsparser-stats — statistics of an EDS / l-EDS file (+ optional .seds), computed on the GPU.
// Flags and the printed text / JSON are the reference tool's contract (src/cpp/tools/stats.cpp:76-235, flags :238-330);
// here a report is DATA - sections of (label, value) rows - walked by one text and one JSON emitter.  The numbers come from
// edsx_eds_stats (device reductions over the tokenised text: EDS::calculate_statistics eds.cpp:361-470,
// calculate_source_statistics :472-505).  Both storage modes of the reference print the same statistics; the mode line
// and the memory section follow the flag as they do there.
#include "edsx.h"
#include "../cli_util.hpp"
#include "../device.hpp"
#include "tool_common.hpp"

#include <cstdio>
#include <locale>
#include <sstream>
#include <vector>

using namespace edsparser;

namespace {

// ---- number formatting ---------------------------------------------------------------------------------------
// Counts are printed with the user's locale grouping (the reference imbues std::locale("") into a stream; the same
// digits come out of the locale's numpunct facet applied by hand).
std::string grouped(uint64_t v)
{
    std::string digits = std::to_string(v);
    std::locale loc;
    try { loc = std::locale(""); } catch (...) { return digits; }
    const auto& np = std::use_facet<std::numpunct<char>>(loc);
    const std::string rule = np.grouping();
    if (rule.empty()) return digits;
    std::string out;
    size_t gi = 0, left = digits.size();
    while (left > 0) {
        const unsigned char g = static_cast<unsigned char>(rule[std::min(gi, rule.size() - 1)]);
        if (g == 0 || g >= 127 || g >= left) { out.insert(0, digits, 0, left); break; }
        out.insert(0, digits, left - g, g);
        out.insert(out.begin(), np.thousands_sep());
        left -= g; gi++;
    }
    return out;
}
std::string fixed(double v, int places)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.*f", places, v);
    return buf;
}
std::string human_bytes(uint64_t bytes)                          // "12.3 MB", powers of 1024 up to TB
{
    static const char* const unit[] = {"B", "KB", "MB", "GB", "TB"};
    double v = static_cast<double>(bytes);
    int u = 0;
    for (; u < 4 && v >= 1024.0; u++) v /= 1024.0;
    return fixed(v, 1) + " " + unit[u];
}

// ---- memory model of the reference's two storage modes (closed formulas of n symbols, m strings, N characters) --
struct MemoryModel {
    uint64_t full, metadata;
    MemoryModel(uint64_t N, uint64_t m, uint64_t n)
    {
        const uint64_t payload = N + 32 * m + 24 * n;            // characters + std::string headers + vector headers
        full = payload + payload / 5;                            // + 20 % allocator bookkeeping
        const uint64_t index = (8 + 4 + 4 + 1) * n + 4 * m + 64;           // per-symbol tables, per-string lengths, fixed part
        metadata = index + index / 10;
    }
    double reduction() const { return static_cast<double>(full) / static_cast<double>(metadata); }
};

// ---- the report as data --------------------------------------------------------------------------------------
// A report is a list of sections; a section is a title and (label, value) rows.  The text emitter pads labels to a
// fixed column and right-aligns values; the JSON emitter prints (key, literal) rows.  What the reference tool prints
// (src/cpp/tools/stats.cpp:76-235) is the contract; only the two row tables below know it.
struct Row { std::string label, value; };
struct Section { std::string title; std::vector<Row> rows; };

struct Facts {
    const edsx_eds_statistics& st;
    std::filesystem::path file;
    uint64_t file_bytes;
    bool full_mode, verbose, sources_given;
    MemoryModel mem;
    const char* mode_name() const { return full_mode ? "FULL" : "METADATA_ONLY"; }
    uint64_t current_mem() const { return full_mode ? mem.full : mem.metadata; }
    bool short_context() const { return st.min_context_length < 5; }
};

void emit_text(const Facts& f)
{
    const edsx_eds_statistics& st = f.st;
    const std::string rule(40, '=');
    std::vector<Section> doc;
    doc.push_back({"Structure", {{"Number of symbols (n):", grouped(st.n_symbols)}, {"Total characters (N):", grouped(st.n_chars)},
                                 {"Total strings (m):", grouped(st.n_strings)}, {"Degenerate symbols:", grouped(st.num_degenerate_symbols)},
                                 {"Regular symbols:", grouped(st.n_symbols - st.num_degenerate_symbols)}}});
    doc.push_back({"Context Lengths (non-degenerate symbols)",
                   {{"Minimum:", std::to_string(st.min_context_length)}, {"Maximum:", std::to_string(st.max_context_length)},
                    {"Average:", fixed(st.avg_context_length, 2)}}});
    doc.push_back({"Variations", {{"Total change size:", grouped(st.total_change_size)}, {"Common characters:", grouped(st.num_common_chars)},
                                  {"Empty strings:", grouped(st.num_empty_strings)}}});
    if (f.verbose)
        doc.push_back({"Detailed Metrics",
                       {{"Avg strings per symbol:", fixed(static_cast<double>(st.n_strings) / st.n_symbols, 2)},
                        {"Avg chars per string:", fixed(static_cast<double>(st.n_chars) / st.n_strings, 2)},
                        {"Degenerate ratio:", fixed(100.0 * st.num_degenerate_symbols / st.n_symbols, 2) + " %"}}});
    if (st.has_sources)
        doc.push_back({"Sources (pangenome paths)",
                       {{"Strings with source info:", grouped(st.n_strings)}, {"Total paths (genomes):", grouped(st.num_paths)},
                        {"Max paths per string:", grouped(st.max_paths_per_string)}, {"Avg paths per string:", fixed(st.avg_paths_per_string, 2)}}});
    Section memsec{"Memory Usage", {}};
    memsec.rows.push_back({std::string("Current (") + f.mode_name() + "):", human_bytes(f.current_mem())});
    if (!f.full_mode) {
        memsec.rows.push_back({"Estimated FULL mode:", human_bytes(f.mem.full)});
        memsec.rows.push_back({"Reduction factor:", fixed(f.mem.reduction(), 1) + "x"});
    }

    std::ostream& os = std::cout;
    os << rule << "\nEDS Statistics\n" << rule << "\n";
    os << "File: " << f.file.filename().string() << "\nSize: " << human_bytes(f.file_bytes) << "\n";
    os << "Storage Mode: " << (f.full_mode ? "FULL (all data in RAM)" : "METADATA_ONLY (memory-efficient)") << "\n\n";
    auto put = [&](const Section& sec, bool pad_label) {
        os << sec.title << ":\n";
        for (const Row& r : sec.rows) {
            std::string label = "  " + r.label;
            if (pad_label) label.resize(32, ' '); else label += ' ';
            // right-align on the value's leading part; a unit suffix (" %", "x") trails the 12-column field
            size_t cut = r.value.size();
            if (cut >= 2 && r.value.compare(cut - 2, 2, " %") == 0) cut -= 2;
            else if (cut >= 1 && r.value.back() == 'x') cut -= 1;
            const std::string num = r.value.substr(0, cut);
            os << label << std::string(num.size() < 12 ? 12 - num.size() : 0, ' ') << num << r.value.substr(cut) << "\n";
        }
    };
    for (const Section& sec : doc) { put(sec, true); os << "\n"; }
    // (the memory block: its first label carries the mode name and is not padded to the label column)
    os << memsec.title << ":\n";
    for (size_t i = 0; i < memsec.rows.size(); i++) {
        const Row& r = memsec.rows[i];
        std::string label = "  " + r.label;
        if (i == 0) label += ' '; else label.resize(32, ' ');
        size_t cut = r.value.size();
        if (r.value.back() == 'x') cut -= 1;
        const std::string num = r.value.substr(0, cut);
        os << label << std::string(num.size() < 12 ? 12 - num.size() : 0, ' ') << num << r.value.substr(cut) << "\n";
    }
    os << "\nRecommendations:\n";
    const std::string lmin = std::to_string(st.min_context_length);
    if (f.short_context())
        os << "  \xe2\x9a\xa0\xef\xb8\x8f  Minimum context length (" << lmin << ") < typical l-EDS threshold (5)\n"
           << "  \xe2\x86\x92 Transformation to l-EDS may require merging adjacent symbols\n"
           << "  \xe2\x86\x92 Suggested command:\n"
           << "      edsparser-transform -i " << f.file.filename().string() << " -l 5 --method linear\n";
    else
        os << "  \xe2\x9c\x93 Minimum context length (" << lmin << ") \xe2\x89\xa5 5\n"
           << "  \xe2\x86\x92 Ready for indexing with l \xe2\x89\xa4 " << lmin << "\n";
    os << rule << "\n";
}

void emit_json(const Facts& f)
{
    const edsx_eds_statistics& st = f.st;
    auto q = [](const std::string& s) { return "\"" + s + "\""; };
    auto yes = [](bool b) { return std::string(b ? "true" : "false"); };
    auto n = [](uint64_t v) { return std::to_string(v); };
    auto mb = [](uint64_t b) { return fixed(b / 1024.0 / 1024.0, 1); };
    std::vector<Section> doc;
    doc.push_back({"file", {{"path", q(f.file.string())}, {"size_bytes", n(f.file_bytes)}, {"storage_mode", q(f.mode_name())}}});
    doc.push_back({"structure", {{"n_symbols", n(st.n_symbols)}, {"N_characters", n(st.n_chars)}, {"m_strings", n(st.n_strings)},
                                 {"degenerate_symbols", n(st.num_degenerate_symbols)},
                                 {"regular_symbols", n(st.n_symbols - st.num_degenerate_symbols)}}});
    doc.push_back({"context_lengths", {{"min", n(st.min_context_length)}, {"max", n(st.max_context_length)},
                                       {"avg", fixed(st.avg_context_length, 2)}}});
    doc.push_back({"variations", {{"total_change_size", n(st.total_change_size)}, {"common_characters", n(st.num_common_chars)},
                                  {"empty_strings", n(st.num_empty_strings)}}});
    Section mem{"memory", {{"current_bytes", n(f.current_mem())}, {"current_mb", mb(f.current_mem())}}};
    if (f.full_mode) mem.rows.push_back({"mode", q("FULL")});
    else {
        mem.rows.push_back({"estimated_full_bytes", n(f.mem.full)});
        mem.rows.push_back({"estimated_full_mb", mb(f.mem.full)});
        mem.rows.push_back({"reduction_factor", fixed(f.mem.reduction(), 1)});
    }
    doc.push_back(mem);
    Section src{"sources", {{"loaded", yes(st.has_sources)}, {"file_provided", yes(f.sources_given)}}};
    src.rows.push_back({"num_paths", st.has_sources ? n(st.num_paths) : "0"});
    src.rows.push_back({"max_paths_per_string", st.has_sources ? n(st.max_paths_per_string) : "0"});
    src.rows.push_back({"avg_paths_per_string", st.has_sources ? fixed(st.avg_paths_per_string, 2) : "0.0"});
    doc.push_back(src);
    doc.push_back({"recommendations",
                   {{"needs_transformation", yes(f.short_context())}, {"ready_for_indexing", yes(!f.short_context())},
                    {"min_context_length", n(st.min_context_length)},
                    {"suggested_command", q(f.short_context() ? "edsparser-transform -i " + f.file.filename().string() + " -l 5"
                                                              : std::string("ready for indexing"))}}});
    std::ostream& os = std::cout;
    os << "{\n";
    for (size_t s = 0; s < doc.size(); s++) {
        os << "  \"" << doc[s].title << "\": {\n";
        for (size_t r = 0; r < doc[s].rows.size(); r++)
            os << "    \"" << doc[s].rows[r].label << "\": " << doc[s].rows[r].value << (r + 1 < doc[s].rows.size() ? ",\n" : "\n");
        os << (s + 1 < doc.size() ? "  },\n" : "  }\n");
    }
    os << "}\n";
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
        const Facts facts{st, input_file, static_cast<uint64_t>(std::filesystem::file_size(input_file)), opts.has("full"),
                          opts.has("verbose"), with_sources, MemoryModel(st.n_chars, st.n_strings, st.n_symbols)};
        if (opts.has("json")) emit_json(facts); else emit_text(facts);
        tool::print_performance(timer);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << "\n";
        tool::print_performance(timer);
        return 1;
    }
}
