// edsparser/formats/eds.hpp — in-memory EDS container (hot subset of the reference's class,
// src/cpp/lib/formats/eds.hpp:26-169): parse / save / sources / pairwise merge / metadata.
// Statistics / get_statistics / print_statistics / print as eds.hpp:107-129.  The query-side utilities of the
// reference (pattern sampling, extract, check_position, METADATA_ONLY streaming) are outside the transform
// hot path and are not provided.
#ifndef EDSPARSER_EDS_HPP
#define EDSPARSER_EDS_HPP

#include "../common.hpp"
#include <filesystem>
#include <iostream>
#include <set>
#include <string>
#include <vector>

namespace edsparser {

class EDS {
public:
    enum class StoringMode { FULL, METADATA_ONLY };
    enum class OutputFormat { FULL, COMPACT };

    EDS() = default;
    explicit EDS(std::istream& eds_stream);
    EDS(std::istream& eds_stream, std::istream& seds_stream);
    explicit EDS(const std::string& eds_string);
    EDS(const std::string& eds_string, const std::string& seds_string);

    static EDS load(const std::filesystem::path& path, StoringMode mode = StoringMode::FULL);
    static EDS load(const std::filesystem::path& eds_path, const std::filesystem::path& seds_path,
                    StoringMode mode = StoringMode::FULL);
    static EDS from_string(const std::string& eds_string) { return EDS(eds_string); }
    static EDS from_string(const std::string& eds_string, const std::string& seds_string)
    {
        return EDS(eds_string, seds_string);
    }

    EDS(const EDS&) = delete;
    EDS& operator=(const EDS&) = delete;
    EDS(EDS&&) = default;
    EDS& operator=(EDS&&) = default;

    bool empty() const { return is_empty_; }
    size_t length() const { return n_; }          // symbols
    size_t size() const { return N_; }            // characters
    size_t cardinality() const { return m_; }     // strings
    bool has_sources() const { return has_sources_; }
    StoringMode get_storing_mode() const { return StoringMode::FULL; }

    struct Metadata {
        std::vector<std::streampos> base_positions;
        std::vector<Length> symbol_sizes;
        std::vector<Length> string_lengths;
        std::vector<Length> cum_set_sizes;
        std::vector<bool> is_degenerate;
        Length min_context_length = 0, max_context_length = 0;
        double avg_context_length = 0.0;
        size_t num_degenerate_symbols = 0, num_common_chars = 0, total_change_size = 0, num_empty_strings = 0;
        size_t num_paths = 0, max_paths_per_string = 0;
        double avg_paths_per_string = 0.0;
    };
    const Metadata& get_metadata() const { return metadata_; }

    // The statistics portion of the metadata (reference: eds.hpp:107-123).
    struct Statistics {
        Length min_context_length;
        Length max_context_length;
        double avg_context_length;
        size_t num_degenerate_symbols;
        size_t num_common_chars;
        size_t total_change_size;
        size_t num_empty_strings;
        size_t num_paths;
        size_t max_paths_per_string;
        double avg_paths_per_string;
    };
    Statistics get_statistics() const;
    void print_statistics(std::ostream& os = std::cout) const;     // the reference's summary block (eds.cpp:528-557)
    void print(std::ostream& os = std::cout) const;                // one line per set (eds.cpp:559-598)

    void save(std::ostream& os, OutputFormat format = OutputFormat::FULL) const;
    void save(const std::filesystem::path& path, OutputFormat format = OutputFormat::FULL) const;
    void save_sources(std::ostream& os) const;
    void save_sources(const std::filesystem::path& path) const;
    void load_sources(std::istream& is);
    void load_sources(const std::filesystem::path& path);
    void load_sources(const std::string& seds_string);

    // Merge symbols pos1 and pos1+1: CARTESIAN without sources, LINEAR (source intersection,
    // 0 = universal path) with sources.  Same exceptions as the reference.
    EDS merge_adjacent(size_t pos1, size_t pos2) const;

    const std::vector<StringSet>& get_sets() const { return sets_; }
    const std::vector<bool>& get_is_degenerate() const { return metadata_.is_degenerate; }
    const std::vector<std::set<int>>& get_sources() const { return sources_; }
    StringSet read_symbol(Position pos) const;
    Length get_symbol_size(Position pos) const { return metadata_.symbol_sizes[pos]; }
    std::streampos get_base_position(Position pos) const { return metadata_.base_positions[pos]; }
    Length get_string_length(size_t string_id) const { return metadata_.string_lengths[string_id]; }

private:
    void parse(const std::string& text);
    void parse_sources(const std::string& text);
    void rebuild_metadata();

    bool is_empty_ = true;
    size_t n_ = 0, N_ = 0, m_ = 0;
    Metadata metadata_;
    std::vector<StringSet> sets_;
    bool has_sources_ = false;
    std::vector<std::set<int>> sources_;
};

} // namespace edsparser

#endif
