// Host-side EDS container: parse / save / sources / pairwise merge.
// Behaviour follows the reference's src/cpp/lib/formats/eds.cpp — parse :39-155, normalize
// :831-881, parse_sources :268-355, save :600-631, save_sources :641-659, merge_adjacent
// :1425-1695 — including its error texts, which callers and tests match on.
#include "edsparser/formats/eds.hpp"

#include <algorithm>
#include <cctype>
#include <fstream>
#include <iterator>
#include <sstream>
#include <stdexcept>

namespace edsparser {

namespace {

std::string read_all(std::istream& is) { return std::string(std::istreambuf_iterator<char>(is), {}); }

std::string drop_whitespace(const std::string& in)
{
    std::string out;
    out.reserve(in.size());
    for (unsigned char c : in) if (!std::isspace(c)) out.push_back(static_cast<char>(c));
    return out;
}

// "ACGT{A,C}T" -> "{ACGT}{A,C}{T}": text outside braces becomes a one-string symbol
std::string to_full_brackets(const std::string& in)
{
    std::string out, run;
    int depth = 0;
    for (char ch : in) {
        if (ch == SET_OPEN) {
            if (!run.empty() && depth == 0) { out += SET_OPEN; out += run; out += SET_CLOSE; run.clear(); }
            out += ch;
            ++depth;
        } else if (ch == SET_CLOSE) { out += ch; --depth; }
        else if (depth > 0) out += ch;
        else run += ch;
    }
    if (!run.empty() && depth == 0) { out += SET_OPEN; out += run; out += SET_CLOSE; }
    return out;
}

std::set<int> source_intersection(const std::set<int>& a, const std::set<int>& b)
{
    const bool ua = a.count(0) > 0, ub = b.count(0) > 0;   // 0 = universal path
    if (ua && ub) return {0};
    if (ua) return b;
    if (ub) return a;
    std::set<int> r;
    std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::inserter(r, r.begin()));
    return r;
}

} // namespace

EDS::EDS(std::istream& eds_stream) { parse(read_all(eds_stream)); }
EDS::EDS(std::istream& eds_stream, std::istream& seds_stream)
{
    parse(read_all(eds_stream));
    parse_sources(read_all(seds_stream));
}
EDS::EDS(const std::string& eds_string) { parse(eds_string); }
EDS::EDS(const std::string& eds_string, const std::string& seds_string)
{
    parse(eds_string);
    parse_sources(seds_string);
}

EDS EDS::load(const std::filesystem::path& path, StoringMode)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Failed to open file: " + path.string());
    return EDS(f);
}
EDS EDS::load(const std::filesystem::path& eds_path, const std::filesystem::path& seds_path, StoringMode)
{
    std::ifstream f(eds_path), s(seds_path);
    if (!f) throw std::runtime_error("Failed to open file: " + eds_path.string());
    if (!s) throw std::runtime_error("Failed to open file: " + seds_path.string());
    return EDS(f, s);
}

void EDS::parse(const std::string& text)
{
    sets_.clear();
    sources_.clear();
    has_sources_ = false;
    std::string in = drop_whitespace(text);
    if (in.empty()) { is_empty_ = true; n_ = N_ = m_ = 0; metadata_ = Metadata(); return; }
    in = to_full_brackets(in);
    metadata_ = Metadata();
    size_t pos = 0;
    while (pos < in.size()) {
        metadata_.base_positions.push_back(static_cast<std::streampos>(pos));
        if (in[pos] != SET_OPEN) throw std::runtime_error("Expected '{' at position " + std::to_string(pos));
        ++pos;
        StringSet set;
        std::string cur;
        while (pos < in.size() && in[pos] != SET_CLOSE) {
            if (in[pos] == SET_SEPARATOR) { set.push_back(cur); cur.clear(); }
            else cur += in[pos];
            ++pos;
        }
        set.push_back(cur);
        if (pos >= in.size() || in[pos] != SET_CLOSE)
            throw std::runtime_error("Expected '}' at position " + std::to_string(pos));
        ++pos;
        sets_.push_back(std::move(set));
    }
    rebuild_metadata();
}

void EDS::rebuild_metadata()
{
    auto base = std::move(metadata_.base_positions);
    metadata_ = Metadata();
    metadata_.base_positions = std::move(base);
    n_ = sets_.size();
    N_ = m_ = 0;
    size_t ndeg_free = 0, sum_ctx = 0;
    bool first_ctx = true;
    for (const auto& set : sets_) {
        metadata_.symbol_sizes.push_back(static_cast<Length>(set.size()));
        metadata_.cum_set_sizes.push_back(static_cast<Length>(m_));
        const bool deg = set.size() > 1;
        metadata_.is_degenerate.push_back(deg);
        for (const auto& s : set) {
            metadata_.string_lengths.push_back(static_cast<Length>(s.size()));
            N_ += s.size();
            if (s.empty()) metadata_.num_empty_strings++;
        }
        if (deg) { metadata_.num_degenerate_symbols++; metadata_.total_change_size += set.size() - 1; }   // eds.cpp:397-399
        else {
            const Length len = static_cast<Length>(set[0].size());
            metadata_.num_common_chars += len;
            if (first_ctx) { metadata_.min_context_length = metadata_.max_context_length = len; first_ctx = false; }
            metadata_.min_context_length = std::min(metadata_.min_context_length, len);
            metadata_.max_context_length = std::max(metadata_.max_context_length, len);
            sum_ctx += len;
            ndeg_free++;
        }
        m_ += set.size();
    }
    metadata_.avg_context_length = ndeg_free ? static_cast<double>(sum_ctx) / ndeg_free : 0.0;
    is_empty_ = (n_ == 0);
    if (has_sources_) {
        std::set<int> all;
        size_t total = 0;
        for (const auto& s : sources_) {
            all.insert(s.begin(), s.end());
            metadata_.max_paths_per_string = std::max(metadata_.max_paths_per_string, s.size());
            total += s.size();
        }
        metadata_.num_paths = all.size();
        metadata_.avg_paths_per_string = sources_.empty() ? 0.0 : static_cast<double>(total) / sources_.size();
    }
}

EDS::Statistics EDS::get_statistics() const
{
    const Metadata& md = metadata_;
    return Statistics{md.min_context_length, md.max_context_length, md.avg_context_length, md.num_degenerate_symbols,
                      md.num_common_chars, md.total_change_size, md.num_empty_strings, md.num_paths,
                      md.max_paths_per_string, md.avg_paths_per_string};
}

// The summary block of the reference (eds.cpp:528-557): titled groups of "label value" lines, labels padded to 32 columns.
void EDS::print_statistics(std::ostream& os) const
{
    const Statistics st = get_statistics();
    struct Line { const char* label; std::string value; };
    struct Group { const char* title; std::vector<Line> lines; };
    auto num = [](auto v) { std::ostringstream o; o << v; return o.str(); };
    const Group groups[] = {
        {"Structure", {{"Number of sets (n):", num(n_)}, {"Total characters (N):", num(N_)}, {"Total strings (m):", num(m_)},
                       {"Degenerate symbols:", num(st.num_degenerate_symbols)},
                       {"Regular symbols:", num(n_ - st.num_degenerate_symbols)}}},
        {"Context Lengths", {{"Minimum:", num(st.min_context_length)}, {"Maximum:", num(st.max_context_length)},
                             {"Average:", num(st.avg_context_length)}}},
        {"Variations", {{"Total change size:", num(st.total_change_size)}, {"Common characters:", num(st.num_common_chars)},
                        {"Empty strings:", num(st.num_empty_strings)}}},
    };
    const std::string rule(40, '=');
    os << rule << "\nEDS Statistics\n" << rule << "\n";
    for (const Group& g : groups) {
        os << g.title << ":\n";
        for (const Line& ln : g.lines) {
            std::string label = std::string("  ") + ln.label;
            label.resize(32, ' ');
            os << label << ln.value << "\n";
        }
        os << "\n";
    }
    if (has_sources_) os << "Sources: Loaded (" << sources_.size() << " strings with source info)\n";
    else os << "Sources: Not loaded\n";
    os << rule << "\n";
}

void EDS::print(std::ostream& os) const
{
    if (is_empty_) { os << "(empty EDS)\n"; return; }
    os << "EDS with " << n_ << " sets, " << m_ << " total strings:\n";
    for (size_t i = 0; i < sets_.size(); ++i) {
        os << "Set " << i << ": {";
        const char* sep = "";
        for (const std::string& str : sets_[i]) {
            os << sep;
            if (str.empty()) os << "\xce\xb5";                 // epsilon for the empty string
            else os << '"' << str << '"';
            sep = ", ";
        }
        os << (metadata_.is_degenerate[i] ? "} [degenerate]\n" : "}\n");
    }
}

void EDS::parse_sources(const std::string& text)
{
    const std::string in = drop_whitespace(text);
    if (in.empty()) throw std::runtime_error("sEDS input is empty");
    std::vector<std::set<int>> parsed;
    size_t pos = 0;
    while (pos < in.size()) {
        if (in[pos] != SET_OPEN) throw std::runtime_error("sEDS: Expected '{' at position " + std::to_string(pos));
        ++pos;
        std::set<int> ids;
        std::string num;
        auto flush = [&] {
            if (num.empty()) return;
            int id = std::stoi(num);
            if (id < 0) throw std::runtime_error("sEDS: Invalid path ID (must be >= 0): " + num);
            ids.insert(id);
            num.clear();
        };
        while (pos < in.size() && in[pos] != SET_CLOSE) {
            if (in[pos] == SET_SEPARATOR) flush();
            else if (std::isdigit(static_cast<unsigned char>(in[pos]))) num += in[pos];
            else
                throw std::runtime_error("sEDS: Invalid character '" + std::string(1, in[pos]) + "' at position " +
                                         std::to_string(pos));
            ++pos;
        }
        flush();
        if (pos >= in.size() || in[pos] != SET_CLOSE)
            throw std::runtime_error("sEDS: Expected '}' at position " + std::to_string(pos));
        ++pos;
        if (ids.empty()) throw std::runtime_error("sEDS: Empty path set at string " + std::to_string(parsed.size()));
        parsed.push_back(std::move(ids));
    }
    if (parsed.size() != m_)
        throw std::runtime_error("sEDS: Source count (" + std::to_string(parsed.size()) +
                                 ") does not match EDS cardinality (" + std::to_string(m_) + ")");
    sources_ = std::move(parsed);
    has_sources_ = true;
    rebuild_metadata();
}

void EDS::load_sources(std::istream& is) { parse_sources(read_all(is)); }
void EDS::load_sources(const std::string& seds_string) { parse_sources(seds_string); }
void EDS::load_sources(const std::filesystem::path& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Failed to open file: " + path.string());
    parse_sources(read_all(f));
}

void EDS::save(std::ostream& os, OutputFormat format) const
{
    for (size_t i = 0; i < sets_.size(); ++i) {
        const bool brackets = format == OutputFormat::FULL || metadata_.is_degenerate[i];
        if (brackets) os << SET_OPEN;
        for (size_t j = 0; j < sets_[i].size(); ++j) {
            if (j) os << SET_SEPARATOR;
            os << sets_[i][j];
        }
        if (brackets) os << SET_CLOSE;
    }
    os << "\n";
}
void EDS::save(const std::filesystem::path& path, OutputFormat format) const
{
    std::ofstream f(path);
    if (!f) throw std::runtime_error("Failed to open file for writing: " + path.string());
    save(f, format);
}

void EDS::save_sources(std::ostream& os) const
{
    if (!has_sources_) throw std::runtime_error("Cannot save sources: no sources loaded");
    for (const auto& ids : sources_) {
        os << SET_OPEN;
        bool first = true;
        for (int id : ids) { if (!first) os << SET_SEPARATOR; os << id; first = false; }
        os << SET_CLOSE;
    }
    os << "\n";
}
void EDS::save_sources(const std::filesystem::path& path) const
{
    std::ofstream f(path);
    if (!f) throw std::runtime_error("Failed to open file for writing: " + path.string());
    save_sources(f);
}

StringSet EDS::read_symbol(Position pos) const
{
    if (pos >= n_) throw std::out_of_range("Position out of range: " + std::to_string(pos));
    return sets_[pos];
}

EDS EDS::merge_adjacent(size_t pos1, size_t pos2) const
{
    if (pos2 != pos1 + 1)
        throw std::invalid_argument("Positions must be adjacent: pos2 (" + std::to_string(pos2) +
                                    ") must equal pos1 + 1 (" + std::to_string(pos1 + 1) + ")");
    if (pos1 >= n_ || pos2 >= n_)
        throw std::out_of_range("Position out of range: pos1=" + std::to_string(pos1) + ", pos2=" +
                                std::to_string(pos2) + ", n=" + std::to_string(n_));
    EDS out;
    out.has_sources_ = has_sources_;
    size_t sid = 0;                                            // flattened string id while walking
    for (size_t i = 0; i < n_; ++i) {
        if (i == pos2) { sid += sets_[i].size(); continue; }   // consumed by the merge
        if (i != pos1) {
            out.sets_.push_back(sets_[i]);
            if (has_sources_)
                for (size_t j = 0; j < sets_[i].size(); ++j) out.sources_.push_back(sources_[sid + j]);
            sid += sets_[i].size();
            continue;
        }
        const StringSet& A = sets_[pos1];
        const StringSet& B = sets_[pos2];
        const size_t a0 = metadata_.cum_set_sizes[pos1], b0 = metadata_.cum_set_sizes[pos2];
        StringSet merged;
        for (size_t x = 0; x < A.size(); ++x)
            for (size_t y = 0; y < B.size(); ++y) {
                if (!has_sources_) { merged.push_back(A[x] + B[y]); continue; }
                std::set<int> I = source_intersection(sources_[a0 + x], sources_[b0 + y]);
                if (I.empty()) continue;
                merged.push_back(A[x] + B[y]);
                out.sources_.push_back(std::move(I));
            }
        if (has_sources_ && merged.empty())
            throw std::runtime_error("Merging positions " + std::to_string(pos1) + " and " + std::to_string(pos2) +
                                     " results in empty set (no valid source intersections)");
        out.sets_.push_back(std::move(merged));
        sid += A.size();
    }
    for (size_t i = 0; i < n_; ++i)
        if (i != pos2) out.metadata_.base_positions.push_back(metadata_.base_positions[i]);
    out.rebuild_metadata();
    return out;
}

} // namespace edsparser
