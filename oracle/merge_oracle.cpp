/*
 * merge_oracle.cpp — CPU restatement of the reference EDS -> l-EDS merge
 * (LINEAR with sources / CARTESIAN without).  TEST INFRASTRUCTURE ONLY (see oracle.h).
 * Parity: pinned by the reference's data/eds/<name>_l<N>.eds goldens and differentially
 * against oracle/_ref (the reference's eds.cpp + eds_transforms.cpp compiled here).
 *
 * Follows, round by round:
 *   EDS::parse                      src/cpp/lib/formats/eds.cpp:39-155
 *   EDS::normalize_eds_format       eds.cpp:831-881
 *   EDS::parse_sources              eds.cpp:268-355
 *   is_leds                         src/cpp/lib/transforms/eds_transforms.cpp:439-468
 *   select_independent_merge_pairs  eds_transforms.cpp:46-107
 *   EDS::merge_adjacent             eds.cpp:1425-1695
 *   reconstruct_eds                 eds_transforms.cpp:207-296 (text round trip is an identity
 *                                   on the parsed structure, so it is done in memory)
 *   EDS::save / save_sources        eds.cpp:600-631 / :641-659
 */
#include "oracle.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

using StringSet = std::vector<std::string>;

struct Eds {
    std::vector<StringSet> sets;
    std::vector<std::set<int>> sources;   // one per string, flattened (empty if no sources)
    bool has_sources = false;
};

std::string strip_ws(const uint8_t* p, size_t n)            // eds.cpp:46
{
    std::string s;
    s.reserve(n);
    for (size_t i = 0; i < n; i++) if (!std::isspace(p[i])) s.push_back(static_cast<char>(p[i]));
    return s;
}

std::string normalize(const std::string& input)              // eds.cpp:831-881
{
    std::string result, cur;
    int depth = 0;
    for (char ch : input) {
        if (ch == '{') {
            if (!cur.empty() && depth == 0) { result += "{" + cur + "}"; cur.clear(); }
            result += ch; depth++;
        } else if (ch == '}') { result += ch; depth--; }
        else if (depth > 0) result += ch;
        else cur += ch;
    }
    if (!cur.empty() && depth == 0) result += "{" + cur + "}";
    return result;
}

void parse_eds(const uint8_t* p, size_t n, Eds& e)           // eds.cpp:39-155
{
    std::string input = strip_ws(p, n);
    if (input.empty()) return;
    input = normalize(input);
    size_t pos = 0;
    while (pos < input.size()) {
        if (input[pos] != '{') throw std::runtime_error("Expected '{' at position " + std::to_string(pos));
        pos++;
        StringSet cur_set;
        std::string cur;
        while (pos < input.size() && input[pos] != '}') {
            if (input[pos] == ',') { cur_set.push_back(cur); cur.clear(); }
            else cur += input[pos];
            pos++;
        }
        cur_set.push_back(cur);
        if (pos >= input.size() || input[pos] != '}')
            throw std::runtime_error("Expected '}' at position " + std::to_string(pos));
        pos++;
        e.sets.push_back(std::move(cur_set));
    }
}

void parse_sources(const uint8_t* p, size_t n, Eds& e)       // eds.cpp:268-355
{
    std::string input = strip_ws(p, n);
    if (input.empty()) throw std::runtime_error("sEDS input is empty");
    size_t pos = 0, m = 0;
    for (auto& s : e.sets) m += s.size();
    while (pos < input.size()) {
        if (input[pos] != '{')
            throw std::runtime_error("sEDS: Expected '{' at position " + std::to_string(pos));
        pos++;
        std::set<int> ids;
        std::string num;
        while (pos < input.size() && input[pos] != '}') {
            if (input[pos] == ',') {
                if (!num.empty()) { ids.insert(std::stoi(num)); num.clear(); }
                pos++;
            } else if (std::isdigit(static_cast<unsigned char>(input[pos]))) {
                num += input[pos]; pos++;
            } else {
                throw std::runtime_error("sEDS: Invalid character '" + std::string(1, input[pos]) +
                                         "' at position " + std::to_string(pos));
            }
        }
        if (!num.empty()) ids.insert(std::stoi(num));
        if (pos >= input.size() || input[pos] != '}')
            throw std::runtime_error("sEDS: Expected '}' at position " + std::to_string(pos));
        pos++;
        if (ids.empty())
            throw std::runtime_error("sEDS: Empty path set at string " + std::to_string(e.sources.size()));
        e.sources.push_back(std::move(ids));
    }
    if (e.sources.size() != m)
        throw std::runtime_error("sEDS: Source count (" + std::to_string(e.sources.size()) +
                                 ") does not match EDS cardinality (" + std::to_string(m) + ")");
    e.has_sources = true;
}

inline bool degenerate(const Eds& e, size_t i) { return e.sets[i].size() > 1; }

bool is_leds(const Eds& e, uint32_t l)                       // eds_transforms.cpp:439-468
{
    if (l == 0) return true;
    const size_t n = e.sets.size();
    for (size_t i = 0; i < n; i++) {
        if (!degenerate(e, i)) {
            size_t len = e.sets[i][0].size();
            if (i > 0 && i < n - 1 && len < l) return false;
        }
        if (i + 1 < n && degenerate(e, i) && degenerate(e, i + 1)) return false;
    }
    return true;
}

std::vector<size_t> select_pairs(const Eds& e, uint32_t l)   // eds_transforms.cpp:46-107
{
    std::vector<size_t> pairs;
    const size_t n = e.sets.size();
    if (n < 2) return pairs;
    std::vector<char> used(n, 0);
    for (size_t i = 0; i + 1 < n; ++i) {
        if (used[i] || used[i + 1]) continue;
        bool should = false;
        if (!degenerate(e, i) && i > 0 && i < n - 1 && e.sets[i][0].size() < l) should = true;
        if (!degenerate(e, i + 1) && (i + 1) < n - 1 && e.sets[i + 1][0].size() < l) should = true;
        if (degenerate(e, i) && degenerate(e, i + 1)) should = true;
        if (should) { pairs.push_back(i); used[i] = used[i + 1] = 1; }
    }
    return pairs;
}

std::set<int> intersect(const std::set<int>& a, const std::set<int>& b)   // eds.cpp:1481-1500
{
    std::set<int> r;
    bool ua = a.count(0) > 0, ub = b.count(0) > 0;
    if (ua && ub) r.insert(0);
    else if (ua) r = b;
    else if (ub) r = a;
    else std::set_intersection(a.begin(), a.end(), b.begin(), b.end(), std::inserter(r, r.begin()));
    return r;
}

Eds merge_round(const Eds& e, const std::vector<size_t>& pairs)   // merge_multiple_pairs + reconstruct_eds
{
    Eds out;
    out.has_sources = e.has_sources;
    std::vector<size_t> cum(e.sets.size() + 1, 0);
    for (size_t i = 0; i < e.sets.size(); i++) cum[i + 1] = cum[i] + e.sets[i].size();
    size_t pi = 0;
    for (size_t pos = 0; pos < e.sets.size(); pos++) {
        if (pi < pairs.size() && pairs[pi] == pos) {
            const StringSet& A = e.sets[pos];
            const StringSet& Bs = e.sets[pos + 1];
            StringSet merged;
            if (!e.has_sources) {                            // eds.cpp:1640-1644
                for (const auto& a : A) for (const auto& b : Bs) merged.push_back(a + b);
            } else {                                         // eds.cpp:1646-1676
                size_t kept = 0;
                for (size_t i = 0; i < A.size(); i++)
                    for (size_t j = 0; j < Bs.size(); j++) {
                        std::set<int> I = intersect(e.sources[cum[pos] + i], e.sources[cum[pos + 1] + j]);
                        if (!I.empty()) { merged.push_back(A[i] + Bs[j]); out.sources.push_back(std::move(I)); kept++; }
                    }
                if (kept == 0)                               // eds.cpp:1513-1519
                    throw std::runtime_error("Merging positions " + std::to_string(pos) + " and " +
                                             std::to_string(pos + 1) +
                                             " results in empty set (no valid source intersections)");
            }
            out.sets.push_back(std::move(merged));
            pos++;                                           // second position consumed
            pi++;
        } else {
            out.sets.push_back(e.sets[pos]);
            if (e.has_sources)
                for (size_t j = 0; j < e.sets[pos].size(); j++) out.sources.push_back(e.sources[cum[pos] + j]);
        }
    }
    return out;
}

char* dup_out(const std::string& s, size_t* n)
{
    char* p = static_cast<char*>(malloc(s.size() + 1));
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *n = s.size();
    return p;
}

} // namespace

extern "C" int oracle_merge(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n,
                            uint32_t l, int compact,
                            char** out, size_t* out_n, char** seds_out, size_t* seds_out_n,
                            char* err, size_t errcap)
{
    try {
        if (l == 0)                                          // eds_transforms.cpp:322-324 / :388-390
            throw std::invalid_argument("context_length must be > 0 for l-EDS transformation");
        Eds e;
        parse_eds(eds, eds_n, e);
        if (seds) parse_sources(seds, seds_n, e);
        size_t iteration = 0;
        const size_t MAX_ITERATIONS = 10000;                 // eds_transforms.cpp:335
        while (iteration < MAX_ITERATIONS) {
            if (is_leds(e, l)) break;
            std::vector<size_t> pairs = select_pairs(e, l);
            if (pairs.empty()) break;
            e = merge_round(e, pairs);
            iteration++;
        }
        if (iteration >= MAX_ITERATIONS)
            throw std::runtime_error("Maximum iterations reached without convergence");

        std::string text;                                    // eds.cpp:600-631
        for (size_t i = 0; i < e.sets.size(); i++) {
            bool br = !compact || degenerate(e, i);
            if (br) text.push_back('{');
            for (size_t j = 0; j < e.sets[i].size(); j++) {
                if (j) text.push_back(',');
                text += e.sets[i][j];
            }
            if (br) text.push_back('}');
        }
        text.push_back('\n');
        *out = dup_out(text, out_n);

        std::string st;                                      // eds.cpp:641-659
        if (e.has_sources) {
            for (const auto& s : e.sources) {
                st.push_back('{');
                bool first = true;
                for (int id : s) { if (!first) st.push_back(','); st += std::to_string(id); first = false; }
                st.push_back('}');
            }
            st.push_back('\n');
        }
        *seds_out = dup_out(st, seds_out_n);
        return 0;
    } catch (const std::exception& ex) {
        if (err && errcap) { strncpy(err, ex.what(), errcap - 1); err[errcap - 1] = 0; }
        return dynamic_cast<const std::invalid_argument*>(&ex) ? 3 : 2;
    }
}

/* One symbol range of a partitioned merge (multi-GPU tests): first / last symbol may be a sentinel shared with the
 * neighbouring range.  Intactness is tracked independently of the device code: every current symbol carries the
 * number of original symbols it covers. */
extern "C" int oracle_merge_range(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n,
                                  uint32_t l, int compact, int head_sentinel, int tail_sentinel,
                                  char** out, size_t* out_n, char** seds_out, size_t* seds_out_n,
                                  int* head_intact, int* tail_intact, char* err, size_t errcap)
{
    try {
        if (l == 0) throw std::invalid_argument("context_length must be > 0 for l-EDS transformation");
        Eds e;
        parse_eds(eds, eds_n, e);
        if (seds) parse_sources(seds, seds_n, e);
        const size_t head_len = e.sets.empty() ? 0 : e.sets[0][0].size();
        std::vector<size_t> span(e.sets.size(), 1);
        size_t iteration = 0;
        while (iteration < 10000) {
            if (is_leds(e, l)) break;
            std::vector<size_t> pairs = select_pairs(e, l);
            if (pairs.empty()) break;
            e = merge_round(e, pairs);
            std::vector<size_t> ns;
            size_t pi = 0;
            for (size_t pos = 0; pos < span.size(); pos++) {
                if (pi < pairs.size() && pairs[pi] == pos) { ns.push_back(span[pos] + span[pos + 1]); pos++; pi++; }
                else ns.push_back(span[pos]);
            }
            span.swap(ns);
            iteration++;
        }
        if (iteration >= 10000) throw std::runtime_error("Maximum iterations reached without convergence");
        *head_intact = !head_sentinel || (span.size() >= 2 && span.front() == 1);
        *tail_intact = !tail_sentinel || (span.size() >= 2 && span.back() == 1);
        std::string text, st;
        size_t first = (head_sentinel && *head_intact) ? 1 : 0, cum = 0;
        for (size_t i = 0; i < e.sets.size(); i++) {
            if (i >= first) {
                bool br = !compact || degenerate(e, i);
                if (br) text.push_back('{');
                for (size_t j = 0; j < e.sets[i].size(); j++) { if (j) text.push_back(','); text += e.sets[i][j]; }
                if (br) text.push_back('}');
                if (e.has_sources)
                    for (size_t j = 0; j < e.sets[i].size(); j++) {
                        st.push_back('{');
                        bool f = true;
                        for (int id : e.sources[cum + j]) { if (!f) st.push_back(','); st += std::to_string(id); f = false; }
                        st.push_back('}');
                    }
            }
            cum += e.sets[i].size();
        }
        (void)head_len;
        if (!tail_sentinel) { text.push_back('\n'); if (e.has_sources) st.push_back('\n'); }
        *out = dup_out(text, out_n);
        *seds_out = dup_out(st, seds_out_n);
        return 0;
    } catch (const std::exception& ex) {
        if (err && errcap) { strncpy(err, ex.what(), errcap - 1); err[errcap - 1] = 0; }
        return dynamic_cast<const std::invalid_argument*>(&ex) ? 3 : 2;
    }
}

// EDS::calculate_statistics eds.cpp:361-470, calculate_source_statistics :472-505, is_leds eds_transforms.cpp:439-468
extern "C" int oracle_eds_stats(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l,
                                oracle_eds_statistics* out, char* err, size_t errcap)
{
    try {
        memset(out, 0, sizeof(*out));
        Eds e;
        parse_eds(eds, eds_n, e);
        if (seds) parse_sources(seds, seds_n, e);
        out->has_sources = seds ? 1 : 0;
        out->is_leds = is_leds(e, l) ? 1 : 0;
        if (e.sets.empty()) return 0;                        // :362-376: everything 0
        uint64_t mn = UINT32_MAX, total_ctx = 0;
        out->n_symbols = e.sets.size();
        for (const auto& st : e.sets) {
            out->n_strings += st.size();
            if (st.size() > 1) { out->num_degenerate_symbols++; out->total_change_size += st.size() - 1; }   // :398-401
            else {                                           // :402-417: a context block
                const uint64_t len = st[0].size();
                if (len < mn) mn = len;
                if (len > out->max_context_length) out->max_context_length = len;
                total_ctx += len; out->num_context_blocks++; out->num_common_chars += len;
            }
            for (const auto& str : st) { out->n_chars += str.size(); if (str.empty()) out->num_empty_strings++; }   // :420-426
        }
        out->avg_context_length = out->num_context_blocks ? (double)total_ctx / (double)out->num_context_blocks : 0.0;   // :429-433
        out->min_context_length = mn == UINT32_MAX ? 0 : mn;                                                              // :436-438
        if (e.has_sources && !e.sources.empty()) {            // :478-503
            std::set<int> all;
            for (const auto& ss : e.sources) {
                if (ss.size() > out->max_paths_per_string) out->max_paths_per_string = ss.size();
                all.insert(ss.begin(), ss.end());
                out->total_paths += ss.size();
            }
            out->num_paths = all.size();
            out->avg_paths_per_string = (double)out->total_paths / (double)e.sources.size();
        }
        return 0;
    } catch (const std::exception& ex) {
        if (err && errcap) { strncpy(err, ex.what(), errcap - 1); err[errcap - 1] = 0; }
        return dynamic_cast<const std::invalid_argument*>(&ex) ? 3 : 2;
    }
}
