/*
 * vcf_oracle.cpp — CPU restatement of the reference VCF(+FASTA) -> EDS / l-EDS transform.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Parity: pinned by the reference's
 * data/vcf/{small,test_overlaps}.{eds,seds} goldens and differentially against oracle/_ref
 * (the reference's vcf_transforms.cpp compiled in this container).
 *
 * Follows src/cpp/lib/transforms/vcf_transforms.cpp:
 *   parse_fasta_metadata :51-86      read_fasta_region :98-129
 *   parse_alt_field :142-176         parse_genotype :190-216      parse_vcf_line :232-326
 *   apply_variant_to_span :356-390   merge_variant_group :396-476
 *   group_overlapping_variants :482-534   generate_eds_from_variants :554-668
 *   parse_vcf_to_eds_streaming :677-729   parse_vcf_to_leds_streaming :735-755
 */
#include "oracle.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace {

struct Fasta {                         // FASTAMetadata :20-25 + the byte image
    const uint8_t* f; size_t n;
    size_t seq_size = 0, line_width = 0;
    int64_t seq_start = 0;
};

struct Variant {                       // VCFVariant :27-33
    std::string chrom; size_t pos = 0; std::string ref;
    std::vector<std::string> alts;
    std::vector<std::vector<int>> genotypes;
};

struct Group {                         // VariantGroup :35-41
    size_t start_pos, end_pos;
    std::vector<std::string> haps;
    std::vector<std::vector<int>> merged_genotypes;
};

// getline over a byte image; returns false when nothing could be extracted
bool next_line(const uint8_t* f, size_t n, size_t& pos, std::string& line, bool* hit_eof = nullptr)
{
    if (pos >= n) return false;
    const uint8_t* nl = static_cast<const uint8_t*>(memchr(f + pos, '\n', n - pos));
    size_t end = nl ? static_cast<size_t>(nl - f) : n;
    line.assign(reinterpret_cast<const char*>(f + pos), end - pos);
    if (hit_eof) *hit_eof = (nl == nullptr);
    pos = nl ? end + 1 : n;
    return true;
}

Fasta parse_fasta(const uint8_t* f, size_t n)                // :51-86
{
    Fasta m{f, n};
    size_t pos = 0;
    std::string line;
    bool eof = false;
    if (!next_line(f, n, pos, line, &eof) || line.empty() || line[0] != '>')
        throw std::runtime_error("Invalid FASTA format: expected header line starting with '>'");
    m.seq_start = eof ? -1 : static_cast<int64_t>(pos);
    if (!next_line(f, n, pos, line)) throw std::runtime_error("FASTA file is empty");
    m.line_width = line.size();
    m.seq_size = line.size();
    while (next_line(f, n, pos, line)) {
        if (line.empty()) continue;
        if (line[0] == '>') break;
        m.seq_size += line.size();
    }
    return m;
}

std::string read_region(const Fasta& m, size_t start, size_t length)      // :98-129
{
    if (start >= m.seq_size) return "";
    if (start + length > m.seq_size) length = m.seq_size - start;
    std::string r;
    int64_t off = m.seq_start + static_cast<int64_t>(start + (start / m.line_width));
    size_t p = off < 0 ? m.n : static_cast<size_t>(off);
    size_t got = 0;
    while (got < length && p < m.n) {
        char c = static_cast<char>(m.f[p++]);
        if (c != '\n' && c != '\r') { r.push_back(c); got++; }
    }
    return r;
}

struct UnsupportedSV : std::runtime_error { using std::runtime_error::runtime_error; };

std::vector<std::string> parse_alt(const std::string& alt_field, const std::string& ref)   // :142-176
{
    std::vector<std::string> alts;
    std::stringstream ss(alt_field);
    std::string a;
    while (std::getline(ss, a, ',')) {
        if (!a.empty() && a[0] == '<' && a[a.size() - 1] == '>') {
            std::string sv = a.substr(1, a.size() - 2);
            if (sv == "DEL") alts.push_back("");
            else if (sv == "INS") alts.push_back(ref);
            else throw UnsupportedSV("Unsupported structural variant type: " + sv);
        } else alts.push_back(a);
    }
    return alts;
}

std::vector<int> parse_gt(const std::string& gt)             // :190-216
{
    std::vector<int> alleles;
    char delim = gt.find('/') != std::string::npos ? '/' : '|';
    std::stringstream ss(gt);
    std::string a;
    while (std::getline(ss, a, delim)) {
        if (a == ".") continue;
        try { alleles.push_back(std::stoi(a)); } catch (...) { continue; }
    }
    return alleles;
}

enum class Skip { NONE, HEADER, MALFORMED, UNSUPPORTED_SV };

bool parse_line(const std::string& line, size_t& n_samples, Skip& skip, Variant& var)   // :232-326
{
    skip = Skip::NONE;
    if (line.empty() || line[0] == '#') {
        if (line.substr(0, 6) == "#CHROM") {
            std::stringstream ss(line);
            std::string tok; size_t cols = 0;
            while (ss >> tok) cols++;
            if (cols > 9) n_samples = cols - 9;
        }
        skip = Skip::HEADER;
        return false;
    }
    std::vector<std::string> fields;
    {
        std::stringstream ss(line);
        std::string tok;
        while (std::getline(ss, tok, '\t')) if (!tok.empty()) fields.push_back(tok);
        if (fields.size() < 5) {
            fields.clear();
            std::stringstream ws(line);
            while (ws >> tok) fields.push_back(tok);
        }
    }
    if (fields.size() < 5) { skip = Skip::MALFORMED; return false; }
    var = Variant();
    var.chrom = fields[0];
    try { var.pos = std::stoull(fields[1]); } catch (...) { skip = Skip::MALFORMED; return false; }
    var.ref = fields[3];
    try { var.alts = parse_alt(fields[4], var.ref); }
    catch (const UnsupportedSV&) { skip = Skip::UNSUPPORTED_SV; return false; }
    if (fields.size() >= 10) {
        for (size_t i = 9; i < fields.size(); i++) {
            std::string gt = fields[i];
            size_t c = gt.find(':');
            if (c != std::string::npos) gt = gt.substr(0, c);
            var.genotypes.push_back(parse_gt(gt));
        }
    }
    return true;
}

std::string apply_variant(const std::string& span, size_t span_start, const Variant& v, int alt_index)  // :356-390
{
    if (alt_index == 0) return span;
    if (alt_index < 1 || alt_index > static_cast<int>(v.alts.size())) return span;
    const std::string& alt = v.alts[alt_index - 1];
    size_t off = (v.pos - 1) - span_start;
    std::string r = span.substr(0, off);                     // throws out_of_range like the reference
    r += alt;
    size_t after = off + v.ref.size();
    if (after < span.size()) r += span.substr(after);
    return r;
}

Group merge_group(const std::vector<const Variant*>& gv, const std::string& span, size_t start)   // :396-476
{
    Group g;
    g.start_pos = start;
    g.end_pos = start + span.size();
    size_t n_samples = gv.empty() ? 0 : gv[0]->genotypes.size();
    g.merged_genotypes.resize(n_samples);
    std::map<std::string, int> idx;
    g.haps.push_back(span);
    idx[span] = 0;
    for (const Variant* v : gv)
        for (size_t a = 0; a < v->alts.size(); a++) {
            std::string h = apply_variant(span, start, *v, static_cast<int>(a) + 1);
            if (idx.find(h) == idx.end()) { idx[h] = static_cast<int>(g.haps.size()); g.haps.push_back(h); }
        }
    for (size_t s = 0; s < n_samples; s++) {
        std::set<int> hs;
        for (const Variant* v : gv) {
            if (s >= v->genotypes.size()) continue;
            for (int allele : v->genotypes[s]) {
                std::string h = apply_variant(span, start, *v, allele);
                auto it = idx.find(h);
                if (it != idx.end()) hs.insert(it->second);
            }
        }
        if (hs.empty()) hs.insert(0);
        g.merged_genotypes[s].assign(hs.begin(), hs.end());
    }
    return g;
}

std::vector<Group> group_variants(const std::vector<Variant>& vars, const Fasta& fa)     // :482-534
{
    std::vector<Group> groups;
    size_t i = 0;
    while (i < vars.size()) {
        std::vector<const Variant*> cur{&vars[i]};
        size_t gs = vars[i].pos - 1;
        size_t ge = gs + vars[i].ref.size();
        size_t j = i + 1;
        while (j < vars.size()) {
            size_t ns = vars[j].pos - 1, ne = ns + vars[j].ref.size();
            if (ns < ge) { cur.push_back(&vars[j]); ge = std::max(ge, ne); j++; }
            else break;
        }
        std::string span = read_region(fa, gs, ge - gs);
        groups.push_back(merge_group(cur, span, gs));
        i = j;
    }
    return groups;
}

// cur0 / next_start: one position range of a partitioned run (multi-GPU tests): the walk starts at cur0 and the
// closing common text is what the reference flushes in front of the next range's first group (:570-578);
// next_start == SIZE_MAX is the whole-file behaviour (:658-665)
void generate(const Fasta& fa, const std::vector<Group>& groups, std::string& eds, std::string& seds,
              size_t cur0 = 0, size_t next_start = SIZE_MAX)   // :554-668
{
    size_t cur = cur0;
    for (const Group& g : groups) {
        if (g.start_pos > cur) {
            std::string r = read_region(fa, cur, g.start_pos - cur);
            if (!r.empty()) { eds += '{'; eds += r; eds += '}'; seds += "{0}"; }
            cur = g.start_pos;
        }
        eds += '{';
        std::map<std::string, std::set<int>> h2s;
        for (size_t s = 0; s < g.merged_genotypes.size(); s++)
            for (int hi : g.merged_genotypes[s])
                if (hi >= 0 && hi < static_cast<int>(g.haps.size())) h2s[g.haps[hi]].insert(static_cast<int>(s) + 1);
        if (h2s.empty()) {                                   // :603-615
            for (size_t i = 0; i < g.haps.size(); i++) { eds += g.haps[i]; if (i + 1 < g.haps.size()) eds += ','; }
            eds += '}';
            seds += "{0}";
            cur = g.end_pos;
            continue;
        }
        std::vector<const std::string*> ordered;             // :619-624
        for (const auto& h : g.haps) if (h2s.find(h) != h2s.end()) ordered.push_back(&h);
        for (size_t i = 0; i < ordered.size(); i++) { eds += *ordered[i]; if (i + 1 < ordered.size()) eds += ','; }
        eds += '}';
        for (const std::string* h : ordered) {               // :636-651
            const std::set<int>& ss = h2s[*h];
            seds += '{';
            size_t k = 0;
            for (int id : ss) { seds += std::to_string(id); if (++k < ss.size()) seds += ','; }
            seds += '}';
        }
        cur = g.end_pos;
    }
    if (next_start != SIZE_MAX) {
        if (next_start > cur) {
            std::string r = read_region(fa, cur, next_start - cur);
            if (!r.empty()) { eds += '{'; eds += r; eds += '}'; seds += "{0}"; }
        }
        return;
    }
    if (cur < fa.seq_size) {                                 // :658-665
        std::string r = read_region(fa, cur, fa.seq_size - cur);
        if (!r.empty()) { eds += '{'; eds += r; eds += '}'; seds += "{0}"; }
    }
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

extern "C" int oracle_vcf(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n,
                          uint32_t l, char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                          oracle_vcf_stats* stats, char* err, size_t errcap)
{
    try {
        Fasta fa = parse_fasta(fasta, fasta_n);
        std::vector<Variant> vars;
        size_t n_samples = 0, pos = 0;
        std::string line;
        oracle_vcf_stats st{};
        while (next_line(vcf, vcf_n, pos, line)) {           // :690-712
            Skip skip; Variant v;
            bool ok = parse_line(line, n_samples, skip, v);
            if (skip == Skip::NONE) { st.total_variants++; st.processed_variants++; }
            else if (skip == Skip::MALFORMED) { st.total_variants++; st.skipped_malformed++; }
            else if (skip == Skip::UNSUPPORTED_SV) { st.total_variants++; st.skipped_unsupported_sv++; }
            if (ok) vars.push_back(std::move(v));
        }
        std::sort(vars.begin(), vars.end(),                  // :715-718 (same libstdc++ sort, same length)
                  [](const Variant& a, const Variant& b) { return a.pos < b.pos; });
        std::vector<Group> groups = group_variants(vars, fa);
        std::string e, s;
        generate(fa, groups, e, s);
        st.variant_groups = groups.size();                   // :724-726
        if (stats) *stats = st;
        if (l == 0) {
            *eds = dup_out(e, eds_n);
            *seds = dup_out(s, seds_n);
            return 0;
        }
        // :735-755 — EDS text fed to the LINEAR merge with defaults (1 thread, COMPACT)
        char *lo = nullptr, *so = nullptr; size_t ln = 0, sn = 0;
        int rc = oracle_merge(reinterpret_cast<const uint8_t*>(e.data()), e.size(),
                              reinterpret_cast<const uint8_t*>(s.data()), s.size(), l, 1,
                              &lo, &ln, &so, &sn, err, errcap);
        if (rc) return rc;
        *eds = lo; *eds_n = ln; *seds = so; *seds_n = sn;
        return 0;
    } catch (const std::exception& ex) {
        if (err && errcap) { strncpy(err, ex.what(), errcap - 1); err[errcap - 1] = 0; }
        return 2;
    }
}

/* One position range of a partitioned run: records already in final order (no sort). */
extern "C" int oracle_vcf_range(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n,
                                uint64_t cur0, uint64_t next_start, char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                                oracle_vcf_stats* stats, char* err, size_t errcap)
{
    try {
        Fasta fa = parse_fasta(fasta, fasta_n);
        std::vector<Variant> vars;
        size_t n_samples = 0, pos = 0;
        std::string line;
        oracle_vcf_stats st{};
        while (next_line(vcf, vcf_n, pos, line)) {
            Skip skip; Variant v;
            bool ok = parse_line(line, n_samples, skip, v);
            if (skip == Skip::NONE) { st.total_variants++; st.processed_variants++; }
            else if (skip == Skip::MALFORMED) { st.total_variants++; st.skipped_malformed++; }
            else if (skip == Skip::UNSUPPORTED_SV) { st.total_variants++; st.skipped_unsupported_sv++; }
            if (ok) vars.push_back(std::move(v));
        }
        std::vector<Group> groups = group_variants(vars, fa);
        std::string e, s;
        generate(fa, groups, e, s, cur0, next_start == UINT64_MAX ? SIZE_MAX : (size_t)next_start);
        st.variant_groups = groups.size();
        if (stats) *stats = st;
        *eds = dup_out(e, eds_n);
        *seds = dup_out(s, seds_n);
        return 0;
    } catch (const std::exception& ex) {
        if (err && errcap) { strncpy(err, ex.what(), errcap - 1); err[errcap - 1] = 0; }
        return 2;
    }
}

/* Index pass (positions, REF lengths, line spans of the accepted records) and the std::sort permutation. */
extern "C" int oracle_vcf_index(const uint8_t* vcf, size_t vcf_n, uint64_t* pos_out, uint64_t* reflen_out,
                                uint64_t* off_out, uint64_t* len_out, size_t cap, size_t* n_out, oracle_vcf_stats* stats)
{
    size_t n_samples = 0, pos = 0, n = 0;
    std::string line;
    oracle_vcf_stats st{};
    while (true) {
        const size_t lo = pos;
        if (!next_line(vcf, vcf_n, pos, line)) break;
        Skip skip; Variant v;
        bool ok = false;
        try { ok = parse_line(line, n_samples, skip, v); } catch (...) { return 2; }
        if (skip == Skip::NONE) { st.total_variants++; st.processed_variants++; }
        else if (skip == Skip::MALFORMED) { st.total_variants++; st.skipped_malformed++; }
        else if (skip == Skip::UNSUPPORTED_SV) { st.total_variants++; st.skipped_unsupported_sv++; }
        if (ok) {
            if (n < cap) { pos_out[n] = v.pos; reflen_out[n] = v.ref.size(); off_out[n] = lo; len_out[n] = line.size(); }
            n++;
        }
    }
    *n_out = n;
    if (stats) *stats = st;
    return n <= cap ? 0 : 1;
}

extern "C" void oracle_vcf_sort_order(const uint64_t* pos, size_t n, uint32_t* order_out)
{
    std::vector<std::pair<uint64_t, uint32_t>> o(n);
    for (size_t i = 0; i < n; i++) o[i] = {pos[i], (uint32_t)i};
    std::sort(o.begin(), o.end(), [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) { return a.first < b.first; });
    for (size_t i = 0; i < n; i++) order_out[i] = o[i].second;
}
