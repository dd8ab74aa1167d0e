/*
 * ref_driver.cpp — thin C entry points over the REAL reference library, for validating the
 * oracle restatement in the build container (TEST INFRASTRUCTURE ONLY).
 *
 * Compiled by oracle/Makefile together with the reference's own, unmodified sources where they
 * lie under /root/reference/src/cpp/lib (common.cpp, formats/eds.cpp, transforms/eds_transforms.cpp,
 * transforms/vcf_transforms.cpp — none of them includes a third-party header).  Output goes to
 * oracle/_ref/ (git-ignored).  transforms/msa_transforms.cpp is NOT part of this build: it includes
 * <sdsl/bit_vectors.hpp>, SDSL is absent from this image, and stand-ins are not allowed, so the
 * MSA path of the reference is unbuildable here (DESIGN.md "Oracle").
 */
#include "transforms/eds_transforms.hpp"
#include "transforms/vcf_transforms.hpp"
#include "formats/eds.hpp"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>

namespace {
char* dup_out(const std::string& s, size_t* n)
{
    char* p = static_cast<char*>(malloc(s.size() + 1));
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    *n = s.size();
    return p;
}
void set_err(char* err, size_t cap, const char* w) { if (err && cap) { strncpy(err, w, cap - 1); err[cap - 1] = 0; } }
}

extern "C" int ref_merge(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n,
                         uint32_t l, int compact, int threads,
                         char** out, size_t* out_n, char** seds_out, size_t* seds_out_n,
                         char* err, size_t errcap)
{
    try {
        std::istringstream in(std::string(reinterpret_cast<const char*>(eds), eds_n));
        std::ostringstream os, ss_out;
        if (seds) {
            std::istringstream sin(std::string(reinterpret_cast<const char*>(seds), seds_n));
            edsparser::eds_to_leds_linear(in, os, l, &sin, &ss_out, static_cast<size_t>(threads), compact != 0);
        } else {
            edsparser::eds_to_leds_cartesian(in, os, l, static_cast<size_t>(threads), compact != 0);
        }
        *out = dup_out(os.str(), out_n);
        *seds_out = dup_out(ss_out.str(), seds_out_n);
        return 0;
    } catch (const std::invalid_argument& ex) { set_err(err, errcap, ex.what()); return 3;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return 2; }
}

struct ref_vcf_stats { uint64_t total_variants, processed_variants, skipped_malformed, skipped_unsupported_sv, variant_groups; };

extern "C" int ref_vcf(const uint8_t* vcf, size_t vcf_n, const uint8_t* fasta, size_t fasta_n,
                       uint32_t l, char** eds, size_t* eds_n, char** seds, size_t* seds_n,
                       ref_vcf_stats* stats, char* err, size_t errcap)
{
    try {
        std::istringstream v(std::string(reinterpret_cast<const char*>(vcf), vcf_n));
        std::istringstream f(std::string(reinterpret_cast<const char*>(fasta), fasta_n));
        edsparser::VCFStats st;
        auto r = (l == 0) ? edsparser::parse_vcf_to_eds_streaming(v, f, &st)
                          : edsparser::parse_vcf_to_leds_streaming(v, f, l, &st);
        if (stats) {
            stats->total_variants = st.total_variants; stats->processed_variants = st.processed_variants;
            stats->skipped_malformed = st.skipped_malformed; stats->skipped_unsupported_sv = st.skipped_unsupported_sv;
            stats->variant_groups = st.variant_groups;
        }
        *eds = dup_out(r.first, eds_n);
        *seds = dup_out(r.second, seds_n);
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return 2; }
}

struct ref_eds_statistics {
    uint64_t n_symbols, n_chars, n_strings, num_degenerate_symbols, total_change_size, num_common_chars,
             num_empty_strings, min_context_length, max_context_length, num_context_blocks;
    double avg_context_length;
    uint64_t has_sources, num_paths, max_paths_per_string, total_paths;
    double avg_paths_per_string;
    int is_leds;
};

// EDS(eds[, seds]) -> get_statistics() + is_leds(l) of the real reference.  num_context_blocks and total_paths are not
// fields of EDS::Statistics: they are reported as 0 and the tests compare the averages instead.
extern "C" int ref_eds_stats(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l,
                             ref_eds_statistics* out, char* err, size_t errcap)
{
    try {
        memset(out, 0, sizeof(*out));
        const std::string es(reinterpret_cast<const char*>(eds), eds_n);
        edsparser::EDS e = seds ? edsparser::EDS(es, std::string(reinterpret_cast<const char*>(seds), seds_n)) : edsparser::EDS(es);
        out->is_leds = edsparser::is_leds(e, l) ? 1 : 0;
        if (e.empty()) return 0;   // an empty EDS never reaches calculate_statistics in the string constructors: its Statistics are uninitialised
        const auto st = e.get_statistics();
        out->n_symbols = e.length(); out->n_chars = e.size(); out->n_strings = e.cardinality();
        out->num_degenerate_symbols = st.num_degenerate_symbols; out->total_change_size = st.total_change_size;
        out->num_common_chars = st.num_common_chars; out->num_empty_strings = st.num_empty_strings;
        out->min_context_length = st.min_context_length; out->max_context_length = st.max_context_length;
        out->avg_context_length = st.avg_context_length;
        out->has_sources = e.has_sources() ? 1 : 0;
        if (e.has_sources()) {   // without sources the reference leaves these three fields uninitialised (eds.cpp:116 never runs :472)
            out->num_paths = st.num_paths;
            out->max_paths_per_string = st.max_paths_per_string; out->avg_paths_per_string = st.avg_paths_per_string;
        }
        out->is_leds = edsparser::is_leds(e, l) ? 1 : 0;
        return 0;
    } catch (const std::invalid_argument& ex) { set_err(err, errcap, ex.what()); return 3;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return 2; }
}

// which = 0: EDS::print_statistics, 1: EDS::print (eds.cpp:528-598) of the real reference, as text
extern "C" int ref_eds_print(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, int which,
                             char** out, size_t* out_n, char* err, size_t errcap)
{
    try {
        const std::string es(reinterpret_cast<const char*>(eds), eds_n);
        edsparser::EDS e = seds ? edsparser::EDS(es, std::string(reinterpret_cast<const char*>(seds), seds_n)) : edsparser::EDS(es);
        std::ostringstream os;
        if (which == 0) e.print_statistics(os); else e.print(os);
        *out = dup_out(os.str(), out_n);
        return 0;
    } catch (const std::exception& ex) { set_err(err, errcap, ex.what()); return 2; }
}

extern "C" void ref_free(void* p) { free(p); }
