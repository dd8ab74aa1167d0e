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

extern "C" void ref_free(void* p) { free(p); }
