// GPU test of the drop-in C++ API: same inputs and expected strings as the reference's
// tests/cpp/test_msa.cpp (:20-60, :63-103, :106-138, :141-171, :174-201, :204-229), through
// edsparser::parse_msa_to_{eds,leds}_streaming, plus the merge/VCF entry points when built.
#include "edsparser/transforms/eds_transforms.hpp"
#include "edsparser/transforms/msa_transforms.hpp"
#include "edsparser/transforms/vcf_transforms.hpp"

#include <cstdio>
#include <sstream>

using namespace edsparser;

static int failures = 0;
static void expect(const std::string& what, const std::string& got, const std::string& want)
{
    if (got != want) { std::printf("FAIL %s\n  got:      '%s'\n  expected: '%s'\n", what.c_str(), got.c_str(), want.c_str()); failures++; }
}

int main(int argc, char** argv)
{
    const bool with_merge = argc > 1 && std::string(argv[1]) == "all";
    const std::string small = ">seq1\nAGTC--TCTATA\n>seq2\nAGTCCCTATATA\n>seq3\nAGTC--TATATA\n";
    try {
        { std::istringstream in(small); auto r = parse_msa_to_eds_streaming(in);
          expect("eds", r.first, "{AGTC}{,CC}{T}{C,A}{TATA}"); expect("seds", r.second, "{0}{1,3}{2}{0}{1}{2,3}{0}"); }
        { std::istringstream in(small); auto r = parse_msa_to_leds_streaming(in, 4);
          expect("leds l=4", r.first, "{AGTC}{TC,CCTA,TA}{TATA}"); expect("seds l=4", r.second, "{0}{1}{2}{3}{0}"); }
        { std::istringstream in(">seq1\nAGTCTA\n>seq2\nAGTCTA\n>seq3\nAGTCTA\n"); auto r = parse_msa_to_eds_streaming(in);
          expect("identical", r.first, "{AGTCTA}"); expect("identical seds", r.second, "{0}"); }
        { std::istringstream in(">seq1\nAGTC\n>seq2\nAGCC\n"); auto r = parse_msa_to_eds_streaming(in);
          expect("single variant", r.first, "{AG}{T,C}{C}"); expect("single variant seds", r.second, "{0}{1}{2}{0}"); }
        { std::istringstream in(">seq1\n--AGTC\n>seq2\nCCAGTC\n"); auto r = parse_msa_to_eds_streaming(in);
          expect("gap at beginning", r.first, "{,CC}{AGTC}"); expect("gap at beginning seds", r.second, "{1}{2}{0}"); }
        { std::istringstream in(">seq1\nAGTC--\n>seq2\nAGTCGG\n"); auto r = parse_msa_to_eds_streaming(in);
          expect("gap at end", r.first, "{AGTC}{,GG}"); expect("gap at end seds", r.second, "{0}{1}{2}"); }
        { std::istringstream in(">only\nACGT\n"); bool threw = false;
          try { parse_msa_to_eds_streaming(in); } catch (const std::runtime_error&) { threw = true; }
          if (!threw) { std::printf("FAIL single-sequence MSA must throw\n"); failures++; } }
        if (with_merge) {
            { std::istringstream in("{AAAA}{A,T}{CG}{G,C}{TTTT}"), src("{0}{1}{2}{0}{1}{2}{0}"); std::ostringstream out, so;
              eds_to_leds_linear(in, out, 4, &src, &so);
              expect("linear", out.str(), "AAAA{ACGG,TCGC}TTTT\n"); expect("linear seds", so.str(), "{0}{1}{2}{0}\n"); }
            { std::istringstream in("{G,T}{X}{AAAAAAAAAA}{C}{G,T}{LLLLLLLLLL}"); std::ostringstream out;
              eds_to_leds_cartesian(in, out, 5);
              expect("cartesian", out.str(), "{GX,TX}AAAAAAAAAAC{G,T}LLLLLLLLLL\n"); }
            { std::istringstream in("{A}"); std::ostringstream out; bool threw = false;
              try { eds_to_leds_cartesian(in, out, 0); } catch (const std::invalid_argument&) { threw = true; }
              if (!threw) { std::printf("FAIL l=0 must throw invalid_argument\n"); failures++; } }
            { std::istringstream vcf("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\tS2\n"
                                     "chr1\t3\t.\tG\tT\t.\t.\t.\tGT\t0|1\t0|0\nchr1\t4\t.\tT\tTA\t.\t.\t.\tGT\t1|1\t0|1\n"),
                                 fa(">chr1\nACGTACGTAC\nGTACGTACGT\n");
              VCFStats st; auto r = parse_vcf_to_eds_streaming(vcf, fa, &st);
              expect("vcf eds", r.first, "{AC}{G,T}{T,TA}{ACGTACGTACGTACGT}"); expect("vcf seds", r.second, "{0}{1,2}{1}{2}{1,2}{0}");
              if (st.variant_groups != 2 || st.processed_variants != 2) { std::printf("FAIL vcf stats\n"); failures++; } }
        }
    } catch (const std::exception& e) {
        std::printf("FAIL exception: %s\n", e.what());
        return 1;
    }
    if (failures) return 1;
    std::printf("C++ API tests passed\n");
    return 0;
}
