// Host-only test of the edsparser::EDS container (no GPU): the vectors are the expectations of the
// reference's tests/cpp/test_merge.cpp (:22-38, :86-110, :166-256), test_eds.cpp (whitespace /
// compact parsing :117-132, :461-533) and test_sources.cpp (error cases).
#include "edsparser/formats/eds.hpp"
#include "edsparser/transforms/eds_transforms.hpp"

#include <cstdio>
#include <fstream>
#include <sstream>
#include <stdexcept>

using namespace edsparser;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } } while (0)

template <class F> static std::string what_of(F&& f)
{
    try { f(); } catch (const std::exception& e) { return e.what(); }
    return "";
}

// print_statistics / print against the reference's own text (tests/golden/print_cases.txt, written by
// tests/golden/make_golden3.py from the compiled reference; format: see there)
static void check_print_cases(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    CHECK(f.good());
    std::string eds, seds, line;
    int ncases = 0;
    auto block = [&]() {
        std::getline(f, line);
        std::string b((size_t)std::stoul(line), '\0');
        f.read(&b[0], (std::streamsize)b.size());
        f.get();                                   // the newline behind the block
        return b;
    };
    while (std::getline(f, eds) && std::getline(f, seds)) {
        const std::string want_stats = block(), want_print = block();
        EDS e = seds == "-" ? EDS(eds) : EDS(eds, seds);
        std::ostringstream a, b;
        e.print_statistics(a);
        e.print(b);
        CHECK(a.str() == want_stats);
        CHECK(b.str() == want_print);
        const EDS::Statistics st = e.get_statistics();
        CHECK(st.min_context_length == e.get_metadata().min_context_length && st.num_paths == e.get_metadata().num_paths);
        ncases++;
    }
    CHECK(ncases >= 6);
}

int main(int argc, char** argv)
{
    if (argc > 1) check_print_cases(argv[1]);
    {   // cartesian merges
        EDS eds("{G,C}{T}");
        EDS m = eds.merge_adjacent(0, 1);
        CHECK(m.length() == 1 && m.cardinality() == 2);
        CHECK(m.get_sets()[0][0] == "GT" && m.get_sets()[0][1] == "CT");
        EDS e3("{G,C}{T}{A,C}");
        EDS s2 = e3.merge_adjacent(0, 1).merge_adjacent(0, 1);
        CHECK(s2.cardinality() == 4);
        CHECK(s2.get_sets()[0][0] == "GTA" && s2.get_sets()[0][1] == "GTC" && s2.get_sets()[0][2] == "CTA" &&
              s2.get_sets()[0][3] == "CTC");
        EDS ee("{,A}{T}");
        EDS me = ee.merge_adjacent(0, 1);
        CHECK(me.get_sets()[0][0] == "T" && me.get_sets()[0][1] == "AT");
    }
    {   // linear merges
        EDS a(std::string("{G,C}{T}"), std::string("{1,2}{2,3}{2}"));
        EDS m = a.merge_adjacent(0, 1);
        CHECK(m.cardinality() == 2 && m.has_sources());
        CHECK(m.get_sources()[0] == std::set<int>{2} && m.get_sources()[1] == std::set<int>{2});
        EDS b(std::string("{A,B}{C,D}"), std::string("{1}{2}{1}{3}"));
        EDS mb = b.merge_adjacent(0, 1);
        CHECK(mb.cardinality() == 1 && mb.get_sets()[0][0] == "AC" && mb.get_sources()[0] == std::set<int>{1});
        EDS c(std::string("{A,B}{C}"), std::string("{0}{2}{1}"));
        EDS mc = c.merge_adjacent(0, 1);
        CHECK(mc.cardinality() == 1 && mc.get_sets()[0][0] == "AC" && mc.get_sources()[0] == std::set<int>{1});
        EDS d(std::string("{A}{B}"), std::string("{0}{0}"));
        CHECK(d.merge_adjacent(0, 1).get_sources()[0] == std::set<int>{0});
        EDS e(std::string("{A,B}{C,D}"), std::string("{1}{2}{3}{4}"));
        CHECK(what_of([&] { e.merge_adjacent(0, 1); }) ==
              "Merging positions 0 and 1 results in empty set (no valid source intersections)");
        CHECK(!what_of([&] { e.merge_adjacent(0, 2); }).empty());
        CHECK(!what_of([&] { e.merge_adjacent(1, 2); }).empty());
    }
    {   // parsing, whitespace, compact <-> full
        EDS w("{AC GT}\n{A,\tC}");
        CHECK(w.length() == 2 && w.get_sets()[0][0] == "ACGT");
        EDS c("ACGT{A,ACA}CGT");
        CHECK(c.length() == 3 && c.cardinality() == 4 && c.size() == 11);
        std::ostringstream full, compact;
        c.save(full, EDS::OutputFormat::FULL);
        c.save(compact, EDS::OutputFormat::COMPACT);
        CHECK(full.str() == "{ACGT}{A,ACA}{CGT}\n");
        CHECK(compact.str() == "ACGT{A,ACA}CGT\n");
        CHECK(what_of([] { EDS bad("{A,C"); }) == "Expected '}' at position 4");
        EDS empty("");
        CHECK(empty.empty() && empty.length() == 0);
        std::ostringstream es;
        empty.save(es);
        CHECK(es.str() == "\n");
    }
    {   // sources
        EDS s(std::string("{ACGT}{A,ACA}{CGT}{T,TG}"), std::string("{0}{1,3}{2}{0}{1}{2,3}"));
        std::ostringstream os;
        s.save_sources(os);
        CHECK(os.str() == "{0}{1,3}{2}{0}{1}{2,3}\n");
        CHECK(what_of([] { EDS x(std::string("{A,C}"), std::string("{1}")); }) ==
              "sEDS: Source count (1) does not match EDS cardinality (2)");
        CHECK(what_of([] { EDS x(std::string("{A}"), std::string("{}")); }) == "sEDS: Empty path set at string 0");
        CHECK(what_of([] { EDS x(std::string("{A}"), std::string("{1")); }) == "sEDS: Expected '}' at position 2");
        CHECK(what_of([] { EDS x(std::string("{A}"), std::string("{-1}")); }).find("Invalid character") != std::string::npos);
    }
    {   // is_leds
        CHECK(is_leds(EDS("{ACGT}{A,C}{ACGT}"), 4));
        CHECK(!is_leds(EDS("{ACGT}{A,C}{AC}{G,T}{ACGT}"), 4));
        CHECK(!is_leds(EDS("{ACGT}{A,C}{G,T}{ACGT}"), 1));
        CHECK(is_leds(EDS("{A}{A,C}{AC}"), 4));   // edges are exempt
        CHECK(is_leds(EDS("{A,C}{G,T}"), 0));
    }
    if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
    std::printf("container tests passed\n");
    return 0;
}
