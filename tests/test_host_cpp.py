"""The C++ host layer: container tests run anywhere, API + CLI tests need the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "edsparser_amd", "host")
BUILD = os.path.join(HOST, "build")
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "edsparser_amd")


def _build_host():
    import edsparser_amd.build as b
    b.build()
    subprocess.run(["make", "-s", "-C", HOST], check=True)


def _compile(src, out):
    _build_host()
    cmd = ["g++", "-std=c++17", "-O1", "-I", INC, src, os.path.join(BUILD, "libedsparser_lib.a"),
           "-L", LIBDIR, "-ledsx", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    subprocess.run(cmd, check=True)
    return out


def test_container_cpu():
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "test_container.cpp"), os.path.join(BUILD, "test_container"))
    # + EDS::print_statistics / print against the reference's own text (tests/golden/make_golden3.py)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "print_cases.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_cli_argument_errors_cpu(tmp_path):
    _build_host()
    r = subprocess.run([os.path.join(BUILD, "msa2eds")], capture_output=True, text=True)
    assert r.returncode == 1 and "the option '--input' is required but missing" in r.stderr
    assert "[Performance] Runtime:" in r.stderr
    bad = tmp_path / "x.txt"
    bad.write_text(">a\nAC\n>b\nAC\n")
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Input file must be an MSA file (.msa)" in r.stderr
    eds = tmp_path / "x.eds"
    eds.write_text("{A}")
    r = subprocess.run([os.path.join(BUILD, "eds2leds"), "-i", str(eds), "-l", "0"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Context length must be > 0" in r.stderr
    r = subprocess.run([os.path.join(BUILD, "eds2leds"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--context-length" in r.stdout


@pytest.mark.gpu
def test_cpp_api_gpu():
    exe = _compile(os.path.join(ROOT, "tests", "cpp", "test_msa_api.cpp"), os.path.join(BUILD, "test_msa_api"))
    r = subprocess.run([exe, "all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_msa2eds_cli_gpu(tmp_path):
    _build_host()
    msa = tmp_path / "small.msa"
    msa.write_bytes(open(os.path.join(ROOT, "tests", "golden", "ref_data", "msa", "small.msa"), "rb").read())
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(msa)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "MSA → EDS transformation" in r.stdout and "Transformation complete!" in r.stdout
    assert (tmp_path / "small.eds").read_text() == "{AGTC}{,CC}{T}{C,A}{TATAAAT}{AA,GG}{ATA}{,GGGG}"
    assert (tmp_path / "small.seds").read_text() == "{0}{1,3}{2}{0}{1}{2,3}{0}{1,2}{3}{0}{1,3}{2}"
    r = subprocess.run([os.path.join(BUILD, "msa2eds"), "-i", str(msa), "-l", "4"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "small_l4.leds").read_text() == "{AGTC}{TC,CCTA,TA}{TATAAAT}{AAATA,AAATAGGGG,GGATA}"
    assert (tmp_path / "small_l4.seds").read_text() == "{0}{1}{2}{3}{0}{1}{2}{3}"


@pytest.mark.gpu
def test_vcf2eds_and_eds2leds_cli_gpu(tmp_path):
    _build_host()
    g = os.path.join(ROOT, "tests", "golden", "ref_data")
    for f in ("vcf/small.vcf", "vcf/small.fa", "eds/test_iterative.eds"):
        (tmp_path / os.path.basename(f)).write_bytes(open(os.path.join(g, f), "rb").read())
    r = subprocess.run([os.path.join(BUILD, "vcf2eds"), "-i", str(tmp_path / "small.vcf"), "-r", str(tmp_path / "small.fa")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "small.eds").read_bytes() == open(os.path.join(g, "vcf/small.eds"), "rb").read()
    assert (tmp_path / "small.seds").read_bytes() == open(os.path.join(g, "vcf/small.seds"), "rb").read()
    assert "Variant groups created:     10" in r.stdout and "Success rate:               100.0%" in r.stdout
    r = subprocess.run([os.path.join(BUILD, "eds2leds"), "-i", str(tmp_path / "test_iterative.eds"), "-l", "4"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "test_iterative_l4.leds").read_bytes() == open(os.path.join(g, "eds/test_iterative_l4.eds"), "rb").read()
    assert "Output mode: compact" in r.stdout and "Threads: 1 (sequential)" in r.stdout


@pytest.mark.gpu
def test_edsparser_stats_cli_gpu(tmp_path):
    """edsparser-stats: the numbers of the text and JSON reports against the oracle's EDS::Statistics restatement
    (itself pinned by the reference's numbers in tests/golden/gen2_stats.json)."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as o
    _build_host()
    eds = b"{ACGTAC}{A,ACA,}{CGTTTTT}{,T}{GG}{C,G}{TTTTTTTTTT}"
    seds = b"{0}{1,3}{2}{4}{0}{1,2}{3,4}{0}{1,2,3}{4}{0}"
    (tmp_path / "x.eds").write_bytes(eds)
    (tmp_path / "x.seds").write_bytes(seds)
    exe = os.path.join(BUILD, "edsparser-stats")
    want = o.eds_stats(eds, seds, 0)
    r = subprocess.run([exe, "-i", str(tmp_path / "x.eds"), "-s", str(tmp_path / "x.seds"), "--json"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    assert j["structure"] == {"n_symbols": want["n_symbols"], "N_characters": want["n_chars"], "m_strings": want["n_strings"],
                              "degenerate_symbols": want["num_degenerate_symbols"],
                              "regular_symbols": want["n_symbols"] - want["num_degenerate_symbols"]}
    assert j["context_lengths"]["min"] == want["min_context_length"] and j["context_lengths"]["max"] == want["max_context_length"]
    assert abs(j["context_lengths"]["avg"] - want["avg_context_length"]) < 0.006
    assert j["variations"] == {"total_change_size": want["total_change_size"], "common_characters": want["num_common_chars"],
                               "empty_strings": want["num_empty_strings"]}
    assert j["sources"]["loaded"] is True and j["sources"]["num_paths"] == want["num_paths"]
    assert j["sources"]["max_paths_per_string"] == want["max_paths_per_string"]
    assert j["recommendations"]["needs_transformation"] is True and j["file"]["storage_mode"] == "METADATA_ONLY"
    r = subprocess.run([exe, "-i", str(tmp_path / "x.eds"), "--full", "--verbose"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Storage Mode: FULL (all data in RAM)" in r.stdout and "Detailed Metrics:" in r.stdout
    assert "  Number of symbols (n):        %12d" % want["n_symbols"] in r.stdout
    assert "  Minimum:                      %12d" % want["min_context_length"] in r.stdout
    assert "Sources (pangenome paths):" not in r.stdout and "[Performance] Runtime:" in r.stderr
    r = subprocess.run([exe, "-i", str(tmp_path / "missing.eds")], capture_output=True, text=True)
    assert r.returncode == 1 and "not found" in r.stderr
    (tmp_path / "bad.eds").write_bytes(b"{A,C}{G")
    r = subprocess.run([exe, "-i", str(tmp_path / "bad.eds")], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: Expected '}'" in r.stderr


def test_cmake_package_cpu(tmp_path):
    """The installed CMake package: find_package(EDSParser) + EDSParser::EDSParser, as a user of the reference writes it
    (README "target_link_libraries(your_target EDSParser::EDSParser)"); the consumer only touches the host container."""
    import shutil
    if not shutil.which("cmake"):
        pytest.skip("cmake not installed")
    _build_host()
    b, prefix = tmp_path / "b", tmp_path / "prefix"
    for cmd in (["cmake", "-S", HOST, "-B", str(b), "-DCMAKE_INSTALL_PREFIX=" + str(prefix)],
                ["cmake", "--build", str(b), "-j4"], ["cmake", "--install", str(b)]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for f in ("lib/libedsparser_lib.a", "lib/libedsx.so", "lib/cmake/EDSParser/EDSParserConfig.cmake", "bin/msa2eds",
              "bin/edsparser-stats", "include/edsparser/transforms/msa_transforms.hpp", "include/edsx.h"):
        assert (prefix / f).exists(), f
    user = tmp_path / "user"
    user.mkdir()
    (user / "CMakeLists.txt").write_text(
        "cmake_minimum_required(VERSION 3.16)\nproject(user CXX)\nfind_package(EDSParser REQUIRED)\n"
        "add_executable(user main.cpp)\ntarget_link_libraries(user EDSParser::EDSParser)\n")
    (user / "main.cpp").write_text(
        '#include "edsparser/formats/eds.hpp"\n#include "edsparser/transforms/eds_transforms.hpp"\n'
        'int main() { edsparser::EDS e(std::string("{ACGT}{A,C}{GTTT}")); '
        'return (e.length() == 3 && e.cardinality() == 4 && edsparser::is_leds(e, 4)) ? 0 : 1; }\n')
    for cmd in (["cmake", "-S", str(user), "-B", str(user / "b"), "-DCMAKE_PREFIX_PATH=" + str(prefix)],
                ["cmake", "--build", str(user / "b")]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    env = dict(os.environ, LD_LIBRARY_PATH=str(prefix / "lib") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([str(user / "b" / "user")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
