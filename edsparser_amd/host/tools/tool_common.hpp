// shared by the three tools: the "[Performance]" line the reference prints to stderr
// (e.g. src/cpp/tools/msa2eds.cpp:18-27) and small file helpers.
#pragma once
#include "edsparser/common.hpp"

#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <stdexcept>
#include <string>

namespace tool {

inline void print_performance(edsparser::Timer& timer)
{
    timer.stop();
    std::cerr << "[Performance] Runtime: " << std::fixed << std::setprecision(2) << timer.elapsed_seconds() << "s";
    const double mb = edsparser::get_peak_memory_mb();
    if (mb > 0.0) std::cerr << " | Peak Memory: " << std::fixed << std::setprecision(1) << mb << " MB";
    std::cerr << "\n";
}

inline void write_file(const std::filesystem::path& path, const std::string& data, const char* what)
{
    std::ofstream out(path);
    if (!out) throw std::runtime_error(std::string("Failed to open ") + what + " file: " + path.string());
    out << data;
}

} // namespace tool
