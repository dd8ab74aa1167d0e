// shared by the three tools: the "[Performance]" line the reference prints to stderr
// (e.g. src/cpp/tools/msa2eds.cpp:18-27) and small file helpers.
#pragma once
#include "edsparser/common.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

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

// Read-only memory map of an input file: the tools hand the mapped bytes to the C ABI as they are (no
// ifstream -> std::string copy in front of the host-to-device copy; GB-sized alignments and VCFs).
class MappedFile {
public:
    MappedFile(const std::filesystem::path& path, const char* what)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error(std::string("Failed to open ") + what + " file: " + path.string());
        struct stat st;
        if (::fstat(fd_, &st) != 0) { ::close(fd_); throw std::runtime_error(std::string("Failed to open ") + what + " file: " + path.string()); }
        size_ = static_cast<size_t>(st.st_size);
        if (size_) {
            void* p = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (p == MAP_FAILED) { ::close(fd_); throw std::runtime_error(std::string("Failed to map ") + what + " file: " + path.string()); }
            (void)::madvise(p, size_, MADV_SEQUENTIAL);
            data_ = static_cast<const uint8_t*>(p);
        }
    }
    ~MappedFile()
    {
        if (data_) ::munmap(const_cast<uint8_t*>(data_), size_);
        if (fd_ >= 0) ::close(fd_);
    }
    MappedFile(const MappedFile&) = delete;
    MappedFile& operator=(const MappedFile&) = delete;
    const uint8_t* data() const { return data_ ? data_ : reinterpret_cast<const uint8_t*>(""); }
    size_t size() const { return size_; }

private:
    int fd_ = -1;
    const uint8_t* data_ = nullptr;
    size_t size_ = 0;
};

inline void write_bytes(const std::filesystem::path& path, const uint8_t* data, size_t n, const char* what)
{
    std::ofstream out(path, std::ios::binary);
    if (!out) throw std::runtime_error(std::string("Failed to open ") + what + " file: " + path.string());
    out.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(n));
}

} // namespace tool
