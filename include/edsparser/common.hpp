// edsparser/common.hpp — shared typedefs, format constants, timer.
// Drop-in for the reference's src/cpp/lib/common.hpp (same names and values: typedefs :15-18,
// format characters :21-25, extensions :28-34, ErrorCode :37-45, Timer :50-64,
// get_peak_memory_mb :70); written for the MI355X build, which links libedsx.so.
#ifndef EDSPARSER_COMMON_HPP
#define EDSPARSER_COMMON_HPP

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace edsparser {

constexpr const char* VERSION = "1.0.0";

using String = std::string;
using StringSet = std::vector<String>;
using Position = uint64_t;
using Length = uint32_t;

constexpr char SET_OPEN = '{';
constexpr char SET_CLOSE = '}';
constexpr char SET_SEPARATOR = ',';
constexpr char CHANGE_SEPARATOR = '#';
constexpr char EMPTY_STRING_MARKER = '\0';

constexpr const char* EXT_MSA = ".msa";
constexpr const char* EXT_VCF = ".vcf";
constexpr const char* EXT_EDS = ".eds";
constexpr const char* EXT_EDZ = ".edz";
constexpr const char* EXT_SEDS = ".seds";
constexpr const char* EXT_LEDS = ".leds";
constexpr const char* EXT_EDP = ".edp";

// Same values as the C ABI's edsx_status (include/edsx.h).
enum class ErrorCode {
    SUCCESS = 0,
    FILE_NOT_FOUND = 1,
    INVALID_FORMAT = 2,
    INVALID_PARAMETER = 3,
    BUILD_FAILED = 4,
    QUERY_FAILED = 5,
    UNKNOWN_ERROR = 99
};

class Timer {
public:
    Timer();
    ~Timer();
    void start();
    void stop();
    double elapsed_seconds() const;
    double elapsed_milliseconds() const;
    double elapsed_microseconds() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// Peak resident set of this process in MB (VmHWM), 0.0 when /proc is unavailable.
double get_peak_memory_mb();

} // namespace edsparser

#endif
