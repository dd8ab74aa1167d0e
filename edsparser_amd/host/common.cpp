// Timer + peak-memory reader (reference: src/cpp/lib/common.cpp:10-41, :44-77).
#include "edsparser/common.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>

namespace edsparser {

struct Timer::Impl {
    std::chrono::steady_clock::time_point t0, t1;
    bool running = false;
    double seconds() const
    {
        auto end = running ? std::chrono::steady_clock::now() : t1;
        return std::chrono::duration<double>(end - t0).count();
    }
};

Timer::Timer() : impl_(new Impl) {}
Timer::~Timer() = default;
void Timer::start() { impl_->t0 = std::chrono::steady_clock::now(); impl_->running = true; }
void Timer::stop() { impl_->t1 = std::chrono::steady_clock::now(); impl_->running = false; }
double Timer::elapsed_seconds() const { return impl_->seconds(); }
double Timer::elapsed_milliseconds() const { return impl_->seconds() * 1e3; }
double Timer::elapsed_microseconds() const { return impl_->seconds() * 1e6; }

double get_peak_memory_mb()
{
    FILE* f = std::fopen("/proc/self/status", "r");
    if (!f) return 0.0;
    char line[256];
    double mb = 0.0;
    while (std::fgets(line, sizeof line, f)) {
        if (std::strncmp(line, "VmHWM:", 6) == 0) {
            long kb = 0;
            if (std::sscanf(line + 6, "%ld", &kb) == 1) mb = kb / 1024.0;
            break;
        }
    }
    std::fclose(f);
    return mb;
}

} // namespace edsparser
