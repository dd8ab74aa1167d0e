// device.hpp — the one place where the edsparser:: shims meet the C ABI (include/edsx.h).
#pragma once
#include "edsx.h"

#include <istream>
#include <stdexcept>
#include <string>

namespace edsparser::detail {

// One context per host thread on GPU $EDSX_DEVICE (default 0), created on first use.
edsx_ctx* context();

// Map an edsx status to the exception type the reference throws on the same condition.
[[noreturn]] void throw_status(int status, edsx_ctx* ctx);

// Whole stream from its current position (the reference's transforms need seekable streams and
// read them to the end as well).
std::string slurp(std::istream& is);

struct Buf {
    edsx_buf b{nullptr, 0};
    ~Buf() { edsx_buf_free(&b); }
    std::string str() const { return std::string(reinterpret_cast<const char*>(b.data), b.size); }
};

} // namespace edsparser::detail
