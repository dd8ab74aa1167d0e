#include "device.hpp"

#include <cstdlib>
#include <iterator>

namespace edsparser::detail {

namespace {
struct Holder {
    edsx_ctx* ctx = nullptr;
    ~Holder() { if (ctx) edsx_ctx_destroy(ctx); }
};
}

edsx_ctx* context()
{
    thread_local Holder h;
    if (!h.ctx) {
        int dev = 0;
        if (const char* e = std::getenv("EDSX_DEVICE")) dev = std::atoi(e);
        int rc = edsx_ctx_create(dev, &h.ctx);
        if (rc != EDSX_OK)
            throw std::runtime_error("edsparser: no usable MI355X (gfx950) device " + std::to_string(dev) +
                                     " — this build has no CPU fallback");
    }
    return h.ctx;
}

void throw_status(int status, edsx_ctx* ctx)
{
    std::string msg = ctx ? edsx_last_error(ctx) : "edsx failure";
    if (status == EDSX_ERR_INVALID_PARAMETER) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

std::string slurp(std::istream& is)
{
    return std::string(std::istreambuf_iterator<char>(is), std::istreambuf_iterator<char>());
}

} // namespace edsparser::detail
