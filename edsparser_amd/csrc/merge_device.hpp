// merge_device.hpp — host driver of the EDS -> l-EDS merge pipeline (see merge_device.hip).
#pragma once

#include "msa_device.hpp"

#include <string>

namespace edsx {

class MergePipeline {
public:
    // eds/seds are host buffers (seds == nullptr => CARTESIAN); outputs end in '\n' like EDS::save.
    void run(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l, bool compact,
             std::string& out, std::string& seds_out, hipStream_t st);

private:
    DevBuf d_chars_, d_str_off_, left_, right_, elen_, bits_, size_[2], ent_off_[2], len1_[2], a_, b_, c_, d_, e_,
           scan_tmp_, ctl_, fin_ent_, fin_flag_, fbytes_, fsbytes_, d_out_, d_sout_;
};

} // namespace edsx
