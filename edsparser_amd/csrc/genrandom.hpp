// genrandom.hpp — genrandomeds-shaped .eds / .seds text generated in HBM (see genrandom.hip).
#pragma once
#include "msa_device.hpp"

namespace edsx {

struct GenParams {
    u64 total_bp; double variability; u32 min_alt, max_alt, var_len_max; double snp_ratio;
    uint8_t alphabet[64]; u32 alpha_n; u64 min_context; u64 seed;
};

class GenPipeline {
public:
    // host outputs; n_sites = degenerate symbols written
    void run(const GenParams& p, HostBytes& eds, HostBytes& seds, u64& n_sites, hipStream_t st);

private:
    DevBuf ebytes_, sbytes_, scan_tmp_, ctl_, out_e_, out_s_;
};

} // namespace edsx
