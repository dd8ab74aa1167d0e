// genvcf.hpp — synthetic VCF + FASTA of BASELINE configs[3]'s shape generated in HBM (see genvcf.hip).
#pragma once
#include "msa_device.hpp"

namespace edsx {

class GenVcfPipeline {
public:
    void run(u64 ref_len, u64 n_records, u32 n_samples, u64 seed, HostBytes& vcf, HostBytes& fasta, hipStream_t st);
private:
    DevBuf len_, scan_tmp_, ctl_, out_;
};

} // namespace edsx
