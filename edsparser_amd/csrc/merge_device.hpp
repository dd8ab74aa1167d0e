// merge_device.hpp — host driver of the EDS -> l-EDS merge pipeline (see merge_device.hip).
#pragma once

#include "msa_device.hpp"

#include <string>

namespace edsx {

// One symbol range of a partitioned merge.  Neighbouring ranges share a sentinel: a single-string
// symbol of at least l characters between two degenerate symbols.  No pair with the sentinel is mergeable while its
// neighbours stay degenerate (eds_transforms.cpp:75-97), so runs of mergeable pairs never cross it and both sides
// evolve exactly as they do inside the whole EDS.  A range reports whether its sentinels came through unmerged
// (a LINEAR product can collapse a neighbour to one short string, which then pulls the sentinel in); the caller
// falls back to the unpartitioned merge when one did not.  The left range prints the sentinel, the right one drops it;
// only the last range ends in '\n'.
struct MergeShard {
    bool head_sentinel = false, tail_sentinel = false;   // in
    bool head_intact = true, tail_intact = true;         // out
};

class MergePipeline {
public:
    // eds/seds are host buffers (seds == nullptr => CARTESIAN); outputs end in '\n' like EDS::save.
    void run(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l, bool compact,
             HostBytes& out, HostBytes& seds_out, hipStream_t st, MergeShard* shard = nullptr);

    bool tokenised_on_device() const { return tokenised_on_device_; }

private:
    bool tokenised_on_device_ = false;
    bool tokenize_device(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, bool linear, hipStream_t st,
                         u64& n0, u64& m, u32& W, bool& head_single, bool& tail_single, u64& head_len);
    DevBuf d_raw_, tk_a_, tk_b_, tk_c_, tk_d_, tk_e_, d_sym_first_;
    DevBuf d_chars_, d_str_off_, left_, right_, elen_, bits_, size_[2], ent_off_[2], len1_[2], a_, b_, c_, d_, e_,
           scan_tmp_, ctl_, fin_ent_, fin_flag_, fbytes_, fsbytes_, d_out_, d_sout_;
};

} // namespace edsx
