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

// EDS::calculate_statistics / calculate_source_statistics (eds.cpp:361-470, :472-505) and is_leds
// (eds_transforms.cpp:439-468) of an .eds (+ .seds) text, as reductions over the tokenised arrays in HBM.
struct EdsStats {
    u64 n_symbols, n_chars, n_strings;                 // n, N, m
    u64 num_degenerate, total_change_size, num_common_chars, num_context_blocks, min_context, max_context, num_empty_strings;
    u64 has_sources, num_paths, max_paths_per_string, total_paths;
    u64 is_leds;                                       // for the given context length
};

class MergePipeline {
public:
    // statistics + l-EDS validity of an .eds (+ .seds) text; seds == nullptr: no sources
    void stats(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l, EdsStats& out, hipStream_t st);

    // eds/seds are host buffers (seds == nullptr => CARTESIAN); outputs end in '\n' like EDS::save.
    void run(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, uint32_t l, bool compact,
             HostBytes& out, HostBytes& seds_out, hipStream_t st, MergeShard* shard = nullptr);

    bool tokenised_on_device() const { return tokenised_on_device_; }

private:
    struct Loaded { u64 n0 = 0, m = 0, head_len = 0; u32 W = 1; bool head_single = false, tail_single = false; };
    void prepare(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, bool linear, hipStream_t st, Loaded& L);
    bool tokenised_on_device_ = false;
    bool tokenize_device(const uint8_t* eds, size_t eds_n, const uint8_t* seds, size_t seds_n, bool linear, hipStream_t st,
                         u64& n0, u64& m, u32& W, bool& head_single, bool& tail_single, u64& head_len);
    DevBuf d_raw_, tk_a_, tk_b_, tk_c_, d_sym_first_, fin_spill_;
    u64 rounds_run_ = 0;                     // merge rounds of the current call (bounds the depth of the entry trees)
    DevBuf d_chars_, d_str_off_, left_, right_, elen_, bits_, size_[2], ent_off_[2], len1_[2], a_, b_, c_, d_, e_,
           scan_tmp_, ctl_, fin_ent_, fin_flag_, fbytes_, fsbytes_, d_out_, d_sout_;
};

} // namespace edsx
