// multi_gpu.hpp — MSA -> EDS over several GPUs of one node from C++: column slabs + boundary stitch over RCCL.
//
// One host thread per GPU.  Every rank cuts its column slab [L*r/N, L*(r+1)/N) out of every row of the host FASTA image
// (column c of row s is byte start[s] + c + c / line_width, msa_transforms.cpp:268-269), uploads it as a
// one-line-per-row image (2D copies: a row's slab is one contiguous piece), transforms it with its own MsaPipeline as
// if it were a whole alignment, and the runs that cross a slab boundary are repaired with KB-sized exchanges
// (SURVEY §8(e)): an all-gather of twelve numbers per rank (the edge descriptors), and - only when a VARIANT run
// crosses a boundary - an all-gather of the raw boundary columns, from which the left-most rank recomputes that
// segment.  With a context length l > 0 every boundary is recomputed between the nearest standalone common runs of at
// least l columns on either side (run_rank_leds in multi_gpu.hip).  The exchanges are ncclAllGather calls on one communicator per rank (RcclExchange); LocalExchange moves the
// same bytes between the rank threads directly and exists so that the whole path can be tested with several contexts
// sharing ONE GPU (RCCL does not run two ranks on one device).
#pragma once

#include "msa_device.hpp"

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace edsx {

struct SlabEdges {              // twelve u64, exchanged as they are
    u64 n_segments, cols, eds_bytes, seds_bytes;
    u64 first_is_variant, first_cols, first_eds_bytes, first_seds_bytes;
    u64 last_is_variant, last_cols, last_eds_bytes, last_seds_bytes;
};
struct SlabChain { int first, last; bool variant; };      // a run of one type over slabs first .. last
struct SlabAction {             // what a rank does to its slab text: bytes dropped at both ends, chains it recomputes
    u64 front_eds = 0, front_seds = 0, back_eds = 0, back_seds = 0;
    std::vector<int> owns;
};
struct StitchPlan { std::vector<SlabChain> chains; std::vector<SlabAction> actions; };
StitchPlan plan_stitch(const std::vector<SlabEdges>& edges);               // pure host logic, the same on every rank

// geometry of a plain uniform FASTA alignment image; ok == false: not partitioned (the unpartitioned transform words the error)
struct MsaLayout { bool ok = false; std::vector<u64> start; u64 draw = 0, lw = 0, L = 0; };
MsaLayout msa_layout(const uint8_t* f, size_t n);

// columns [c0, c1) of every row as a one-line-per-row image in HBM, every row on a multiple of 128 bytes (see multi_gpu.hip)
struct RowImage { u64 rows = 0, cols = 0, hdr0 = 0, hdr = 0, bytes = 0; };
RowImage upload_row_image(const uint8_t* fasta, const MsaLayout& lay, u64 c0, u64 c1, DevBuf& d_img, std::vector<uint8_t>& host_tmp,
                          hipStream_t st);

// msa2eds in K column batches on one GPU (working set of one slab; see multi_gpu.hip).  false: not batched - the image is
// not a plain uniform alignment, the batches would be too narrow, or (l > 0) a slab has no standalone common runs to
// anchor the stitch on; nothing has been written to eds / seds then.
struct BatchResources { MsaPipeline* slab; MsaPipeline* mini; DevBuf* d_img; DevBuf* d_eds; DevBuf* d_seds; DevBuf* d_mini; std::vector<uint8_t>* host_tmp; };
bool msa_transform_batched(const BatchResources& R, const uint8_t* fasta, const MsaLayout& lay, uint32_t l, int K,
                           HostBytes& eds, HostBytes& seds, hipStream_t st);

// all ranks call with `bytes` bytes each; `all` receives world * bytes bytes in rank order
class Exchange {
public:
    virtual ~Exchange() = default;
    virtual void all_gather(int rank, const void* mine, size_t bytes, void* all) = 0;
    virtual const char* name() const = 0;
};

class RankBarrier {             // threads of one process; carries the first failure text to every rank
public:
    explicit RankBarrier(int n) : n_(n) {}
    void arrive(int rank, const std::string* failure);     // returns when all n have arrived
    bool failed() const { return failed_; }
    std::string message() const { return msg_; }
    void reset() { failed_ = false; msg_.clear(); }
private:
    int n_, count_ = 0; unsigned long gen_ = 0;
    bool failed_ = false; std::string msg_; int failed_rank_ = -1;
    std::mutex mu_; std::condition_variable cv_;
};

class MultiMsa {
public:
    MultiMsa(const std::vector<int>& devices, bool use_rccl);
    ~MultiMsa();
    // msa2eds: column slabs (context length > 0: stitched between standalone common runs); an odd file, or an l-EDS
    // without such runs near the slab ends: rank 0 transforms the whole image
    void transform(const uint8_t* fasta, size_t n, uint32_t context_len, HostBytes& eds, HostBytes& seds);
    int world() const { return (int)devices_.size(); }
    bool partitioned() const { return partitioned_; }
    int chains() const { return chains_; }
    const char* exchange_name() const { return xch_ ? xch_->name() : "none"; }
private:
    struct Rank;
    void run_rank(int r, const uint8_t* fasta, size_t n, const MsaLayout& lay, HostBytes& eds, HostBytes& seds);
    void run_rank_leds(int r, const uint8_t* fasta, const MsaLayout& lay, uint32_t l, HostBytes& eds, HostBytes& seds);
    std::vector<int> devices_;
    std::vector<std::unique_ptr<Rank>> ranks_;
    std::unique_ptr<Exchange> xch_;
    std::unique_ptr<RankBarrier> bar_;
    bool partitioned_ = false;
    bool no_anchor_ = false;             // l > 0: some slab has no standalone common runs to anchor the stitch on
    int chains_ = 0;
    // shared between the rank threads of one call
    std::vector<u64> piece_e_, piece_s_;
};

} // namespace edsx
