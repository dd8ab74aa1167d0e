// msa_device.hpp — host-side driver of the MSA -> EDS / l-EDS device pipeline.
#pragma once

#include "dev_util.hpp"

#include <vector>

namespace edsx {

struct FormatError : std::runtime_error { using std::runtime_error::runtime_error; };   // EDSX_ERR_INVALID_FORMAT
struct ParamError : std::runtime_error { using std::runtime_error::runtime_error; };    // EDSX_ERR_INVALID_PARAMETER

// grow-only device allocation
struct DevBuf {
    void* ptr = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes)
    {
        if (bytes <= cap) return;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
        size_t want = (bytes + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(&ptr, want);
        if (e != hipSuccess) { ptr = nullptr; throw DeviceError(std::string("hipMalloc: ") + hipGetErrorString(e)); }
        cap = want;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(ptr); }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// device header block: geometry, counters and status, all u64
struct MsaHdr {
    u64 hdr_end, first_nl, first_hdr2;
    u64 S, L, lw, Draw;
    u64 nwords, nwords_raw;
    u64 status;
    u64 nv;            // variant columns (slot allocator)
    u64 R, nseg, E, Q;
    u64 tmp_total;
};

// column access through V / vc (see msa_device.hip)
struct MsaView {
    const uint8_t* file; const u64* row_start; const u64* V; const u64* Vraw; const u64* word_slot;
    const uint8_t* vc; const MsaHdr* hdr; u64 L, lw; u32 S, Spad;
    __device__ __forceinline__ u64 raw(u64 c) const { return lw ? c + c / lw : c; }
    __device__ __forceinline__ u32 vbit(u64 c) const { return (u32)(V[c >> 6] >> (c & 63)) & 1u; }
    __device__ __forceinline__ u64 slot(u64 c) const
    {
        u64 q = raw(c);
        u64 bits = Vraw[q >> 6] & ((1ull << (q & 63)) - 1ull);
        return word_slot[q >> 6] + __builtin_popcountll(bits);
    }
    __device__ __forceinline__ u32 ref_byte(u64 c) const { return file[row_start[0] + raw(c)]; }
};

class MsaPipeline {
public:
    static constexpr u64 MAX_ROWS = 8192;      // LDS budget of the grouping kernels (18 B / row)
    static constexpr u64 ROW_CAP = 65536;

    ~MsaPipeline();
    void plan(const uint8_t* d_msa, size_t n, uint32_t l, hipStream_t st, uint64_t* eds_bytes, uint64_t* seds_bytes);
    void emit(uint8_t* d_eds, uint8_t* d_seds, hipStream_t st);
    const MsaHdr& header() const { return h_; }
    bool planned() const { return planned_; }
    size_t msa_bytes() const { return n_; }

    void set_timing(bool on);
    int get_timing(const char** names, float* ms, int* counts, int cap);

private:
    struct TimedKernel { const char* name; hipEvent_t t0, t1; };
    struct TimeAcc { const char* name; double total_ms; int count; };
    std::vector<TimeAcc> acc_;
    void launch_timer_begin(const char* name, hipStream_t st);
    void launch_timer_end(hipStream_t st);
    void clear_timers();
    void plan_body(hipStream_t st);
    unsigned seg_grid() const;

    const uint8_t* file_ = nullptr;
    size_t n_ = 0;
    uint32_t l_ = 0;
    bool planned_ = false, timing_ = false;
    MsaHdr h_{};
    std::vector<TimedKernel> timed_;

    DevBuf hdr_, rows_, vraw_, v_, wslot_, vc_, hrun_, hseg_, cnt_, wbase_, segbase_, scan_tmp_,
           run_start_, flag_, seg_start_, eds_len_, seds_len_;
    u64 vc_cap_cols_ = 0;

    // emit-time view
    MsaView mv_{};
    const u64* seg_start_p_ = nullptr; const u64* hseg_p_ = nullptr; const u64* segbase_p_ = nullptr;
    const u64* nseg_p_ = nullptr;
    size_t seg_lds_ = 0;
};

} // namespace edsx
