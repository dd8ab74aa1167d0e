// msa_device.hpp — host-side driver of the MSA -> EDS / l-EDS device pipeline.
#pragma once
#include <cstdlib>
#include <sys/mman.h>
#include <thread>
#include <vector>
#include <cstring>
#include <new>

#include "dev_util.hpp"

#include <vector>

namespace edsx {

struct FormatError : std::runtime_error { using std::runtime_error::runtime_error; };   // EDSX_ERR_INVALID_FORMAT
struct ParamError : std::runtime_error { using std::runtime_error::runtime_error; };    // EDSX_ERR_INVALID_PARAMETER
// a well-formed input beyond a limit of this build (more than MsaPipeline::MAX_ROWS sequences): EDSX_ERR_BUILD_FAILED,
// not a format error - the reference has no such limit
struct LimitError : std::runtime_error { using std::runtime_error::runtime_error; };

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
        if (e != hipSuccess) {
            ptr = nullptr;
            (void)hipGetLastError();                              // (the failure is reported here, not by the next launch check)
            const std::string what = std::string("hipMalloc of ") + std::to_string(want) + " bytes: " + hipGetErrorString(e);
            if (e == hipErrorOutOfMemory) throw OutOfDeviceMemory(what);
            throw DeviceError(what);
        }
        cap = want;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    template <class T> T* as() const { return static_cast<T*>(ptr); }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

// malloc'ed host bytes (what an edsx_buf hands to the caller); not value-initialised
struct HostBytes {
    uint8_t* data = nullptr;
    size_t size = 0;
    void take(size_t n)
    {
        std::free(data);
        data = alloc(n);
        size = n;
    }
    // Large outputs: 2 MB-aligned and advised for transparent huge pages, so that the first touch of a GB of fresh
    // memory (by the device-to-host copy) is a few hundred page faults instead of a few hundred thousand.  Still
    // released with free().
    static uint8_t* alloc(size_t n)
    {
        void* p = nullptr;
        if (n >= ((size_t)8 << 20)) {
            const size_t huge = (size_t)2 << 20, len = (n + huge - 1) & ~(huge - 1);
            if (posix_memalign(&p, huge, len) == 0 && p) { (void)madvise(p, len, MADV_HUGEPAGE); return static_cast<uint8_t*>(p); }
            p = nullptr;
        }
        p = std::malloc(n ? n : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<uint8_t*>(p);
    }
    void drop_front(size_t k) { if (k > size) k = size; std::memmove(data, data + k, size - k); size -= k; }
    uint8_t* release() { uint8_t* p = data; data = nullptr; size = 0; return p; }
    HostBytes() = default;
    HostBytes(const HostBytes&) = delete;
    HostBytes& operator=(const HostBytes&) = delete;
    ~HostBytes() { std::free(data); }
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
    u64 slow_n;        // variant segments left to the generic (workgroup-per-segment) kernels: by shape
    u64 slow_n2;       // ... and those the fast kernel gave up on (more than KCAP distinct strings)
    u64 idx_done;      // the speculative parallel row index validated: k_index_rows skips its chain
    u64 idx_bad;       // first row whose speculative header position did not validate
    u64 wide_n;        // variant segments of 5..64 strings (k_seg_group -> wide emitter)
    u64 cnt_n, heavy_n; // lengths of the grouping kernels' work lists (adjacent to wide_n: cleared together)
    u64 nvs;           // number of variant segments
    u64 long_n;        // common segments the variant emitters leave to k_emit_common_long (very long ones, and a final one)
    u64 heavy2_n;      // segments the light grouping kernel hands to a second heavy pass (other alphabets)
    u64 wide16_n;      // variant segments of 9..64 strings (wide_n: those of 5..8); adjacent to heavy2_n: cleared together
    u64 rl_next;       // row-loop kernels: the next variant segment a wave takes
};

// One vc column holds the bytes of one variant column in NATURAL row order: byte r = row r, pitch =
// S rounded up to 16 plus 16 bytes of slack (a lane's 16-byte load at 16*lane never leaves the column).
// The wave-per-segment kernels give lane l the rows 16l .. 16l+15 (one 16-byte load), so row order is
// (lane, byte) and the ids of a lane's rows are consecutive numbers.
__host__ __device__ inline u32 vc_pitch(u32 S) { return (S + 15u) / 16u * 16u + 16u; }

// grouping record of a variant segment (wave-per-segment kernels, S <= 1024), see msa_device.hip
constexpr u32 REC_HDR = 192;      // u32 k | ncol << 16; u64 first slot; u16 first row of each of up to 64 strings
__host__ __device__ inline u32 rec_gid_bytes(u32 S) { return 16u * ((S + 15u) / 16u); }       // up to 8 bits per row
__host__ __device__ inline u32 rec_stride(u32 S) { return (rec_gid_bytes(S) + REC_HDR + 63u) & ~63u; }

// column access through V / vc (see msa_device.hip)
struct MsaView {
    const uint8_t* file; const u64* row_start; const u64* V; const u64* Vraw; const u64* word_slot;
    const uint8_t* vc; MsaHdr* hdr; u64 L, lw; u32 S, Spad;
    u32 tileW;                              // columns per tile of the column scan: slots are consecutive inside a tile
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

// parameters of the wave-per-segment fast kernels (S <= 1024), see msa_device.hip
struct FastParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; u64* segmeta;
    u64* eds_len; u64* seds_len;            // sizes (count pass) == offsets (emit pass, after the scans)
    u64* slow_list; u64* slow_count;        // variant segments left to the generic kernels (k_seg_meta)
    u64* slow_list2; u64* slow_count2;      // ... added by k_seg_group
    u64* wide_list; u64* wide_count;        // variant segments (ordinals) of 5..8 strings: work list of the wide emitter's K <= 8 instantiation
    u64* wide16_list; u64* wide16_count;    // ... of 9..64 strings
    u64* wide_flag; u64* wide16_flag;       // per variant segment: grouped by the column scan with 5..8 / 9..16 strings (-> those lists)
    u64* cnt_meta; u64* cnt_flag;           // per variant segment: column descriptor (0: not for k_seg_group), flag / list position
    u64* cnt_vi; u64* cnt_cm; u64* cnt_n;   // work list of k_seg_group: ordinal among the variant segments, column descriptor
    u64* heavy_vi; u64* heavy_cm; u64* heavy_n;   // ... of its heavy instantiation (11..64 columns, mixed segments)
    u64* heavy_flag;                        // per variant segment: on the heavy list (scanned into the list position)
    u64* heavy2_vi; u64* heavy2_cm; u64* heavy2_n;   // what the light instantiation gives up on (another alphabet): second heavy pass
    uint8_t* eds; uint8_t* seds; u64 tok_total;
    uint8_t* rec; u32 rec_stride, rec_gid;  // grouping records (count -> emit): stride, bytes of the group-id area
    const u64* Fraw; const u32* rec_info;   // column scan's own groupings: first-column bitmap, k | textlen << 8 | ok << 31 per slot
    const uint8_t* recf; u32 recf_stride, recf_gid;   // ... and their records (indexed by slot)
    u64* long_list; u64* long_count;        // common segments for k_emit_common_long
};

// parameters of the row-loop kernels (more than 1024 rows), see msa_rowloop_kernels.hpp
struct RlParams {
    MsaView mv; const u64* seg_start; const u64* nseg_ptr; u64* eds_len; u64* seds_len; u64 tok_total;
    u64* segmeta;                             // per segment: META_REC | vi when the row-loop kernels own it, else 0
    u64* slow_list; u64* slow_count;          // variant segments for the generic kernels
    u64* long_list; u64* long_count;          // common segments for k_emit_common_long
    uint8_t* rec; u64 rec_stride;
    uint8_t* eds; uint8_t* seds;              // emit pass (eds_len / seds_len hold the offsets then)
    u64* next;                                // work counter: the next variant segment (zeroed before each launch)
};

class MsaPipeline {
public:
    static constexpr u64 MAX_ROWS = 9999999;   // sequence ids of up to seven digits (an id + ',' is one 8-byte token)
    static constexpr u64 LDS_ROWS = 8192;      // LDS budget of the generic kernels (18 B / row); more rows: tables in HBM scratch
    static constexpr u64 ROW_CAP = 65536;      // first size of the row table (grown once when an alignment has more rows)
    static constexpr u64 ROW_PAD = 4200;       // copies of the last row behind it (16-row loads of the column scan)

    ~MsaPipeline();
    void plan(const uint8_t* d_msa, size_t n, uint32_t l, hipStream_t st, uint64_t* eds_bytes, uint64_t* seds_bytes);
    void emit(uint8_t* d_eds, uint8_t* d_seds, hipStream_t st);
    const MsaHdr& header() const { return h_; }
    struct Edges { u64 nseg, fvar, fcols, feds, fseds, lvar, lcols, leds, lseds; };
    Edges edge_info(hipStream_t st);
    void copy_columns(u64 col0, u64 ncols, uint8_t* host_out, hipStream_t st);
    // first / last common segment of at least min_cols columns: segment numbers, the last one's first column and text
    // offsets, the first one's end column and the text offsets behind it
    struct Anchors { u64 nseg, found, first_seg, last_seg, last_col, last_eds, last_seds, first_end, first_eds_end, first_seds_end; };
    Anchors anchor_info(u64 min_cols, hipStream_t st);
    struct SegLoc { u64 seg, col, eds_off, seds_off; };
    SegLoc locate(u64 col, hipStream_t st);
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
    unsigned persistent_grid(const void* kern, int threads, size_t dyn_lds) const;

    const uint8_t* file_ = nullptr;
    size_t n_ = 0;
    uint32_t l_ = 0;
    bool planned_ = false, timing_ = false;
    MsaHdr h_{};
    std::vector<TimedKernel> timed_;

    DevBuf hdr_, rows_, vraw_, v_, wslot_, vc_, hrun_, hseg_, cnt_, wbase_, segbase_, scan_tmp_,
           run_start_, flag_, seg_start_, eds_len_, seds_len_, segmeta_, slow_list_, cnt_list_, rec_, recf_, rec_info_, fraw_, colbuf_, idx_tmp_, gcache_,
           long_list_;
    u64 vc_cap_cols_ = 0;

    // emit-time view
    MsaView mv_{};
    const u64* seg_start_p_ = nullptr; const u64* hseg_p_ = nullptr; const u64* segbase_p_ = nullptr;
    const u64* nseg_p_ = nullptr;
    size_t seg_lds_ = 0;
    u32 stage_off_ = 0, stage_cols_ = 0;   // generic kernels: column staging area in their LDS (offset, capacity; 0: none)
    bool fast_ = false, fuse_ = false;
    u32 recf_stride_ = 0, recf_gid_ = 0;
    RlParams rl_{};                          // more than 1024 rows: the row-loop kernels' parameters (plan -> emit)
    bool big_ = false;                       // more than LDS_ROWS rows
    unsigned big_grid_ = 0;
    DevBuf seg_scratch_; size_t seg_scratch_stride_ = 0;      // ... their row tables: one slice per workgroup
    u64 gc_stride_ = 0;                      // grouping cache of the generic kernels: entry size, bytes per region
    size_t gc_region_ = 0;
    FastParams fp_{};
    int cus_ = 0;
    // side streams of emit(): the small emitters run next to each other
    hipStream_t side_[2] = {nullptr, nullptr};
    hipEvent_t side_ev_[3] = {nullptr, nullptr, nullptr};
    bool side_ready_ = false;
    void ensure_side_streams();
};

} // namespace edsx
